"""Parity of the pooled megakernel (two pixels per lane, rays traced from a per-wave LDS pool with lane refill) with
the CPU oracle: bit-exact accumulators, identical traversal counters."""
import numpy as np
import pytest
from test_gpu_parity import STAT_KEYS, assert_bit_equal

pytestmark = pytest.mark.gpu
POOLED = 3


def render(trt, desc, spp, depth, seed=1, stats=True, **over):
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, depth, False, desc["background"], seed=seed, backend=POOLED)
    img = r.render(pcam, pw, collect_stats=False, **over)
    gst = r.last_stats
    if stats:
        counted = r.render(pcam, pw, collect_stats=True, **over)
        assert_bit_equal(counted.data, img.data, "pooled counting kernel vs production kernel")
        gst = r.last_stats
    return img.data, gst


def oracle(orc, desc, spp, depth, seed=1):
    ow, ocam = orc.world_from_description(desc)
    return orc.render(ow, ocam, spp, depth, desc["background"], seed=seed, nthreads=8)


@pytest.mark.parametrize("scene,spp,depth", [("cornell", 8, 8), ("cornell_deep", 16, 50), ("spheres", 8, 50), ("dummy", 16, 10),
                                             ("quads", 10, 10), ("grid", 4, 50)])
def test_scenes_bit_exact_with_counters(trt, orc, scene, spp, depth):
    desc = {"cornell": lambda: trt.scenes.cornell(400, 400), "cornell_deep": lambda: trt.scenes.cornell(96, 96),
            "spheres": lambda: trt.scenes.random_spheres(240, 135), "dummy": lambda: trt.scenes.dummy_spheres("renderer", 200, 150),
            "quads": lambda: trt.scenes.quad_test(200, 150), "grid": lambda: trt.scenes.sphere_grid(4000, 160, 90)}[scene]()
    gpu, gst = render(trt, desc, spp, depth)
    cpu, cst = oracle(orc, desc, spp, depth)
    assert_bit_equal(gpu, cpu, f"pooled {scene}")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


@pytest.mark.parametrize("wh", [(2, 2), (17, 5), (33, 47), (32, 16), (130, 3), (200, 70)])
def test_ragged_image_sizes(trt, orc, wh):
    desc = trt.scenes.cornell(*wh)
    gpu, gst = render(trt, desc, 3, 6)
    cpu, _ = oracle(orc, desc, 3, 6)
    assert_bit_equal(gpu, cpu, f"pooled image {wh}")
    assert gst["samples"] == wh[0] * wh[1] * 3


@pytest.mark.parametrize("serve_min", ["1", "8", "40", "64"])
def test_serve_threshold_never_changes_the_frame(trt, orc, serve_min, monkeypatch):
    monkeypatch.setenv("TRT_POOL_SERVE_MIN", serve_min)
    desc = trt.scenes.random_spheres(96, 64)
    gpu, gst = render(trt, desc, 4, 20)
    cpu, cst = oracle(orc, desc, 4, 20)
    assert_bit_equal(gpu, cpu, f"pooled serve_min {serve_min}")
    assert gst["node_tests"] == cst["node_tests"]


def test_progressive_bands_and_backend_agreement(trt, orc):
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    desc = trt.scenes.cornell(40, 70)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(8, 1, 10, False, desc["background"], backend=POOLED)
    one = r.render(pcam, pw).data
    a = r.render(pcam, pw, sample_begin=0, sample_end=3).data
    b = r.render(pcam, pw, accum=a.copy(), sample_begin=3, sample_end=8, accumulate=1).data
    assert_bit_equal(b, one, "pooled progressive")
    assert_bit_equal(one, trt.Renderer(8, 1, 10, False, desc["background"], backend=0).render(pcam, pw).data, "pooled vs megakernel")
    out = np.zeros_like(one)
    for rank in range(3):
        lay = tiles.band_layout(70, 3, rank, 16)
        out[lay["rows"]] = r.render(pcam, pw, band_rows=16, band_stride=3, band_offset=rank, rows_local=lay["rows_local"]).data
    assert_bit_equal(out, one, "pooled bands")
    for depth in (1, 2):
        gpu, _ = render(trt, desc, 4, depth, stats=False)
        cpu, _ = oracle(orc, desc, 4, depth)
        assert_bit_equal(gpu, cpu, f"pooled depth {depth}")
