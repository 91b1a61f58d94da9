"""Parity of the wavefront backend (workgroup-resident ray queues, material-sorted shading) with the CPU oracle and
with the megakernel: same bar as test_gpu_parity.py — bit-exact linear f32 accumulators and identical traversal counters."""
import numpy as np
import pytest
from test_gpu_parity import STAT_KEYS, assert_bit_equal

pytestmark = pytest.mark.gpu
WAVEFRONT = 1


def render_wf(trt, desc, spp, depth, seed=1, stats=True, **over):
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, depth, False, desc["background"], seed=seed, backend=WAVEFRONT)
    img = r.render(pcam, pw, collect_stats=False, **over)          # production kernel: culling tree
    gst = r.last_stats
    if stats:
        counted = r.render(pcam, pw, collect_stats=True, **over)   # counting kernel: reference tree, oracle's counters
        assert_bit_equal(counted.data, img.data, "wavefront counting kernel vs production kernel")
        gst = r.last_stats
    return img.data, gst


def oracle(orc, desc, spp, depth, seed=1):
    ow, ocam = orc.world_from_description(desc)
    return orc.render(ow, ocam, spp, depth, desc["background"], seed=seed, nthreads=8)


def test_cornell_config1_bit_exact_with_counters(trt, orc):
    desc = trt.scenes.cornell(400, 400)
    gpu, gst = render_wf(trt, desc, 8, 8)
    cpu, cst = oracle(orc, desc, 8, 8)
    assert_bit_equal(gpu, cpu, "wavefront cornell 400x400")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


def test_random_spheres_all_materials(trt, orc):
    """BASELINE config 3's scene (the wavefront path's target): Lambertian / metal / dielectric / sky misses."""
    desc = trt.scenes.random_spheres(240, 135)
    gpu, gst = render_wf(trt, desc, 8, 50)
    cpu, cst = oracle(orc, desc, 8, 50)
    assert_bit_equal(gpu, cpu, "wavefront random spheres")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


def test_dummy_spheres_and_quads(trt, orc):
    for desc, spp, depth in ((trt.scenes.dummy_spheres("renderer", 200, 150), 16, 10), (trt.scenes.quad_test(200, 150), 10, 10)):
        gpu, gst = render_wf(trt, desc, spp, depth)
        cpu, cst = oracle(orc, desc, spp, depth)
        assert_bit_equal(gpu, cpu, desc["name"])
        assert gst["rays"] == cst["rays"] and gst["node_tests"] == cst["node_tests"]


def test_large_scene_from_global_memory(trt, orc):
    desc = trt.scenes.sphere_grid(4000, 160, 90)
    gpu, gst = render_wf(trt, desc, 4, 50)
    cpu, cst = oracle(orc, desc, 4, 50)
    assert_bit_equal(gpu, cpu, "wavefront sphere grid 4000")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


@pytest.mark.parametrize("wh", [(2, 2), (17, 5), (65, 33), (130, 3), (64, 32), (200, 70)])
def test_ragged_image_sizes(trt, orc, wh):
    desc = trt.scenes.cornell(*wh)
    gpu, gst = render_wf(trt, desc, 3, 6)
    cpu, cst = oracle(orc, desc, 3, 6)
    assert_bit_equal(gpu, cpu, f"wavefront image {wh}")
    assert gst["samples"] == wh[0] * wh[1] * 3


@pytest.mark.parametrize("serve_min", [1, 8, 48, 64])
def test_refill_threshold_never_changes_the_frame(trt, orc, serve_min):
    """The EXTEND phase's serve threshold (trt_tuning.wf_serve_min) is a scheduling knob: every value gives the same bits."""
    desc = trt.scenes.random_spheres(96, 64)
    gpu, gst = render_wf(trt, desc, 4, 20, tuning={"wf_serve_min": serve_min})
    cpu, cst = oracle(orc, desc, 4, 20)
    assert_bit_equal(gpu, cpu, f"serve_min {serve_min}")
    assert gst["node_tests"] == cst["node_tests"]


def test_bounce_budgets_and_progressive(trt, orc):
    desc = trt.scenes.cornell(48, 40)
    for depth in (1, 2):
        gpu, _ = render_wf(trt, desc, 4, depth, stats=False)
        cpu, _ = oracle(orc, desc, 4, depth)
        assert_bit_equal(gpu, cpu, f"wavefront depth {depth}")
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(8, 1, 10, False, desc["background"], backend=WAVEFRONT)
    one = r.render(pcam, pw).data
    a = r.render(pcam, pw, sample_begin=0, sample_end=3).data
    b = r.render(pcam, pw, accum=a.copy(), sample_begin=3, sample_end=8, accumulate=1).data
    assert_bit_equal(b, one, "wavefront progressive")
    mega = trt.Renderer(8, 1, 10, False, desc["background"], backend=0).render(pcam, pw).data
    assert_bit_equal(one, mega, "wavefront vs megakernel")


def test_row_bands(trt):
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    desc = trt.scenes.cornell(40, 70)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(4, 1, 8, False, desc["background"], backend=WAVEFRONT)
    full = r.render(pcam, pw).data
    out = np.zeros_like(full)
    for rank in range(3):
        lay = tiles.band_layout(70, 3, rank, 16)
        part = r.render(pcam, pw, band_rows=16, band_stride=3, band_offset=rank, rows_local=lay["rows_local"]).data
        out[lay["rows"]] = part
    assert_bit_equal(out, full, "wavefront bands")


def test_full_size_random_spheres_1080p_properties(trt):
    """BASELINE config 3's image size (1920x1080, depth 50): accounting, determinism, backend agreement."""
    desc = trt.scenes.random_spheres(1920, 1080)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(4, 1, 50, False, desc["background"], backend=WAVEFRONT)
    full = r.render(pcam, pw).data
    st = r.last_stats
    assert st["samples"] == 1920 * 1080 * 4 and st["samples"] <= st["rays"] <= 50 * st["samples"]
    again = r.render(pcam, pw).data
    assert_bit_equal(again, full, "wavefront determinism")
    mega = trt.Renderer(4, 1, 50, False, desc["background"], backend=0).render(pcam, pw).data
    assert_bit_equal(full, mega, "wavefront vs megakernel 1080p")
