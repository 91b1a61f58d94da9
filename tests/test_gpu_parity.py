"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT linear f32 accumulators (tighter than north_star's per-channel |delta| < 1e-3): one lane walks one
pixel's samples in order, every f32 operation is the reference's, so any differing bit is a bug.  The GPU's
traversal counters must equal the oracle's too: the kernel visits exactly the reference's node sequence."""
import ctypes as C
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MEGAKERNEL = 0          # this file pins the megakernel backend; the others have their own files
STAT_KEYS = ("samples", "rays", "node_tests", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(gpu, cpu, what=""):
    if not np.array_equal(bits(gpu), bits(cpu)):
        bad = np.argwhere(bits(gpu) != bits(cpu))
        d = np.abs(gpu.astype(np.float64) - cpu.astype(np.float64))
        raise AssertionError(f"{what}: {len(bad)} of {gpu.size} values differ, max |delta| {np.nanmax(d):.3e}, first at {bad[0]}")


def render_both(trt, orc, desc, spp, depth, seed=1, nthreads=8, stats=True, **over):
    pw, pcam = trt.world_from_description(desc)
    ow, ocam = orc.world_from_description(desc)
    r = trt.Renderer(spp, 1, depth, False, desc["background"], seed=seed, backend=MEGAKERNEL)
    img = r.render(pcam, pw, collect_stats=False, **over)          # production kernel: walks the culling tree
    gst = r.last_stats
    acc, st = orc.render(ow, ocam, spp, depth, desc["background"], seed=seed, nthreads=nthreads)
    if stats:
        # counting kernel on the reference tree: same frame, and its counters are the oracle's
        counted = r.render(pcam, pw, collect_stats=True, **over)
        assert_bit_equal(counted.data, img.data, "counting kernel (reference tree) vs production kernel (culling tree)")
        gst = r.last_stats
        # counting kernel on the culling tree: exactly the reference's primitive tests and hits, whatever the number of
        # postponed-leaf slots (default 4); with one slot (plain while-while) also fewer (or equal) box tests
        own = r.render(pcam, pw, collect_stats=2, **over)
        assert_bit_equal(own.data, img.data, "counting kernel on the culling tree")
        for k in ("samples", "rays", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
            assert r.last_stats[k] == gst[k], k
        plain = r.render(pcam, pw, collect_stats=2, tuning={"leaf_slots": 1}, **over)
        assert_bit_equal(plain.data, img.data, "counting kernel on the culling tree, one leaf slot")
        for k in ("samples", "rays", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
            assert r.last_stats[k] == gst[k], k
        assert r.last_stats["node_tests"] <= gst["node_tests"]
    return img.data, gst, acc, st


def test_library_sees_the_gpu(trt):
    assert trt.lib.trt_device_count() >= 1


def test_cornell_config1_bit_exact_with_counters(trt, orc):
    """BASELINE config 1: Cornell box 400x400, 8 spp, depth 8."""
    desc = trt.scenes.cornell(400, 400)
    gpu, gst, cpu, cst = render_both(trt, orc, desc, 8, 8)
    assert_bit_equal(gpu, cpu, "cornell 400x400")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k
    assert abs(np.nanmean(gpu) - np.nanmean(cpu)) == 0
    # |delta| < 1e-3, the tolerance north_star states, holds trivially
    assert np.nanmax(np.abs(gpu - cpu)) < 1e-3


def test_cornell_deep_paths(trt, orc):
    desc = trt.scenes.cornell(96, 96)
    gpu, gst, cpu, cst = render_both(trt, orc, desc, 32, 50)
    assert_bit_equal(gpu, cpu, "cornell depth 50")
    assert gst["rays"] == cst["rays"] and gst["node_tests"] == cst["node_tests"]


@pytest.mark.parametrize("seed", [1, 2, 12345, 0xFFFFFFFF])
def test_seeds(trt, orc, seed):
    desc = trt.scenes.cornell(48, 40)
    gpu, _, cpu, _ = render_both(trt, orc, desc, 4, 12, seed=seed, stats=False)
    assert_bit_equal(gpu, cpu, f"seed {seed}")


def test_metal_and_dielectric_spheres(trt, orc):
    """The reference's own 5-sphere test world (renderer.rs:87-123): Lambertian + glass shell + fuzzy metal."""
    desc = trt.scenes.dummy_spheres("renderer", 200, 150)
    gpu, gst, cpu, cst = render_both(trt, orc, desc, 16, 10)
    assert_bit_equal(gpu, cpu, "dummy spheres")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


def test_quad_test_scene(trt, orc):
    desc = trt.scenes.quad_test(200, 150)
    gpu, _, cpu, _ = render_both(trt, orc, desc, 10, 10)
    assert_bit_equal(gpu, cpu, "quad_test")


def test_random_spheres_lds_resident_scene(trt, orc):
    """BASELINE config 3's scene (~480 spheres, all four material paths), reduced resolution."""
    desc = trt.scenes.random_spheres(240, 135)
    pw, _ = trt.world_from_description(desc)
    assert pw.get_bvh().info()["lds_bytes"] > 0
    gpu, gst, cpu, cst = render_both(trt, orc, desc, 8, 50)
    assert_bit_equal(gpu, cpu, "random spheres")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


def test_large_scene_traversed_from_global_memory(trt, orc):
    """BASELINE config 5's generator at 4000 spheres: too big for LDS, everything comes through L1/L2."""
    desc = trt.scenes.sphere_grid(4000, 160, 90)
    pw, _ = trt.world_from_description(desc)
    info = pw.get_bvh().info()
    assert info["device_bytes"] > 64 * 1024 and info["lds_bytes"] == 0
    gpu, gst, cpu, cst = render_both(trt, orc, desc, 4, 50)
    assert_bit_equal(gpu, cpu, "sphere grid 4000")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


@pytest.mark.parametrize("case", ["camera inside", "camera on the domain's edge", "camera far outside", "a far-away sphere inflates the bound", "tiny scene coordinates"])
def test_fused_slab_walk_in_and_out_of_its_domain(trt, orc, case):
    """rt_path.h box_loop_compact's fused slab arithmetic (round 5) is conservative for ray origins with |o| <= 4 B per axis (B = the tree's largest
    |coordinate| on the axis); rays that start further out walk the reference tree.  Frames and ray counts equal the oracle's on both sides of that
    border, with the bound blown up by one far-away sphere (every box then grows by 2^-19 B = 0.19), and on a scene a thousand times smaller."""
    desc = trt.scenes.sphere_grid(3000, 96, 54)                        # walked from global memory: 16-byte nodes
    cam = dict(desc["camera"])
    geos = list(desc["geometries"])
    if case == "camera on the domain's edge":
        cam["position"] = (3999.0, 40.0, 20.0)                         # B_x = 1000 (the ground sphere): |o.x| just inside 4 B
        cam["focus_distance"] = 4000.0
    elif case == "camera far outside":
        cam["position"] = (9000.0, 800.0, 7000.0)                      # primary rays leave the domain: reference-tree walk; bounces are inside it
        cam["focus_distance"] = 11000.0
        cam["vertical_fov"] = 1.0
    elif case == "a far-away sphere inflates the bound":
        geos.append(("sphere", (1.0e5, 50.0, -3.0e4), 10.0, desc["materials"][3][0]))
    elif case == "tiny scene coordinates":
        k = 1.0e-3
        geos = [(g[0], tuple(c * k for c in g[1]), g[2] * k, g[3]) for g in geos]
        cam["position"] = tuple(c * k for c in cam["position"])
        cam["focus_distance"] = cam["focus_distance"] * k
    d2 = dict(desc, geometries=geos, camera=cam)
    ow, ocam = orc.world_from_description(d2)
    cpu, cst = orc.render(ow, ocam, 3, 50, d2["background"], seed=11, nthreads=8)
    pw, pcam = trt.world_from_description(d2)
    assert pw.get_bvh().info()["lds_bytes"] == 0
    r = trt.Renderer(3, 1, 50, False, d2["background"], seed=11)
    if not any(k.startswith("TRT_") and k not in ("TRT_LIB_PATH",) for k in os.environ):
        assert r.launch_plan(pcam, pw.get_bvh())["walk"] == 3             # (default tuning: the 16-byte-node walk; tools/test_knobs.sh runs this under others)
    gpu = r.render(pcam, pw).data
    assert_bit_equal(gpu, cpu, case)
    assert r.last_stats["rays"] == cst["rays"]
    for tuning in ({"runtime_walk": 1}, {"dual_walk": 1}):            # the general kernels and the two-paths-per-lane kernel run the same loop: same domain
        r2 = trt.Renderer(3, 1, 50, False, d2["background"], seed=11)
        r2.tuning = tuning
        assert_bit_equal(r2.render(pcam, pw).data, cpu, f"{case} {tuning}")
    if case != "tiny scene coordinates":
        assert gpu.mean() > 0.05                                       # (the frame shows something)


@pytest.mark.parametrize("wh", [(2, 2), (17, 5), (33, 47), (64, 16), (130, 3)])
def test_ragged_image_sizes(trt, orc, wh):
    desc = trt.scenes.cornell(*wh)
    gpu, gst, cpu, cst = render_both(trt, orc, desc, 3, 6)
    assert_bit_equal(gpu, cpu, f"image {wh}")
    assert gst["samples"] == wh[0] * wh[1] * 3 == cst["samples"]


def test_bounce_budgets(trt, orc):
    desc = trt.scenes.cornell(32, 32)
    for depth in (1, 2, 3):
        gpu, _, cpu, _ = render_both(trt, orc, desc, 4, depth, stats=False)
        assert_bit_equal(gpu, cpu, f"depth {depth}")
    # max_bounces = 0: the loop never runs (cpu.rs:47), colour 0
    pw, pcam = trt.world_from_description(desc)
    img = trt.Renderer(4, 1, 0, False, desc["background"], backend=MEGAKERNEL).render(pcam, pw)
    assert not img.data.any()


def test_progressive_passes_equal_one_pass(trt, orc):
    """Samples [0,3) then [3,8) continuing the same sums == samples [0,8) in one launch == oracle."""
    desc = trt.scenes.cornell(64, 48)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(8, 1, 10, False, desc["background"], backend=MEGAKERNEL)
    one = r.render(pcam, pw).data
    a = r.render(pcam, pw, sample_begin=0, sample_end=3).data
    b = r.render(pcam, pw, accum=a.copy(), sample_begin=3, sample_end=8, accumulate=1).data
    assert_bit_equal(b, one, "progressive")
    ow, ocam = orc.world_from_description(desc)
    cpu, _ = orc.render(ow, ocam, 8, 10, desc["background"], nthreads=8)
    assert_bit_equal(one, cpu, "progressive vs oracle")


@pytest.mark.parametrize("world_size", [2, 3, 8])
def test_row_bands_assemble_to_the_full_frame(trt, orc, world_size):
    """Image tiles across GPUs: each rank's bands, rendered separately, are the rows of the full frame bit for bit."""
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    desc = trt.scenes.cornell(40, 70)                   # 70 rows: ragged last band
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(4, 1, 8, False, desc["background"], backend=MEGAKERNEL)
    full = r.render(pcam, pw).data
    out = np.zeros_like(full)
    for rank in range(world_size):
        lay = tiles.band_layout(70, world_size, rank, 16)
        if lay["rows_local"] == 0:
            continue
        part = r.render(pcam, pw, band_rows=16, band_stride=world_size, band_offset=rank, rows_local=lay["rows_local"]).data
        out[lay["rows"]] = part
    assert_bit_equal(out, full, f"bands ws={world_size}")


def test_sample_batch_liveness_kat_and_parity(trt, orc, golden):
    """trait Sampler in batch form; the reference's test_dummy_sampling (cpu.rs:131-180) + parity of the colours."""
    g = golden["sampler_liveness"]
    desc = trt.scenes.dummy_spheres("sampler")
    pw, _ = trt.world_from_description(desc)
    ow, _ = orc.world_from_description(desc)
    n = g["num_samples"]
    gp, op = (trt.SamplePoint * n)(), (orc.SamplePoint * n)()
    for i in range(n):
        t = np.float32(i) / np.float32(n)
        x = np.float32(-1.0) * (np.float32(1.0) - t) + np.float32(1.0) * t
        ray = orc.lib.orc_ray_new(orc.Vec3(x, 0.0, 0.0), orc.Vec3(0.0, 0.0, -1.0))
        op[i].x, op[i].y, op[i].ray = i, 0, ray
        C.memmove(C.byref(gp[i]), C.byref(op[i]), 32)
    for depth, bg in ((g["max_bounces"], (0, 0, 0)), (12, (0.7, 0.8, 1.0))):
        gout, gst = trt.sample_batch(pw.get_bvh(), gp, depth, bg, seed=5)
        oout, ost = orc.sample_batch(ow, op, depth, bg, seed=5)
        fast, _ = trt.sample_batch(pw.get_bvh(), gp, depth, bg, seed=5, collect_stats=False)       # production walk, no counters
        assert bytes(fast) == bytes(gout)
        assert [gout[i].x for i in range(n)] == list(range(n))            # every sample answered (liveness)
        assert bytes(gout) == bytes(oout)
        for k in STAT_KEYS:
            assert gst[k] == ost[k], k


def test_degenerate_rays_take_the_exact_slab_path(trt, orc):
    """Axis-parallel rays starting on box planes (0 * inf = NaN in aabb.rs:45-46), zero and NaN directions."""
    desc = trt.scenes.cornell(8, 8)
    pw, _ = trt.world_from_description(desc)
    ow, _ = orc.world_from_description(desc)
    nan, inf = float("nan"), float("inf")
    rays = []
    for o in [(0.0, 50.0, 50.0), (100.0, 50.0, 50.0), (50.0, 0.0, 50.0), (50.0, 100.0, 50.0), (25.0, 30.0, 50.0),
              (55.0, 60.0, 80.0), (50.0, 50.0, -140.0), (45.0, 30.0, 10.0), (50.0, 99.99995, 50.0)]:
        for d in [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0.0, 0.6, 0.8), (0.6, 0.0, -0.8),
                  (-0.0, 1.0, 0.0), (0.0, -0.0, 1.0), (1e-39, 1.0, 0.0)]:
            rays.append((o, d))
    rays += [((50, 50, 50), (nan, nan, nan)), ((50, 50, 50), (0, 0, 0)), ((50, 50, 50), (inf, 0, 0)),
             ((nan, 50, 50), (0, 0, 1)), ((1e38, 50, 50), (-1, 0, 0))]
    n = len(rays)
    gp, op = (trt.SamplePoint * n)(), (orc.SamplePoint * n)()
    for i, (o, d) in enumerate(rays):
        op[i].x, op[i].y = i, 1
        op[i].ray = orc.Ray(orc.Vec3(*o), orc.Vec3(*d))                  # used as given (not re-normalised)
        C.memmove(C.byref(gp[i]), C.byref(op[i]), 32)
    gout, gst = trt.sample_batch(pw.get_bvh(), gp, 6, (0.1, 0.2, 0.3), seed=9)
    oout, ost = orc.sample_batch(ow, op, 6, (0.1, 0.2, 0.3), seed=9)
    assert bytes(gout) == bytes(oout)
    for k in STAT_KEYS:
        assert gst[k] == ost[k], k


@pytest.mark.parametrize("scene", ["cornell", "random_spheres", "sphere_grid"])
def test_nan_rays_hit_nothing_without_walking(trt, orc, scene):
    """A NaN anywhere in a ray's origin or direction: the reference walks every box the NaN lets through and finds nothing (every primitive
    test fails on a NaN t: sphere.rs:40,42, quad.rs:37).  The production walk answers "miss" at once (rt_path.h ray_has_nan; round 5: such
    rays - one Lambertian scatter in 2^23 - each held a wave for a whole-tree walk); the counting walk still performs the reference's tests.
    All three give the same colours; the counting walk's counters equal the oracle's."""
    desc = {"cornell": lambda: trt.scenes.cornell(8, 8), "random_spheres": lambda: trt.scenes.random_spheres(8, 8),
            "sphere_grid": lambda: trt.scenes.sphere_grid(3000, 8, 8)}[scene]()
    pw, _ = trt.world_from_description(desc)
    ow, _ = orc.world_from_description(desc)
    nan = float("nan")
    o0 = desc["camera"]["position"]
    good_d = (-0.6, -0.3, -0.74) if scene != "cornell" else (0.0, 0.0, 1.0)
    rays = [(o0, good_d)]                                                  # a sane ray among them
    for k in range(1, 8):                                                  # every non-empty subset of NaN direction components
        rays.append((o0, tuple(nan if k >> a & 1 else good_d[a] for a in range(3))))
    for k in range(1, 8):                                                  # ... and of NaN origin components
        rays.append((tuple(nan if k >> a & 1 else o0[a] for a in range(3)), good_d))
    rays.append(((nan, nan, nan), (nan, nan, nan)))
    rays.append((o0, (-nan, 0.0, 1.0)))
    n = len(rays)
    gp, op = (trt.SamplePoint * n)(), (orc.SamplePoint * n)()
    for i, (o, d) in enumerate(rays):
        op[i].x, op[i].y = i, 2
        op[i].ray = orc.Ray(orc.Vec3(*o), orc.Vec3(*d))                  # used as given (not re-normalised)
        C.memmove(C.byref(gp[i]), C.byref(op[i]), 32)
    bg = (0.1, 0.2, 0.3)
    counted, gst = trt.sample_batch(pw.get_bvh(), gp, 6, bg, seed=9)
    fast, _ = trt.sample_batch(pw.get_bvh(), gp, 6, bg, seed=9, collect_stats=False)
    oout, ost = orc.sample_batch(ow, op, 6, bg, seed=9)
    assert bytes(counted) == bytes(oout) and bytes(fast) == bytes(oout)
    for k in STAT_KEYS:
        assert gst[k] == ost[k], k
    for i in range(1, n):                                                  # a NaN ray sees the background, unattenuated
        assert oout[i].color.tolist() == [np.float32(c) for c in bg], i


def test_sample_batch_empty_and_zero_budget(trt):
    pw, _ = trt.world_from_description(trt.scenes.cornell(8, 8))
    out, st = trt.sample_batch(pw.get_bvh(), (trt.SamplePoint * 0)(), 4, (0, 0, 0))
    assert st["samples"] == 0
    pts = (trt.SamplePoint * 2)()
    pts[1].x = 7
    out, st = trt.sample_batch(pw.get_bvh(), pts, 0, (1, 1, 1))
    assert out[1].x == 7 and out[1].color.tolist() == [0.0, 0.0, 0.0]


def test_render_on_device_buffers(trt, orc):
    """trt_render_device: accumulators and counters stay in HBM (torch owns the memory and the stream)."""
    import torch
    desc = trt.scenes.cornell(64, 64)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(6, 1, 8, False, desc["background"], backend=MEGAKERNEL)
    dev = torch.device("cuda:0")
    acc = torch.zeros((64, 64, 3), dtype=torch.float32, device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=0, sample_end=2)
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=2, sample_end=6,
                        accumulate=1)
    stream.synchronize()
    ow, ocam = orc.world_from_description(desc)
    cpu, st = orc.render(ow, ocam, 6, 8, desc["background"], nthreads=8)
    assert_bit_equal(acc.cpu().numpy(), cpu, "device buffers")
    assert int(ctr[0]) == st["samples"] and int(ctr[1]) == st["rays"]


def test_full_size_properties_cornell_2048(trt):
    """BASELINE's headline image size (Cornell 2048x2048, depth 50) through size-independent properties: sample and ray
    accounting, determinism, progressive passes == one pass, a band render == the same rows of the full frame."""
    desc = trt.scenes.cornell(2048, 2048)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(4, 1, 50, False, desc["background"], backend=MEGAKERNEL)
    full = r.render(pcam, pw).data
    st = r.last_stats
    assert st["samples"] == 2048 * 2048 * 4
    assert st["samples"] <= st["rays"] <= 50 * st["samples"]
    assert not np.isnan(full).any() and full.min() >= 0
    again = r.render(pcam, pw).data
    assert_bit_equal(again, full, "determinism")
    a = r.render(pcam, pw, sample_begin=0, sample_end=1).data
    b = r.render(pcam, pw, accum=a, sample_begin=1, sample_end=4, accumulate=1).data
    assert_bit_equal(b, full, "progressive 2048")
    band = r.render(pcam, pw, band_rows=16, band_stride=8, band_offset=3, rows_local=256).data
    rows = [((q // 16) * 8 + 3) * 16 + q % 16 for q in range(256)]
    assert_bit_equal(band, full[rows], "band 3 of 8")
    # the frame's mean radiance agrees with a low-resolution render of the same scene (law of large numbers)
    small = trt.Renderer(64, 1, 50, False, desc["background"], backend=MEGAKERNEL).render(*reversed(trt.world_from_description(trt.scenes.cornell(128, 128)))).data
    assert abs(full.mean() / small.mean() - 1.0) < 0.05


def test_cpp_mirror_renders_the_same_frame(trt, tmp_path):
    """examples/cornell.cpp (the reference binary src/main.rs written against include/tinyrt.hpp) produces the same
    quantised frame as the Python mirror: both are thin callers of one C ABI."""
    import subprocess
    from test_host_boundary import _build_cpp_example
    exe = _build_cpp_example(tmp_path)
    r = subprocess.run([exe, "72", "56", "5"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    from PIL import Image as PILImage
    got = np.asarray(PILImage.open(tmp_path / "output.png").convert("RGB"))          # src/main.rs:20 saves a PNG
    desc = trt.scenes.cornell(72, 56)
    pw, pcam = trt.world_from_description(desc)
    img = trt.Renderer(5, 8, 20, True, (0.001, 0.001, 0.001)).render(pcam, pw)
    assert got.shape == (56, 72, 3) and np.array_equal(got, img.to_u8())
    # the same binary over three shards (tinyrt::Renderer::render_multi -> trt_render_multi): the same PNG
    r = subprocess.run([exe, "72", "56", "5", "3"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    again = np.asarray(PILImage.open(tmp_path / "output.png").convert("RGB"))
    assert np.array_equal(again, got)


def test_default_backend_is_auto_and_matches_the_oracle(trt, orc):
    """Renderer's default (TRT_BACKEND_AUTO) picks the fastest measured backend; same bits as every other one."""
    desc = trt.scenes.cornell(80, 60)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(6, 1, 12, False, desc["background"])
    assert r.backend == trt.BACKEND_AUTO
    gpu = r.render(pcam, pw).data
    ow, ocam = orc.world_from_description(desc)
    cpu, st = orc.render(ow, ocam, 6, 12, desc["background"], nthreads=8)
    assert_bit_equal(gpu, cpu, "auto backend")
    assert r.last_stats["rays"] == st["rays"]
