"""The scene beyond the caches: scenes.sphere_field - the deep-BVH scene of BASELINE configs[4] at a million spheres (203 MB packed, hot part
~100 MB: beyond the chip's 32 MiB of L2, where the launch plan takes two paths per lane) - under test, in the shape of test_gpu_cfg5.py:

 * 16-row bands at 2 spp rendered AS bands against the oracle, bit for bit, with the ray count (top / middle / bottom of a 1920x1080 frame);
 * the counting kernel's node / sphere / shade counters against the oracle's on those bands (reference-order walk of the 2 000 001-node tree);
 * the whole frame: default plan (two paths per lane, 6 waves) == one path per lane (dual_walk = 2: 8 waves) == the megakernel, frames and ray counts;
   progressive passes == one pass.
bench.py reports the 4 M-sphere version (809 MB) in `other_scenes`; its frames go through the same kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, DEPTH, N = 1920, 1080, 50, 1_000_000


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def field(trt):
    return trt.scenes.sphere_field(N, W, H)


def test_field_plan_takes_two_paths_per_lane_beyond_l2(trt, field):
    import os
    if any(k.startswith("TRT_") and k not in ("TRT_LIB_PATH", "TRT_BENCH_REHEARSAL") for k in os.environ):
        pytest.skip("TRT_* variables are set: the library's default tuning was overridden at load (tools/test_knobs.sh)")
    pw, pcam = trt.world_from_description(field)
    info = pw.get_bvh().info()
    assert info["num_spheres"] == N + 1 and info["num_nodes"] == 2 * N + 1 and info["lds_bytes"] == 0 and info["device_bytes"] > 150e6
    r = trt.Renderer(4, 1, DEPTH, False, field["background"])
    pl = r.launch_plan(pcam, pw.get_bvh())
    assert (pl["walk"], pl["dual_walk"], pl["waves_per_simd"], pl["specialised"]) == (3, 1, 6, 1)
    r.tuning = {"dual_walk": 2}
    pl = r.launch_plan(pcam, pw.get_bvh())
    assert (pl["walk"], pl["dual_walk"], pl["waves_per_simd"], pl["specialised"]) == (3, 0, 8, 1)
    small, scam = trt.world_from_description(trt.scenes.sphere_grid(4000, 64, 48))                 # fits L2: one path per lane at 7 waves
    pl = trt.Renderer(4, 1, DEPTH, False, field["background"]).launch_plan(scam, small.get_bvh())
    assert (pl["walk"], pl["dual_walk"], pl["waves_per_simd"]) == (3, 0, 7)


def test_field_bands_and_counters_against_the_oracle(trt, orc, field):
    desc = field
    pw, pcam = trt.world_from_description(desc)
    ow, ocam = orc.world_from_description(desc)
    r = trt.Renderer(2, 1, DEPTH, False, desc["background"], seed=7)
    n_bands = (H + 15) // 16
    for band in (0, n_bands // 2, n_bands - 2):
        cpu, cst = orc.render(ow, ocam, 2, DEPTH, desc["background"], seed=7, nthreads=16, row_begin=16 * band, row_end=16 * band + 16)
        want = cpu[16 * band:16 * band + 16]
        for knobs in ({}, {"dual_walk": 2}):
            r.tuning = knobs
            got = r.render(pcam, pw, band_rows=16, band_stride=n_bands, band_offset=band, rows_local=16)
            assert np.array_equal(bits(got.data), bits(want)), (band, knobs)
            assert r.last_stats["rays"] == cst["rays"], (band, knobs)
        r.tuning = {}
        counted = r.render(pcam, pw, collect_stats=True, band_rows=16, band_stride=n_bands, band_offset=band, rows_local=16)
        assert np.array_equal(bits(counted.data), bits(want)), band
        for k in ("samples", "rays", "node_tests", "sphere_tests", "shades"):
            assert r.last_stats[k] == cst[k], (band, k)


def test_field_whole_frame_agrees_across_plans_backends_and_passes(trt, field):
    desc = field
    pw, pcam = trt.world_from_description(desc)
    frames, rays = {}, {}
    for name, backend, knobs in (("two paths per lane (default plan)", trt.BACKEND_STREAMED, {}), ("one path per lane", trt.BACKEND_STREAMED, {"dual_walk": 2}),
                                 ("megakernel", trt.BACKEND_MEGAKERNEL, {})):
        r = trt.Renderer(3, 1, DEPTH, False, desc["background"], seed=3, backend=backend)
        r.tuning = knobs
        frames[name] = r.render(pcam, pw).data
        rays[name] = r.last_stats["rays"]
        assert r.last_stats["samples"] == W * H * 3
    first = "two paths per lane (default plan)"
    for name in frames:
        assert rays[name] == rays[first] and np.array_equal(bits(frames[name]), bits(frames[first])), name
    assert np.isfinite(frames[first]).all() and frames[first].mean() > 0.1 and rays[first] > 5 * W * H * 3
    r = trt.Renderer(3, 1, DEPTH, False, desc["background"], seed=3)
    acc = np.zeros((H, W, 3), np.float32)
    r.render(pcam, pw, accum=acc, sample_begin=0, sample_end=1)
    r.render(pcam, pw, accum=acc, sample_begin=1, sample_end=3, accumulate=1)
    assert np.array_equal(bits(acc), bits(frames[first]))
