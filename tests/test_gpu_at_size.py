"""BASELINE configs[1], configs[2] and configs[3] at FULL image size on the DEFAULT backend: Cornell 800x800 and
2048x2048 and the random-spheres scene at 1920x1080, depth 50, through the streamed backend's production plans (lock-step leaf list + ray pool; 768-lane
workgroups with the LDS leaf stack) - the shapes `bench.py --scene ...` measures.  In the form of tests/test_gpu_cfg5.py:

 * oracle parity, bit for bit, on the whole frame at a few spp (the oracle needs a second or two) and on 16-row bands rendered
   AS bands (top / middle / ragged bottom), with the counting kernel's counters equal to the oracle's;
 * full-frame properties at the config's depth: accounting (every pixel gets every sample exactly once), determinism,
   progressive passes == one pass, multi-shard render == one render, and the plan that ran is the one the bench measures."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEPTH = 50
CASES = {
    "cfg2_cornell_800": dict(scene="cornell", w=800, h=800, plan=dict(walk=2, threads_per_workgroup=256, ray_pool=1, specialised=1)),
    "cfg3_random_spheres_1080p": dict(scene="random_spheres", w=1920, h=1080, plan=dict(walk=1, threads_per_workgroup=768, ray_pool=0, specialised=1)),
    # BASELINE configs[3] at its own image size (round 4): the frame bench.py times, against the oracle pixel for pixel
    "cfg4_cornell_2048": dict(scene="cornell", w=2048, h=2048, plan=dict(walk=2, threads_per_workgroup=256, ray_pool=1, specialised=1)),
}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def describe(trt, case):
    c = CASES[case]
    return getattr(trt.scenes, c["scene"])(c["w"], c["h"]), c


@pytest.mark.parametrize("case", sorted(CASES))
def test_at_size_full_frame_bit_exact_against_the_oracle(trt, orc, case):
    desc, c = describe(trt, case)
    spp = 2
    ow, ocam = orc.world_from_description(desc)
    cpu, cst = orc.render(ow, ocam, spp, DEPTH, desc["background"], seed=5, nthreads=16)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, DEPTH, False, desc["background"], seed=5)            # backend: the library's default
    plan = trt._lib.LaunchPlan()
    trt._lib.check(trt.lib.trt_streamed_launch_plan(pw.get_bvh()._h, C.byref(pcam.pod), C.byref(r.params()), C.byref(plan)))
    for k, v in c["plan"].items():
        assert getattr(plan, k) == v, (case, k, plan.as_dict())                 # the production plan bench.py measures
    got = r.render(pcam, pw).data
    assert r.last_stats["samples"] == c["w"] * c["h"] * spp and r.last_stats["rays"] == cst["rays"]
    assert np.array_equal(bits(got), bits(cpu)), f"{case}: default backend differs from the oracle on the full frame"
    counted = r.render(pcam, pw, collect_stats=True)
    assert np.array_equal(bits(counted.data), bits(cpu))
    for k in ("samples", "rays", "node_tests", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
        assert r.last_stats[k] == cst[k], (case, k)


@pytest.mark.parametrize("case", sorted(CASES))
def test_at_size_bands_rendered_as_bands(trt, orc, case):
    desc, c = describe(trt, case)
    pw, pcam = trt.world_from_description(desc)
    ow, ocam = orc.world_from_description(desc)
    r = trt.Renderer(4, 1, DEPTH, False, desc["background"], seed=11)
    n_bands = (c["h"] + 15) // 16                                               # 1080 rows: 67 full bands + one of 8 rows
    for band in (0, n_bands // 2, n_bands - 1):
        y0, y1 = 16 * band, min(16 * band + 16, c["h"])
        cpu, cst = orc.render(ow, ocam, 4, DEPTH, desc["background"], seed=11, nthreads=16, row_begin=y0, row_end=y1)
        got = r.render(pcam, pw, band_rows=16, band_stride=n_bands, band_offset=band, rows_local=y1 - y0)
        assert np.array_equal(bits(got.data), bits(cpu[y0:y1])), (case, band)
        assert r.last_stats["rays"] == cst["rays"], (case, band)


@pytest.mark.parametrize("case", sorted(CASES))
def test_at_size_frame_properties(trt, case):
    desc, c = describe(trt, case)
    W, H, spp = c["w"], c["h"], 16
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, DEPTH, False, desc["background"], seed=3)
    whole = r.render(pcam, pw).data
    st = dict(r.last_stats)
    assert st["samples"] == W * H * spp and spp * W * H <= st["rays"] <= DEPTH * spp * W * H      # accounting
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and whole.mean() > 0.01
    again = r.render(pcam, pw).data                                                              # determinism
    assert np.array_equal(bits(again), bits(whole)) and r.last_stats["rays"] == st["rays"]
    acc = np.zeros((H, W, 3), np.float32)                                                        # progressive == one pass
    r.render(pcam, pw, accum=acc, sample_begin=0, sample_end=5)
    r.render(pcam, pw, accum=acc, sample_begin=5, sample_end=6, accumulate=1)
    r.render(pcam, pw, accum=acc, sample_begin=6, sample_end=16, accumulate=1)
    assert np.array_equal(bits(acc), bits(whole))
    for devices in ([0, 0], [0, 0, 0, 0, 0]):                                                    # multi-shard == one render
        sharded = r.render_multi(pcam, pw, devices=devices)
        assert np.array_equal(bits(sharded.data), bits(whole)) and r.last_stats["rays"] == st["rays"], devices
    mega = trt.Renderer(spp, 1, DEPTH, False, desc["background"], seed=3, backend=trt.BACKEND_MEGAKERNEL).render(pcam, pw).data
    assert np.array_equal(bits(mega), bits(whole))                                               # another backend, same frame


SCHEDULES = {
    # (scene at bench size, spp) -> variants whose frames must be the default's, bit for bit: trt_tuning fields (per render) or, under
    # "scene", trt_scene_options fields (per scene compilation).  Each changes scheduling / placement only: the hand-written box-step loops
    # with and without the resumable-walk exit, other leaf-stack depths, the other workgroup size, the register-slot walk (a C++ loop), the
    # 32-byte-node walk, other wave budgets, two paths per lane.
    "random_spheres_1080p": (lambda trt: trt.scenes.random_spheres(1920, 1080), 32,
                             [{"lds_stragglers": 0}, {"lds_stragglers": 24}, {"leaf_slots": 3}, {"stream_big_threads": 512},
                              {"lds_leaf_stack": 0}, {"stream_waves_per_simd": 5}, {"runtime_walk": 1}]),
    "sphere_grid100k_2160p": (lambda trt: trt.scenes.sphere_grid(100000, 3840, 2160), 4,
                              [{"stragglers": 0}, {"stragglers": 40}, {"leaf_slots": 2}, {"scene": {"compact_nodes": 0}},
                               {"stream_waves_per_simd": 6}, {"ray_pool": 0}, {"dual_walk": 1}, {"dual_walk": 1, "stream_waves_per_simd": 6}]),
}


@pytest.mark.parametrize("case", sorted(SCHEDULES))
def test_at_size_schedules_agree(trt, case):
    """Round 3: the box-step loops of the two tree walks are hand-written assembly and both walks are resumable.  At the size and depth
    bench.py measures (1.8e8 / 2.0e8 rays per render), every scheduling variant gives the default's frame and ray count."""
    import torch
    dev = torch.device("cuda:0")
    mk, spp, variants = SCHEDULES[case]
    desc = mk(trt)
    stream = torch.cuda.current_stream()
    ref = ref_rays = None
    for env in [{}] + variants:
        knobs = {k: v for k, v in env.items() if k != "scene"}
        pw, pcam = trt.world_from_description(desc, **env.get("scene", {}))     # scene compiled with the variant's options
        W, H = pcam.get_image_size()
        r = trt.Renderer(spp, 1, DEPTH, False, desc["background"], seed=1)
        r.tuning = knobs
        acc = torch.zeros((H, W, 3), device=dev)
        ctr = torch.zeros(16, dtype=torch.int64, device=dev)
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr())
        torch.cuda.synchronize()
        rays = int(ctr[1].item())
        if ref is None:
            ref, ref_rays = acc, rays
            assert rays > 1.5e8 and int(ctr[0].item()) == W * H * spp
            continue
        assert rays == ref_rays, (case, env)
        assert torch.equal(acc.view(torch.int32), ref.view(torch.int32)), (case, env)


FULL_SPP = {
    # BASELINE configs[1], configs[2] and configs[4] at their FULL sample counts, once each: (scene, W, H, spp, split) - `split` lies inside
    # the first launch's sample range, so the two progressive passes cut the frame's launches (512 spp = 2 x 256 at 1080p, 256 spp = 2 x 128
    # at 2160p: streamed.hip streamed_chunk_spp) at other places than the one-pass render does
    # configs[1] too (the one BASELINE sample count that had never been rendered in full under test): 1024 spp = 4 x 256 at 800x800
    "cfg2_cornell_800_1024spp": (lambda trt: trt.scenes.cornell(800, 800), 1024, 300),
    "cfg3_random_spheres_1080p_512spp": (lambda trt: trt.scenes.random_spheres(1920, 1080), 512, 300),
    "cfg5_sphere_grid100k_2160p_256spp": (lambda trt: trt.scenes.sphere_grid(100000, 3840, 2160), 256, 100),
}


@pytest.mark.parametrize("case", sorted(FULL_SPP))
def test_full_sample_count_streamed_equals_megakernel_and_progressive(trt, case):
    """The whole config, every sample: the default (streamed) backend's chunked launches against the megakernel - one lane per pixel walking
    its samples in order, no radiance buffer, no fold pass - bit for bit with equal ray counts, and two progressive passes that straddle
    the launch boundary against the one-pass render."""
    import torch
    dev = torch.device("cuda:0")
    mk, spp, split = FULL_SPP[case]
    desc = mk(trt)
    pw, pcam = trt.world_from_description(desc)
    W, H = pcam.get_image_size()
    stream = torch.cuda.current_stream()
    chunk = trt.lib.trt_streamed_chunk_spp(W, H)
    assert chunk < spp and split % chunk != 0, "the case no longer crosses a launch boundary"
    frames, rays = {}, {}
    for name, backend, passes in (("streamed", trt.BACKEND_STREAMED, [(0, spp)]), ("streamed, two passes", trt.BACKEND_STREAMED, [(0, split), (split, spp)]),
                                  ("megakernel", trt.BACKEND_MEGAKERNEL, [(0, spp)])):
        r = trt.Renderer(spp, 1, DEPTH, False, desc["background"], seed=1, backend=backend)
        acc = torch.zeros((H, W, 3), device=dev)
        ctr = torch.zeros(16, dtype=torch.int64, device=dev)
        for k, (s0, s1) in enumerate(passes):
            r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=s0, sample_end=s1, accumulate=int(k > 0))
        torch.cuda.synchronize()
        frames[name], rays[name] = acc, int(ctr[1].item())
        assert int(ctr[0].item()) == W * H * spp, name
    assert rays["streamed"] == rays["megakernel"] == rays["streamed, two passes"] and rays["streamed"] > W * H * spp
    for name in ("streamed, two passes", "megakernel"):
        assert torch.equal(frames[name].view(torch.int32), frames["streamed"].view(torch.int32)), (case, name)
    assert torch.isfinite(frames["streamed"]).all() and float(frames["streamed"].mean()) > 0.01
