"""The streamed backend's out-of-memory path on the real runtime: with most of HBM held by somebody else the
render asks for shorter launches instead of failing (capi.hip enqueue_render's halving loop), renders the SAME frame, keeps the granted
workspace while the pressure lasts, and is back at one launch per step once the memory is free again.  Until round 5 these lines had only
ever run against the simulated runtime (tests/native/hipstub)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = H = 2048
SPP = 256                     # one full-size launch: 2048 x 2048 x 256 records of 12 bytes = 12.9 GB of scratch
DEPTH = 50


def _launches(trt, fn):
    """Dominant-kernel launches made by fn() (trt_kernel_timing_*: HIP events around every launch of the tracing kernel)."""
    trt._lib.check(trt.lib.trt_kernel_timing_begin())
    fn()
    ms, n = C.c_double(0.0), C.c_uint32(0)
    trt._lib.check(trt.lib.trt_kernel_timing_end(C.byref(ms), C.byref(n)))
    return n.value


def test_render_with_hbm_held_by_torch_halves_its_launches_and_renders_the_same_frame(trt):
    import torch
    dev = torch.device("cuda:0")
    desc = trt.scenes.cornell(W, H)
    world, cam = trt.world_from_description(desc)
    scene = world.get_bvh()                                        # a fresh scene handle: no scratch cached on it yet
    r = trt.Renderer(4096, 1, DEPTH, False, desc["background"], seed=1)
    full_bytes = r.launch_plan(cam, scene, sample_begin=0, sample_end=SPP)["workspace_bytes"]
    assert r.launch_plan(cam, scene, sample_begin=0, sample_end=SPP)["chunk_spp"] == SPP and full_bytes > 12 * 2**30
    stream = torch.cuda.current_stream()
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)

    def render(target, accumulate=0, s0=0):
        r.render_device(cam, scene, target.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=s0, sample_end=s0 + SPP, accumulate=accumulate)
        torch.cuda.synchronize()

    # ---- under pressure: leave room for about 70 % of the full-size scratch (so the first halving, to 128 spp = 6.4 GB, fits) ----
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info()
    hold = torch.empty(int(free_b - 0.7 * full_bytes), dtype=torch.uint8, device=dev)
    try:
        n_first = _launches(trt, lambda: render(acc))
        assert n_first >= 2, "the render was granted its full-size scratch although HBM was held"
        rays_short = int(ctr[1].item())
        # lasting pressure: the granted workspace is kept - the next step makes the same number of launches and no new allocation
        before = torch.cuda.mem_get_info()[0]
        acc2 = torch.zeros_like(acc)                                # (allocated before the measurement: torch's own allocation, not the library's)
        before = torch.cuda.mem_get_info()[0]
        ctr.zero_()
        n_second = _launches(trt, lambda: render(acc2))
        assert n_second == n_first and torch.cuda.mem_get_info()[0] == before
        assert torch.equal(acc2.view(torch.int32), acc.view(torch.int32))
    finally:
        del hold
        torch.cuda.empty_cache()
    # ---- the memory is back: one launch per step again, and the frame of the short launches is the frame of the full-size launch ----
    ctr.zero_()
    ref = torch.zeros_like(acc)
    n_after = _launches(trt, lambda: render(ref))
    assert n_after == 1, "the render did not return to its full-size launch after the memory was freed"
    assert int(ctr[1].item()) == rays_short
    assert torch.equal(ref.view(torch.int32), acc.view(torch.int32)), "short launches rendered another frame"
    # a non-power-of-two grant (a chunk that comes from halving 13 -> 7 had never run on hardware): a progressive pass of 13 samples
    # continued under pressure that fits 7 of them
    trt._lib.check(trt.lib.trt_scene_trim(scene._h))
    torch.cuda.empty_cache()
    small = r.launch_plan(cam, scene, sample_begin=0, sample_end=13)["workspace_bytes"]
    a13 = torch.zeros_like(acc)
    r.render_device(cam, scene, a13.data_ptr(), stream.cuda_stream, 0, sample_begin=0, sample_end=13)
    torch.cuda.synchronize()
    trt._lib.check(trt.lib.trt_scene_trim(scene._h))
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info()
    b13 = torch.zeros_like(acc)
    hold = torch.empty(int(free_b - 0.6 * small), dtype=torch.uint8, device=dev)
    try:
        n13 = _launches(trt, lambda: (r.render_device(cam, scene, b13.data_ptr(), stream.cuda_stream, 0, sample_begin=0, sample_end=13), torch.cuda.synchronize()))
    finally:
        del hold
        torch.cuda.empty_cache()
    assert n13 == 2                                                 # 13 -> 7: launches of 7 and 6 samples
    assert torch.equal(a13.view(torch.int32), b13.view(torch.int32))
