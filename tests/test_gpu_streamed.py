"""Parity of the streamed backend (samples as work items pulled by persistent waves, radiances folded per pixel in sample order) with
the CPU oracle: bit-exact accumulators, identical traversal counters."""
import numpy as np
import pytest
from test_gpu_parity import STAT_KEYS, assert_bit_equal

pytestmark = pytest.mark.gpu
STREAMED = 3


def render(trt, desc, spp, depth, seed=1, stats=True, **over):
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, depth, False, desc["background"], seed=seed, backend=STREAMED)
    img = r.render(pcam, pw, collect_stats=False, **over)
    gst = r.last_stats
    if stats:
        counted = r.render(pcam, pw, collect_stats=True, **over)
        assert_bit_equal(counted.data, img.data, "streamed counting kernel vs production kernel")
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
    assert_bit_equal(gpu, cpu, f"streamed {scene}")
    for k in STAT_KEYS:
        assert gst[k] == cst[k], k


@pytest.mark.parametrize("wh", [(2, 2), (17, 5), (33, 47), (32, 16), (130, 3), (200, 70)])
def test_ragged_image_sizes(trt, orc, wh):
    desc = trt.scenes.cornell(*wh)
    gpu, gst = render(trt, desc, 3, 6)
    cpu, _ = oracle(orc, desc, 3, 6)
    assert_bit_equal(gpu, cpu, f"streamed image {wh}")
    assert gst["samples"] == wh[0] * wh[1] * 3


@pytest.mark.parametrize("minw", [5, 6, 7])
def test_wave_budget_never_changes_the_frame(trt, orc, minw):
    desc = trt.scenes.random_spheres(96, 64)
    gpu, gst = render(trt, desc, 4, 20, tuning={"stream_waves_per_simd": minw})
    cpu, cst = oracle(orc, desc, 4, 20)
    assert_bit_equal(gpu, cpu, f"streamed minw {minw}")
    assert gst["node_tests"] == cst["node_tests"]


def test_progressive_bands_and_backend_agreement(trt, orc):
    from importlib import import_module
    tiles = import_module("tiny-raytracer_amd.tiles")
    desc = trt.scenes.cornell(40, 70)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(8, 1, 10, False, desc["background"], backend=STREAMED)
    one = r.render(pcam, pw).data
    a = r.render(pcam, pw, sample_begin=0, sample_end=3).data
    b = r.render(pcam, pw, accum=a.copy(), sample_begin=3, sample_end=8, accumulate=1).data
    assert_bit_equal(b, one, "streamed progressive")
    assert_bit_equal(one, trt.Renderer(8, 1, 10, False, desc["background"], backend=0).render(pcam, pw).data, "streamed vs megakernel")
    out = np.zeros_like(one)
    for rank in range(3):
        lay = tiles.band_layout(70, 3, rank, 16)
        out[lay["rows"]] = r.render(pcam, pw, band_rows=16, band_stride=3, band_offset=rank, rows_local=lay["rows_local"]).data
    assert_bit_equal(out, one, "streamed bands")
    # more samples than one chunk (256 spp at most): several sample/fold launch pairs continue the same sums
    desc2 = trt.scenes.cornell(24, 20)
    gpu, gst = render(trt, desc2, 300, 6)
    cpu, cst = oracle(orc, desc2, 300, 6)
    assert_bit_equal(gpu, cpu, "streamed 300 spp (2 chunks, ragged last batch)")
    assert gst["samples"] == 24 * 20 * 300 and gst["rays"] == cst["rays"]
    for depth in (1, 2):
        gpu, _ = render(trt, desc, 4, depth, stats=False)
        cpu, _ = oracle(orc, desc, 4, depth)
        assert_bit_equal(gpu, cpu, f"streamed depth {depth}")


def test_full_size_properties_cornell_2048_and_device_buffers(trt):
    """BASELINE's headline image size on the default backend: accounting, determinism, progressive passes == one pass,
    a band == the same rows of the frame, agreement with the megakernel; buffers and counters resident in HBM."""
    import torch
    desc = trt.scenes.cornell(2048, 2048)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(4, 1, 50, False, desc["background"], backend=STREAMED)
    full = r.render(pcam, pw).data
    st = r.last_stats
    assert st["samples"] == 2048 * 2048 * 4 and st["samples"] <= st["rays"] <= 50 * st["samples"]
    assert not np.isnan(full).any() and full.min() >= 0
    assert_bit_equal(r.render(pcam, pw).data, full, "streamed determinism")
    assert_bit_equal(trt.Renderer(4, 1, 50, False, desc["background"], backend=0).render(pcam, pw).data, full, "streamed vs megakernel 2048^2")
    dev = torch.device("cuda:0")
    acc = torch.zeros((2048, 2048, 3), dtype=torch.float32, device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=0, sample_end=1)
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=1, sample_end=4, accumulate=1)
    stream.synchronize()
    assert_bit_equal(acc.cpu().numpy(), full, "streamed progressive on device buffers")
    assert int(ctr[0]) == st["samples"] and int(ctr[1]) == st["rays"]
    band = r.render(pcam, pw, band_rows=16, band_stride=8, band_offset=3, rows_local=256).data
    rows = [((q // 16) * 8 + 3) * 16 + q % 16 for q in range(256)]
    assert_bit_equal(band, full[rows], "streamed band 3 of 8")


def test_frames_match_the_reference_renders_statistically(trt):
    """The product's frames against the PNGs the reference ships (same scene, spp, depth and background as the
    reference source records; tests/golden/reference_png_blocks.json): block means of the quantised frame agree.
    (The oracle passes the same check on CPU; this one goes GPU -> tonemap -> PNG statistics directly.)"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "reference_png_blocks.json")) as f:
        fx = json.load(f)

    def lin_blocks(u8, block):
        lin = ((u8.astype(np.float64) + 0.5) / 255.0) ** 2.2
        h, w, _ = u8.shape
        bh, bw = h // block, w // block
        return lin[: bh * block, : bw * block].reshape(bh, block, bw, block, 3).mean(axis=(1, 3))

    cases = [("cornell", trt.scenes.cornell(300, 300), 300, 20, 0.985), ("quad_test", trt.scenes.quad_test(), 10, 10, 0.999),
             ("render_test", trt.scenes.dummy_spheres("renderer"), 3, 10, 0.998)]
    for key, desc, spp, depth, min_corr in cases:
        f = fx[key]
        ref = np.array(f["mean"])
        ok = ~np.array(f["saturated"]).astype(bool)
        pw, pcam = trt.world_from_description(desc)
        img = trt.Renderer(spp, 8, depth, False, desc["background"], seed=21).render(pcam, pw)     # the reference's own settings
        mine = lin_blocks(img.to_u8(), f["block"])
        assert abs(mine[ok].mean() / ref[ok].mean() - 1.0) < 0.03, key
        assert np.corrcoef(mine[ok].ravel(), ref[ok].ravel())[0, 1] > min_corr, key


@pytest.mark.parametrize("scene", ["random_spheres", "cornell"])
def test_leaf_slots_are_scheduling_only(trt, scene):
    """The postponed-leaf walk (rt_path.h walk_fast / walk_fast_lds) must run exactly the reference's primitive tests:
    frames, ray counts and primitive-test counters are identical for 1 slot (plain while-while), 2, 4 (default) and 8,
    with the slots in registers or in LDS, on scenes large enough for grazing hits to occur (an earlier version that
    tested postponed leaves without re-checking their box against the current t_best differed on random-spheres in
    ~1e-7 of the rays, and only there)."""
    desc = trt.scenes.random_spheres(960, 540) if scene == "random_spheres" else trt.scenes.cornell(512, 512)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(16, 1, 50, False, desc["background"], seed=5, backend=STREAMED)
    ref_img = ref_stats = None
    for slots, lds in ((1, 0), (2, 0), (4, 0), (1, 2), (4, 2), (8, 2), (4, 1)):
        # lds_leaf_stack: 0 registers, 2 LDS, 1 LDS where it costs no occupancy
        img = r.render(pcam, pw, collect_stats=2, tuning={"leaf_slots": slots, "lds_leaf_stack": lds})
        st = r.last_stats
        if ref_img is None:
            ref_img, ref_stats = img.data.copy(), dict(st)
            continue
        assert np.array_equal(img.data.view(np.uint32), ref_img.view(np.uint32)), (slots, lds)
        for k in ("samples", "rays", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
            assert st[k] == ref_stats[k], (slots, lds, k)
    plain = r.render(pcam, pw)                                                      # production (non-counting) kernel, defaults
    assert np.array_equal(plain.data.view(np.uint32), ref_img.view(np.uint32))
    assert r.last_stats["rays"] == ref_stats["rays"]


def test_device_tonemap_is_bit_exact_against_the_oracle(trt, orc):
    """SURVEY 8 f1 on the device: the frame is rendered, gamma-corrected and quantised without leaving HBM.  Byte output:
    the device frame must EQUAL the oracle's (orc_tonemap_u8) - both evaluate trt-math v2's powf (trt_pow.h / rt_oracle.c
    m_powf), as does the host form."""
    import torch
    dev = torch.device("cuda:0")
    desc = trt.scenes.cornell(301, 203)                                    # odd sizes: the 4-channel vector path has a tail
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(16, 1, 12, False, desc["background"], seed=3)
    acc = torch.zeros((203, 301, 3), device=dev)
    stream = torch.cuda.current_stream()
    r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream)
    rgb = torch.zeros((203, 301, 3), dtype=torch.uint8, device=dev)
    for gamma in (2.2, 1.0, 1.8, 0.45):
        trt.tonemap_u8_device(acc.data_ptr(), 203 * 301, rgb.data_ptr(), gamma, stream.cuda_stream)
        torch.cuda.synchronize()
        want = orc.tonemap_u8(acc.cpu().numpy(), gamma)
        assert np.array_equal(rgb.cpu().numpy(), want), gamma
        assert np.array_equal(trt.Image(acc.cpu().numpy(), gamma).to_u8(), want), gamma
    # a dense sweep of channel values incl. every special case, through the same kernel: 3 M floats
    rng = np.random.default_rng(11)
    vals = np.concatenate([rng.random(1_000_000).astype(np.float32), np.exp(rng.uniform(-100, 90, 1_000_000)).astype(np.float32),
                           rng.integers(0, 2**32, 999_000, dtype=np.uint32).view(np.float32),            # any bit pattern
                           np.resize(np.array([np.nan, -1.0, 0.0, -0.0, 1e-9, 0.5, 0.999, 1.0, 7.0, np.inf, -np.inf, 1e-45, 3.4e38], np.float32), 1000)])
    d_vals = torch.from_numpy(vals).to(dev)
    d_out = torch.zeros(vals.size, dtype=torch.uint8, device=dev)
    trt.tonemap_u8_device(d_vals.data_ptr(), vals.size // 3, d_out.data_ptr(), 2.2, stream.cuda_stream)
    torch.cuda.synchronize()
    with np.errstate(all="ignore"):
        assert np.array_equal(d_out.cpu().numpy(), orc.tonemap_u8(vals.reshape(1, -1, 3), 2.2).ravel())
    # unaligned buffers (scalar path): NaN -> 0, negative -> 0, >= 1 -> 254, inf -> 254
    special = torch.tensor([float("nan"), -1.0, 0.0, 1e-9, 0.5, 0.999, 1.0, 7.0, float("inf"), 0.25, 0.125, 0.73], device=dev)
    buf = torch.zeros(16, device=dev)
    buf[1:13] = special
    out = torch.zeros(16, dtype=torch.uint8, device=dev)
    trt.tonemap_u8_device(buf.data_ptr() + 4, 4, out.data_ptr() + 1, 2.2, stream.cuda_stream)
    torch.cuda.synchronize()
    ref = orc.tonemap_u8(special.cpu().numpy().reshape(1, 4, 3), 2.2).ravel()
    assert np.array_equal(out.cpu().numpy()[1:13], ref)
    assert list(out.cpu().numpy()[[1, 2, 3, 7, 8, 9]]) == [0, 0, 0, 254, 254, 254]


def test_lockstep_leaf_list_is_scheduling_only(trt, orc):
    """Scenes with at most 32 primitives are walked as a lock-step leaf list (rt_path.h walk_flat, wave-uniform nodes
    from the scalar cache) instead of through the culling tree: same frame, same primitive tests, as the tree walk,
    the plain one-slot walk and the oracle."""
    desc = trt.scenes.cornell(320, 320)
    ow, ocam = orc.world_from_description(desc)
    cpu, cst = orc.render(ow, ocam, 8, 50, desc["background"], seed=9, nthreads=8)
    results = []
    for flat, slots in ((0, 1), (1, 0), (1, 2), (1, 12), (0, 0)):
        pw, pcam = trt.world_from_description(desc, flat_walk=flat)                 # trt_scene_options: read when the scene is compiled
        r = trt.Renderer(8, 1, 50, False, desc["background"], seed=9, backend=STREAMED)
        r.tuning = {"leaf_slots": slots}                                            # 0 = by launch plan
        img = r.render(pcam, pw, collect_stats=2)
        results.append((flat, slots, img.data.copy(), dict(r.last_stats)))
        plain = r.render(pcam, pw)                                                  # production kernel
        assert_bit_equal(plain.data, cpu, f"flat {flat} slots {slots} vs oracle")
    for flat, slots, data, st in results:
        assert_bit_equal(data, cpu, f"counting kernel, flat {flat} slots {slots}")
        for k in ("samples", "rays", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
            assert st[k] == cst[k], (flat, slots, k)
    tree, flat_default = results[4][3], results[1][3]
    # every ray steps every leaf box (but for the handful of NaN-prone rays that take the reference tree)
    assert abs(flat_default["node_tests"] - 18 * flat_default["rays"]) < 1e-4 * flat_default["node_tests"]
    assert tree["node_tests"] < flat_default["node_tests"]


def test_full_size_schedules_agree_cornell_2048(trt):
    """BASELINE's full frame size, 64 spp (1.9e9 rays per render): the shipped schedule (lock-step leaf list, 6 LDS
    slots), the culling-tree walk with LDS slots, the plain one-slot tree walk and the megakernel give the same frame
    and the same ray count - every scheduling layer added on top of the reference's walk is checked at full size."""
    import torch
    dev = torch.device("cuda:0")
    desc = trt.scenes.cornell(2048, 2048)
    stream = torch.cuda.current_stream()
    ref = ref_rays = None
    for backend, flat, slots, lds in ((STREAMED, 1, None, None), (STREAMED, 0, None, None), (STREAMED, 0, 1, 0), (0, 0, 1, 0)):
        pw, pcam = trt.world_from_description(desc, flat_walk=flat)
        r = trt.Renderer(64, 1, 50, False, desc["background"], seed=1, backend=backend)
        r.tuning = {k: v for k, v in (("leaf_slots", slots), ("lds_leaf_stack", lds)) if v is not None}
        acc = torch.zeros((2048, 2048, 3), device=dev)
        ctr = torch.zeros(16, dtype=torch.int64, device=dev)
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr())
        torch.cuda.synchronize()
        rays = int(ctr[1].item())
        if ref is None:
            ref, ref_rays = acc, rays
            assert rays > 1.8e9
            continue
        assert rays == ref_rays, (backend, flat, slots, lds)
        assert torch.equal(acc.view(torch.int32), ref.view(torch.int32)), (backend, flat, slots, lds)


def test_full_baseline_config_bit_identical_across_backends(trt):
    """BASELINE configs[3] in full - Cornell 2048x2048, 4096 spp, depth 50, 1.2e11 rays - rendered by the shipped schedule
    (streamed, lock-step leaf list, LDS slots) and by the megakernel with the plain one-slot tree walk: the two frames
    are bit-identical and trace the same number of rays."""
    import torch
    dev = torch.device("cuda:0")
    desc = trt.scenes.cornell(2048, 2048)
    stream = torch.cuda.current_stream()
    frames, rays = [], []
    for backend, knobs, options in ((STREAMED, {}, {}), (0, {"leaf_slots": 1}, {"flat_walk": 0})):
        pw, pcam = trt.world_from_description(desc, **options)
        r = trt.Renderer(4096, 1, 50, False, desc["background"], seed=1, backend=backend)
        r.tuning = knobs
        acc = torch.zeros((2048, 2048, 3), device=dev)
        ctr = torch.zeros(16, dtype=torch.int64, device=dev)
        r.render_device(pcam, pw.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr())
        torch.cuda.synchronize()
        frames.append(acc)
        rays.append(int(ctr[1].item()))
    assert rays[0] == rays[1] and rays[0] > 1.2e11
    assert torch.equal(frames[0].view(torch.int32), frames[1].view(torch.int32))
    assert torch.isfinite(frames[0]).all() and 0.05 < float(frames[0].mean()) < 5.0


@pytest.mark.parametrize("scene", ["cornell", "random_spheres", "grid4000"])
def test_out_of_range_tunings_still_render_the_same_frame(trt, scene):
    """trt_tuning is caller data: values outside every sensible range (ABI v3 took them out of the environment, where nobody typed 10^6) are
    clamped by the library, never trusted by a launch - same frame, same ray count, no fault."""
    desc = {"cornell": lambda: trt.scenes.cornell(160, 120), "random_spheres": lambda: trt.scenes.random_spheres(160, 90),
            "grid4000": lambda: trt.scenes.sphere_grid(4000, 128, 72)}[scene]()
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(6, 1, 20, False, desc["background"], seed=2, backend=STREAMED)
    ref = r.render(pcam, pw).data
    rays = r.last_stats["rays"]
    wild = [dict(stream_batch_spp=1), dict(stream_batch_spp=4000000000), dict(leaf_slots=100000, lds_leaf_stack=7), dict(stragglers=4000000000, lds_stragglers=99999),
            dict(stream_waves_per_simd=99, stream_big_threads=123, ray_pool=77), dict(stream_waves_per_simd=1, radiance_gb=4000000000, runtime_walk=9),
            dict(dual_walk=5, stream_waves_per_simd=3, leaf_slots=1), dict(dual_walk=1, stragglers=1000, stream_batch_spp=300),
            {k: 4294967295 for k in trt.Tuning.FIELDS if k != "xcd_remap"}]
    for knobs in wild:
        got = r.render(pcam, pw, tuning=knobs).data
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and r.last_stats["rays"] == rays, knobs
    for backend, knobs in ((0, dict(mega_waves_per_simd=4000000000, mega_threads=7, mega_global_waves8=3)), (1, dict(wf_waves_per_simd=12345, wf_serve_min=4000000000))):
        rb = trt.Renderer(6, 1, 20, False, desc["background"], seed=2, backend=backend)
        got = rb.render(pcam, pw, tuning=knobs).data
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and rb.last_stats["rays"] == rays, (backend, knobs)
