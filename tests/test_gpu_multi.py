"""Multi-GPU behind the C ABI (trt_render_multi: renderer.rs:37-79 is ONE call that returns the whole Image) and
concurrent renders of one scene handle.

A GPU box has one MI355X, so the N-shard layout is exercised with the same ordinal listed several times: every shard
is driven by its own host thread on its own stream with its own device buffers, renders its round-robin bands and
copies them to their place in the frame - everything the 8-GPU run does except that the copies stay on one device.
Shards of one device render the same scene handle concurrently, which is exactly the case the per-render workspace
pool (capi.hip workspace_acquire) exists for."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("scene,w,h", [("cornell", 200, 150), ("random_spheres", 160, 90)])
def test_render_multi_equals_render_for_every_shard_count(trt, orc, scene, w, h):
    desc = getattr(trt.scenes, scene)(w, h)                       # h is not a multiple of 16: the last band is ragged
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(8, 1, 10, False, desc["background"], seed=5)
    one = r.render(pcam, pw)
    rays = r.last_stats["rays"]
    ow, ocam = orc.world_from_description(desc)
    cpu, cst = orc.render(ow, ocam, 8, 10, desc["background"], seed=5, nthreads=8)
    assert np.array_equal(bits(one.data), bits(cpu)) and rays == cst["rays"]
    for devices in ([0], [0, 0], [0, 0, 0], [0] * 8, [0] * 13, None):     # 13 shards > 10 bands: some shards own nothing
        img = r.render_multi(pcam, pw, devices=devices)
        assert np.array_equal(bits(img.data), bits(cpu)), devices
        assert r.last_stats["rays"] == rays and r.last_stats["samples"] == 8 * w * h, devices
    # counting variant through the multi entry: the shards' counters add up to the oracle's
    r.render_multi(pcam, pw, devices=[0, 0, 0], collect_stats=True)
    for k in ("samples", "rays", "node_tests", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"):
        assert r.last_stats[k] == cst[k], k


def test_render_multi_progressive_passes_and_backends(trt):
    desc = trt.scenes.cornell(128, 100)
    pw, pcam = trt.world_from_description(desc)
    for backend in (trt.BACKEND_STREAMED, trt.BACKEND_MEGAKERNEL, trt.BACKEND_WAVEFRONT):
        r = trt.Renderer(12, 1, 8, False, desc["background"], seed=2, backend=backend)
        whole = r.render(pcam, pw).data
        acc = np.zeros((100, 128, 3), np.float32)
        r.render_multi(pcam, pw, devices=[0, 0], accum=acc, sample_begin=0, sample_end=5)
        r.render_multi(pcam, pw, devices=[0, 0, 0], accum=acc, sample_begin=5, sample_end=12, accumulate=1)
        assert np.array_equal(bits(acc), bits(whole)), backend


def test_render_multi_device_gathers_into_hbm(trt):
    import torch
    desc = trt.scenes.cornell(256, 208)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(6, 1, 12, False, desc["background"], seed=9)
    want = r.render(pcam, pw).data
    frame = torch.full((208, 256, 3), -1.0, device="cuda:0")
    for devices in ([0], [0, 0], [0] * 4):
        frame.fill_(-1.0)
        torch.cuda.synchronize()
        r.render_multi(pcam, pw, devices=devices, d_accum_ptr=frame.data_ptr())
        assert np.array_equal(bits(frame.cpu().numpy()), bits(want)), devices
    # continuing sums that live in HBM
    frame.zero_()
    torch.cuda.synchronize()
    r.render_multi(pcam, pw, devices=[0, 0], d_accum_ptr=frame.data_ptr(), sample_begin=0, sample_end=2)
    r.render_multi(pcam, pw, devices=[0, 0, 0], d_accum_ptr=frame.data_ptr(), sample_begin=2, sample_end=6, accumulate=1)
    assert np.array_equal(bits(frame.cpu().numpy()), bits(want))


def test_render_multi_rejects_bad_arguments(trt):
    desc = trt.scenes.cornell(64, 64)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(2, 1, 4, False, desc["background"])
    with pytest.raises(trt.TinyRTError) as e:
        r.render_multi(pcam, pw, devices=[0, 99])
    assert e.value.code == -1
    with pytest.raises(trt.TinyRTError) as e:
        r.render_multi(pcam, pw, devices=[0], band_rows=16, band_stride=1, band_offset=0, rows_local=64)
    assert e.value.code == -1
    with pytest.raises(trt.TinyRTError) as e:
        r.render_multi(pcam, pw, devices=[0], sample_begin=3, sample_end=2)
    assert e.value.code == -1


@pytest.mark.parametrize("backend", ["streamed", "wavefront"])
def test_concurrent_renders_of_one_scene_equal_serial_ones(trt, backend):
    """Round 1 cached ONE device workspace (radiance records + batch counter) on the scene handle, shared by every render:
    two renders of one scene at the same time overwrote each other's records.  Now every render owns its scratch until
    its last kernel has run.  Six host threads render the same scene handle at once, each with its own seed and sample
    count (ctypes releases the GIL during the call; trt_render uses a stream of its own); every frame must equal the one
    rendered alone."""
    be = {"streamed": trt.BACKEND_STREAMED, "wavefront": trt.BACKEND_WAVEFRONT}[backend]
    desc = trt.scenes.cornell(384, 384)
    pw, pcam = trt.world_from_description(desc)
    scene = pw.get_bvh()
    jobs = [(seed, 4 + 2 * (seed % 3)) for seed in range(1, 7)]
    serial = {}
    for seed, spp in jobs:
        serial[seed] = trt.Renderer(spp, 1, 12, False, desc["background"], seed=seed, backend=be).render(pcam, scene).data.copy()
    for _ in range(3):
        got, errors = {}, []

        def work(seed, spp):
            try:
                got[seed] = trt.Renderer(spp, 1, 12, False, desc["background"], seed=seed, backend=be).render(pcam, scene).data
            except Exception as ex:                                  # noqa: BLE001 - reported below
                errors.append(ex)

        threads = [threading.Thread(target=work, args=j) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for seed, _ in jobs:
            assert np.array_equal(bits(got[seed]), bits(serial[seed])), (backend, seed)


def test_concurrent_device_renders_on_two_streams(trt):
    """trt_render_device is asynchronous on the caller's stream: two renders of one scene enqueued on two streams (no host
    synchronisation in between) and a third that needs a LARGER workspace while those may still run."""
    import torch
    dev = torch.device("cuda:0")
    desc = trt.scenes.cornell(512, 512)
    pw, pcam = trt.world_from_description(desc)
    scene = pw.get_bvh()
    big_desc = trt.scenes.cornell(1024, 1024)
    _, big_cam = trt.world_from_description(big_desc)               # same world, larger image -> larger scratch
    ra, rb = trt.Renderer(8, 1, 16, False, desc["background"], seed=3), trt.Renderer(8, 1, 16, False, desc["background"], seed=4)
    want_a, want_b = ra.render(pcam, scene).data, rb.render(pcam, scene).data
    want_big = ra.render(big_cam, scene).data
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros((512, 512, 3), device=dev)
    b = torch.zeros((512, 512, 3), device=dev)
    big = torch.zeros((1024, 1024, 3), device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        ra.render_device(pcam, scene, a.data_ptr(), s1.cuda_stream)
        rb.render_device(pcam, scene, b.data_ptr(), s2.cuda_stream)
        ra.render_device(big_cam, scene, big.data_ptr(), s3.cuda_stream)
        ra.render_device(pcam, scene, a.data_ptr(), s1.cuda_stream)          # again on the same stream, back to back
    torch.cuda.synchronize()
    assert np.array_equal(bits(a.cpu().numpy()), bits(want_a))
    assert np.array_equal(bits(b.cpu().numpy()), bits(want_b))
    assert np.array_equal(bits(big.cpu().numpy()), bits(want_big))


def test_kernel_timing_brackets_the_dominant_kernel(trt):
    import ctypes as C
    desc = trt.scenes.cornell(256, 256)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(16, 1, 8, False, desc["background"])
    trt._lib.check(trt.lib.trt_kernel_timing_begin())
    r.render(pcam, pw)
    r.render(pcam, pw)
    ms, n = C.c_double(0.0), C.c_uint32(0)
    trt._lib.check(trt.lib.trt_kernel_timing_end(C.byref(ms), C.byref(n)))
    assert n.value == 2 and 0.0 < ms.value < 1000.0
    r.render(pcam, pw)                                              # disabled again: nothing recorded
    trt._lib.check(trt.lib.trt_kernel_timing_end(C.byref(ms), C.byref(n)))
    assert n.value == 0
    # several rendering threads at once (one per shard of a multi render): every launch's two events are paired on the thread
    # that makes the launch, so brackets never mix streams (round 2 paired them by parity in one process-wide list)
    trt._lib.check(trt.lib.trt_kernel_timing_begin())
    r.render_multi(pcam, pw, devices=[0, 0, 0])
    r.render_multi(pcam, pw, devices=[0, 0])
    trt._lib.check(trt.lib.trt_kernel_timing_end(C.byref(ms), C.byref(n)))
    assert n.value == 5 and 0.0 < ms.value < 1000.0


def _bench_line(cmd, env, root):
    import json
    import subprocess
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith("{"), "the JSON line is the last line of stdout"
    return json.loads(lines[-1])


BENCH_N2_ARGS = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "512", "--height", "500", "--spp-per-step", "64", "--cpu-seconds", "0"]


def _check_n2_line(d):
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["unit"] == "Mray/s"
    assert d["config"]["samples"] == 3 * 64 * 512 * 500                 # both ranks' samples, every pixel once per sample
    assert d["config"]["image_rows_per_gpu"] == 256                     # 500 rows = 32 bands of 16 (the last one 4 rows): rank 0 owns 16 full bands
    assert "REHEARSAL" in d["config"]["parallelism"] and d["roofline"]["launches_per_step"] >= 1
    assert d["config"]["world_size"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert len(d["roofline_per_rank"]) == 2


def test_bench_n2_rehearsal_on_one_gpu():
    """bench.py's N > 1 path (band layout per rank, kernel timing, counter all-reduce, gather + un-interleave, JSON line) with two
    ranks on this one GPU over gloo (TRT_BENCH_REHEARSAL=1): everything the driver's multi-GPU run does except RCCL itself.
    Started the way the DRIVER starts a bench - the plain command, no external launcher: the parent spawns
    torch.distributed.run as a child before it touches the GPU and relays rank 0's line."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TRT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = _bench_line([sys.executable, os.path.join(root, "bench.py")] + BENCH_N2_ARGS, env, root)
    _check_n2_line(d)
    assert "bench.py itself" in d["config"]["launched_by"]


def test_bench_n2_rehearsal_under_an_external_launcher():
    """The same run started by the launcher line of the build contract (python -m torch.distributed.run ... bench.py --gpus 2)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = _bench_line(bench.launcher_argv(2, bench.free_port(), BENCH_N2_ARGS), env, root)
    _check_n2_line(d)
    assert d["config"]["launched_by"] == "external launcher"


def test_bench_plain_n2_without_a_second_gpu_fails_with_one_line():
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the plain command would run for real")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRT_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + BENCH_N2_ARGS, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs 2 visible GPUs" in r.stderr


def test_multi_gpu_c_example_renders_identical_frames(tmp_path):
    """examples/multi_gpu.c: every visible device (two shards on device 0 when there is only one) through trt_render_multi_device
    into a frame in HBM, ragged last band, then a second accumulating pass: identical to the one-device frame."""
    import subprocess
    from test_host_boundary import _build_multi_gpu_example
    exe = _build_multi_gpu_example(tmp_path)
    r = subprocess.run([exe, "512", "500", "16"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr
    # the mode a new multi-GPU box is brought up with: every visible ordinal renders the whole frame alone before anything is gathered
    r = subprocess.run([exe, "--verify-each-device", "320", "200", "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "device 0 alone" in r.stdout and "DIFFERS" not in r.stdout and "IDENTICAL" in r.stdout, r.stdout + r.stderr
    assert "one strided 2-D copy per shard" in r.stdout                   # host or same-device gather: never band by band



def test_mixed_entry_points_from_eight_threads_on_two_scenes(trt):
    """Every blocking entry point at once, on two scene handles: trt_render (streamed, wavefront, megakernel; several image sizes, so
    workspaces and context frames are regrown while others are in flight), trt_render_multi with three and five shards, and
    trt_sample_batch - eight host threads, four repetitions, with the idle-scratch cap forcing trims in between.  Every result equals
    the one computed alone.  (The round-3 host layer under its real runtime; its logic alone runs under TSan / ASan on the simulated
    runtime in tests/test_host_sanitizers.py.)"""
    import ctypes as C
    cornell = {sz: trt.scenes.cornell(*sz) for sz in ((192, 160), (320, 200), (96, 96))}
    spheres = trt.scenes.random_spheres(240, 135)
    wc, _ = trt.world_from_description(cornell[(192, 160)])
    cams = {sz: trt.world_from_description(d)[1] for sz, d in cornell.items()}
    ws, cam_s = trt.world_from_description(spheres)
    scene_c, scene_s = wc.get_bvh(), ws.get_bvh()
    bg_c, bg_s = cornell[(192, 160)]["background"], spheres["background"]

    def batch_points(n):
        pts = (trt.SamplePoint * n)()
        for i in range(n):
            pts[i].x, pts[i].y = i % 16, i // 16
            pts[i].ray.origin = trt.Vec3(50.0, 50.0, -140.0)
            pts[i].ray.direction = trt.Vec3(0.01 * (i % 16) - 0.08, 0.01 * (i // 16) - 0.08, 1.0)
        return pts

    jobs = []
    for k, (sz, be) in enumerate((((192, 160), trt.BACKEND_STREAMED), ((320, 200), trt.BACKEND_WAVEFRONT), ((96, 96), trt.BACKEND_MEGAKERNEL),
                                  ((320, 200), trt.BACKEND_STREAMED))):
        jobs.append(("render", lambda sz=sz, be=be, k=k: trt.Renderer(6, 1, 10, False, bg_c, seed=20 + k, backend=be).render(cams[sz], scene_c).data))
    jobs.append(("multi3", lambda: trt.Renderer(5, 1, 10, False, bg_c, seed=31).render_multi(cams[(320, 200)], scene_c, devices=[0, 0, 0]).data))
    jobs.append(("multi5", lambda: trt.Renderer(4, 1, 12, False, bg_s, seed=32).render_multi(cam_s, scene_s, devices=[0] * 5).data))
    jobs.append(("spheres", lambda: trt.Renderer(8, 1, 12, False, bg_s, seed=33).render(cam_s, scene_s).data))

    def batch():
        out, _ = trt.sample_batch(scene_c, batch_points(256), 8, bg_c, seed=5, collect_stats=False)
        return np.array([[c.color.x, c.color.y, c.color.z] for c in out], np.float32)
    jobs.append(("batch", batch))
    serial = [fn() for _, fn in jobs]
    for rep in range(4):
        got, errors = [None] * len(jobs), []

        def work(i):
            try:
                got[i] = jobs[i][1]()
            except Exception as ex:                                      # noqa: BLE001 - reported below
                errors.append((jobs[i][0], ex))

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for i, (name, _) in enumerate(jobs):
            assert np.array_equal(bits(got[i]), bits(serial[i])), (rep, name)
        if rep == 1:
            trt._lib.check(trt.lib.trt_scene_trim(scene_c._h))            # frees every idle cached buffer; the next repetition re-grows them


def test_minimal_c_example_renders_and_its_two_tunings_agree(trt, tmp_path):
    """examples/minimal.c on the GPU: plain C11 against include/tinyrt.h, one render with the default tuning and one under another trt_tuning -
    the program itself compares the two frames and exits non-zero if they differ."""
    from test_host_boundary import test_header_is_valid_c_and_the_c_example_fails_loudly_without_gpu as build_and_run
    assert trt.lib.trt_device_count() > 0
    build_and_run(trt, tmp_path)
