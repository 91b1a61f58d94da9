"""bench.py's step arithmetic on CPU: any --steps/--warmup must stay inside the renderer's sample range
(round 1 crashed on the driver's `--steps 20 --warmup 5`: step 16 asked for samples 4096..4352 of a 4096-spp frame,
which to_render_args in capi.hip rejects)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("steps,warmup", [(4, 1), (20, 5), (40, 10), (1, 0), (16, 0), (17, 16), (1000, 333)])
@pytest.mark.parametrize("spp", [256, 64, 16, 4096, 1])
def test_every_step_is_a_slice_of_the_frame(bench, steps, warmup, spp):
    frame = bench.FRAME_SPP
    seen_in_frame = 0
    for i in range(warmup + steps):
        s0, s1, accumulate = bench.step_range(i, spp)
        assert 0 <= s0 < s1 <= frame, "what capi.hip to_render_args accepts: begin <= end <= samples_per_pixel"
        assert s1 - s0 == spp, "every step is the same amount of work"
        # a frame starts by overwriting the sums and every later slice continues them, in order
        assert accumulate == (0 if s0 == 0 else 1)
        if s0 == 0:
            seen_in_frame = 0
        assert s0 == seen_in_frame
        seen_in_frame = s1


def test_step_size_must_divide_the_frame(bench):
    for bad in (0, -1, 3, 4097, 8192, 1000):
        with pytest.raises(ValueError):
            bench.step_range(0, bad)


def test_library_accepts_every_range_bench_produces(bench, trt):
    """The C ABI's own validation (no GPU needed: argument errors come before the device check would matter)."""
    import ctypes as C
    desc = trt.scenes.cornell(64, 64)
    world, cam = trt.world_from_description(desc)
    scene = world.get_bvh()
    r = trt.Renderer(bench.FRAME_SPP, 1, 50, False, desc["background"], seed=1)
    for i in (0, 15, 16, 24, 39, 49):
        s0, s1, acc = bench.step_range(i, 256)
        p = r.params(sample_begin=s0, sample_end=s1, accumulate=acc)
        name = trt.lib.trt_dominant_kernel(scene._h, C.byref(cam.pod), C.byref(p))      # runs to_render_args; "" on a rejected range
        assert name.decode() != ""
    p = r.params(sample_begin=4096, sample_end=4352, accumulate=1)                       # round 1's failing request
    assert trt.lib.trt_dominant_kernel(scene._h, C.byref(cam.pod), C.byref(p)).decode() == ""


def test_kernel_source_digest_is_stable(bench):
    assert bench.kernel_source_digest() == bench.kernel_source_digest()
    assert len(bench.kernel_source_digest()) == 16


def test_committed_pmc_profile_is_well_formed_and_staleness_is_reported(bench):
    """bench.py refuses a PMC profile of other kernel sources (round 1's 100 k-sphere profile had gone stale unnoticed) and says
    so in its JSON line (`pmc_stale`).  Here every committed entry must be well-formed; an entry taken from other kernel sources
    than the tree's is REPORTED (skip with the stale keys), not failed: a CPU-side edit under csrc/ must not turn the suite red
    until a GPU re-profile (tools/pmc_bench.sh, tools/pmc_collect.py) has been committed."""
    import json
    import pytest
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_kernels.json")))
    entry, _ = bench.pmc_profile("cornell_2048x2048_d50_spp256_streamed")
    assert entry is not None, "the bench workload has no PMC entry at all"
    stale = []
    for key, e in table.items():
        if key.startswith("_"):
            continue
        assert e["SQ_INSTS_VALU"] > 0 and e["hbm_bytes_per_launch"] > 0 and e["trace_avg_ns"] > 0, key
        assert len(e["kernel_source_digest"]) == 16, key
        if e["kernel_source_digest"] != bench.kernel_source_digest():
            stale.append(key)
    if stale:
        pytest.skip("PMC profile taken from other kernel sources (bench.py reports pmc_stale for them): " + ", ".join(stale))


def test_plain_multi_gpu_command_builds_the_drivers_launcher_line(bench, monkeypatch):
    """`python3 bench.py --gpus N --steps K --warmup W` (how the driver starts benches) with N > 1 and no WORLD_SIZE:
    the parent builds exactly the contract's torch.distributed.run line around its own arguments and starts it as a child."""
    import sys
    argv = bench.launcher_argv(8, 29511, ["--gpus", "8", "--steps", "20", "--warmup", "5"], python="python3")
    assert argv == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                    "--master-port", "29511", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    # the child's stdout is relayed with rank 0's JSON line LAST; the exit code is the child's
    started = {}

    class FakeChild:
        def __init__(self, cmd, **kw):
            started["cmd"], started["kw"] = cmd, kw
            self.stdout = iter(["rank chatter\n", '{"metric": "Mray/s"}\n', "late line from rank 3\n"])

        def wait(self):
            return 7

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeChild)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 4)
    out = []
    monkeypatch.setattr("builtins.print", lambda *a, **k: out.append((a, k.get("file"))))
    assert bench.self_launch(4, ["--gpus", "4", "--steps", "3"]) == 7
    assert started["cmd"][:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"] and started["cmd"][4:6] == ["--nproc-per-node", "4"]
    assert started["cmd"][-4:] == ["--gpus", "4", "--steps", "3"] and started["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert [a[0] for a, f in out if f is None] == ["rank chatter", "late line from rank 3", '{"metric": "Mray/s"}']


def test_plain_multi_gpu_command_without_the_gpus_exits_nonzero_with_one_line():
    """Fewer visible devices than --gpus and no rehearsal flag: one clear line on stderr, a non-zero exit code, nothing started.
    (This container has no GPU at all; the parent must get that far without importing the package or touching a device.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRT_BENCH_REHEARSAL")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout == ""
    assert len(r.stderr.strip().splitlines()) == 1 and "needs 2 visible GPUs" in r.stderr


def test_isa_event_costs_table_is_current_and_sane(bench):
    """profiles/isa_event_costs.json (tools/isa_event_costs.py; refreshed by __graft_entry__.build()): made from the kernel sources in the
    tree, the hand-written loops' counts are the documented ones, every event costs something, a light costs less than a scatter."""
    table, stale = bench.isa_event_costs()
    assert table is not None and not stale, "run python tools/isa_event_costs.py (or __graft_entry__.build())"
    loops, ev = table["asm_loops"], table["events"]
    assert (loops["box_step_flat"], loops["box_step_lds"], loops["box_step_compact"], loops["slab_test_alone"]) == (23, 27, 21, 21)    # DESIGN 5.2-5.4 (round 5: TRT_SLAB_MED3; fused slab arithmetic and byte-offset cursor in the 16-byte-node walk)
    assert all(v > 0 for v in ev.values())
    assert ev["shade_light_quad"] < ev["shade_dielectric_sphere"] < ev["shade_metal_sphere"] <= ev["shade_lambertian_sphere"]
    assert ev["shade_miss"] < 20 and 20 <= ev["box_test"] <= 40 and 100 < ev["primary_ray"] < 400


def test_useful_lane_instructions_adds_up(bench):
    table, _ = bench.isa_event_costs()
    ev, loops = table["events"], table["asm_loops"]
    c = dict(samples=10, rays=70, node_tests=1260, sphere_tests=0, quad_plane_tests=76, quad_inside_tests=70, shades=65, shade_lambertian=56,
             shade_metal=0, shade_dielectric=0, shade_light=9)
    total, parts, box = bench.useful_lane_instructions(c, 2, 18, 0, table)                  # walk 2 = the lock-step leaf list
    assert box == loops["box_step_flat"] and parts["box_tests"] == 1260 * 23
    assert parts["shades"] == 56 * ev["shade_lambertian_quad"] + 9 * ev["shade_light_quad"] + 5 * ev["shade_miss"]
    assert abs(total - sum(parts.values())) < 1e-6 and parts["ray_setup"] == 70 * ev["ray_setup"]
    _, parts_s, box_s = bench.useful_lane_instructions(dict(c, sphere_tests=100, quad_plane_tests=0), 3, 0, 1000, table)
    assert box_s == loops["box_step_compact"] and parts_s["primitive_tests"] == 100 * ev["sphere_test"]
