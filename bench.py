#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: Mray/s (primary + secondary rays) of the path-tracing
hot path on the Cornell box at 2048x2048, depth 50 (BASELINE.json configs[3]; it fits one GPU), plus the dominant
kernel's roofline block (VALU lane-slot fraction: the binding resource; physical HBM traffic; SURVEY 8(d)'s algorithmic bytes,
labelled) and the CPU oracle timed on this box's host cores (all cores and one thread).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts the line below as a child and relays it)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of --spp-per-step samples per pixel over the whole image, continuing the running f32 sums that
stay resident in HBM (the full config's 4096 spp is 16 such steps of 256).  With N > 1 the image's 16-row bands are
dealt round-robin to the ranks (strong scaling: the image is fixed), every rank runs the same steps on its rows, and
one RCCL gather of the accumulators to rank 0 closes the frame inside the timed region.  Rays are counted on the
device (one ray = one closest-hit query = one `world.hit` call, reference cpu.rs:48).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); 6290 measured-achievable
# VALU issue peak: one wave64 VALU instruction per 2 cycles per SIMD-32 (guides/MI355X_MICROARCH.md, cycle constants:
# `v_fma_f32` (wave64) 2 cyc), 4 SIMDs x 256 CUs, 2.4 GHz
N_SIMD = 1024
CLOCK_GHZ = 2.4
VALU_PEAK_GINST = N_SIMD * CLOCK_GHZ / 2.0          # 1228.8 G wave-instructions per second
FRAME_SPP = 4096              # BASELINE configs[3]: the full frame is 4096 spp


def step_range(i, spp_per_step, frame_spp=FRAME_SPP):
    """Sample range of step `i` (warm-up steps included in the numbering) and whether it continues the running sums.

    The frame has `frame_spp` samples per pixel and a step renders `spp_per_step` of them; after the frame's last
    slice the next step starts a new frame (sums overwritten instead of continued), so any --steps/--warmup works:
    every step is one slice of the BASELINE frame, exactly the same work whichever frame it belongs to."""
    if spp_per_step <= 0 or spp_per_step > frame_spp or frame_spp % spp_per_step:
        raise ValueError(f"--spp-per-step must divide the frame's {frame_spp} spp (got {spp_per_step})")
    per_frame = frame_spp // spp_per_step
    k = i % per_frame
    return k * spp_per_step, (k + 1) * spp_per_step, (0 if k == 0 else 1)


_COMMENT_OR_STRING = None


def strip_comments(text):
    """C / C++ source without its comments and with runs of blanks collapsed (string literals - the hand-written asm blocks - kept as they are):
    what the compiler sees.  A comment-only edit must not turn a stored PMC profile stale."""
    import re
    global _COMMENT_OR_STRING
    if _COMMENT_OR_STRING is None:
        _COMMENT_OR_STRING = re.compile(r'''("(?:\\.|[^"\\\n])*"|'(?:\\.|[^'\\\n])*')|//[^\n]*|/\*.*?\*/''', re.S)
    code = _COMMENT_OR_STRING.sub(lambda m: m.group(1) or " ", text)
    return "\n".join(" ".join(line.split()) for line in code.splitlines() if line.strip())


def kernel_source_digest():
    """sha256 over the kernel sources AS THE COMPILER SEES THEM (comments stripped, blanks collapsed): a PMC profile is only valid for the build it was
    taken from.  (Until round 5 the raw bytes were hashed, so rewording a comment cost a re-profile.)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tiny-raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name == "capi.hip":
            continue                  # host-side C ABI only: no device code, nothing a kernel's counters depend on
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            with open(os.path.join(d, name), "r", encoding="utf-8", errors="replace") as f:
                text = f.read()
            h.update(name.encode() + b"\0" + (text if name == "Makefile" else strip_comments(text)).encode("utf-8"))
    return h.hexdigest()[:16]


def pmc_profile(key):
    """profiles/pmc_kernels.json[key]: per-launch PMC means of the dominant kernel (tools/pmc_bench.sh).  Returns
    (entry, stale): an entry taken from other kernel sources than the ones in the tree is reported as stale and its
    numbers are not used."""
    path = os.path.join(ROOT, "profiles", "pmc_kernels.json")
    if not os.path.exists(path):
        return None, False
    with open(path) as f:
        e = json.load(f).get(key)
    if e is None:
        return None, False
    if e.get("kernel_source_digest") != kernel_source_digest():
        return e, True
    return e, False


def algorithmic_bytes(c, pixels):
    """bytes the reference-order traversal has to read, + 12 B per pixel written per launch."""
    return (32 * c["node_tests"] + 16 * c["sphere_tests"] + 16 * c["quad_plane_tests"] + 48 * c["quad_inside_tests"]
            + 20 * c["shades"] + 12 * pixels)


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows all
    of the host's CPUs in the mask but grants one GPU slot a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, quota // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(trt, desc, depth, budget_s):
    """The CPU oracle (a port of the reference's CPU path) on a bounded sample of the same workload: whole-image passes of 1 spp
    until the budget is spent - about two thirds of it on all host cores (the headline CPU figure), the rest on ONE thread
    (the reference's configs[0] is a single-thread run).  Returns (block, frame, spp, rays): the all-core leg's
    frame holds samples [0, spp) of the 4096-spp frame - main() renders the same samples on the GPU and compares (`parity`)."""
    from oracle import orc
    import numpy as np
    cores = usable_cores()
    world, cam = orc.world_from_description(desc)
    orc.lib.orc_world_build(world._h)

    def run(nthreads, seconds, row_end=None):
        acc = np.zeros((cam.height, cam.width, 3), np.float32)
        rays = spp_done = 0
        kw = {} if row_end is None else dict(row_begin=0, row_end=row_end)
        t0 = time.perf_counter()
        while True:
            _, st = orc.render(world, cam, 4096, depth, desc["background"], seed=1, nthreads=nthreads, sample_begin=spp_done,
                               sample_end=spp_done + 1, accum=acc, **kw)
            rays += st["rays"]
            spp_done += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or spp_done >= 64:
                return rays, spp_done, dt, acc

    rays, spp_done, dt, frame = run(cores, budget_s * 2.0 / 3.0)
    out = {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
           "sample": f"{spp_done} spp of the {cam.width}x{cam.height} depth-{depth} frame ({rays} rays, {dt:.1f} s)"}
    # one thread: a 1-spp pass over the top rows of the frame, sized from the all-core rate to fit the rest of the budget
    per_thread = rays / dt / max(cores, 1)
    rows = int(min(cam.height, max(16, (budget_s / 3.0) * per_thread / max(rays / spp_done / cam.height, 1.0))))
    r1, s1, d1, _ = run(1, budget_s / 3.0, row_end=rows)
    out["single_thread"] = {"value": r1 / d1 / 1e6, "unit": "Mray/s", "cores": 1,
                            "sample": f"{s1} spp of the top {rows} rows ({r1} rays, {d1:.1f} s)"}
    return out, frame, spp_done, rays


def valu_roofline(key, avg_ms, rays_per_launch, warnings, single_gpu=True):
    """The VALU lane-slot block of one kernel from its stored PMC profile (per-ray instruction counts) and its launch time measured live."""
    prof, stale = pmc_profile(key)
    out = {"frac": None, "achieved": None, "pmc_key": key, "pmc_stale": stale}
    if prof is not None and not stale and avg_ms and rays_per_launch > 0:
        prof_rays = prof.get("rays_per_launch")
        if not prof_rays:
            if single_gpu:
                prof_rays = rays_per_launch              # entries older than the field: same workload, same rays
            else:
                warnings.append("PMC entry has no rays_per_launch: cannot scale it to this rank's share")
        if prof_rays:
            scale = rays_per_launch / prof_rays
            insts = prof["SQ_INSTS_VALU"] * scale
            lanes = prof["SQ_THREAD_CYCLES_VALU"] / prof["SQ_INSTS_VALU"] if prof.get("SQ_THREAD_CYCLES_VALU") else None   # active lanes per VALU instruction
            hbm = prof["hbm_bytes_per_launch"] * scale
            issue_rate = insts / (avg_ms * 1e-3) / 1e9                                    # G wave-instructions / s
            issue_frac = issue_rate / VALU_PEAK_GINST
            out.update({
                "issue_frac": round(issue_frac, 4), "issue_achieved_Ginst_s": round(issue_rate, 1), "issue_peak_Ginst_s": round(VALU_PEAK_GINST, 1),
                "valu_wave_insts_per_launch": int(insts), "valu_wave_insts_per_ray": round(prof["SQ_INSTS_VALU"] / prof_rays, 2) if prof_rays else None,
                "cycles_per_valu_inst_per_simd": round(avg_ms * 1e-3 * CLOCK_GHZ * 1e9 * N_SIMD / insts, 3),
                "peak_cycles_per_valu_inst_per_simd": 2.0, "mean_active_lanes": round(lanes, 1) if lanes else None,
                "salu_insts_per_launch": int(prof.get("SQ_INSTS_SALU", 0) * scale), "traffic": int(hbm),
                "hbm_physical_GBps": round(hbm / (avg_ms * 1e-3) / 1e9, 1), "hbm_physical_frac": round(hbm / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "pmc_source": prof.get("source")})
            if lanes:
                out["achieved"] = round(issue_rate * lanes / 1e3, 2)
                out["frac"] = round(issue_frac * lanes / 64.0, 4)
                out["lane_slot_frac"] = out["frac"]
            # HBM side of the same launch (north_star: "rocprof-reported achieved HBM GB/s against the chip's peak").  FETCH_SIZE / WRITE_SIZE count the
            # L2's fabric-side requests in KiB: what misses L2 - Infinity Cache hits INCLUDED (the guide: no counter separates them), so this
            # bounds the HBM traffic from above; FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950, hence x 2.
            if prof.get("FETCH_SIZE") is not None:
                t_s = avg_ms * 1e-3
                rd = 2.0 * prof["FETCH_SIZE"] * 1024.0 * scale
                wr = prof.get("WRITE_SIZE", 0.0) * 1024.0 * scale
                hbm_block = {"read_GBps": round(rd / t_s / 1e9, 1), "write_GBps": round(wr / t_s / 1e9, 1),
                             "read_frac_of_peak": round(rd / t_s / 1e9 / HBM_PEAK_GBPS, 4), "read_bytes_per_ray": round(rd / max(rays_per_launch, 1), 1),
                             "peak_GBps": HBM_PEAK_GBPS, "counts": "L2 misses (fabric requests): Infinity Cache hits included, no counter separates them"}
                if prof.get("TCC_HIT_sum") is not None and prof.get("TCC_MISS_sum") is not None:
                    hbm_block["l2_hit_rate"] = round(prof["TCC_HIT_sum"] / max(prof["TCC_HIT_sum"] + prof["TCC_MISS_sum"], 1.0), 4)
                if prof.get("SQ_WAIT_ANY") and prof.get("SQ_WAVE_CYCLES"):
                    hbm_block["wave_cycles_waiting"] = round(prof["SQ_WAIT_ANY"] / prof["SQ_WAVE_CYCLES"], 3)
                out["hbm"] = hbm_block
            if prof.get("GRBM_GUI_ACTIVE") and prof.get("trace_avg_ns"):
                # the clock the chip really ran at while the profile was taken: GRBM_GUI_ACTIVE sums busy cycles over the 8 XCDs
                eff = prof["GRBM_GUI_ACTIVE"] / 8.0 / prof["trace_avg_ns"]
                out["effective_clock_ghz"] = round(eff, 3)
                if out["frac"] is not None and eff > 0:
                    out["frac_at_effective_clock"] = round(out["frac"] * CLOCK_GHZ / eff, 4)      # peak re-priced at that clock instead of the nominal 2.4 GHz
            if issue_frac > 1.0 or out["hbm_physical_frac"] > 1.0:
                warnings.append(f"{key}: a fraction above 1: the stored PMC profile does not describe this run (other box clock, other build?)")
    elif prof is None:
        warnings.append(f"no PMC profile for {key} under profiles/pmc_kernels.json: frac is null")
    elif stale:
        warnings.append(f"the PMC profile of {key} was taken from other kernel sources (pmc_stale): frac is null until tools/pmc_bench.sh is re-run")
    return out


def isa_event_costs():
    """profiles/isa_event_costs.json (tools/isa_event_costs.py): VALU lane-instructions one lane needs per event, counted in the gfx950 ISA
    of the product's own device functions.  (table, stale): a table made from other kernel sources is not used."""
    path = os.path.join(ROOT, "profiles", "isa_event_costs.json")
    if not os.path.exists(path):
        return None, False
    with open(path) as f:
        t = json.load(f)
    return t, t.get("kernel_source_digest") != kernel_source_digest()


UNIT_DISK_EXTRA_ITERATIONS = 4.0 / 3.141592653589793 - 1.0        # the rejection loop of vec3extend.rs:45-53 runs 4/pi times on average
UNIT_DISK_LOOP_VALU = 14                                          # its body: two draws (rng_next 6 + mapping 2 each), the squared length and the compare


def useful_lane_instructions(c, walk, n_quads, n_spheres, table):
    """Lane-instructions the PATH'S ARITHMETIC needs for the events the counting pass counted: every event x what one lane
    needs for it (isa_event_costs).  `c`: counters of a collect_stats=2 pass (box tests of the walk the production kernel performs);
    `walk`: trt_launch_plan.walk of the production launch (which hand-written box-step loop runs)."""
    ev, loops = table["events"], table["asm_loops"]
    box = {1: loops["box_step_lds"], 2: loops["box_step_flat"], 3: loops["box_step_compact"]}.get(walk, ev["box_test"])
    quads = n_quads >= n_spheres                                   # which primitive the shades' HitRecord is built for (bench scenes hold one kind)
    shade = (c["shade_lambertian"] * ev["shade_lambertian_quad" if quads else "shade_lambertian_sphere"]
             + c["shade_metal"] * ev["shade_metal_sphere"] + c["shade_dielectric"] * ev["shade_dielectric_sphere"]
             + c["shade_light"] * ev["shade_light_quad"] + (c["rays"] - c["shades"]) * ev["shade_miss"])
    parts = {"ray_setup": c["rays"] * ev["ray_setup"], "box_tests": c["node_tests"] * box,
             "primitive_tests": c["quad_plane_tests"] * ev["quad_test"] + c["sphere_tests"] * ev["sphere_test"], "shades": shade,
             "primary_rays": c["samples"] * (ev["primary_ray"] + UNIT_DISK_EXTRA_ITERATIONS * UNIT_DISK_LOOP_VALU)}
    return sum(parts.values()), parts, box


OTHER_SCENES = (   # BASELINE configs[2] and configs[4] at their own sizes: short untimed-by-the-driver runs reported beside the headline,
    # and (round 5) the deep-BVH scene at a size BEYOND the 256 MiB Infinity Cache: 4 M spheres, 809 MB packed, ~300 MB of it
    # touched per frame - the one scene whose walk reads HBM, where north_star's HBM fraction is a number
    # (scene, spheres, width, height, spp per step, steps, warm-up)
    ("random_spheres", 0, 1920, 1080, 256, 3, 1),
    ("sphere_grid", 0, 3840, 2160, 16, 3, 2),
    ("sphere_field", 4_000_000, 3840, 2160, 4, 3, 1),
)


def short_run(trt, torch, dev, scene_name, spheres, W, H, S, steps, warmup, depth, backend_name, backend):
    """A few steps of another scene at its own size on this GPU, after the timed region of the headline workload: value,
    ms per step, the dominant kernel's launch time (HIP events on its launch stream), its lane-slot fraction and its HBM numbers."""
    t_scene = time.perf_counter()
    desc = {"random_spheres": trt.scenes.random_spheres, "sphere_grid": lambda w, h: trt.scenes.sphere_grid(spheres or 100000, w, h),
            "sphere_field": lambda w, h: trt.scenes.sphere_field(spheres or 4_000_000, w, h)}[scene_name](W, H)
    world, cam = trt.world_from_description(desc)
    scene = world.get_bvh()
    t_scene = time.perf_counter() - t_scene
    renderer = trt.Renderer(FRAME_SPP, 1, depth, False, desc["background"], seed=1, backend=backend)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()

    def step(i):
        s0, s1, accumulate = step_range(i, S)
        renderer.render_device(cam, scene, acc.data_ptr(), stream.cuda_stream, ctr.data_ptr(), sample_begin=s0, sample_end=s1, accumulate=accumulate)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    ctr.zero_()
    torch.cuda.synchronize()
    trt._lib.check(trt.lib.trt_kernel_timing_begin())
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = C.c_double(0.0), C.c_uint32(0)
    trt._lib.check(trt.lib.trt_kernel_timing_end(C.byref(k_ms), C.byref(k_n)))
    rays, samples = int(ctr[1].item()), int(ctr[0].item())
    n_launch = int(k_n.value)
    avg_ms = k_ms.value / n_launch if n_launch else None
    warnings = []
    key = f"{scene_name}{spheres or ''}_{W}x{H}_d{depth}_spp{S}_{backend_name}"
    rf = valu_roofline(key, avg_ms, rays / n_launch if n_launch else 0.0, warnings)
    kernel = trt.lib.trt_dominant_kernel(scene._h, C.byref(cam.pod), C.byref(renderer.params())).decode()
    info = scene.info()
    # useful-work fraction of this scene too (main() has the definition): one untimed step of the counting kernel on the culling tree
    useful = None
    table, table_stale = isa_event_costs()
    if table is not None and not table_stale and avg_ms and n_launch:
        try:
            uctr = torch.zeros(16, dtype=torch.int64, device=dev)
            s0, s1, _ = step_range(warmup, S)
            renderer.render_device(cam, scene, acc.data_ptr(), stream.cuda_stream, uctr.data_ptr(), sample_begin=s0, sample_end=s1, accumulate=0, collect_stats=2)
            torch.cuda.synchronize()
            names = ("samples", "rays", "node_tests", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades")
            uc = dict(zip(names, [int(v) for v in uctr[:7].tolist()]))
            uc.update(zip(("shade_lambertian", "shade_metal", "shade_dielectric", "shade_light"), [int(v) for v in uctr[12:16].tolist()]))
            walk = renderer.launch_plan(cam, scene)["walk"]
            total, parts, box_cost = useful_lane_instructions(uc, walk, info["num_quads"], info["num_spheres"], table)
            per_ray = total / max(uc["rays"], 1)
            rate = per_ray * (rays / n_launch) / (avg_ms * 1e-3) / 1e12
            useful = {"useful_frac": round(rate / (VALU_PEAK_GINST * 64.0 / 1e3), 4), "lane_instructions_per_ray": round(per_ray, 1),
                      "box_tests_per_ray": round(uc["node_tests"] / max(uc["rays"], 1), 2), "valu_per_box_test": box_cost,
                      "per_ray_by_event": {k: round(v / max(uc["rays"], 1), 1) for k, v in parts.items()}}
            if rf.get("valu_wave_insts_per_ray") and rf.get("mean_active_lanes"):
                useful["useful_over_issued"] = round(per_ray / (rf["valu_wave_insts_per_ray"] * rf["mean_active_lanes"]), 3)
        except Exception as e:                                   # noqa: BLE001 - the bench line must still be printed
            useful = {"error": repr(e)}
    out = {"workload": f"{scene_name}{' ' + str(spheres) + ' spheres' if spheres else ''} {W}x{H}, depth {depth}, {S} spp per step (of 4096), {backend_name}, seed 1",
           "scene": {"primitives": info["num_spheres"] + info["num_quads"], "packed_bytes": info["device_bytes"], "build_s": round(t_scene, 2)},
           "value": round(rays / elapsed / 1e6, 2),
           "unit": "Mray/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4), "rays": rays, "samples": samples,
           "roofline": {"kernel": kernel, "avg_launch_ms": round(avg_ms, 4) if avg_ms else None, "launches_timed": n_launch,
                        "frac": rf["frac"], "issue_frac": rf.get("issue_frac"), "mean_active_lanes": rf.get("mean_active_lanes"),
                        "cycles_per_valu_inst_per_simd": rf.get("cycles_per_valu_inst_per_simd"), "traffic": rf.get("traffic"),
                        "hbm": rf.get("hbm"), "useful": useful, "pmc_key": key, "pmc_stale": rf["pmc_stale"], "warnings": warnings}}
    del acc, scene, world
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_argv(n, port, bench_args, python=None):
    """The command the driver's contract names for N > 1, built by the parent when `bench.py --gpus N` is started plainly."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(bench_args)


def visible_gpus():
    """Devices this process could use, WITHOUT initialising the GPU (torch.cuda.device_count() does not, on this image): the parent of
    a self-launched run must stay a process that never touched the card."""
    import torch
    return torch.cuda.device_count()


def self_launch(n, bench_args):
    """`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE: start one rank per GPU under torch.distributed.run as a CHILD process
    (never an exec), relay its stdout - rank 0's JSON line is the last line - and return its exit code.  Nothing here imports the
    package, creates a HIP context or makes a GPU call."""
    import subprocess
    rehearsal = os.environ.get("TRT_BENCH_REHEARSAL") == "1"
    have = visible_gpus()
    if have < n and not rehearsal:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this box shows {have} (TRT_BENCH_REHEARSAL=1 runs all ranks on cuda:0 over gloo: "
              "a rehearsal, not a measurement)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), TRT_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = launcher_argv(n, free_port(), bench_args)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    last = None
    for line in child.stdout:
        line = line.rstrip("\n")
        if line.startswith("{"):
            if last is not None:
                print(last, flush=True)
            last = line                   # held back so that the JSON line is the LAST thing on stdout whatever else a rank prints
        else:
            print(line, flush=True)
    rc = child.wait()
    if last is not None:
        print(last, flush=True)
    return rc


def main():
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        n = pre.parse_known_args()[0].gpus
        if n > 1:
            sys.exit(self_launch(n, sys.argv[1:]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--backend", default="auto", choices=["auto", "megakernel", "wavefront", "streamed"],
                    help="auto = the library's default (streamed)")
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=2048)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "random_spheres", "sphere_grid", "sphere_field"])
    ap.add_argument("--spheres", type=int, default=0, help="sphere_grid (default 100000) / sphere_field (default 4000000): number of spheres")
    ap.add_argument("--tuning", default="", help="trt_tuning fields for the timed renders, e.g. dual_walk=1,stream_waves_per_simd=6 (scheduling only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="budget of the cpu_baseline leg; 0 = measurement only: no CPU leg, no parity block, no other scenes (what the tools/ scripts use)")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the untimed counter pass")
    ap.add_argument("--no-other-scenes", action="store_true", help="skip the short runs of the other two BASELINE scenes after the timed region")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:                        # under a launcher the launcher's world size is the truth
        args.gpus = world_size
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    # Rehearsal of the N > 1 code path on a box with ONE GPU (TRT_BENCH_REHEARSAL=1): every rank renders its bands on
    # cuda:0 and the collectives run on gloo through host copies.  Not a measurement - it only proves the band layout,
    # the counters' all-reduce and the gather before the driver runs the real thing on RCCL.
    rehearsal = world_size > 1 and os.environ.get("TRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    trt = importlib.import_module("tiny-raytracer_amd")
    tiles = importlib.import_module("tiny-raytracer_amd.tiles")
    trt._lib.check(trt.lib.trt_set_device(local_rank))

    W, H = args.width, args.height
    t_scene = time.perf_counter()
    desc = {"cornell": trt.scenes.cornell, "random_spheres": trt.scenes.random_spheres,
            "sphere_grid": lambda w, h: trt.scenes.sphere_grid(args.spheres or int(os.environ.get("TRT_BENCH_SPHERES", "100000")), w, h),
            "sphere_field": lambda w, h: trt.scenes.sphere_field(args.spheres or 4_000_000, w, h)}[args.scene](W, H)
    world, cam = trt.world_from_description(desc)
    scene = world.get_bvh()
    t_scene = time.perf_counter() - t_scene
    total_spp = FRAME_SPP
    try:
        step_range(0, args.spp_per_step)
    except ValueError as e:
        sys.exit(str(e))
    if args.backend == "auto":
        args.backend = "streamed"
    backend = {"wavefront": trt.BACKEND_WAVEFRONT, "streamed": trt.BACKEND_STREAMED}.get(args.backend, trt.BACKEND_MEGAKERNEL)
    renderer = trt.Renderer(total_spp, 1, args.depth, False, desc["background"], seed=1, backend=backend)
    if args.tuning:
        renderer.tuning = {k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))}
    kernel_name = trt.lib.trt_dominant_kernel(scene._h, C.byref(cam.pod), C.byref(renderer.params())).decode()      # what a kernel trace of a step shows

    lay = tiles.band_layout(H, world_size, rank)
    band = dict(band_rows=lay["band_rows"], band_stride=lay["band_stride"], band_offset=lay["band_offset"],
                rows_local=lay["rows_local"]) if world_size > 1 else {}
    rows_local = lay["rows_local"]
    acc = torch.zeros((rows_local, W, 3), dtype=torch.float32, device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    S = args.spp_per_step

    def step(i, stats=False, counters=ctr, target=acc):
        s0, s1, accumulate = step_range(i, S)
        renderer.render_device(cam, scene, target.data_ptr(), stream.cuda_stream, counters.data_ptr(),
                               sample_begin=s0, sample_end=s1, accumulate=accumulate, collect_stats=int(stats),
                               **band)

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather(local):
        if rehearsal:
            torch.cuda.synchronize()
            return tiles.gather_image(local.cpu(), H, W, world_size, rank)
        return tiles.gather_image(local, H, W, world_size, rank)

    def reduce(t, op):
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    for i in range(args.warmup):
        step(i)
    if world_size > 1:                                  # warm the RCCL gather too
        gather(acc)
    barrier()
    ctr.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    trt._lib.check(trt.lib.trt_kernel_timing_begin())   # HIP events around every dominant-kernel launch, on the launch stream
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        step(args.warmup + k)
        ev[k][1].record(stream)
    frame = gather(acc)
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = C.c_double(0.0), C.c_uint32(0)
    trt._lib.check(trt.lib.trt_kernel_timing_end(C.byref(k_ms), C.byref(k_n)))
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    counts = ctr.clone()
    if world_size > 1:
        reduce(t, dist.ReduceOp.MAX)
        reduce(counts, dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = int(counts[1].item())
    total_samples = int(counts[0].item())
    launch_ms = [a.elapsed_time(b) for a, b in ev]

    # ---- roofline block of the dominant kernel, per rank ----
    # Live part: the kernel's average launch time from HIP events on its own launch stream (trt_kernel_timing_*), this rank's
    # rays per launch.  Stored part: VALU instruction and HBM byte counts PER RAY from the rocprofv3 PMC passes kept under
    # profiles/ (tools/pmc_bench.sh; counters need their own runs), scaled by this rank's rays - so the block exists at any N.
    # Nothing here can end the run: what does not add up goes into roofline["warnings"].
    warnings = []
    launches_per_step = 1
    if args.backend == "streamed":
        chunk = trt.lib.trt_streamed_chunk_spp(W, max(rows_local, 1))
        launches_per_step = (S + chunk - 1) // chunk
    if rows_local == 0:
        launches_per_step = 0
    n_launch = int(k_n.value)
    if n_launch != launches_per_step * args.steps:
        warnings.append(f"timed {n_launch} dominant-kernel launches, expected {launches_per_step * args.steps}")
    avg_ms = k_ms.value / n_launch if n_launch else None
    my_rays = int(ctr[1].item())
    rays_per_launch = my_rays / n_launch if n_launch else 0.0
    scene_tag = args.scene + (str(args.spheres) if args.spheres else "")
    key = f"{scene_tag}_{W}x{H}_d{args.depth}_spp{S}_{args.backend}" + ("_" + args.tuning.replace(",", "_").replace("=", "") if args.tuning else "")
    roofline = {"bound": "valu", "kernel": kernel_name, "achieved": None, "peak": round(VALU_PEAK_GINST * 64.0 / 1e3, 2),
                "unit": "T f32 lane-instructions/s (VALU wave-instructions x active lanes)", "frac": None, "traffic": None,
                "frac_is": "lane-slot fraction: VALU lane-instructions per second / (1024 SIMDs x 2.4 GHz / 2 cycles x 64 lanes); "
                           "instruction counts per ray from a stored PMC profile of the same kernel sources, launch time measured live",
                "avg_launch_ms": round(avg_ms, 4) if avg_ms else None, "launches_per_step": launches_per_step, "launches_timed": n_launch,
                "rays_per_launch": int(rays_per_launch), "step_ms_by_events": round(sum(launch_ms) / len(launch_ms), 4), "rank": rank}
    roofline.update(valu_roofline(key, avg_ms, rays_per_launch, warnings, single_gpu=world_size == 1))
    if not args.no_roofline_pass and rows_local > 0:
        # untimed: the same K launches with the counting kernel variant -> exact algorithmic bytes of those launches
        sctr = torch.zeros(16, dtype=torch.int64, device=dev)
        scratch = torch.zeros_like(acc)
        for k in range(args.steps):
            step(args.warmup + k, stats=True, counters=sctr, target=scratch)
        torch.cuda.synchronize()
        c = dict(zip(("samples", "rays", "node_tests", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades"),
                     [int(v) for v in sctr[:7].tolist()]))
        if c["rays"] != my_rays:
            warnings.append(f"counting variant traced {c['rays']} rays, the timed launches {my_rays}")
        roofline["rays_per_sample"] = round(c["rays"] / max(c["samples"], 1), 3)
        if avg_ms and n_launch:
            bytes_per_launch = (algorithmic_bytes(c, 0) + 12 * rows_local * W * args.steps) / n_launch
            gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            roofline["algorithmic"] = {
                "bytes_per_launch": int(bytes_per_launch), "GBps": round(gbps, 2), "ratio_to_hbm_peak": round(gbps / HBM_PEAK_GBPS, 5),
                "bytes_per_ray": round(algorithmic_bytes(c, 0) / max(c["rays"], 1), 2),
                "note": "SURVEY 8(d): bytes the reference-order traversal would read per launch / launch time.  NOT a physical "
                        "fraction and not a roofline for this path: the scene is served from SGPRs / LDS / L2, so it exceeds 1 "
                        "(DESIGN.md section 8); the physical HBM traffic is `traffic`"}
    # ---- useful-work fraction: what the path's arithmetic NEEDS against the chip's lane-slots (frac counts every issued lane-instruction) ----
    table, table_stale = isa_event_costs()
    if not args.no_roofline_pass and rows_local > 0 and avg_ms and n_launch and table is not None and not table_stale:
        # one untimed step with the counting variant on the CULLING tree (collect_stats = 2): the box tests the production walk performs
        uctr = torch.zeros(16, dtype=torch.int64, device=dev)
        uscratch = torch.zeros_like(acc)
        step(args.warmup, stats=2, counters=uctr, target=uscratch)
        torch.cuda.synchronize()
        names = ("samples", "rays", "node_tests", "sphere_tests", "quad_plane_tests", "quad_inside_tests", "shades")
        uc = dict(zip(names, [int(v) for v in uctr[:7].tolist()]))
        uc.update(zip(("shade_lambertian", "shade_metal", "shade_dielectric", "shade_light"), [int(v) for v in uctr[12:16].tolist()]))
        plan = renderer.launch_plan(cam, scene) if hasattr(renderer, "launch_plan") else None
        walk = plan["walk"] if plan else 0
        info = scene.info()
        useful, parts, box_cost = useful_lane_instructions(uc, walk, info["num_quads"], info["num_spheres"], table)
        per_ray = useful / max(uc["rays"], 1)
        useful_rate = per_ray * rays_per_launch / (avg_ms * 1e-3) / 1e12             # T lane-instructions / s
        roofline["useful_frac"] = round(useful_rate / (VALU_PEAK_GINST * 64.0 / 1e3), 4)
        roofline["useful"] = {"lane_instructions_per_ray": round(per_ray, 1), "achieved_T_per_s": round(useful_rate, 2),
                              "per_ray_by_event": {k: round(v / max(uc["rays"], 1), 1) for k, v in parts.items()},
                              "box_tests_per_ray": round(uc["node_tests"] / max(uc["rays"], 1), 2), "valu_per_box_test": box_cost,
                              "events_from": "one untimed step of the counting kernel on the culling tree (collect_stats = 2)",
                              "costs_from": "profiles/isa_event_costs.json (tools/isa_event_costs.py: gfx950 ISA of the product's device functions)",
                              "is": "lane-instructions the path's arithmetic needs (events x per-event VALU count) per second / peak lane-slots; "
                                    "frac counts every lane-instruction ISSUED instead"}
        if roofline.get("valu_wave_insts_per_ray") and roofline.get("mean_active_lanes"):
            issued = roofline["valu_wave_insts_per_ray"] * roofline["mean_active_lanes"]
            roofline["useful"]["issued_lane_instructions_per_ray"] = round(issued, 1)
            roofline["useful"]["useful_over_issued"] = round(per_ray / issued, 3)
    elif table is None or table_stale:
        warnings.append("profiles/isa_event_costs.json is missing or was made from other kernel sources: no useful_frac (python tools/isa_event_costs.py)")
    roofline["warnings"] = warnings
    # per-rank summary on rank 0 (N > 1): launch time and rays of every rank
    per_rank = None
    if world_size > 1:
        mine = torch.tensor([avg_ms or 0.0, float(n_launch), float(my_rays), float(rows_local), roofline["frac"] or 0.0], dtype=torch.float64, device=dev)
        table = [torch.zeros_like(mine) for _ in range(world_size)]
        if rehearsal:
            host = [t.cpu() for t in table]
            dist.all_gather(host, mine.cpu())
            table = host
        else:
            dist.all_gather(table, mine)
        per_rank = [{"rank": r, "avg_launch_ms": round(float(t[0]), 4), "launches": int(t[1]), "rays": int(t[2]), "image_rows": int(t[3]),
                     "frac": round(float(t[4]), 4) if float(t[4]) > 0 else None} for r, t in enumerate(table)]

    # ---- outside the timed region, rank 0 at N = 1 only: the CPU oracle on this box's host cores, a pixel-for-pixel parity check of the
    # GPU path against the frame that leg renders, and short runs of the other two BASELINE scenes.  None of it can end the run.
    cpu = parity = None
    other = []
    if rank == 0 and world_size == 1 and args.cpu_seconds > 0:
        cpu, cpu_frame, cpu_spp, cpu_rays = cpu_baseline(trt, desc, args.depth, args.cpu_seconds)
        try:
            import numpy as np
            pctr = torch.zeros(16, dtype=torch.int64, device=dev)
            pacc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            renderer.render_device(cam, scene, pacc.data_ptr(), stream.cuda_stream, pctr.data_ptr(), sample_begin=0, sample_end=cpu_spp, accumulate=0)
            torch.cuda.synchronize()
            gpu_frame = pacc.cpu().numpy()
            same = bool(np.array_equal(gpu_frame.view(np.uint32), np.ascontiguousarray(cpu_frame, np.float32).view(np.uint32)))
            with np.errstate(invalid="ignore"):
                delta = float(np.nanmax(np.abs(gpu_frame - cpu_frame)))
            parity = {"against": "the oracle frame of the cpu_baseline leg (oracle/rt_oracle.c, the CPU restatement of cpu.rs:39-65 / imager.rs:50)",
                      "workload": f"{args.scene} {W}x{H}, depth {args.depth}, samples [0, {cpu_spp}) of 4096, seed 1, whole frame",
                      "spp": cpu_spp, "pixels": W * H, "bit_identical": same, "max_abs_delta": delta, "tolerance": 1e-3,
                      "rays_gpu": int(pctr[1].item()), "rays_oracle": int(cpu_rays), "ray_counts_equal": int(pctr[1].item()) == int(cpu_rays)}
            del pacc, gpu_frame
        except Exception as e:                                   # noqa: BLE001 - the bench line must still be printed
            parity = {"error": repr(e)}
    if rank == 0 and world_size == 1 and args.cpu_seconds > 0 and not args.no_other_scenes:
        for name, ospheres, ow, oh, ospp, osteps, owarm in OTHER_SCENES:
            if name == args.scene and (ow, oh) == (W, H):
                continue
            try:
                other.append(short_run(trt, torch, dev, name, ospheres, ow, oh, ospp, osteps, owarm, args.depth, args.backend, backend))
            except Exception as e:                               # noqa: BLE001
                other.append({"workload": name, "error": repr(e)})

    if rank == 0:
        if frame is not None:
            assert tuple(frame.shape) == (H, W, 3)
        out = {
            "metric": "Mray/s (primary+secondary)", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mray/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, depth {args.depth}, {S} spp per step (of 4096), "
                                   f"{args.backend}, reference-order BVH, seed 1",
                       "rays": total_rays, "samples": total_samples, "image_rows_per_gpu": rows_local,
                       "scene_build_s": round(t_scene, 2), "scene": scene.info(), "tuning": args.tuning or "default",
                       "parallelism": (f"image bands x{world_size}" + (" (REHEARSAL: all ranks on cuda:0, gloo)" if rehearsal else ""))
                                      if world_size > 1 else "single GPU",
                       "world_size": dist.get_world_size() if world_size > 1 else 1,          # what the ranks saw, not what was asked for
                       "collective_backend": (dist.get_backend() if world_size > 1 else None),
                       "launched_by": "bench.py itself (child torch.distributed.run)" if os.environ.get("TRT_BENCH_SELF_LAUNCHED") == "1"
                                      else ("external launcher" if world_size > 1 else "plain")},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "other_scenes": other,
        }
        if per_rank is not None:
            out["roofline_per_rank"] = per_rank
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
