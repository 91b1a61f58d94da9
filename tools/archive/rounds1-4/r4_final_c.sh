#!/bin/bash
# Round 4, final measurement B2 (after tools/pmc_collect.py merged B1's counters into profiles/pmc_kernels.json): the driver's exact command, plain and
# under rocprofv3 --kernel-trace --stats, and the two other scenes' own bench lines.
export TMPDIR=/tmp
out=gpurun_out/r4; mkdir -p $out
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04_driver_command_bench.json 2> $out/r04_driver_command_bench.err; echo "driver command rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04_driver_command_bench_under_rocprof.json 2> $out/trace_driver.err || { tail -3 $out/trace_driver.err; exit 1; }
cp $out/trace_driver/*/*_kernel_stats.csv $out/r04_driver_command_kernel_stats.csv
timeout -k 10 300 python3 bench.py --scene random_spheres --width 1920 --height 1080 --steps 4 --warmup 1 --cpu-seconds 0 > $out/r04_spheres1080_bench_line.json 2>/dev/null
timeout -k 10 300 python3 bench.py --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 4 --warmup 1 --cpu-seconds 0 > $out/r04_grid100k_bench_line.json 2>/dev/null
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/r04_driver_command_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("cornell", d["value"], d["ms_per_step"], r["avg_launch_ms"], "frac", r["frac"], "issue", r.get("issue_frac"), "lanes", r.get("mean_active_lanes"), "cyc", r.get("cycles_per_valu_inst_per_simd"), "hbm", r.get("hbm_physical_frac"), r["warnings"])
print("parity", {k: d["parity"][k] for k in ("spp", "bit_identical", "max_abs_delta", "ray_counts_equal")})
for o in d["other_scenes"]: print(o["workload"][:40], o["value"], o["ms_per_step"], o["roofline"]["avg_launch_ms"], "frac", o["roofline"]["frac"], o["roofline"]["mean_active_lanes"], o["roofline"]["warnings"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["single_thread"]["value"])
for f in ("r04_spheres1080_bench_line", "r04_grid100k_bench_line"):
    e = json.loads(open(f"gpurun_out/r4/{f}.json").read().strip().splitlines()[-1]); print(f, e["value"], e["roofline"]["avg_launch_ms"], e["roofline"]["frac"], e["roofline"].get("algorithmic", {}).get("bytes_per_ray"))
PY
head -8 $out/r04_driver_command_kernel_stats.csv | cut -c1-200
