#!/bin/bash
# Round 4: the whole GPU suite, the driver's bench command (with the parity and other_scenes blocks), then PMC passes of the 100 k-sphere
# scene with one and with two paths per lane.  Stops at the first failing step.
out=gpurun_out/r4; mkdir -p $out
timeout -k 10 1100 python3 -X faulthandler -m pytest tests -x -q -m gpu > $out/gputests_abi3.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 $out/gputests_abi3.log)"; [ $rc -ne 0 ] && { tail -60 $out/gputests_abi3.log; exit $rc; }
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_abi3.json 2> $out/bench_driver_abi3.err; rc=$?
echo "bench rc=$rc"; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_driver_abi3.json").read().strip().splitlines()[-1])
print("cornell", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["warnings"])
print("parity", d["parity"])
for o in d["other_scenes"]: print(o.get("workload"), o.get("value"), o.get("ms_per_step"), o.get("roofline", {}).get("avg_launch_ms"), o.get("roofline", {}).get("frac"), o.get("error"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
[ $rc -ne 0 ] && { tail -20 $out/bench_driver_abi3.err; exit $rc; }
bash tools/pmc_bench.sh r04_grid100k_one_path --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $out/pmc_grid_one.log 2>&1 || { tail -5 $out/pmc_grid_one.log; exit 1; }
tail -1 $out/pmc_grid_one.log | cut -c1-400
TRT_DUAL_WALK=1 TRT_STREAM_MINW=6 TRT_STRAGGLERS=16 bash tools/pmc_bench.sh r04_grid100k_two_paths_w6 --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $out/pmc_grid_dual.log 2>&1 || { tail -5 $out/pmc_grid_dual.log; exit 1; }
tail -1 $out/pmc_grid_dual.log | cut -c1-400
