#!/bin/bash
# round 5, final: the driver's own command plain and under rocprofv3 --kernel-trace --stats, then the knob suite (every fallback path and scheduling
# knob + the C++-loops build through the GPU parity suite).   gpurun --timeout 1200 -- bash tools/r5/final_b.sh
export TMPDIR=/tmp
keep=gpurun_out/prof_keep; mkdir -p $keep gpurun_out/r5/final
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $keep/r05_driver_command_bench.json 2> gpurun_out/r5/final/driver.err || { echo "driver command failed"; tail -5 gpurun_out/r5/final/driver.err; exit 1; }
python3 -c "import json; d=json.loads(open('$keep/r05_driver_command_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('driver command:', d['value'], 'Mray/s', d['ms_per_step'], 'ms/step frac', r['frac'], 'useful_frac', r.get('useful_frac'), 'parity', d['parity']['bit_identical'], [ (o['workload'][:24], o['value']) for o in d['other_scenes']])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/final/trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $keep/r05_driver_command_bench_under_rocprof.json 2> gpurun_out/r5/final/trace.err || { echo "trace failed"; tail -5 gpurun_out/r5/final/trace.err; exit 1; }
cp gpurun_out/r5/final/trace/*/*_kernel_stats.csv $keep/r05_driver_command_kernel_stats.csv && head -6 $keep/r05_driver_command_kernel_stats.csv | cut -c1-220
bash tools/test_knobs.sh > gpurun_out/r5/final/knobs.txt 2>&1; rc=$?; tail -30 gpurun_out/r5/final/knobs.txt; exit $rc
