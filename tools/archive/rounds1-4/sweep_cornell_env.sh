#!/bin/bash
# A/B of one environment knob on Cornell only: VAR=name VALUES="a b" tools/sweep_cornell_env.sh
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for v in $VALUES; do export $VAR=$v; echo "$VAR=$v: cornell $(run --scene cornell --spp-per-step 64 --steps 2 --warmup 1)" | tee -a gpurun_out/cornell_env.txt; done; done
