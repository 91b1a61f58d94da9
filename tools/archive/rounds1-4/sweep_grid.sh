#!/bin/bash
# 100 k-sphere scene (scene in HBM/L2): A/B of one environment knob on one box.  VAR=TRT_COMPACT_NODES VALUES="0 1" tools/sweep_grid.sh
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do for v in $VALUES; do export $VAR=$v; echo "$VAR=$v: $(run)" | tee -a gpurun_out/grid_sweep.txt; done; done
