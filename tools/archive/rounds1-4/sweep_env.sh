#!/bin/bash
# A/B of one environment knob on one box: VAR=name VALUES="a b c" tools/sweep_env.sh   (three BASELINE scenes, streamed backend)
out=gpurun_out/env_sweep.txt; : > $out
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
for v in $VALUES; do
  export $VAR=$v
  echo "$VAR=$v: cornell $(run --scene cornell --spp-per-step 64 --steps 2 --warmup 1) | random_spheres $(run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 2 --warmup 1) | grid $(run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --warmup 1)" | tee -a $out
done; done
