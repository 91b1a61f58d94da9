#!/bin/bash
# Traversal schedule sweep on one box: TRT_LEAF_SLOTS (1..3 postponed-leaf slots) x TRT_LEAF_SERVE (1..63 served, 64 plain).
out=gpurun_out/serve.txt; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
for v in ${COMBOS:-1,64 2,64 3,64 1,6 2,6 2,16 2,64}; do
  sl=${v%,*}; sv=${v#*,}
  export TRT_LEAF_SLOTS=$sl TRT_LEAF_SERVE=$sv
  echo "slots $sl serve $sv: cornell $(run --scene cornell --spp-per-step 64 --steps 2 --warmup 1) | random_spheres $(run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 2 --warmup 1) | grid $(run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --warmup 1)" | tee -a $out
done
