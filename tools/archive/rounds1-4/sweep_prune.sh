#!/bin/bash
# Culling-tree prune factor sweep (TRT_CULL_PRUNE; 0 = flat leaf list) on one box.
out=gpurun_out/prune.txt; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
SC="${SCENE_ARGS:---scene cornell --spp-per-step 64 --steps 2 --warmup 1}"
for pr in ${PRUNES:-0.7 0.0 0.1 0.2 0.3 0.4 0.5 0.7}; do
  echo "prune $pr: $(TRT_CULL_PRUNE=$pr run $SC)" | tee -a $out
done
