#!/bin/bash
# random-spheres: workgroup shapes for the 49.6 KB scene copy, one box
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do
  echo "512 lanes, register slots: $(TRT_BIG_THREADS=512 run)"
  echo "768 lanes, LDS stack:      $(TRT_BIG_THREADS=768 run)"
done
