#!/bin/bash
# round 3: straggler thresholds with the exit check INSIDE the box-step loops (random-spheres: TRT_LDS_STRAGGLERS; 100 k spheres: TRT_STRAGGLERS)
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1"
for rep in 1 2; do
for t in 0 4 8 12 16 20 24 32; do echo "random_spheres TRT_LDS_STRAGGLERS=$t: $(TRT_LDS_STRAGGLERS=$t run $R)"; done
for t in 0 4 8 12 16 24; do echo "grid TRT_STRAGGLERS=$t: $(TRT_STRAGGLERS=$t run $G)"; done
done
