#!/bin/bash
# TRT_STRAGGLERS: a round's walk phase ends once at most that many lanes still walk (0 = never), random-spheres and the 100 k-sphere scene, one box
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do for s in 0 6 12 20 28; do
  echo "stragglers $s: random_spheres $(TRT_STRAGGLERS=$s run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1) | grid100k $(TRT_STRAGGLERS=$s run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1) Mray/s"
done; done
