#!/bin/bash
# Round 3, after the hand-written box loops: do the scheduling defaults still sit at their optimum?  (same box, two repetitions)
run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1"
for rep in 1 2; do
echo "rs default: $(run $R)"
for t in 0 4 12 16 24; do echo "rs TRT_LDS_STRAGGLERS=$t: $(TRT_LDS_STRAGGLERS=$t run $R)"; done
for s in 3 4 6 7; do echo "rs TRT_LEAF_SLOTS=$s: $(TRT_LEAF_SLOTS=$s run $R)"; done
echo "rs TRT_BIG_THREADS=512: $(TRT_BIG_THREADS=512 run $R)"
for p in 0.35 0.7; do echo "rs TRT_CULL_PRUNE=$p: $(TRT_CULL_PRUNE=$p run $R)"; done
echo "grid default: $(run $G)"
for t in 4 12 16 24; do echo "grid TRT_STRAGGLERS=$t: $(TRT_STRAGGLERS=$t run $G)"; done
for s in 2 4 5; do echo "grid TRT_LEAF_SLOTS=$s: $(TRT_LEAF_SLOTS=$s run $G)"; done
done
