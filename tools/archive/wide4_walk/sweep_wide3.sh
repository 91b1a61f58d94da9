#!/bin/bash
# (a) what perfect ray coherence would buy on the 100 k-sphere scene: primary rays only (depth 1) against the full path (depth 50);
# (b) the 4-wide walk on a scene ten times larger (1 M spheres: the tree no longer fits one XCD's L2), against the 16-byte-node walk.
run() { timeout -k 10 600 python bench.py --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s' % d['value'])"; }
for rep in 1 2; do
echo "100k binary depth 50: $(TRT_WIDE_WALK=0 run)"
echo "100k binary depth 1 (primary rays only): $(TRT_WIDE_WALK=0 run --depth 1)"
echo "100k wide6  depth 1 (primary rays only): $(TRT_STREAM_MINW=6 run --depth 1)"
echo "1M binary: $(TRT_BENCH_SPHERES=1000000 TRT_WIDE_WALK=0 run)"
echo "1M wide6:  $(TRT_BENCH_SPHERES=1000000 TRT_STREAM_MINW=6 run)"
echo "1M wide4:  $(TRT_BENCH_SPHERES=1000000 TRT_STREAM_MINW=4 run)"
done
