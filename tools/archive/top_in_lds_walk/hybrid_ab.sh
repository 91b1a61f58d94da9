#!/bin/bash
# round 5: the top-in-LDS walk (rt_path.h walk_hybrid) - parity first, then the budget sweep on the 100 k-sphere scene and on sphere_field, one box.
# gpurun -- bash tools/r5/hybrid_ab.sh
export TMPDIR=/tmp
out=gpurun_out/r5/hybrid; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -m gpu > $out/parity.log 2>&1 || { echo "PARITY FAILED"; tail -40 $out/parity.log; exit 1; }
tail -2 $out/parity.log
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
F="--scene sphere_field --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
run() { timeout -k 10 400 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s %8.2f ms  %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"; }
{
echo "100 k spheres, plain walk (top_nodes=0): $(TRT_TOP_NODES=0 run $G)"
for top in 64 320 640 1280; do for burst in 1 2 4 8; do echo "100 k spheres, top_nodes=$top burst=$burst: $(TRT_TOP_NODES=$top TRT_TOP_BURST=$burst run $G)"; done; done
echo "100 k spheres, plain walk again: $(TRT_TOP_NODES=0 run $G)"
for st in 4 16; do echo "100 k spheres, top_nodes=640 burst=4 stragglers=$st: $(TRT_TOP_NODES=640 run $G --tuning stragglers=$st)"; done
for n in 1000000 4000000; do for top in 0 320 1280; do echo "sphere_field $n, top_nodes=$top: $(TRT_TOP_NODES=$top run $F --spheres $n)"; done; done
} 2>&1 | tee $out/sweep2.txt
