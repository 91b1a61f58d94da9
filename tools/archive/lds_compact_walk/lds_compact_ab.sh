#!/bin/bash
# round 5: the 16-byte-node walk with the fused box step over an LDS image (WALK_LDS_COMPACT; random-spheres' class of scenes): the GPU suite, then a
# same-box A/B against the 32-byte-node LDS walk (TRT_COMPACT_NODES=0: the scene is compiled without the 16-byte nodes and the plan falls back).
#   gpurun --timeout 1200 -- bash tools/r5/lds_compact_ab.sh
out=gpurun_out/r5/lds_compact; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/suite.log 2>&1 || { echo "SUITE FAILED"; tail -40 $out/suite.log; exit 1; }
tail -2 $out/suite.log
run() { timeout -k 10 400 python3 bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %7.2f ms %s' % (d['value'], d['roofline']['avg_launch_ms'], d['config'].get('launch_plan', {}).get('walk', '')))"; }
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 256 --steps 3 --warmup 1"
{
for rep in 1 2 3; do
  echo "32-byte nodes in LDS (box_loop_lds):         $(TRT_COMPACT_NODES=0 run $R)"
  echo "16-byte nodes in LDS (box_loop_compact_lds): $(run $R)"
done
for st in 4 12 16; do echo "16-byte nodes, lds_stragglers=$st: $(run $R --tuning lds_stragglers=$st)"; done
for sl in 3 4; do echo "16-byte nodes, leaf_slots=$sl: $(run $R --tuning leaf_slots=$sl)"; done
} 2>&1 | tee $out/ab.txt
