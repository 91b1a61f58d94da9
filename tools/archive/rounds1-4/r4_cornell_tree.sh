#!/bin/bash
# Round 4: Cornell through the LDS TREE walk (hand-written loop + resumable walks since round 3) against the lock-step leaf list, same box.
run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms  %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"; }
{
echo "lock-step leaf list (default): $(run)"
for st in 0 4 8 12; do for sl in 4 6; do
  echo "tree walk, lds_stragglers=$st leaf_slots=$sl: $(TRT_FLAT_WALK=0 TRT_LDS_STRAGGLERS=$st TRT_LEAF_SLOTS=$sl run)"
done; done
echo "tree walk, defaults: $(TRT_FLAT_WALK=0 run)"
echo "lock-step leaf list (default): $(run)"
} | tee gpurun_out/r4/cornell_tree_walk.txt
