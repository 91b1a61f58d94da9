#!/bin/bash
# 100 k-sphere scene (3840x2160, 16 spp per step): the 4-wide walk at every waves-per-SIMD instantiation and stack split, against the
# 16-byte-node walk, one box, interleaved twice.  Output: gpurun_out/r03_wide_sweep.txt
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %s' % (d['value'], d['roofline']['kernel']))"; }
out=gpurun_out/r03_wide_sweep.txt; : > $out
for rep in 1 2; do
  echo "binary 16-byte nodes (TRT_WIDE_WALK=0):  $(TRT_WIDE_WALK=0 run)" | tee -a $out
  for w in 4 5 6 8; do
    echo "4-wide, $w waves/SIMD:                   $(TRT_STREAM_MINW=$w run)" | tee -a $out
  done
  for cap in 0 8 16; do
    echo "4-wide, 4 waves/SIMD, LDS stack cap $cap:  $(TRT_STREAM_MINW=4 TRT_TRAV_CAP=$cap run)" | tee -a $out
  done
  for st in 0 6 20; do
    echo "4-wide, 4 waves/SIMD, stragglers $st:     $(TRT_STREAM_MINW=4 TRT_STRAGGLERS=$st run)" | tee -a $out
  done
done
