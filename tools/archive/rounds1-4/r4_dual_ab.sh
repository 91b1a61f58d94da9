#!/bin/bash
# Round 4: two walks per lane (stream_dual_kernel) against the one-walk pool kernel on the 100 k-sphere scene, same box.
# Parity first (cfg5 + fuzz + global-memory tests under TRT_DUAL_WALK=1), then the waves-per-SIMD / stragglers / slots sweep.
out=gpurun_out/r4; mkdir -p $out
for w in 8 6; do
  TRT_DUAL_WALK=1 TRT_STREAM_MINW=$w timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "cfg5 or global_memory or fuzz" > $out/dual_parity_w$w.log 2>&1
  rc=$?; echo "parity dual w=$w rc=$rc $(tail -1 $out/dual_parity_w$w.log)"
  [ $rc -ne 0 ] && { tail -40 $out/dual_parity_w$w.log; exit $rc; }
done
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
run() { timeout -k 10 300 python3 bench.py $G 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s %8.2f ms  %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"; }
{
echo "baseline (one walk per lane, 8 waves):"; run; run
for w in 5 6 7 8; do for st in 8 16 32; do
  echo "dual w=$w stragglers=$st: $(TRT_DUAL_WALK=1 TRT_STREAM_MINW=$w TRT_STRAGGLERS=$st run)"
done; done
for w in 6 8; do for sl in 2 3; do
  echo "dual w=$w slots=$sl stragglers=16: $(TRT_DUAL_WALK=1 TRT_STREAM_MINW=$w TRT_LEAF_SLOTS=$sl TRT_STRAGGLERS=16 run)"
done; done
echo "baseline again:"; run
} | tee $out/dual_sweep.txt
