#!/bin/bash
# Round 4: stream_dual_kernel (two paths per lane) at 4..8 waves per SIMD against the one-path pool kernel, 100 k spheres, same box.
out=gpurun_out/r4; mkdir -p $out
for w in 4 6; do
  TRT_DUAL_WALK=1 TRT_STREAM_MINW=$w timeout -k 10 600 python3 -m pytest tests/test_gpu_cfg5.py -x -q -m gpu > $out/dual2_parity_w$w.log 2>&1
  rc=$?; echo "parity dual w=$w rc=$rc $(tail -1 $out/dual2_parity_w$w.log)"
  [ $rc -ne 0 ] && { tail -40 $out/dual2_parity_w$w.log; exit $rc; }
done
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
run() { timeout -k 10 300 python3 bench.py $G 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s %8.2f ms  %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"; }
{
echo "baseline (one path per lane, 8 waves):"; run; run
for w in 4 5 6 7 8; do for st in 4 16; do
  echo "dual w=$w stragglers=$st: $(TRT_DUAL_WALK=1 TRT_STREAM_MINW=$w TRT_STRAGGLERS=$st run)"
done; done
echo "baseline again:"; run
} | tee $out/dual_sweep2.txt
