#!/bin/bash
# round 5: two paths per lane (stream_dual_kernel) with the fused, mixed-precision box step against the one-path kernel (default plan), one box.
out=gpurun_out/r5/dualmix; mkdir -p $out
L=$PWD/build/libtinyrt_dualmix.so
TRT_LIB_PATH=$L TRT_DUAL_WALK=1 TRT_STREAM_MINW=6 timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "cfg5 or global_memory or fuzz or fused_slab" > $out/parity.log 2>&1 || { echo "PARITY FAILED"; tail -30 $out/parity.log; exit 1; }
tail -1 $out/parity.log
run() { TRT_LIB_PATH=$L timeout -k 10 400 python3 bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f %s' % (d['value'], d['roofline']['kernel'][5:]))"; }
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1"
F1="--scene sphere_field --spheres 1000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
F4="--scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
{
echo "one path (default plan): 100k $(run $G)   field1M $(run $F1)   field4M $(run $F4)"
for w in 4 5 6 7; do for st in 8 16; do
  echo "two paths, $w waves, stragglers $st: 100k $(run $G --tuning dual_walk=1,stream_waves_per_simd=$w,stragglers=$st)   field1M $(run $F1 --tuning dual_walk=1,stream_waves_per_simd=$w,stragglers=$st)   field4M $(run $F4 --tuning dual_walk=1,stream_waves_per_simd=$w,stragglers=$st)"
done; done
echo "one path again: 100k $(run $G)   field1M $(run $F1)   field4M $(run $F4)"
} 2>&1 | tee $out/ab.txt
