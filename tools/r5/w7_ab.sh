#!/bin/bash
# round 5: the 16-byte-node pool kernel at 7 and 6 waves per SIMD (72 / 80 VGPRs, no spills) against the shipped 8 (64 VGPRs, 10 spilled), one box.
out=gpurun_out/r5/w7; mkdir -p $out
L=$PWD/build/libtinyrt_w7.so
run() { TRT_LIB_PATH=$L timeout -k 10 400 python3 bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s  %s' % (d['value'], d['roofline']['kernel']))"; }
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1"
F1="--scene sphere_field --spheres 1000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
F4="--scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
{
for rep in 1 2; do for w in 8 7 6; do
  echo "waves=$w: 100k $(run $G --tuning stream_waves_per_simd=$w)   field1M $(run $F1 --tuning stream_waves_per_simd=$w)   field4M $(run $F4 --tuning stream_waves_per_simd=$w)"
done; done
for st in 4 12; do echo "waves=7 stragglers=$st: 100k $(run $G --tuning stream_waves_per_simd=7,stragglers=$st)"; done
for sl in 5 6; do echo "waves=7 leaf_slots=$sl: 100k $(run $G --tuning stream_waves_per_simd=7,leaf_slots=$sl)"; done
} 2>&1 | tee $out/ab.txt
