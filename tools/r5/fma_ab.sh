#!/bin/bash
# round 5: fused slab arithmetic in the 16-byte-node walk (rt_path.h box_loop_compact) on top of TRT_SLAB_MED3: suite, then a same-box A/B of the three
# builds (build/libtinyrt_r5base.so = before both, _med3.so, _fma.so = HEAD) on the bench scenes and on sphere_field.   gpurun -- bash tools/r5/fma_ab.sh
out=gpurun_out/r5/fma; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/suite.log 2>&1 || { echo "SUITE FAILED"; tail -40 $out/suite.log; exit 1; }
tail -2 $out/suite.log
run() { TRT_LIB_PATH=$1 timeout -k 10 400 python3 bench.py --cpu-seconds 0 --no-roofline-pass "${@:2}" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
C="--steps 20 --warmup 3"
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1"
F1="--scene sphere_field --spheres 1000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
F4="--scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
{
for rep in 1 2 3; do
  for lib in r5base med3 fma; do
    L=$PWD/build/libtinyrt_$lib.so
    echo "$lib: cornell $(run $L $C)  random_spheres $(run $L $R)  sphere_grid $(run $L $G)  field1M $(run $L $F1)  field4M $(run $L $F4)"
  done
done
} 2>&1 | tee $out/ab.txt
