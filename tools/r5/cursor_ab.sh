#!/bin/bash
# round 5: byte-offset cursor in the 16-byte-node walks (21 vector instructions per trip) + the fused loop's domain check in the general kernels:
# the whole GPU suite, then a same-box A/B against the build before it.   gpurun --timeout 1200 -- bash tools/r5/cursor_ab.sh
out=gpurun_out/r5/cursor; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/suite.log 2>&1 || { echo "SUITE FAILED"; tail -40 $out/suite.log; exit 1; }
tail -2 $out/suite.log
run() { TRT_LIB_PATH=$1 timeout -k 10 400 python3 bench.py --cpu-seconds 0 --no-roofline-pass "${@:2}" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1"
F1="--scene sphere_field --spheres 1000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
F4="--scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1"
{
for rep in 1 2 3; do for lib in build/libtinyrt_head.so tiny-raytracer_amd/libtinyrt.so; do L=$PWD/$lib
  echo "$(basename $lib): sphere_grid $(run $L $G)  field1M $(run $L $F1)  field4M $(run $L $F4)"
done; done
} 2>&1 | tee $out/ab.txt
