#!/bin/bash
# round 5: v_med3_f32 in the hand-written box-step loops (rt_path.h TRT_SLAB_MED3): parity, then same-box A/B against the build before it.
# gpurun -- bash tools/r5/med3_ab.sh
out=gpurun_out/r5/med3; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/suite.log 2>&1 || { echo "SUITE FAILED"; tail -40 $out/suite.log; exit 1; }
tail -2 $out/suite.log
bash tools/ab_libs.sh $PWD/build/libtinyrt_r5base.so $PWD/build/libtinyrt_med3.so 3 2>&1 | tee $out/ab.txt
