#!/bin/bash
# Round 4: the shared BatchCursor / pool helpers (one copy of the work-queue and ray-pool code for the three streamed kernels) against the kernels
# with their own copies (HEAD's libtinyrt.so): parity suite on the new build, then the same-box A/B on the three scenes.
out=gpurun_out/r4; mkdir -p $out
TRT_LIB_PATH=$PWD/build/libtinyrt_refactor.so timeout -k 10 900 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "not full_baseline and not c_example and not bench" > $out/refactor_parity.log 2>&1; rc=$?
echo "parity rc=$rc $(tail -1 $out/refactor_parity.log)"; [ $rc -ne 0 ] && { tail -40 $out/refactor_parity.log; exit $rc; }
bash tools/ab_libs.sh $PWD/tiny-raytracer_amd/libtinyrt.so $PWD/build/libtinyrt_refactor.so 3 | tee $out/refactor_ab.txt
