#!/bin/bash
# round 5: the lock-step box test with its six subtractions as three v_pk_add_f32 on the planes' SGPR pairs (20 vector instructions per box instead of 23):
# the GPU suite, then a same-box A/B on the bench workload against the build before it.   gpurun --timeout 1200 -- bash tools/r5/flat_pk_ab.sh
out=gpurun_out/r5/flat_pk; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/suite.log 2>&1 || { echo "SUITE FAILED"; tail -40 $out/suite.log; exit 1; }
tail -2 $out/suite.log
run() { TRT_LIB_PATH=$1 timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --steps 20 --warmup 3 "${@:2}" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %7.2f ms' % (d['value'], d['roofline']['avg_launch_ms']))"; }
{
for rep in 1 2 3 4; do for lib in build/libtinyrt_head.so tiny-raytracer_amd/libtinyrt.so; do
  echo "$(basename $lib): cornell 2048x2048 $(run $PWD/$lib)"
done; done
for lib in build/libtinyrt_head.so tiny-raytracer_amd/libtinyrt.so; do echo "$(basename $lib): cornell at 7 waves $(run $PWD/$lib --tuning stream_waves_per_simd=7)"; done
} 2>&1 | tee $out/ab.txt
