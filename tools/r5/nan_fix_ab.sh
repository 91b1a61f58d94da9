#!/bin/bash
# round 5: the NaN-ray shortcut (rt_path.h ray_has_nan).  Parity first, then a same-box A/B of round 4's library (build/libtinyrt_r4.so)
# against HEAD on the three bench scenes, then the field scenes again.  gpurun -- bash tools/r5/nan_fix_ab.sh
export TMPDIR=/tmp
out=gpurun_out/r5/nan_fix; mkdir -p $out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py tests/test_gpu_streamed.py tests/test_gpu_fuzz.py -x -q -m gpu > $out/parity.log 2>&1 || { echo "PARITY FAILED"; tail -30 $out/parity.log; exit 1; }
tail -2 $out/parity.log
bash tools/ab_libs.sh $PWD/build/libtinyrt_r4.so $PWD/tiny-raytracer_amd/libtinyrt.so 3 2>&1 | tee $out/ab.txt
run() { tag=$1; shift; python3 bench.py --scene sphere_field --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $out/$tag.json 2> $out/$tag.err || { echo "$tag FAILED"; tail -5 $out/$tag.err; return 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', d['value'], 'Mray/s', d['ms_per_step'], 'ms/step', r['kernel'], r['avg_launch_ms'], 'ms/launch', 'build', d['config']['scene_build_s'], 's', d['config']['scene']['device_bytes'], 'B', 'alg B/ray', (r.get('algorithmic') or {}).get('bytes_per_ray'))"; }
run n1m_default --spheres 1000000 &&
run n4m_default --spheres 4000000 &&
run n4m_dual5 --spheres 4000000 --tuning dual_walk=1,stream_waves_per_simd=5 &&
run n4m_dual6 --spheres 4000000 --tuning dual_walk=1,stream_waves_per_simd=6 &&
run n16m_default --spheres 16000000 &&
run n16m_dual6 --spheres 16000000 --tuning dual_walk=1,stream_waves_per_simd=6
