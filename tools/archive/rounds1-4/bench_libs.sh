#!/bin/bash
# Same-box comparison of several library builds: LIBS="A C D" -> build/libtinyrt_A.so ... (Cornell + random-spheres, streamed)
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s' % d['value'])"; }
for rep in 1 2; do for v in $LIBS; do
  export TRT_LIB_PATH=$PWD/build/libtinyrt_$v.so
  echo "$v: cornell $(run --scene cornell --spp-per-step 64 --steps 2 --warmup 1) | random_spheres $(run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 2 --warmup 1)"
done; done
