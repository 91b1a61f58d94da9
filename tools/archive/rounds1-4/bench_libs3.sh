#!/bin/bash
# Same-box comparison of several library builds on the three BASELINE scenes: LIBS="base new" -> build/libtinyrt_base.so ...
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do for v in $LIBS; do
  export TRT_LIB_PATH=$PWD/build/libtinyrt_$v.so
  echo "$v: cornell $(run --scene cornell --spp-per-step 64 --steps 3 --warmup 1) | random_spheres $(run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1) | grid100k $(run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1) Mray/s"
done; done
