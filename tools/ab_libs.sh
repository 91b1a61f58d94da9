#!/bin/bash
# Same-box A/B of two builds of libtinyrt.so on the three bench scenes: tools/ab_libs.sh <libA> <libB> [reps]
# (TRT_LIB_PATH picks the build; kernel time only: no CPU leg, no roofline pass.)
A=$1; B=$2; reps=${3:-3}
run() { TRT_LIB_PATH=$1 timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass "${@:2}" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
C="--steps 20 --warmup 3"
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1"
for rep in $(seq $reps); do
  for lib in $A $B; do
    echo "$(basename $lib): cornell $(run $lib $C)  random_spheres $(run $lib $R)  sphere_grid $(run $lib $G)"
  done
done
