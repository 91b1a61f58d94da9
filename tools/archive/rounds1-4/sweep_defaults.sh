#!/bin/bash
# Round 3: are the defaults still the optima at HEAD?  One knob at a time on the scene it matters for, one box, each value twice.
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
C="--scene cornell --spp-per-step 64 --steps 3 --warmup 1"
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1"
sweep() { var=$1; shift; scene=$1; shift; args=$1; shift; for v in "$@"; do a=$(env $var=$v bash -c "$(declare -f run); run $args"); b=$(env $var=$v bash -c "$(declare -f run); run $args"); echo "$scene $var=$v: $a $b"; done; }
echo "defaults: cornell $(run $C) random_spheres $(run $R) grid $(run $G)"
sweep TRT_CULL_PRUNE random_spheres "$R" 0.5 0.6 0.7 0.8 0.9
sweep TRT_CULL_PRUNE grid "$G" 0.5 0.7 0.85 1.0
sweep TRT_LEAF_SLOTS random_spheres "$R" 3 4 5 6
sweep TRT_LEAF_SLOTS grid "$G" 3 4 6 8
sweep TRT_LEAF_SLOTS cornell "$C" 6 7 8 9
sweep TRT_STRAGGLERS grid "$G" 8 12 16 24
sweep TRT_STREAM_BATCH_SPP cornell "$C" 4 8 16
sweep TRT_STREAM_BATCH_SPP random_spheres "$R" 4 8 16
sweep TRT_STREAM_BATCH_SPP grid "$G" 4 8 16
echo "defaults: cornell $(run $C) random_spheres $(run $R) grid $(run $G)"
