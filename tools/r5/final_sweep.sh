#!/bin/bash
# round 5: the scheduling knobs of the 16-byte-node walks swept once more at the final loop (21 vector instructions per trip, 7 waves / two paths at 6):
# the optimum of rounds 2-4 belonged to a loop with 37.   gpurun --timeout 1200 -- bash tools/r5/final_sweep.sh
out=gpurun_out/r5/final_sweep; mkdir -p $out
C="--steps 3 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 $C"
F="--scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 $C"
run() { timeout -k 10 300 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s %8.2f ms  %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"; }
{
echo "== 100 k spheres (one path per lane, 7 waves)"
echo "default: $(run $G)"
for st in 4 6 10 12 16; do echo "stragglers=$st: $(run $G --tuning stragglers=$st)"; done
echo "default: $(run $G)"
for sl in 2 4 5 6 8; do echo "leaf_slots=$sl: $(run $G --tuning leaf_slots=$sl)"; done
for bs in 2 4 16; do echo "batch_spp=$bs: $(run $G --tuning stream_batch_spp=$bs)"; done
echo "default: $(run $G)"
echo "leaf_slots=5,batch_spp=4: $(run $G --tuning leaf_slots=5,stream_batch_spp=4)"
echo "leaf_slots=5,batch_spp=4,stragglers=6: $(run $G --tuning leaf_slots=5,stream_batch_spp=4,stragglers=6)"
echo "== sphere_field 4 M (two paths per lane, 6 waves)"
echo "default: $(run $F)"
for st in 4 12 16 24; do echo "stragglers=$st: $(run $F --tuning stragglers=$st)"; done
for sl in 2 4 5 6; do echo "leaf_slots=$sl: $(run $F --tuning leaf_slots=$sl)"; done
for bs in 2 4 16; do echo "batch_spp=$bs: $(run $F --tuning stream_batch_spp=$bs)"; done
echo "default: $(run $F)"
} 2>&1 | tee $out/sweep.txt
