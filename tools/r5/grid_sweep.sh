#!/bin/bash
# round 5: the 100 k-sphere scene (BASELINE configs[4]) after the NaN-ray shortcut - the scheduling knobs of its walk swept again on one box
# (rounds 2-4 tuned them with ~70 whole-tree walks per launch stalling waves at random).  gpurun -- bash tools/r5/grid_sweep.sh
out=gpurun_out/r5/grid_sweep; mkdir -p $out
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
run() { timeout -k 10 300 python3 bench.py $G "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s %8.2f ms  %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"; }
{
echo "default (one path, 8 waves, 3 slots, 8 stragglers): $(run)"
echo "default again: $(run)"
for st in 0 4 12 16 24 32; do echo "stragglers=$st: $(run --tuning stragglers=$st)"; done
for sl in 2 4 5; do echo "leaf_slots=$sl: $(run --tuning leaf_slots=$sl)"; done
for w in 6 7; do echo "waves=$w (sample kernel): $(run --tuning stream_waves_per_simd=$w)"; done
for w in 4 5 6 7 8; do for st in 8 16; do echo "dual w=$w stragglers=$st: $(run --tuning dual_walk=1,stream_waves_per_simd=$w,stragglers=$st)"; done; done
for bs in 4 16; do echo "batch_spp=$bs: $(run --tuning stream_batch_spp=$bs)"; done
echo "default again: $(run)"
} 2>&1 | tee $out/sweep.txt
