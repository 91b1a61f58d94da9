#!/bin/bash
# round 5, final sources: the host layer's concurrency soak (4 minutes), then random-spheres' scheduling knobs once more.   gpurun --timeout 900 -- bash tools/r5/final_e.sh
out=gpurun_out/r5/final_e; mkdir -p $out
timeout -k 10 400 python3 tools/soak_concurrency.py 240 > $out/soak.txt 2>&1 || { echo "SOAK FAILED"; tail -20 $out/soak.txt; exit 1; }
tail -1 $out/soak.txt
run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --scene random_spheres --width 1920 --height 1080 --spp-per-step 256 --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %7.2f ms' % (d['value'], d['roofline']['avg_launch_ms']))"; }
{
echo "random-spheres default (768 lanes, 6 waves, 5 slots, 8 stragglers): $(run)"
for st in 4 6 12 16; do echo "lds_stragglers=$st: $(run --tuning lds_stragglers=$st)"; done
for sl in 3 4; do echo "leaf_slots=$sl: $(run --tuning leaf_slots=$sl)"; done
echo "512 lanes: $(run --tuning stream_big_threads=512)"
for bs in 4 16; do echo "batch_spp=$bs: $(run --tuning stream_batch_spp=$bs)"; done
echo "default again: $(run)"
} 2>&1 | tee $out/spheres_sweep.txt
