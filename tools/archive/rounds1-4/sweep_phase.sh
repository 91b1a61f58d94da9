#!/bin/bash
# Phase-schedule sweep on one box: thresholds "S,L,G" x waves per SIMD, against the round schedule (no TRT_PHASE_THR).
out=gpurun_out/phase.txt; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.2f ms' % (d['value'], d['ms_per_step']))"; }
SC="${SCENE_ARGS:---scene cornell --spp-per-step 64 --steps 2 --warmup 1}"
echo "round schedule: $(run $SC)" | tee -a $out
for thr in ${THRS:-40,16,16 32,16,16 48,16,16 40,8,16 40,24,16 40,16,32 24,16,16}; do
  for w in ${WAVES:-7 6 5}; do
    echo "thr $thr waves $w: $(TRT_PHASE_THR=$thr TRT_STREAM_MINW=$w run $SC)" | tee -a $out
  done
done
echo "round schedule: $(run $SC)" | tee -a $out
