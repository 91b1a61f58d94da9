#!/bin/bash
# TRT_FLAT_STRAGGLERS: lanes that may carry the rest of their final leaf phase into the next round (lock-step leaf list, Cornell), one box
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --spp-per-step 64 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do for s in 0 2 4 6 8 12 16; do echo "flat stragglers $s: cornell $(TRT_FLAT_STRAGGLERS=$s run) Mray/s"; done; done
