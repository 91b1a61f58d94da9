#!/bin/bash
# Cornell: waves per SIMD of the streamed pool kernel (the LDS leaf stack shrinks to fit: 7 / 5 / 4 slots), one box
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --spp-per-step 64 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do for w in 6 7 8; do echo "waves/SIMD $w: $(TRT_STREAM_MINW=$w run) Mray/s"; done; echo "default: $(run)"; done
