#!/bin/bash
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do for s in 2 3 4 6 8; do echo "leaf slots $s: $(TRT_LEAF_SLOTS=$s run)"; done; echo "default: $(run)"; done
