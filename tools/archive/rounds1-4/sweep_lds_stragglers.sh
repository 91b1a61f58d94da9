#!/bin/bash
# TRT_LDS_STRAGGLERS: lanes of a wave that may carry an unfinished LDS tree walk into the next round (0 = every walk runs to its end).
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do for t in 0 2 4 6 8 12 16 24; do echo "random_spheres TRT_LDS_STRAGGLERS=$t: $(TRT_LDS_STRAGGLERS=$t run)"; done; done
