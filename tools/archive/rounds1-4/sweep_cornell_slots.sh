#!/bin/bash
# Cornell: postponed-leaf slots of the lock-step walk and waves per SIMD, same box, two repetitions.
run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do
echo "cornell default: $(run)"
for s in 2 3 4 5 6 8; do echo "cornell TRT_LEAF_SLOTS=$s: $(TRT_LEAF_SLOTS=$s run)"; done
for w in 5 7; do echo "cornell TRT_STREAM_MINW=$w: $(TRT_STREAM_MINW=$w run)"; done
echo "cornell TRT_STREAM_BATCH_SPP=16: $(TRT_STREAM_BATCH_SPP=16 run)"
echo "cornell TRT_STREAM_BATCH_SPP=4: $(TRT_STREAM_BATCH_SPP=4 run)"
done
