#!/bin/bash
# stream_gen_kernel (a wave traces one batch generation by generation) against the product's pool kernel, scenes in global memory, one box.
run() { timeout -k 10 600 python bench.py --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s  %s' % (d['value'], d['roofline']['kernel']))"; }
for rep in 1 2; do
echo "100k pool kernel (product):        $(run)"
echo "100k generations, 8 waves, batch 8:  $(TRT_GENERATIONS=1 run)"
echo "100k generations, 8 waves, batch 16: $(TRT_GENERATIONS=1 TRT_STREAM_BATCH_SPP=16 run)"
echo "100k generations, 8 waves, batch 4:  $(TRT_GENERATIONS=1 TRT_STREAM_BATCH_SPP=4 run)"
echo "100k generations, 6 waves, batch 8:  $(TRT_GENERATIONS=1 TRT_STREAM_MINW=6 run)"
echo "100k generations, 6 waves, batch 16: $(TRT_GENERATIONS=1 TRT_STREAM_MINW=6 TRT_STREAM_BATCH_SPP=16 run)"
done
echo "1M pool kernel:                    $(TRT_BENCH_SPHERES=1000000 run)"
echo "1M generations, 8 waves, batch 16: $(TRT_BENCH_SPHERES=1000000 TRT_GENERATIONS=1 TRT_STREAM_BATCH_SPP=16 run)"
