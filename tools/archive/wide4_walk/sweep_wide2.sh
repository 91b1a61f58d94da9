run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mray/s' % d['value'])"; }
for rep in 1 2; do
echo "binary: $(TRT_WIDE_WALK=0 run)"; for w in 4 5 6; do echo "wide bypass w=$w: $(TRT_STREAM_MINW=$w run)"; done
done
echo "phase split, wide w=6:"; TRT_STREAM_MINW=6 TRT_LIB_PATH=$PWD/build/libtinyrt_clock.so python3 tools/phase_clock.py sphere_grid100k
echo "phase split, binary:"; TRT_WIDE_WALK=0 TRT_LIB_PATH=$PWD/build/libtinyrt_clock.so python3 tools/phase_clock.py sphere_grid100k
