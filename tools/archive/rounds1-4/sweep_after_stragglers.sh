run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1"
for rep in 1 2; do
echo "rs default: $(run $R)"
for s in 3 4 5; do echo "rs TRT_LEAF_SLOTS=$s: $(TRT_LEAF_SLOTS=$s run $R)"; done
echo "rs TRT_BIG_THREADS=512: $(TRT_BIG_THREADS=512 run $R)"
for p in 0.4 0.5 0.6; do echo "rs TRT_CULL_PRUNE=$p: $(TRT_CULL_PRUNE=$p run $R)"; done
echo "grid default: $(run $G)"
for s in 3 4 5 6; do echo "grid TRT_LEAF_SLOTS=$s: $(TRT_LEAF_SLOTS=$s run $G)"; done
for p in 0.4 0.5; do echo "grid TRT_CULL_PRUNE=$p: $(TRT_CULL_PRUNE=$p run $G)"; done
for w in 6 7 8; do echo "grid TRT_STREAM_MINW=$w: $(TRT_STREAM_MINW=$w run $G)"; done
done
