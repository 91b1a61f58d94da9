run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 3 --warmup 1"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1"
for rep in 1 2; do
for pr in 0.3 0.4 0.5 0.7; do for sl in 4 5; do echo "random_spheres prune $pr slots $sl: $(TRT_CULL_PRUNE=$pr TRT_LEAF_SLOTS=$sl run $R)"; done; done
for pr in 0.3 0.4 0.5 0.7; do echo "grid prune $pr: $(TRT_CULL_PRUNE=$pr run $G)"; done
done
