run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
export TRT_SCRATCH_CAP_MB=60000
for rep in 1 2; do
for g in 4 8 16; do echo "cornell TRT_RADIANCE_GB=$g: $(TRT_RADIANCE_GB=$g run)"; done
for g in 4 8; do echo "random_spheres TRT_RADIANCE_GB=$g: $(TRT_RADIANCE_GB=$g run --scene random_spheres --width 1920 --height 1080 --steps 4 --warmup 1)"; done
done
