run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f' % d['value'])"; }
for rep in 1 2; do
for s in 5 6 7 8 9 10 12; do echo "cornell MINW=5 TRT_LEAF_SLOTS=$s: $(TRT_STREAM_MINW=5 TRT_LEAF_SLOTS=$s run)"; done
done
