#!/bin/bash
# round 5: the default bench line (no flags) with useful_frac for every scene; Cornell's waves / slots once more after TRT_SLAB_MED3; a soak of the host layer.
out=gpurun_out/r5/final; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { echo "bench failed"; tail -5 $out/bench_default.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default bench:', d['value'], 'Mray/s frac', r['frac'], 'useful', r.get('useful_frac'))
for o in d['other_scenes']: print('  ', o['workload'][:50], o['value'], o['roofline']['frac'], (o['roofline'].get('useful') or {}).get('useful_frac'), (o['roofline'].get('useful') or {}).get('useful_over_issued'), (o['roofline'].get('useful') or {}).get('error'))
"
run() { timeout -k 10 300 python3 bench.py --cpu-seconds 0 --no-roofline-pass --steps 20 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s' % d['value'])"; }
{
echo "cornell default (6 waves, 7 slots): $(run)"
echo "cornell 7 waves: $(run --tuning stream_waves_per_simd=7)"
echo "cornell 8 waves: $(run --tuning stream_waves_per_simd=8)"
echo "cornell 6 waves, 6 slots: $(run --tuning leaf_slots=6)"
echo "cornell 6 waves, 8 slots: $(run --tuning leaf_slots=8)"
echo "cornell batch_spp 4: $(run --tuning stream_batch_spp=4)"
echo "cornell batch_spp 16: $(run --tuning stream_batch_spp=16)"
echo "cornell default again: $(run)"
} 2>&1 | tee $out/cornell_sweep.txt
timeout -k 10 400 python3 tools/soak_concurrency.py 240 > $out/soak.txt 2>&1; tail -3 $out/soak.txt
