#!/bin/bash
# round 5, last look at HEAD on a GPU box: smoke(), then the default bench command.   gpurun --timeout 900 -- bash tools/r5/last_check.sh
out=gpurun_out/r5/last; mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -20 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
python3 bench.py > $out/bench.json 2> $out/bench.err || { echo "BENCH FAILED"; tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r.get('useful_frac'), 'stale', r.get('pmc_stale'), 'parity', d['parity']['bit_identical'], [(o['value'], o['roofline'].get('pmc_stale')) for o in d['other_scenes']])"
