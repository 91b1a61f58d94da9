#!/bin/bash
# round 5, final validation: the whole GPU suite at HEAD, then seeded fuzz campaigns against the oracle (default tuning; 8 waves for the 16-byte-node walk;
# two paths per lane).   gpurun --timeout 1200 -- bash tools/r5/final_a.sh
out=gpurun_out/r5/final; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/suite.log 2>&1 || { echo "SUITE FAILED"; tail -40 $out/suite.log; exit 1; }
tail -2 $out/suite.log
{
python3 tests/fuzz_campaign.py 300 5000
python3 tests/fuzz_campaign.py 150 6000 '{"stream_waves_per_simd": 8, "stragglers": 3, "leaf_slots": 3}'
python3 tests/fuzz_campaign.py 120 7000 '{"dual_walk": 1, "stragglers": 24}'
} 2>&1 | grep -v "^$" | tee $out/fuzz.txt
