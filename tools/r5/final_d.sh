#!/bin/bash
# round 5, last validation at the final sources: seeded fuzz campaigns against the oracle - default plan, two paths per lane everywhere, 8 waves.
out=gpurun_out/r5/final; mkdir -p $out
{
python3 tests/fuzz_campaign.py 240 8000
python3 tests/fuzz_campaign.py 200 9000 '{"dual_walk": 1, "stragglers": 12}'
python3 tests/fuzz_campaign.py 100 10000 '{"dual_walk": 1, "stream_waves_per_simd": 6, "leaf_slots": 3}'
python3 tests/fuzz_campaign.py 100 11000 '{"dual_walk": 2, "stream_waves_per_simd": 8}'
} 2>&1 | grep --line-buffered "fuzz campaign:" | tee $out/fuzz2.txt
