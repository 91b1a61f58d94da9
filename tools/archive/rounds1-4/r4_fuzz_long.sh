#!/bin/bash
# Round 4: longer fuzz campaigns under three tunings (default; 8 waves per SIMD with 3 stragglers; two paths per lane at 8 waves), other seeds than the short runs.
out=gpurun_out/r4; mkdir -p $out
timeout -k 10 500 python3 tests/fuzz_campaign.py 420 200000 > $out/fuzz_long_default.txt 2>&1; rc=$?; tail -1 $out/fuzz_long_default.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tests/fuzz_campaign.py 240 300000 '{"stream_waves_per_simd": 8, "stragglers": 3, "lds_stragglers": 3, "leaf_slots": 3}' > $out/fuzz_long_w8.txt 2>&1; rc=$?; tail -1 $out/fuzz_long_w8.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tests/fuzz_campaign.py 240 400000 '{"dual_walk": 1, "stragglers": 24}' > $out/fuzz_long_dual8.txt 2>&1; rc=$?; tail -1 $out/fuzz_long_dual8.txt; exit $rc
