#!/bin/bash
# Round 4, final validation A: the whole GPU suite, the knob suite (one process per environment knob + the C++-loop build), two fuzz campaigns.
out=gpurun_out/r4; mkdir -p $out
timeout -k 10 900 python3 -X faulthandler -m pytest tests -x -q -m gpu > $out/gputests_final.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 $out/gputests_final.log)"; [ $rc -ne 0 ] && { tail -60 $out/gputests_final.log; exit $rc; }
bash tools/test_knobs.sh > $out/knobs_final.txt 2>&1; rc=$?; tail -25 $out/knobs_final.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tests/fuzz_campaign.py 240 7000 > $out/fuzz_default.txt 2>&1; rc=$?; tail -1 $out/fuzz_default.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tests/fuzz_campaign.py 120 9000 '{"dual_walk": 1, "stream_waves_per_simd": 6, "stragglers": 16}' > $out/fuzz_dual.txt 2>&1; rc=$?; tail -1 $out/fuzz_dual.txt; exit $rc
