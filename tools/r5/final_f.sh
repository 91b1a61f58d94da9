#!/bin/bash
# round 5, final sources: three more fuzz campaigns - the general (run-time walk) kernels, which reach the fused loop through closest_hit's run-time
# domain check; the build with the C++ box-step loops; the default plan on new seeds.   gpurun --timeout 1200 -- bash tools/r5/final_f.sh
out=gpurun_out/r5/final_f; mkdir -p $out
{
python3 tests/fuzz_campaign.py 150 12000 '{"runtime_walk": 1}'
TRT_LIB_PATH=$PWD/build/libtinyrt_cxxloops.so python3 tests/fuzz_campaign.py 120 13000
python3 tests/fuzz_campaign.py 240 14000
} 2>&1 | grep --line-buffered "fuzz campaign:" | tee $out/fuzz3.txt
