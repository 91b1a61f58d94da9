#!/bin/bash
# Round-4 start: the GPU parity suite and the three bench scenes at HEAD on one box.
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4/gputests_start.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4/gputests_start.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_cornell_start.json 2> gpurun_out/r4/bench_cornell_start.err; tail -c 600 gpurun_out/r4/bench_cornell_start.json | cut -c1-300
timeout -k 10 200 python3 bench.py --scene random_spheres --width 1920 --height 1080 --steps 4 --warmup 1 --cpu-seconds 0 > gpurun_out/r4/bench_rs_start.json 2>/dev/null; cut -c1-200 gpurun_out/r4/bench_rs_start.json
timeout -k 10 200 python3 bench.py --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 4 --warmup 1 --cpu-seconds 0 > gpurun_out/r4/bench_grid_start.json 2>/dev/null; cut -c1-200 gpurun_out/r4/bench_grid_start.json
