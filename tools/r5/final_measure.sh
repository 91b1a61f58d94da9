#!/bin/bash
# round 5, final measurement: PMC passes + kernel trace of the four bench workloads at HEAD's kernel sources, then the driver's own command plain and
# under rocprofv3 --kernel-trace --stats.   gpurun --timeout 1200 -- bash tools/r5/final_measure.sh
export TMPDIR=/tmp
mkdir -p gpurun_out/r5/final
bash tools/pmc_bench.sh r05_cornell2048 > gpurun_out/r5/final/pmc_cornell.log 2>&1; tail -c 300 gpurun_out/r5/final/pmc_cornell.log; echo
bash tools/pmc_bench.sh r05_spheres1080 --scene random_spheres --width 1920 --height 1080 > gpurun_out/r5/final/pmc_spheres.log 2>&1; tail -c 300 gpurun_out/r5/final/pmc_spheres.log; echo
bash tools/pmc_bench.sh r05_grid100k --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > gpurun_out/r5/final/pmc_grid.log 2>&1; tail -c 300 gpurun_out/r5/final/pmc_grid.log; echo
bash tools/pmc_bench.sh r05_field4m --scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 > gpurun_out/r5/final/pmc_field.log 2>&1; tail -c 300 gpurun_out/r5/final/pmc_field.log; echo
