#!/bin/bash
# Round 4, final measurement B1: the PMC passes + kernel trace + plain run of the three BASELINE scenes at HEAD (tools/pmc_bench.sh).
out=gpurun_out/r4; mkdir -p $out
bash tools/pmc_bench.sh r04_cornell2048 > $out/pmc_cornell.log 2>&1 || { tail -5 $out/pmc_cornell.log; exit 1; }
bash tools/pmc_bench.sh r04_spheres1080 --scene random_spheres --width 1920 --height 1080 > $out/pmc_rs.log 2>&1 || { tail -5 $out/pmc_rs.log; exit 1; }
bash tools/pmc_bench.sh r04_grid100k --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $out/pmc_grid.log 2>&1 || { tail -5 $out/pmc_grid.log; exit 1; }
for t in r04_cornell2048 r04_spheres1080 r04_grid100k; do tail -1 $out/pmc_*.log | grep -c $t > /dev/null; done
ls gpurun_out/prof_keep | grep r04_
