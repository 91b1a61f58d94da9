#!/bin/bash
# Round 4: tile-major radiance records against round 3's [sample][pixel] layout (build/libtinyrt_rowmajor.so), same box: parity suite on the
# new layout, A/B on the three scenes, then the PMC passes (WRITE_SIZE, SQ_INSTS_VMEM_WR among them) of the three scenes at HEAD.
out=gpurun_out/r4; mkdir -p $out
timeout -k 10 900 python3 -X faulthandler -m pytest tests -x -q -m gpu -k "not full_baseline and not full_sample_count" > $out/gputests_tile.log 2>&1; rc=$?
echo "pytest rc=$rc $(tail -1 $out/gputests_tile.log)"; [ $rc -ne 0 ] && { tail -60 $out/gputests_tile.log; exit $rc; }
bash tools/ab_libs.sh $PWD/build/libtinyrt_rowmajor.so $PWD/tiny-raytracer_amd/libtinyrt.so 2 | tee $out/tile_ab.txt
bash tools/pmc_bench.sh r04_cornell2048 > $out/pmc_cornell.log 2>&1 || { tail -5 $out/pmc_cornell.log; exit 1; }
bash tools/pmc_bench.sh r04_spheres1080 --scene random_spheres --width 1920 --height 1080 > $out/pmc_rs.log 2>&1 || { tail -5 $out/pmc_rs.log; exit 1; }
bash tools/pmc_bench.sh r04_grid100k --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $out/pmc_grid.log 2>&1 || { tail -5 $out/pmc_grid.log; exit 1; }
TRT_DUAL_WALK=1 TRT_STREAM_MINW=6 TRT_STRAGGLERS=16 bash tools/pmc_bench.sh r04_grid100k_two_paths_w6 --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $out/pmc_grid_dual.log 2>&1 || { tail -5 $out/pmc_grid_dual.log; exit 1; }
for t in r04_cornell2048 r04_spheres1080 r04_grid100k r04_grid100k_two_paths_w6; do python3 - $t <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/prof_keep/{sys.argv[1]}_pmc.json"))
print(sys.argv[1], {k: (round(v / 1e9, 3) if isinstance(v, float) and v > 1e6 else v) for k, v in d.items() if k in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "WRITE_SIZE", "FETCH_SIZE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "trace_avg_ns", "rays_per_launch", "SQ_THREAD_CYCLES_VALU")})
PY
done
