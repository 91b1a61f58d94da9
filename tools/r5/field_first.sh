#!/bin/bash
# round 5, first look at the scene beyond the Infinity Cache (scenes.sphere_field): one-path kernel vs two paths per lane, 1M / 4M spheres,
# + FETCH_SIZE / TCC hit passes of the default launch.  Run on the GPU box: gpurun -- bash tools/r5/field_first.sh
export TMPDIR=/tmp
out=gpurun_out/r5/field_first; mkdir -p $out
run() { tag=$1; shift; python3 bench.py --scene sphere_field --width 3840 --height 2160 --spp-per-step 4 --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $out/$tag.json 2> $out/$tag.err || { echo "$tag FAILED"; tail -5 $out/$tag.err; return 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', d['value'], 'Mray/s', d['ms_per_step'], 'ms/step', r['kernel'], r['avg_launch_ms'], 'ms/launch', 'build', d['config']['scene_build_s'], 's', d['config']['scene']['device_bytes'], 'B', 'alg B/ray', (r.get('algorithmic') or {}).get('bytes_per_ray'))"; }
run n1m_default --spheres 1000000 &&
run n4m_default --spheres 4000000 &&
run n4m_dual5 --spheres 4000000 --tuning dual_walk=1,stream_waves_per_simd=5 &&
run n4m_dual6 --spheres 4000000 --tuning dual_walk=1,stream_waves_per_simd=6 &&
run n4m_w6 --spheres 4000000 --tuning stream_waves_per_simd=6 &&
run n1m_dual6 --spheres 1000000 --tuning dual_walk=1,stream_waves_per_simd=6
rocprofv3 -L 2>/dev/null | grep -i -E "mall|TCC_EA0_RDREQ|TCC_HIT|TCC_MISS|TCC_REQ|TCP_TCC_READ" | head -40 > $out/counters_avail.txt
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc_$name -- python3 bench.py --scene sphere_field --spheres 4000000 --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass > $out/pmc_$name.json 2> $out/pmc_$name.err || { echo "pmc $name failed"; tail -3 $out/pmc_$name.err; }
  python3 - "$out/pmc_$name" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stream_" in r["Kernel_Name"] and "fold" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("  pmc", k, "mean per launch", sum(v) / len(v), "launches", len(v))
PY
done
