#!/bin/bash
# TCP / TCC / TA counters of the dominant streamed kernel on the 100 k-sphere scene (per launch): tools/pmc_tcp_grid.sh <tag> [env assignments]
#   e.g. tools/pmc_tcp_grid.sh pool      tools/pmc_tcp_grid.sh gen TRT_GENERATIONS=1 TRT_STREAM_BATCH_SPP=16
export TMPDIR=/tmp
tag=${1:-pool}; shift
for kv in "$@"; do export "$kv"; done
base=gpurun_out/prof/tcp_grid_$tag; mkdir -p $base
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $base/$n -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $base/$n.json 2> $base/$n.err || { echo "pass $n failed"; tail -3 $base/$n.err; }
done
python3 - "$base" <<'PY'
import csv, glob, collections, sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stream_pool_kernel" in r["Kernel_Name"] or "stream_gen_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v))
if "TCP_TOTAL_CACHE_ACCESSES_sum" in agg and "TA_FLAT_READ_WAVEFRONTS_sum" in agg:
    print("L1 lines per wave load:", round(sum(agg["TCP_TOTAL_CACHE_ACCESSES_sum"])/len(agg["TCP_TOTAL_CACHE_ACCESSES_sum"]) / (sum(agg["TA_FLAT_READ_WAVEFRONTS_sum"])/len(agg["TA_FLAT_READ_WAVEFRONTS_sum"])), 2))
PY
