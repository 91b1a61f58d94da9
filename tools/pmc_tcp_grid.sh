#!/bin/bash
export TMPDIR=/tmp
base=gpurun_out/prof/tcp_grid; mkdir -p $base
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $base/$n -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 > $base/$n.json 2> $base/$n.err || { echo "pass $n failed"; tail -3 $base/$n.err; }
done
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/tcp_grid/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stream_pool_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v))
PY
