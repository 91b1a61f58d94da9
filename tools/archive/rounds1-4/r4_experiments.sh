#!/bin/bash
# Round 4 experiments in one call: (1) Cornell's leaf phase in numbers; (2) the culling-tree nodes' two halves in two LDS planes
# (build/libtinyrt_ldssplit.so, -DTRT_LDS_SPLIT_NODES=1) against HEAD on random-spheres: parity, A/B, LDS counters; (3) WRITE_SIZE of a
# Cornell launch against the batch size (samples per pixel of one work batch).
export TMPDIR=/tmp
out=gpurun_out/r4; mkdir -p $out
timeout -k 10 300 python3 tools/leaf_phase_budget.py > $out/leaf_phase_budget.txt 2>&1; echo "leaf budget rc=$?"; cat $out/leaf_phase_budget.txt
TRT_LIB_PATH=$PWD/build/libtinyrt_ldssplit.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "not full_baseline and not full_sample_count and not cfg5 and not c_example and not bench" > $out/ldssplit_parity.log 2>&1; rc=$?
echo "ldssplit parity rc=$rc $(tail -1 $out/ldssplit_parity.log)"; [ $rc -ne 0 ] && { tail -40 $out/ldssplit_parity.log; exit $rc; }
R="--scene random_spheres --width 1920 --height 1080 --spp-per-step 256 --steps 4 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
run() { TRT_LIB_PATH=$1 timeout -k 10 300 python3 bench.py $R 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%9.1f Mray/s %8.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))"; }
{ for rep in 1 2 3; do echo "HEAD (32-byte nodes in LDS): $(run $PWD/tiny-raytracer_amd/libtinyrt.so)"; echo "two planes of 16-byte halves: $(run $PWD/build/libtinyrt_ldssplit.so)"; done; } | tee $out/ldssplit_ab.txt
for v in head ldssplit; do
  lib=$PWD/tiny-raytracer_amd/libtinyrt.so; [ $v = ldssplit ] && lib=$PWD/build/libtinyrt_ldssplit.so
  TRT_LIB_PATH=$lib rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $out/prof_lds_$v -- python3 bench.py --steps 2 --warmup 1 $R > /dev/null 2> $out/prof_lds_$v.err || { tail -3 $out/prof_lds_$v.err; exit 1; }
  python3 - $out/prof_lds_$v $v <<'PY' | tee -a $out/ldssplit_ab.txt
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stream_sample_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(sys.argv[2], {k: round(v / 1e9, 3) for k, v in sorted(m.items())}, "conflict share of LDS cycles %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
PY
done
C="--steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
for b in 2 4 8 16; do
  TRT_STREAM_BATCH_SPP=$b rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_write_b$b -- python3 bench.py $C > $out/prof_write_b$b.json 2> $out/prof_write_b$b.err || { tail -3 $out/prof_write_b$b.err; exit 1; }
  python3 - $out/prof_write_b$b $b <<'PY' | tee -a $out/batch_spp_write_size.txt
import csv, glob, json, sys
v = []
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stream_pool_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE": v.append(float(r["Counter_Value"]))
d = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])
print("batch spp %2s: WRITE_SIZE %.2f GB per 256-spp launch (records 12.88 GB), %.1f Mray/s under the profiler" % (sys.argv[2], sum(v) / len(v) * 1024 / 1e9, d["value"]))
PY
done
