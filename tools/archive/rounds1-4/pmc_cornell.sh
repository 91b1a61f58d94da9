#!/bin/bash
# rocprofv3 PMC passes over the bench workload (run on the GPU box). Usage: tools/pmc_cornell.sh <tag> [bench args...]
export TMPDIR=/tmp
tag=$1; shift
base=gpurun_out/prof/$tag; mkdir -p $base
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $base/$name -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass $BENCH_ARGS > $base/$name.json 2> $base/$name.err; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32
pass sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FLOPS_FP32
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
rocprofv3 --kernel-trace --stats --output-format csv -d $base/trace -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 $BENCH_ARGS > $base/trace.json 2> $base/trace.err
python3 - "$base" <<'PY'
import csv, glob, collections, json, sys
base=sys.argv[1]; out={}
for f in glob.glob(base+"/*/*/*_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("megakernel", "wavefront_kernel", "stream_sample_kernel", "stream_pool_kernel")):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): out[k]=sum(v)/len(v)
json.dump(out, open(base+"/pmc_summary.json","w"), indent=1)
print(json.dumps(out))
for f in glob.glob(base+"/trace/*/*_kernel_stats.csv"): print(open(f).read()[:900])
PY
