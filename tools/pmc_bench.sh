#!/bin/bash
# rocprofv3 PMC passes + kernel trace over one bench.py workload (run on the GPU box through gpurun).
#   tools/pmc_bench.sh <tag> [bench.py args...]      e.g.  tools/pmc_bench.sh r02_cornell2048
#                                                           tools/pmc_bench.sh r02_spheres1080 --scene random_spheres --width 1920 --height 1080
# Counters are collected in passes of their own (never combined with tracing); every pass profiles
# `python3 bench.py --steps 2 --warmup 1` directly (no wrapper between rocprofv3 and the program).
# Output: gpurun_out/prof/<tag>/pmc_summary.json = per-launch means of the dominant kernel + the digest of the kernel
# sources they were taken from; tools/pmc_collect.py merges such summaries into profiles/pmc_kernels.json.
export TMPDIR=/tmp
tag=$1; shift
base=gpurun_out/prof/$tag; mkdir -p $base
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $base/$name -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass $BENCH_ARGS > $base/$name.json 2> $base/$name.err || { echo "pass $name failed"; tail -3 $base/$name.err; exit 1; }; echo "pass $name ok"; }
BENCH_ARGS="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32
pass sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FLOPS_FP32
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
rocprofv3 --kernel-trace --stats --output-format csv -d $base/trace -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 $BENCH_ARGS > $base/trace.json 2> $base/trace.err || { echo "trace failed"; tail -3 $base/trace.err; exit 1; }
python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 $BENCH_ARGS > $base/bench_plain.json 2> $base/bench_plain.err
python3 - "$base" "$tag" <<'PY'
import csv, glob, collections, json, sys, os
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("bench_module", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
base, tag = sys.argv[1], sys.argv[2]
out = {}
dominant = ("megakernel", "wavefront_kernel", "stream_sample_kernel", "stream_pool_kernel", "stream_dual_kernel")
for f in glob.glob(base + "/*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in dominant):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
for f in glob.glob(base + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in dominant) and "true" not in r["Name"].split("<")[1].split(",")[1]:
            out["trace_kernel"] = r["Name"].split("(")[0]
            out["trace_avg_ns"] = float(r["AverageNs"]); out["trace_calls"] = int(r["Calls"])
    os.makedirs("gpurun_out/prof_keep", exist_ok=True)
    open(f"gpurun_out/prof_keep/{tag}_kernel_stats.csv", "w").write(open(f).read())
line = json.loads(open(base + "/trace.json").read().strip().splitlines()[-1])
out["pmc_key"] = line["roofline"]["pmc_key"]
out["rays_per_launch"] = line["roofline"]["rays_per_launch"]      # bench.py scales the per-launch counts by rays (any N, any share of the frame)
out["kernel_source_digest"] = b.kernel_source_digest()
# gfx950 (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE tallies 128-byte requests at 64 B -> x2
out["hbm_bytes_per_launch"] = (2.0 * out.get("FETCH_SIZE", 0.0) + out.get("WRITE_SIZE", 0.0)) * 1024.0
out["tag"] = tag
json.dump(out, open(base + "/pmc_summary.json", "w"), indent=1)
json.dump(out, open(f"gpurun_out/prof_keep/{tag}_pmc.json", "w"), indent=1)
open(f"gpurun_out/prof_keep/{tag}_bench_under_rocprof.json", "w").write(json.dumps(line) + "\n")
open(f"gpurun_out/prof_keep/{tag}_bench.json", "w").write(open(base + "/bench_plain.json").read())
print(json.dumps(out))
PY
