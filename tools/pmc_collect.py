#!/usr/bin/env python3
"""Merge gpurun_out/prof_keep/<tag>_* (made by tools/pmc_bench.sh on the GPU box) into profiles/: copies the per-tag
files and (re)writes the entry of profiles/pmc_kernels.json that bench.py reads for its roofline block.
  python tools/pmc_collect.py r02_cornell2048 [more tags...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
keep = os.path.join(ROOT, "gpurun_out", "prof_keep")
dst = os.path.join(ROOT, "profiles")
path = os.path.join(dst, "pmc_kernels.json")
table = json.load(open(path)) if os.path.exists(path) else {
    "_about": "per-launch PMC means of the dominant kernel, keyed like bench.py's roofline.pmc_key; made by tools/pmc_bench.sh "
              "(separate rocprofv3 --pmc passes) and merged by tools/pmc_collect.py. hbm_bytes_per_launch = (2 x FETCH_SIZE + "
              "WRITE_SIZE) KiB (gfx950 correction, guides/MI355X_MICROARCH.md). bench.py refuses an entry whose "
              "kernel_source_digest differs from the sources in the tree."}
for tag in sys.argv[1:]:
    for suffix in ("_pmc.json", "_kernel_stats.csv", "_bench_under_rocprof.json", "_bench.json"):
        src = os.path.join(keep, tag + suffix)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, tag + suffix))
    e = json.load(open(os.path.join(keep, tag + "_pmc.json")))
    e["source"] = f"profiles/{tag}_pmc.json"
    table[e["pmc_key"]] = e
    print(tag, "->", e["pmc_key"], "digest", e["kernel_source_digest"])
json.dump(table, open(path, "w"), indent=1)
