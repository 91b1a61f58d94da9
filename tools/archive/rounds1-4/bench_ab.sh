#!/bin/bash
# A/B of two library builds (build/libtinyrt_old.so, build/libtinyrt_new.so) on one box, interleaved.
out=gpurun_out/ab.jsonl; : > $out
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 >> $out; }
for rep in 1 2; do
for v in old new; do
  echo "{\"top\": \"$v\"}" >> $out
  TRT_LIB_PATH=$PWD/build/libtinyrt_$v.so run --scene cornell --spp-per-step 128 --steps 2 --warmup 1
  TRT_LIB_PATH=$PWD/build/libtinyrt_$v.so run --scene random_spheres --width 1920 --height 1080 --spp-per-step 128 --steps 2 --warmup 1
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/ab.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"]); continue
    print("   ", d["config"]["workload"][:40].ljust(42), "%9.1f Mray/s  %7.2f ms/step" % (d["value"], d["ms_per_step"]))
PY
