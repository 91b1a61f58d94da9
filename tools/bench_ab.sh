#!/bin/bash
# A/B of two library builds in one process-sequence on one box: tools/bench_ab.sh <bench args...>
out=gpurun_out/ab.jsonl; : > $out
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 >> $out; }
for rep in 1 2; do
for v in old new new5; do
  lib=build/libtinyrt_new.so; [ $v = old ] && lib=build/libtinyrt_old.so
  unset TRT_MINW5; [ $v = new5 ] && export TRT_MINW5=1
  echo "{\"top\": \"$v\"}" >> $out
  TRT_LIB_PATH=$PWD/$lib run --scene cornell --spp-per-step 128 --steps 2 --warmup 1
  TRT_LIB_PATH=$PWD/$lib run --scene random_spheres --width 1920 --height 1080 --spp-per-step 128 --steps 2 --warmup 1
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/ab.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"]); continue
    print("   ", d["config"]["workload"][:40].ljust(42), "%9.1f Mray/s  %7.2f ms/step" % (d["value"], d["ms_per_step"]))
PY
