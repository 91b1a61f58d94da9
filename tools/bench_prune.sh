#!/bin/bash
out=gpurun_out/prune.jsonl; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 "$@" 2>/dev/null | tail -1 >> $out; }
for pr in 1.01 0.9 0.8 0.7 0.6 0.5 0.35 0.2 0.0; do
  export TRT_CULL_PRUNE=$pr
  echo "{\"top\": \"prune=$pr\"}" >> $out
  run --scene cornell --spp-per-step 64 --steps 2 --warmup 1 --no-roofline-pass
  run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 2 --warmup 1 --no-roofline-pass
done
python - <<'PY'
import json
for ln in open("gpurun_out/prune.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"]); continue
    print("   ", d["config"]["workload"][:40].ljust(42), "%9.1f Mray/s  %7.2f ms/step" % (d["value"], d["ms_per_step"]))
PY
