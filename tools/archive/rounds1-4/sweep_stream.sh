#!/bin/bash
out=gpurun_out/sweep.jsonl; : > $out
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 >> $out; }
for ls in 2 4 6 8 12 16; do
  export TRT_LEAF_SERVE=$ls
  echo "{\"top\": \"leaf_serve $ls\"}" >> $out
  run --scene random_spheres --width 1920 --height 1080 --steps 2 --warmup 1
  run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --warmup 1
done
python - <<'PY'
import json
for ln in open("gpurun_out/sweep.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"]); continue
    print("   ", d["config"]["workload"][:30].ljust(32), "%9.1f Mray/s  %7.2f ms/step" % (d["value"], d["ms_per_step"]))
PY
