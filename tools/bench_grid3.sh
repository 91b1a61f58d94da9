#!/bin/bash
out=gpurun_out/grid3.jsonl; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 "$@" 2>/dev/null | tail -1 >> $out; }
for w8 in "" 1; do
  if [ -n "$w8" ]; then export TRT_MINW8=1; fi
  echo "{\"top\": \"minw8=$w8\"}" >> $out
  run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --backend megakernel
  TRT_WF_SERVE_MIN=8 run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --backend wavefront
  TRT_WF_SERVE_MIN=8 run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --warmup 1 --backend wavefront
done
python - <<'PY'
import json
for ln in open("gpurun_out/grid3.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"]); continue
    r=d["roofline"]; print(d["config"]["workload"][:70].ljust(72), "%9.1f Mray/s  %7.2f ms/step  frac %.3f" % (d["value"], d["ms_per_step"], r["frac"]))
PY
