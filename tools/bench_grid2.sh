#!/bin/bash
out=gpurun_out/grid2.jsonl; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 "$@" 2>/dev/null | tail -1 >> $out; }
for top in 0 512 1024 2048; do
  export TRT_TOP_NODES=$top
  echo "{\"top\": $top}" >> $out
  run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --backend megakernel
  TRT_WF_SERVE_MIN=8 run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --backend wavefront
done
python - <<'PY'
import json
for ln in open("gpurun_out/grid2.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print("TRT_TOP_NODES", d["top"]); continue
    r=d["roofline"]; print(d["config"]["workload"][:70].ljust(72), "%9.1f Mray/s  %7.2f ms/step  frac %.3f" % (d["value"], d["ms_per_step"], r["frac"]))
PY
