#!/bin/bash
out=gpurun_out/sweep.jsonl; : > $out
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 >> $out; }
for rep in 1 2; do for w in 1 6 7 8; do
  export TRT_WF_MINW=$w
  echo "{\"top\": \"minw $w\"}" >> $out
  run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --warmup 1 --backend wavefront
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/sweep.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"], end=": "); continue
    print("%9.1f Mray/s  %7.2f ms/step" % (d["value"], d["ms_per_step"]))
PY
