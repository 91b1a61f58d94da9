#!/bin/bash
# The backends on the three BASELINE scenes, one box, one call (run on the GPU box).
out=gpurun_out/matrix.jsonl; : > $out
run() { timeout -k 10 400 python bench.py --cpu-seconds 0 "$@" 2>/dev/null | tail -1 >> $out; }
run --scene cornell --backend streamed
run --scene cornell --backend megakernel
run --scene cornell --spp-per-step 64 --backend wavefront
run --scene random_spheres --width 1920 --height 1080 --backend streamed
run --scene random_spheres --width 1920 --height 1080 --backend megakernel
run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --backend wavefront
run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --backend streamed
run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --backend wavefront
run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 1 --backend megakernel
python - <<'PY'
import json
for ln in open("gpurun_out/matrix.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    r=d["roofline"]; print(d["config"]["workload"][:78].ljust(80), "%9.1f Mray/s  %8.2f ms/step  frac %.3f" % (d["value"], d["ms_per_step"], r["frac"]))
PY
