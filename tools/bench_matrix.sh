#!/bin/bash
# Diagnostic sweep: both backends on the three BASELINE scenes (run on the GPU box).
out=gpurun_out/matrix.jsonl; : > $out
run() { timeout -k 10 200 python bench.py --cpu-seconds 0 "$@" 2>/dev/null | tail -1 >> $out; }
run --scene cornell --spp-per-step 64 --steps 2 --warmup 1 --backend megakernel
for sm in 16 32 48 64; do TRT_WF_SERVE_MIN=$sm run --scene cornell --spp-per-step 64 --steps 2 --warmup 1 --backend wavefront; done
run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 2 --warmup 1 --backend megakernel
for sm in 8 16 32 48 64; do TRT_WF_SERVE_MIN=$sm run --scene random_spheres --width 1920 --height 1080 --spp-per-step 64 --steps 2 --warmup 1 --backend wavefront; done
run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --backend megakernel
for sm in 4 8 16 32 48; do TRT_WF_SERVE_MIN=$sm run --scene sphere_grid --width 3840 --height 2160 --spp-per-step 4 --steps 2 --warmup 1 --backend wavefront; done
python - <<'PY'
import json
for ln in open("gpurun_out/matrix.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    r=d["roofline"]; print(d["config"]["workload"][:70].ljust(72), "%9.1f Mray/s  %7.2f ms/step  frac %.3f" % (d["value"], d["ms_per_step"], r["frac"]))
PY
