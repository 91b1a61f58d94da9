#!/bin/bash
out=gpurun_out/ab.jsonl; : > $out
run() { timeout -k 10 300 python bench.py --cpu-seconds 0 --no-roofline-pass "$@" 2>/dev/null | tail -1 >> $out; }
echo "{\"top\": \"megakernel\"}" >> $out
run --scene cornell --spp-per-step 128 --steps 2 --warmup 1 --backend megakernel
run --scene random_spheres --width 1920 --height 1080 --spp-per-step 128 --steps 2 --warmup 1 --backend megakernel
for mw in 1 4 5; do for sm in 8 16 32; do
  export TRT_POOL_MINW=$mw TRT_POOL_SERVE_MIN=$sm
  echo "{\"top\": \"pooled minw=$mw serve_min=$sm\"}" >> $out
  run --scene cornell --spp-per-step 128 --steps 2 --warmup 1 --backend pooled
  run --scene random_spheres --width 1920 --height 1080 --spp-per-step 128 --steps 2 --warmup 1 --backend pooled
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/ab.jsonl"):
    try: d=json.loads(ln)
    except Exception: print("bad line", ln[:80]); continue
    if "top" in d: print(d["top"]); continue
    print("   ", d["config"]["workload"][:40].ljust(42), "%9.1f Mray/s  %7.2f ms/step" % (d["value"], d["ms_per_step"]))
PY
