#!/bin/bash
# Round 4: does the power-of-two stride between tiles' record blocks (n_spp x 768 B) alias in the L2?  WRITE_SIZE and speed with the stride padded by 1 and by 5
# tile-samples (build/libtinyrt_pad1.so, _pad5.so: -DTRT_TILE_PAD) against HEAD, Cornell 2048^2 256 spp per step and the 100 k-sphere scene.
export TMPDIR=/tmp
out=gpurun_out/r4; mkdir -p $out
C="--steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
G="--scene sphere_grid --width 3840 --height 2160 --spp-per-step 16 --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline-pass"
for v in head pad1 pad5; do
  lib=$PWD/tiny-raytracer_amd/libtinyrt.so; [ $v != head ] && lib=$PWD/build/libtinyrt_$v.so
  for sc in cornell grid; do
    args=$C; kern=stream_pool_kernel; [ $sc = grid ] && args=$G
    TRT_LIB_PATH=$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/prof_pad_${v}_$sc -- python3 bench.py $args > $out/prof_pad_${v}_$sc.json 2> $out/prof_pad_${v}_$sc.err || { tail -3 $out/prof_pad_${v}_$sc.err; exit 1; }
    python3 - $out/prof_pad_${v}_$sc $v $sc <<'PY' | tee -a $out/tile_pad.txt
import csv, glob, json, sys
v = []
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "stream_pool_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE": v.append(float(r["Counter_Value"]))
d = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])
print("%-5s %-8s WRITE_SIZE %.2f GB per launch, %.1f Mray/s under the profiler" % (sys.argv[2], sys.argv[3], sum(v) / len(v) * 1024 / 1e9, d["value"]))
PY
  done
done
bash tools/ab_libs.sh $PWD/tiny-raytracer_amd/libtinyrt.so $PWD/build/libtinyrt_pad1.so 2 | tee -a $out/tile_pad.txt
