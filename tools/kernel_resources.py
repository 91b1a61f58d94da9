#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel from `make -C tiny-raytracer_amd/csrc asm` (build/*.resource.txt)."""
import re, sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in ["streamed", "kernels", "wavefront"]:
    txt = open(os.path.join(root, "build", f + ".resource.txt")).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("void trt::", "").split("(")[0]
        print("%-9s VGPR %3d SGPR %3d (spilled %2d) VGPR spilled %2d scratch %4d B/lane occ %d  %s" % (
            f, g("VGPRs"), g("TotalSGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), dem))
