#!/usr/bin/env python3
"""The reference binary (src/main.rs) through the Python mirror: Cornell box -> output.png.
   python examples/render_cornell.py [width height spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyrt_amd as trt

w, h, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (300, 300, 300)
world, camera = trt.world_from_description(trt.scenes.cornell(w, h))          # build_world + Camera::new, src/main.rs:7-16
instance = trt.Renderer(spp, 8, 20, True, (0.001, 0.001, 0.001))             # Renderer::new(300, 8, 20, true, Some(0.001))
t0 = time.perf_counter()
image = instance.render(camera, world)
dt = time.perf_counter() - t0
image.save("output.png")
st = instance.last_stats
print(f"{w}x{h}, {spp} spp: {st['rays']} rays, kernel {st['kernel_ms']:.1f} ms ({st['rays'] / st['kernel_ms'] / 1e3:.0f} Mray/s), "
      f"call {dt * 1e3:.1f} ms -> output.png")
