#!/usr/bin/env python3
"""Where a streamed render's wave time goes: fetch/generate | box steps | leaf phase | shade, from the s_memtime instrumentation of a
-DTRT_PHASE_CLOCK build (tools/phase_clock.sh builds it as build/libtinyrt_clock.so and runs this with TRT_LIB_PATH set)."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
trt = importlib.import_module("tiny-raytracer_amd")

CASES = [("cornell", lambda: trt.scenes.cornell(2048, 2048), 64), ("random_spheres", lambda: trt.scenes.random_spheres(1920, 1080), 64),
         ("sphere_grid100k", lambda: trt.scenes.sphere_grid(100000, 3840, 2160), 8)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
dev = torch.device("cuda:0")
for name, mk, spp in CASES:
    desc = mk()
    w, cam = trt.world_from_description(desc)
    scene = w.get_bvh()
    r = trt.Renderer(spp, 1, 50, False, desc["background"])
    W, H = cam.get_image_size()
    acc = torch.zeros((H, W, 3), device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()
    r.render_device(cam, scene, acc.data_ptr(), st.cuda_stream, ctr.data_ptr())       # warm
    torch.cuda.synchronize()
    ctr.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    r.render_device(cam, scene, acc.data_ptr(), st.cuda_stream, ctr.data_ptr())
    e1.record(st)
    torch.cuda.synchronize()
    c = ctr.tolist()
    t = c[12:16]
    tot = sum(t) or 1
    print(json.dumps(dict(scene=name, ms=round(e0.elapsed_time(e1), 2), rays=c[1], mray_s=round(c[1] / e0.elapsed_time(e1) / 1e3, 1),
                          fetch_generate=round(t[0] / tot, 3), box_steps=round(t[1] / tot, 3), leaf_phase=round(t[2] / tot, 3),
                          shade=round(t[3] / tot, 3), wave_cycles_per_ray=round(tot / max(c[1], 1), 1))), flush=True)
