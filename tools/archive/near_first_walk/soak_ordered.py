#!/usr/bin/env python3
"""Soak: the opt-in near-first walk against the fixed-order walk on the 100 k-sphere scene at BASELINE cfg5's frame size,
bit for bit.  python tools/soak_ordered.py [spp]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
trt = importlib.import_module("tiny-raytracer_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 3840, 2160
desc = trt.scenes.sphere_grid(100000, W, H)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
frames, rays = [], []
for ordered in ("0", "1"):
    os.environ["TRT_ORDERED_WALK"] = ordered
    world, cam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, 50, False, desc["background"], seed=1)
    acc = torch.zeros((H, W, 3), device=dev)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    r.render_device(cam, world.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr())     # warm-up of the first chunk included
    torch.cuda.synchronize()
    acc.zero_(); ctr.zero_()
    t0 = time.perf_counter()
    r.render_device(cam, world.get_bvh(), acc.data_ptr(), stream.cuda_stream, ctr.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames.append(acc); rays.append(int(ctr[1].item()))
    print(f"TRT_ORDERED_WALK={ordered}: {rays[-1]} rays in {dt:.2f} s = {rays[-1] / dt / 1e9:.2f} Gray/s", flush=True)
same = torch.equal(frames[0].view(torch.int32), frames[1].view(torch.int32))
print("ray counts equal:", rays[0] == rays[1], " frames bit-identical:", same)
sys.exit(0 if same and rays[0] == rays[1] else 1)
