#!/usr/bin/env python3
"""How loose are the 16-byte nodes' f16 boxes?  (round 5)  Counting kernel on the culling tree (collect_stats = 2) with 32-byte exact nodes and with
16-byte f16 nodes: the difference in box tests per ray = exact-box re-tests of leaves whose COARSE box passed; sphere tests = leaves whose exact
box passed.  f16 spacing is 0.125 at |x| in [128, 256) and 0.5 at [512, 1024): on large scenes the coarse leaf boxes are several sphere
diameters wide."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
trt = importlib.import_module("tiny-raytracer_amd")
dev = torch.device("cuda:0")
for name, desc in (("sphere_grid 100k", trt.scenes.sphere_grid(100000, 1920, 1080)), ("sphere_field 1M", trt.scenes.sphere_field(1_000_000, 1920, 1080)),
                   ("sphere_field 4M", trt.scenes.sphere_field(4_000_000, 1920, 1080))):
    row = {}
    for label, opts in (("32-byte exact nodes", dict(compact_nodes=0)), ("16-byte f16 nodes", dict(compact_nodes=1))):
        w, cam = trt.world_from_description(desc, **opts)
        r = trt.Renderer(1, 1, 50, False, desc["background"], seed=1)
        acc = torch.zeros((1080, 1920, 3), device=dev)
        ctr = torch.zeros(16, dtype=torch.int64, device=dev)
        r.render_device(cam, w.get_bvh(), acc.data_ptr(), 0, ctr.data_ptr(), collect_stats=2)
        torch.cuda.synchronize()
        c = ctr.tolist()
        row[label] = (c[1], c[2] / c[1], c[3] / c[1])
        del w
    (rays, n32, s32), (_, n16, s16) = row["32-byte exact nodes"], row["16-byte f16 nodes"]
    print(f"{name}: rays {rays}  box tests per ray: exact nodes {n32:.1f}, f16 nodes {n16:.1f} (of which exact re-tests of coarse-passing leaves <= {n16 - n32 + s32:.1f}); "
          f"sphere tests per ray {s32:.2f} / {s16:.2f}")
