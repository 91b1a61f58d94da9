#!/usr/bin/env python3
"""Dumps the 100 k-sphere scene's culling tree (as the library builds it, host only) and runs tools/proto/bvh4_walk_model.c on it:
box steps per ray of the binary walk against node loads / stack depth of a 4- and 8-wide collapse.  CPU only."""
import importlib, os, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
trt = importlib.import_module("tiny-raytracer_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W, H, depth = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (160, 90, 4)
desc = trt.scenes.sphere_grid(n, 3840, 2160)
w, cam = trt.world_from_description(desc)
bbox, prim, skip = w.get_bvh().cull_nodes()
sph = np.array([[*g[1], g[2]] for g in desc["geometries"]], np.float32)
c = cam.pod
camv = np.array([c.position.x, c.position.y, c.position.z, c.viewport_upper_left.x, c.viewport_upper_left.y, c.viewport_upper_left.z,
                 c.horizontal.x, c.horizontal.y, c.horizontal.z, c.vertical.x, c.vertical.y, c.vertical.z], np.float32)
path = "/tmp/bvh4_scene.bin"
with open(path, "wb") as f:
    f.write(struct.pack("ii", len(prim), len(sph)))
    f.write(np.ascontiguousarray(bbox, np.float32).tobytes()); f.write(prim.astype(np.int32).tobytes()); f.write(skip.astype(np.int32).tobytes())
    f.write(sph.tobytes()); f.write(camv.tobytes())
exe = "/tmp/bvh4_walk_model"
subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "proto", "bvh4_walk_model.c"), "-lm"], check=True)
subprocess.run([exe, path, str(W), str(H), str(depth)], check=True)
