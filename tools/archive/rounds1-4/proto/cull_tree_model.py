#!/usr/bin/env python3
"""Dumps a scene's leaf sequence (exact leaf boxes in walk order, spheres, camera) and runs tools/proto/cull_tree_model.c on it.
   python3 tools/proto/cull_tree_model.py random_spheres|grid [rule factor]..."""
import importlib, os, struct, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
trt = importlib.import_module("tiny-raytracer_amd")
which = sys.argv[1]
desc = trt.scenes.random_spheres(1920, 1080) if which == "random_spheres" else trt.scenes.sphere_grid(100000, 3840, 2160)
w, cam = trt.world_from_description(desc)
bbox, prim, skip = w.get_bvh().nodes()                     # the reference tree: its leaves in walk order
leaves = prim >= 0
sph = np.array([[*g[1], g[2]] for g in desc["geometries"]], np.float32)
c = cam.pod
camv = np.array([c.position.x, c.position.y, c.position.z, c.viewport_upper_left.x, c.viewport_upper_left.y, c.viewport_upper_left.z,
                 c.horizontal.x, c.horizontal.y, c.horizontal.z, c.vertical.x, c.vertical.y, c.vertical.z], np.float32)
path = os.path.join(ROOT, "build", "model", f"cull_{which}.bin")
with open(path, "wb") as f:
    f.write(struct.pack("ii", int(leaves.sum()), len(sph)))
    f.write(np.ascontiguousarray(bbox[leaves], np.float32).tobytes()); f.write(prim[leaves].astype(np.int32).tobytes())
    f.write(sph.tobytes()); f.write(camv.tobytes())
exe = os.path.join(ROOT, "build", "model", "cull_tree_model")
subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "proto", "cull_tree_model.c"), "-lm"], check=True)
W, H = (1920, 1080) if which == "random_spheres" else (3840, 2160)
params = sys.argv[2:] or ["0", "0.3", "0", "0.4", "0", "0.5", "0", "0.6", "0", "0.7", "0", "0.9", "0", "1e9", "1", "0.5", "1", "1", "1", "2", "1", "4", "1", "8"]
subprocess.run([exe, path, str(W), str(H), "5" if which == "random_spheres" else "4", *params], check=True)
