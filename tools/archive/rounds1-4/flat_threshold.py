#!/usr/bin/env python3
"""Where does the lock-step leaf list (walk_flat) stop paying?  Random closed-ish scenes of n primitives, streamed backend,
trt_scene_options.flat_walk = 0 vs 1 (read when the scene is compiled)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
trt = importlib.import_module("tiny-raytracer_amd")


def scene(n, seed=1):
    rng = np.random.default_rng(seed)
    f = lambda a: tuple(float(np.float32(v)) for v in a)
    mats = [("white", 0, (0.73, 0.73, 0.73), 0.0), ("light", 3, (6.0, 6.0, 6.0), 0.0), ("metal", 1, (0.8, 0.8, 0.9), 0.1)]
    geos = [("quad", (-6.0, -2.0, -6.0), (12.0, 0.0, 0.0), (0.0, 0.0, 12.0), "white"),          # floor
            ("quad", (-2.0, 5.0, -2.0), (4.0, 0.0, 0.0), (0.0, 0.0, 4.0), "light")]
    for i in range(n - 2):
        c = rng.uniform(-4, 4, 3)
        m = "metal" if i % 5 == 0 else "white"
        if i % 2:
            geos.append(("sphere", f(c), float(np.float32(rng.uniform(0.3, 0.9))), m))
        else:
            geos.append(("quad", f(c), f(rng.uniform(-1.5, 1.5, 3)), f(rng.uniform(-1.5, 1.5, 3)), m))
    cam = dict(focus_distance=10.0, defocus_angle=0.0, position=(0.0, 2.0, 12.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
               vertical_fov=45.0, width=1024, height=1024)
    return dict(name=f"n{n}", materials=mats, geometries=geos, camera=cam, background=(0.3, 0.35, 0.4))


for n in (8, 16, 24, 32, 40, 48, 64):
    desc = scene(n)
    out = []
    for flat in (0, 1):
        w, cam = trt.world_from_description(desc, flat_walk=flat)
        r = trt.Renderer(64, 1, 20, False, desc["background"], seed=2, backend=trt.BACKEND_STREAMED)
        best = 0.0
        for rep in range(3):
            r.render(cam, w)
            st = r.last_stats
            best = max(best, st["rays"] / st["kernel_ms"] / 1e6)
        out.append(best)
    print(f"n={n:3d}: tree {out[0]:7.2f} Gray/s   leaf list {out[1]:7.2f} Gray/s   ratio {out[1] / out[0]:.3f}", flush=True)
