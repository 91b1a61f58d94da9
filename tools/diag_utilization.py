#!/usr/bin/env python3
"""Diagnostic: SIMD utilisation of the megakernel's loops, from the counting kernel variant.
lane-level counts / (64 x wave-level loop trips) = fraction of lanes doing useful work per trip."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
trt = importlib.import_module("tiny-raytracer_amd")

CASES = [("cornell", lambda: trt.scenes.cornell(1024, 1024), 64), ("random_spheres", lambda: trt.scenes.random_spheres(960, 540), 64),
         ("sphere_grid100k", lambda: trt.scenes.sphere_grid(100000, 960, 540), 8)]
BACKEND = int(os.environ.get("DIAG_BACKEND", "3"))
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
for name, mk, spp in CASES:
    desc = mk()
    w, cam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, 50, False, desc["background"], backend=BACKEND)
    r.render(cam, w, collect_stats=2)               # counting kernel on the culling tree: the tests actually performed
    s = r.last_stats
    rounds, steps, leafs, gens = s["wave_trips"]
    prim = s["sphere_tests"] + s["quad_plane_tests"]
    out = dict(scene=name, rays=s["rays"], ms=round(s["kernel_ms"], 2), mray_s=round(s["rays"] / s["kernel_ms"] / 1e3, 1),
               nodes_per_ray=round(s["node_tests"] / s["rays"], 2), prims_per_ray=round(prim / s["rays"], 3),
               util_round=round(s["rays"] / (64 * rounds), 3), util_box_step=round(s["node_tests"] / (64 * steps), 3),
               util_leaf=round(prim / (64 * leafs), 3), util_gen=round(s["samples"] / (64 * gens), 3),
               box_steps_per_round=round(steps / rounds, 2), leaf_phases_per_round=round(leafs / rounds, 2),
               gen_per_round=round(gens / rounds, 3))
    print(json.dumps(out), flush=True)
