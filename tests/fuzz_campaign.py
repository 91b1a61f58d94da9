#!/usr/bin/env python3
"""One-off extended fuzz campaign (not collected by pytest): seeded random scenes of every size class - lock-step leaf list (<= 32
primitives), LDS tree walk, 512-lane fallback, global-memory 16-byte-node walk - rendered by the default (streamed) backend and compared
with the CPU oracle bit for bit, frame and ray count, until the time budget is spent.
    python3 tests/fuzz_campaign.py [seconds=420] [first_seed=1000] [tuning as JSON, e.g. '{"dual_walk": 1, "stream_waves_per_simd": 6}']
Test infrastructure (it drives the oracle); the scenes are tests/test_gpu_fuzz.py's random_scene."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
from test_gpu_fuzz import random_scene
from oracle import orc
trt = importlib.import_module("tiny-raytracer_amd")

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 420.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
import json
knobs = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
t0 = time.time()
classes = [(1, 8), (9, 32), (33, 120), (121, 420), (421, 700), (1500, 4000)]
done = rays = 0
per_class = [0] * len(classes)
while time.time() - t0 < budget:
    k = done % len(classes)
    rng = np.random.default_rng(seed)
    lo, hi = classes[k]
    n = int(rng.integers(lo, hi + 1))
    desc = random_scene(seed, n_prims=n, width=int(rng.choice([33, 48, 64, 97])), height=int(rng.choice([17, 32, 40])),
                        degenerate=bool(rng.random() < 0.15), p_sphere=float(rng.choice([0.0, 0.5, 0.5, 1.0])))
    spp, depth = int(rng.choice([1, 3, 8])), int(rng.choice([4, 12, 50]))
    ow, ocam = orc.world_from_description(desc)
    cpu, cst = orc.render(ow, ocam, spp, depth, desc["background"], seed=seed & 0xffff, nthreads=16)
    pw, pcam = trt.world_from_description(desc)
    r = trt.Renderer(spp, 1, depth, False, desc["background"], seed=seed & 0xffff)
    r.tuning = knobs
    gpu = r.render(pcam, pw).data
    same = np.array_equal(np.ascontiguousarray(gpu, np.float32).view(np.uint32), np.ascontiguousarray(cpu, np.float32).view(np.uint32))
    if not same or r.last_stats["rays"] != cst["rays"]:
        print(f"MISMATCH seed {seed} prims {n} spp {spp} depth {depth}: rays {r.last_stats['rays']} vs {cst['rays']}", flush=True)
        sys.exit(1)
    done += 1; per_class[k] += 1; rays += cst["rays"]; seed += 1
    if done % 50 == 0:
        print(f"{done} scenes, {rays} rays, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz campaign: {done} random scenes ({', '.join(f'{a}-{b} prims: {c}' for (a, b), c in zip(classes, per_class))}), {rays} rays, "
      f"default backend{' with ' + json.dumps(knobs) if knobs else ''} == oracle bit for bit on every frame and ray count ({time.time() - t0:.0f} s)")
