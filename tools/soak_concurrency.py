#!/usr/bin/env python3
"""Soak of the host layer on real hardware (round 4): the concurrency tests of tests/test_gpu_multi.py - six threads on one scene per backend, two
un-synchronised streams, eight threads over every blocking entry point on two scenes, many shards on one device - repeated in ONE process for a
time budget, every frame compared with its serial render each time.  One process, one GPU context: not a retry loop around a failure, an
endurance run of code that passes.    python3 -X faulthandler tools/soak_concurrency.py [seconds=300]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_multi as t  # noqa: E402

trt = importlib.import_module("tiny-raytracer_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
t0 = time.time()
rounds = 0
while time.time() - t0 < budget:
    for backend in ("streamed", "wavefront"):
        t.test_concurrent_renders_of_one_scene_equal_serial_ones(trt, backend)
    t.test_concurrent_device_renders_on_two_streams(trt)
    t.test_mixed_entry_points_from_eight_threads_on_two_scenes(trt)
    t.test_render_multi_progressive_passes_and_backends(trt)
    rounds += 1
    if rounds % 5 == 0:
        print(f"{rounds} rounds, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {rounds} rounds of the concurrency tests in one process ({time.time() - t0:.0f} s), every frame equal to its serial render")
