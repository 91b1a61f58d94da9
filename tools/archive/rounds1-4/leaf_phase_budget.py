#!/usr/bin/env python3
"""Cornell's leaf phase in numbers (round 4): how many leaves a lane puts aside per walk, how many of them it tests, how many
trips the wave's leaf phase runs (= the tests of its busiest lane), and what a redistribution of ALL pending tests over the wave's lanes
(test every pending leaf speculatively at full occupancy, fold the results in walk order) could save at best - against what the exchange
costs (tools/micro/leaf_exchange.hip).  Counting kernel on the culling tree's lock-step list (collect_stats = 2), bench-size frame."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

trt = importlib.import_module("tiny-raytracer_amd")
W = H = int(os.environ.get("LEAF_SIZE", "2048"))
SPP = int(os.environ.get("LEAF_SPP", "16"))
desc = trt.scenes.cornell(W, H)
world, cam = trt.world_from_description(desc)
dev = torch.device("cuda:0")
acc = torch.zeros((H, W, 3), device=dev)
for slots in (7, 4, 12):
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    r = trt.Renderer(SPP, 1, 50, False, desc["background"], seed=1)
    r.tuning = {"leaf_slots": slots}
    r.render_device(cam, world.get_bvh(), acc.data_ptr(), torch.cuda.current_stream().cuda_stream, ctr.data_ptr(), collect_stats=2)
    torch.cuda.synchronize()
    c = [int(v) for v in ctr.tolist()]
    rays, tests, pend, rounds, steps, leaf_trips = c[1], c[4], c[7], c[8], c[9], c[10]
    lanes = rays / rounds
    print(f"leaf slots {slots}: {rays} rays, {rounds} wave rounds ({lanes:.1f} lanes walking per round)")
    print(f"  per ray: {c[2] / rays:.2f} box steps, {pend / rays:.3f} leaves put aside, {tests / rays:.3f} quad tests run (the others fail the re-check t_best > start)")
    print(f"  per wave round: {leaf_trips / rounds:.2f} leaf-phase trips with a test in them (the busiest lane's tests), {tests / rounds:.1f} tests = {tests / leaf_trips:.1f} lanes per trip ({tests / leaf_trips / 64:.0%})")
    print(f"  all pending leaves tested speculatively and spread over 64 lanes: {pend / rounds / 64:.2f} trips at 100 % - saves {leaf_trips / rounds - pend / rounds / 64:.2f} trips per round before the exchange")
    print(f"  only the tests that the fold would accept, spread over 64 lanes (unknowable in advance: a lower bound): {tests / rounds / 64:.2f} trips")
