#!/usr/bin/env python3
"""Round 4: what could MORE THAN ONE PATH PER LANE buy the LDS tree walk (random-spheres), whose loss is the spread of walk
lengths inside a wave?  A replay of measured walk lengths (tools/proto/cull_tree_model.py with DUMP_STEPS: box steps per ray of the shipped
culling tree, by bounce) through a model of one wave, in wave-instructions per ray - the currency of a VALU-issue-bound kernel:

  A  the shipped schedule: a round's walk phase ends once at most T lanes still walk (they carry their walk on), the finished lanes are
     shaded together and get their next ray.
  B  two paths per lane, BOTH stepped in every trip under their own masks (the shape of stream_dual_kernel): a trip costs two box steps.
  D  paths decoupled from lanes: 192 paths per wave in a store, a lane whose walk has ended hands its hit over and takes the next ready
     ray - in a switch block that runs once K lanes wait (wave-level code: executed for all 64 lanes whoever needs it) - and hits are
     shaded 64 at a time.

  python3 tools/proto/cull_tree_model.py random_spheres 0 0.5   (with DUMP_STEPS=build/model/steps_rs.txt)
  python3 tools/proto/two_paths_model.py build/model/steps_rs.txt
"""
import sys

import numpy as np

C_T = 29        # vector instructions per box-step trip (rt_path.h box_loop_lds)
C_S = 450       # per shade pass (sin, cos x2, acos, cbrt, 3 sqrt, 9 divisions per Lambertian bounce)
C_SW = 60       # D: the switch block - push the hit, pop a ready ray, load origin and direction, three 1/d, reset the cursor
C_IO = 75       # D: per shade pass - the path store's loads and stores (~35 vector-memory instructions + addressing)
P_END = 0.37    # a ray is its path's last with this probability (random-spheres: 2.7 rays per sample)

rng = np.random.default_rng(7)
steps_by_bounce = {}
for line in open(sys.argv[1] if len(sys.argv) > 1 else "build/model/steps_rs.txt"):
    b, s = map(int, line.split())
    steps_by_bounce.setdefault(min(b, 1), []).append(s)          # primary rays | every later bounce
pools = {b: np.array(v) for b, v in steps_by_bounce.items()}
N_RAYS = 400_000


def draw(primary):
    p = pools[0 if primary else 1]
    return int(p[rng.integers(len(p))])


def shipped(T, two_masks=False):
    """A (and B: the same lanes x 2 slots, every trip costing two box steps)."""
    slots = 128 if two_masks else 64
    remain = np.array([draw(True) for _ in range(slots)])
    cost = rays = busy = trips = 0
    while rays < N_RAYS:
        # walk phase: until at most T walks are still under way (at least one must finish)
        order = np.sort(remain)
        few = min(T * (2 if two_masks else 1), slots - 1)
        t = int(order[slots - 1 - few]) if few < slots else int(order[-1])
        t = max(t, 1)
        cost += t * C_T * (2 if two_masks else 1)
        trips += t
        busy += int(np.minimum(remain, t).sum())
        remain = remain - t
        done = remain <= 0
        n_done = int(done.sum())
        rays += n_done
        cost += C_S * (2 if two_masks else 1) if n_done else 0          # B shades slot A's and slot B's finished lanes one after the other
        for i in np.nonzero(done)[0]:
            remain[i] = draw(rng.random() < P_END)
    return cost / rays, busy / (trips * slots)


def decoupled(K, paths=192):
    ready = paths - 64                                   # rays waiting for a lane
    hits = 0                                             # hits waiting for a shade pass
    lane = np.array([draw(True) for _ in range(64)])     # remaining steps of the lane's walk; 0 = waiting; -1 = no path
    cost = rays = busy = trips = 0
    while rays < N_RAYS:
        walking = lane > 0
        waiting = int((lane == 0).sum())
        idle = int((lane < 0).sum())
        if waiting >= K or (not walking.any() and waiting):
            cost += C_SW                                 # the switch block
            for i in np.nonzero(lane == 0)[0]:
                hits += 1
                rays += 1
                if ready:
                    ready -= 1
                    lane[i] = draw(rng.random() < P_END)
                else:
                    lane[i] = -1
            continue
        if hits >= 64 or (hits and not walking.any()):
            n = min(hits, 64)
            cost += C_S + C_IO                           # one shade pass: n hits become n ready rays (next bounce, or a new sample's primary ray)
            hits -= n
            ready += n
            for i in np.nonzero(lane < 0)[0]:            # lanes without a path take one at once (inside the same switch code)
                if ready:
                    ready -= 1
                    lane[i] = draw(rng.random() < P_END)
            continue
        # box-step trips until the next event: a lane finishes
        t = int(lane[walking].min())
        cost += t * C_T
        trips += t
        busy += t * int(walking.sum())
        lane[walking] -= t
    return cost / rays, busy / (trips * 64)


print(f"random-spheres walk lengths: {sum(len(v) for v in pools.values())} rays, primary mean {pools[0].mean():.1f}, secondary mean {pools[1].mean():.1f}, "
      f"p95 {np.percentile(np.concatenate(list(pools.values())), 95):.0f}")
print(f"costs: box-step trip {C_T}, shade pass {C_S}, switch block {C_SW}, path-store traffic per shade pass {C_IO} (wave-instructions)")
for T in (0, 8, 16):
    c, u = shipped(T)
    print(f"A  shipped schedule, {T:2d} stragglers:                       {c:6.1f} wave-instructions per ray, box-step lane utilisation {u:.3f}")
c, u = shipped(8, two_masks=True)
print(f"B  two paths per lane, both stepped every trip, 8 stragglers: {c:6.1f} wave-instructions per ray, box-step lane utilisation {u:.3f}")
for K in (1, 4, 8, 16, 32):
    c, u = decoupled(K)
    print(f"D  192 paths per wave, switch block once {K:2d} lanes wait:     {c:6.1f} wave-instructions per ray, box-step lane utilisation {u:.3f}")
c, u = decoupled(8, paths=128)
print(f"D  128 paths per wave, switch block once  8 lanes wait:     {c:6.1f} wave-instructions per ray, box-step lane utilisation {u:.3f}")
