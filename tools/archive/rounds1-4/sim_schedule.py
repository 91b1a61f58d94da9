#!/usr/bin/env python3
"""Scheduling simulation: what lane occupancy can a wave reach on the Cornell box under different phase schedules?

The kernels are VALU-issue bound (profiles/README.md), so the cost of a wave is the number of phase bodies it executes
times their instruction counts, whatever the number of lanes that had work in them. This script replays real per-ray
operation sequences (box steps T, leaf tests L, shade S, primary-ray generation G) of Cornell paths through a model of one
wave and compares
  A  the shipped schedule: while-while traversal to completion for all 64 lanes, then one shade for all;
  B  dynamic phases: run S (or L) as soon as enough lanes wait for it, otherwise keep stepping boxes.
Rays come from a small float64 path tracer over the same quads (diffuse walls, light, open front), walked through the
culling tree the library builds (trt_scene_get_cull_nodes) - CPU only, no GPU needed.
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
trt = importlib.import_module("tiny-raytracer_amd")

C_T, C_L, C_S, C_G = 36, 75, 450, 200          # VALU instructions per phase body (ISA of the stream kernels; S, G from measured totals)
C_SW = int(os.environ.get("SIM_SWITCH", 0))      # VALU instructions per scheduling decision that changes phase (votes, register shuffles)
rng = np.random.default_rng(5)


def quads_of(desc):
    qs = []
    for g in desc["geometries"]:
        assert g[0] == "quad"
        c, u, v = (np.array(x, np.float64) for x in g[1:4])
        n = np.cross(u, v)
        qs.append(dict(c=c, u=u, v=v, n=n / np.linalg.norm(n), w=n / n.dot(n), d=(n / np.linalg.norm(n)).dot(c), mat=g[4]))
    return qs


def hit_quads(qs, o, d, tmax=np.inf):
    best, bi = tmax, -1
    for i, q in enumerate(qs):
        dn = q["n"].dot(d)
        if abs(dn) < 1e-12:
            continue
        t = (q["d"] - q["n"].dot(o)) / dn
        if not (1e-3 <= t < best):
            continue
        p = o + t * d - q["c"]
        a, b = q["w"].dot(np.cross(p, q["v"])), q["w"].dot(np.cross(q["u"], p))
        if 0 <= a < 1 and 0 <= b < 1:
            best, bi = t, i
    return best, bi


def walk(nodes, o, d):
    """Operation sequence of one ray through the pre-order skip-link tree: 'T' per node visited, 'L' per leaf tested."""
    bbox, prim, skip, qs = nodes
    inv = 1.0 / np.where(d == 0, 1e-300, d)
    ops, i, tbest, n = [], 0, np.inf, len(prim)
    while i < n:
        ops.append("T")
        t0 = (bbox[i, :3] - o) * inv
        t1 = (bbox[i, 3:] - o) * inv
        lo, hi = max(np.minimum(t0, t1).max(), 1e-3), min(np.maximum(t0, t1).min(), tbest)
        if lo <= hi:
            if prim[i] >= 0:
                ops.append("L")
                t, _ = hit_quads([qs[prim[i] & 0x3FFFFFFF]], o, d, tbest)
                tbest = min(tbest, t)
                i = skip[i]
            else:
                i += 1
        else:
            i = skip[i]
    return ops


def make_paths(desc, nodes, n_paths, max_depth=50):
    qs = nodes[3]
    cam = desc["camera"]
    pos = np.array(cam["position"], np.float64)
    half = np.tan(np.radians(cam["vertical_fov"]) / 2)
    paths = []
    for _ in range(n_paths):
        px, py = rng.uniform(-1, 1, 2)
        d = np.array([px * half, py * half, 1.0])
        o, d = pos, d / np.linalg.norm(d)
        rays = []
        for _depth in range(max_depth):
            rays.append(walk(nodes, o, d))
            t, qi = hit_quads(qs, o, d)
            if qi < 0 or qs[qi]["mat"] == "light":
                break
            n = qs[qi]["n"]
            if n.dot(d) > 0:
                n = -n
            while True:
                r = rng.uniform(-1, 1, 3)
                if 1e-6 < r.dot(r) <= 1:
                    break
            nd = n + r / np.linalg.norm(r)
            o, d = o + t * d, nd / np.linalg.norm(nd)
        paths.append(rays)
    return paths


class Lane:
    """One lane; like the streamed kernel's lanes it takes its next path from the wave's shared queue."""
    __slots__ = ("queue", "path", "ri", "oi", "state")

    def __init__(self, queue):
        self.queue, self.path, self.ri, self.oi = queue, None, 0, 0
        self.state = "G" if queue else "X"

    def advance_after(self, phase):
        """state after the lane's pending operation of kind `phase` has been executed"""
        if phase == "G":
            if not self.queue:
                self.state = "X"
                return
            self.path, self.ri, self.oi = self.queue.pop(), 0, 0
        elif phase == "S":
            self.ri += 1
            self.oi = 0
            if self.ri >= len(self.path):
                self.state = "G" if self.queue else "X"
                return
        else:
            self.oi += 1
        ops = self.path[self.ri]
        self.state = ops[self.oi] if self.oi < len(ops) else "S"


def simulate(wave_paths, policy):
    queue = list(wave_paths)
    lanes = [Lane(queue) for _ in range(64)]
    cost = dict(T=0, L=0, S=0, G=0, W=0)
    trips = dict(T=0, L=0, S=0, G=0)
    last = None
    served = dict(T=0, L=0, S=0, G=0)
    price = dict(T=C_T, L=C_L, S=C_S, G=C_G)
    while True:
        n = dict(T=0, L=0, S=0, G=0, X=0)
        for l in lanes:
            n[l.state] += 1
        if n["X"] == len(lanes):
            break
        ph = policy(n)
        assert n[ph] > 0, (ph, n)
        cost[ph] += price[ph]
        if not (ph == "T" and last == "T"):
            cost["W"] += C_SW
        last = ph
        trips[ph] += 1
        served[ph] += n[ph]
        for l in lanes:
            if l.state == ph:
                l.advance_after(ph)
    return cost, trips, served


def policy_shipped():
    """The shipped round: primary rays for the lanes that need one, while-while traversal until every lane is through,
    one shade for all."""
    st = dict(in_round=False)

    def pick(n):
        if n["G"] and not st["in_round"]:
            return "G"
        if n["T"] or n["L"]:
            st["in_round"] = True
            return "T" if n["T"] else "L"
        st["in_round"] = False
        return "S" if n["S"] else "G"
    return pick


def policy_dynamic(thr_s, thr_l, thr_g):
    def pick(n):
        busy = n["T"] + n["L"]
        if n["S"] >= thr_s or (busy == 0 and n["G"] == 0 and n["S"]):
            return "S"
        if n["G"] >= thr_g or (busy == 0 and n["G"]):
            return "G"
        if n["L"] >= thr_l or (n["T"] == 0 and n["L"]):
            return "L"
        if n["T"]:
            return "T"
        return "S" if n["S"] else ("G" if n["G"] else "L")
    return pick


def report(name, res, n_rays):
    cost, trips, served = res
    tot = sum(cost.values())
    util = {k: (served[k] / (64 * trips[k]) if trips[k] else 0) for k in trips}
    print(f"{name:34s} cost/ray {tot / n_rays:7.1f}   " + "  ".join(f"{k}: {cost[k] / n_rays:6.1f} (u {util[k]:.2f})" for k in "TLSG") + f"  switch {cost['W'] / n_rays:5.1f}", flush=True)
    return tot / n_rays


def main():
    desc = trt.scenes.cornell(256, 256)
    world, _cam = trt.world_from_description(desc)
    bbox, prim, skip = world.get_bvh().cull_nodes()
    nodes = (bbox.astype(np.float64), prim, skip, quads_of(desc))
    n_waves, per_lane = int(os.environ.get("SIM_WAVES", 4)), int(os.environ.get("SIM_PATHS", 24))
    all_paths = make_paths(desc, nodes, n_waves * 64 * per_lane)
    n_rays = sum(len(p) for p in all_paths)
    n_t = sum(op == "T" for p in all_paths for r in p for op in r)
    n_l = sum(op == "L" for p in all_paths for r in p for op in r)
    print(f"{len(all_paths)} paths, {n_rays} rays ({n_rays / len(all_paths):.2f}/path), {n_t / n_rays:.2f} box steps and {n_l / n_rays:.2f} leaf tests per ray")
    ideal = (n_t * C_T + n_l * C_L + n_rays * C_S + len(all_paths) * C_G) / 64 / n_rays
    print(f"ideal (every phase body full): {ideal:.1f} per ray")

    def run(policy_factory):
        tot = [dict(T=0, L=0, S=0, G=0, W=0), dict(T=0, L=0, S=0, G=0), dict(T=0, L=0, S=0, G=0)]
        for w in range(n_waves):
            res = simulate(all_paths[w * 64 * per_lane:(w + 1) * 64 * per_lane], policy_factory())
            for acc, r in zip(tot, res):
                for k in acc:
                    acc[k] += r[k]
        return tot

    base = report("shipped (while-while, then shade)", run(policy_shipped), n_rays)
    for thr_s in (32, 40, 48, 56):
        for thr_l in (16, 24, 32, 48):
            c = report(f"dynamic S>={thr_s} L>={thr_l} G>=16", run(lambda: policy_dynamic(thr_s, thr_l, 16)), n_rays)
            print(f"{'':34s} -> x{base / c:.3f}")


if __name__ == "__main__" and not os.environ.get("SIM_SPEC"):
    main()


# ---- speculative while-while: a lane that finds a leaf keeps walking with up to K leaves postponed ----
def walk_spec(nodes, o, d, k_pending):
    """Segments [(box steps, leaf tests)] of one ray when up to k_pending leaf tests are postponed: the lane keeps stepping
    boxes with the t_best it had at the last flush (so it passes boxes the up-to-date t_best would have culled), then
    runs the postponed leaf tests in order.  The hit is the same (a leaf test can only succeed if its box
    passes with the current t_best)."""
    bbox, prim, skip, qs = nodes
    inv = 1.0 / np.where(d == 0, 1e-300, d)
    segs, i, tbest, n = [], 0, np.inf, len(prim)
    steps, pend = 0, []
    while True:
        while i < n and len(pend) < k_pending:
            steps += 1
            t0 = (bbox[i, :3] - o) * inv
            t1 = (bbox[i, 3:] - o) * inv
            lo, hi = max(np.minimum(t0, t1).max(), 1e-3), min(np.maximum(t0, t1).min(), tbest)
            if lo <= hi:
                if prim[i] >= 0:
                    pend.append(prim[i] & 0x3FFFFFFF)
                    i = skip[i]
                else:
                    i += 1
            else:
                i = skip[i]
        segs.append((steps, len(pend)))
        for q in pend:
            t, _ = hit_quads([qs[q]], o, d, tbest)
            tbest = min(tbest, t)
        steps, pend = 0, []
        if i >= n:
            break
    return segs


def make_paths_spec(desc, nodes, n_paths, ks, max_depth=50):
    """paths[k] for each k in ks over the SAME rays"""
    qs = nodes[3]
    cam = desc["camera"]
    pos = np.array(cam["position"], np.float64)
    half = np.tan(np.radians(cam["vertical_fov"]) / 2)
    out = {k: [] for k in ks}
    for _ in range(n_paths):
        px, py = rng.uniform(-1, 1, 2)
        d = np.array([px * half, py * half, 1.0])
        o, d = pos, d / np.linalg.norm(d)
        rays = {k: [] for k in ks}
        for _depth in range(max_depth):
            for k in ks:
                rays[k].append(walk_spec(nodes, o, d, k))
            t, qi = hit_quads(qs, o, d)
            if qi < 0 or qs[qi]["mat"] == "light":
                break
            nrm = qs[qi]["n"]
            if nrm.dot(d) > 0:
                nrm = -nrm
            while True:
                r = rng.uniform(-1, 1, 3)
                if 1e-6 < r.dot(r) <= 1:
                    break
            nd = nrm + r / np.linalg.norm(r)
            o, d = o + t * d, nd / np.linalg.norm(nd)
        for k in ks:
            out[k].append(rays[k])
    return out


def simulate_spec(wave_paths):
    """Round schedule over rays given as segments: gen, [box steps until every lane finished its segment, leaf trips]*, shade."""
    queue = list(wave_paths)
    cost = dict(T=0, L=0, S=0, G=0)
    trips = dict(T=0, L=0, S=0, G=0)
    served = dict(T=0, L=0, S=0, G=0)
    lanes = [dict(path=None, ri=0, alive=False) for _ in range(64)]
    while True:
        took = 0
        for l in lanes:
            if not l["alive"] and queue:
                l.update(path=queue.pop(), ri=0, alive=True)
                took += 1
        if took:
            cost["G"] += C_G; trips["G"] += 1; served["G"] += took
        act = [l for l in lanes if l["alive"]]
        if not act:
            break
        segs = [l["path"][l["ri"]] for l in act]
        for j in range(max(len(sg) for sg in segs)):
            cur = [sg[j] for sg in segs if j < len(sg)]
            t_trips = max(c[0] for c in cur)
            cost["T"] += C_T * t_trips; trips["T"] += t_trips; served["T"] += sum(c[0] for c in cur)
            l_trips = max(c[1] for c in cur)
            cost["L"] += C_L * l_trips; trips["L"] += l_trips; served["L"] += sum(c[1] for c in cur)
        cost["S"] += C_S; trips["S"] += 1; served["S"] += len(act)
        for l in act:
            l["ri"] += 1
            if l["ri"] >= len(l["path"]):
                l["alive"] = False
    return cost, trips, served


def main_spec():
    desc = trt.scenes.cornell(256, 256)
    world, _cam = trt.world_from_description(desc)
    bbox, prim, skip = world.get_bvh().cull_nodes()
    nodes = (bbox.astype(np.float64), prim, skip, quads_of(desc))
    n_waves, per_lane = int(os.environ.get("SIM_WAVES", 4)), int(os.environ.get("SIM_PATHS", 24))
    ks = (1, 2, 3, 4)
    paths = make_paths_spec(desc, nodes, n_waves * 64 * per_lane, ks)
    for k in ks:
        all_paths = paths[k]
        n_rays = sum(len(p) for p in all_paths)
        n_t = sum(sg[0] for p in all_paths for r in p for sg in r)
        n_l = sum(sg[1] for p in all_paths for r in p for sg in r)
        tot = [dict(T=0, L=0, S=0, G=0), dict(T=0, L=0, S=0, G=0), dict(T=0, L=0, S=0, G=0)]
        for w in range(n_waves):
            res = simulate_spec(all_paths[w * 64 * per_lane:(w + 1) * 64 * per_lane])
            for acc, r in zip(tot, res):
                for kk in acc:
                    acc[kk] += r[kk]
        cost, trips, served = tot
        c = sum(cost.values()) / n_rays
        print(f"postpone {k}: {n_t / n_rays:5.2f} box steps, {n_l / n_rays:4.2f} leaf tests per ray; cost/ray {c:6.1f}  " + "  ".join(
            f"{p}: {cost[p] / n_rays:5.1f} (u {served[p] / (64 * max(trips[p], 1)):.2f})" for p in "TLSG"), flush=True)


if __name__ == "__main__" and os.environ.get("SIM_SPEC"):
    main_spec()
