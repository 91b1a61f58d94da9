#!/usr/bin/env python3
"""Prototype: box tests per ray of a near-first (ordered) traversal over a free-order binned-SAH BVH2, against the
fixed-order culling tree.  CPU/numpy, a few hundred rays per scene."""
import math, sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import tinyrt_amd as t
from proto_recluster import leaf_order_and_boxes, make_prim_fn, gen_rays, secondary, count_tests, build_sah

def sa(lo, hi):
    d = np.maximum(hi - lo, 0); return 2*(d[0]*d[1] + d[1]*d[2] + d[2]*d[0])

def build_bvh2(boxes, max_leaf=1):
    lo = boxes[:, :3]; hi = boxes[:, 3:]; cen = 0.5*(lo+hi)
    nodes = []   # dict: lo,hi,left,right,prim
    sys.setrecursionlimit(100000)
    def rec(idx):
        blo = lo[idx].min(0); bhi = hi[idx].max(0)
        me = len(nodes); nodes.append(None)
        if len(idx) <= max_leaf:
            nodes[me] = (blo, bhi, -1, -1, int(idx[0])); return me
        best = None
        clo = cen[idx].min(0); chi = cen[idx].max(0)
        for ax in range(3):
            if chi[ax] - clo[ax] <= 0: continue
            if len(idx) <= 32:
                order = idx[np.argsort(cen[idx, ax], kind="stable")]
                plo = np.minimum.accumulate(lo[order], 0); phi = np.maximum.accumulate(hi[order], 0)
                slo = np.minimum.accumulate(lo[order][::-1], 0)[::-1]; shi = np.maximum.accumulate(hi[order][::-1], 0)[::-1]
                for k in range(1, len(order)):
                    c = sa(plo[k-1], phi[k-1])*k + sa(slo[k], shi[k])*(len(order)-k)
                    if best is None or c < best[0]: best = (c, order[:k], order[k:])
            else:
                nb = 16
                b = np.minimum(((cen[idx, ax]-clo[ax])/(chi[ax]-clo[ax])*nb).astype(int), nb-1)
                for k in range(1, nb):
                    L = idx[b < k]; R = idx[b >= k]
                    if len(L) == 0 or len(R) == 0: continue
                    c = sa(lo[L].min(0), hi[L].max(0))*len(L) + sa(lo[R].min(0), hi[R].max(0))*len(R)
                    if best is None or c < best[0]: best = (c, L, R)
        if best is None:
            h = len(idx)//2; best = (0, idx[:h], idx[h:])
        l = rec(best[1]); r = rec(best[2])
        nodes[me] = (blo, bhi, l, r, -1); return me
    rec(np.arange(len(boxes)))
    return nodes

def traverse_ordered(nodes, pf, o, d):
    inv = 1.0/d; tb = np.inf; tests = 0; leaf_tests = 0; steps = 0
    def slab(n, tb):
        t0 = (n[0]-o)*inv; t1 = (n[1]-o)*inv
        tn = max(np.minimum(t0,t1).max(), 0.001); tf = min(np.maximum(t0,t1).min(), tb)
        return tn if tf > tn else None
    root = nodes[0]; tests += 1
    if slab(root, tb) is None: return tests, leaf_tests, steps
    stack = [0]
    while stack:
        i = stack.pop(); n = nodes[i]
        if n[4] >= 0:
            # leaf: re-test box with current tb (exactness rule) then prim
            tests += 1
            if slab(n, tb) is not None:
                leaf_tests += 1
                tt = pf(n[4], o, d, tb)
                if tt is not None: tb = tt
            continue
        steps += 1
        l, r = nodes[n[2]], nodes[n[3]]
        tl = slab(l, tb); tr = slab(r, tb); tests += 2
        if tl is not None and tr is not None:
            if tl <= tr: stack.append(n[3]); stack.append(n[2])
            else: stack.append(n[2]); stack.append(n[3])
        elif tl is not None: stack.append(n[2])
        elif tr is not None: stack.append(n[3])
    return tests, leaf_tests, steps

rng = np.random.default_rng(0)
for name, desc, nr, scale in [("cornell", t.scenes.cornell(), 300, 30.0), ("random_spheres", t.scenes.random_spheres(), 200, 3.0), ("grid100k", t.scenes.sphere_grid(100000), 100, 25.0)]:
    bbox, prim, skip, lboxes, lprims = leaf_order_and_boxes(desc)
    pf = make_prim_fn(desc, lprims)
    leafidx = np.full(len(prim), -1); c = 0
    for i in range(len(prim)):
        if prim[i] >= 0: leafidx[i] = c; c += 1
    ref = (bbox.astype(np.float64), leafidx, skip)
    rays = gen_rays(desc, nr, rng, scale); rays = rays + secondary(rays, *ref, pf, rng)
    cull = build_sah(lboxes, 0.7)
    r = count_tests(*cull, pf, rays)
    t0 = time.time(); nodes = build_bvh2(lboxes); bt = time.time()-t0
    tot = np.array([traverse_ordered(nodes, pf, o, d) for o, d in rays]).mean(0)
    print("%s: fixed-order culling tree %.1f box tests/ray (%.2f leaf passes) | ordered BVH2: %.1f box tests, %.2f prim tests, %.1f inner steps (build %.1fs)" % (name, r[0], r[1], tot[0], tot[1], tot[2], bt))
