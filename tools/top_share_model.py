#!/usr/bin/env python3
"""Where do the box steps of the culling-tree walk fall?  (round 5, before the top-in-LDS walk was written; no GPU needed)

  python tools/top_share_model.py [n_spheres]        -> profiles/r05_top_share_model.txt

Walks the scene's culling tree (trt_scene_get_cull_nodes) on the host for 600 rays - primary rays and upward scatters from the ground -
with the slab test in float64 and an approximate t_best (the entry distance of the first leaf box hit), and prints, level by level, how many
nodes the tree's upper levels hold and which share of all box steps falls on them.  The step count per ray it arrives at (223 on the
100 k-sphere scene) is the counting kernel's (222), so the model is close enough to size an LDS cache with."""
import sys, importlib, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
trt = importlib.import_module("tiny-raytracer_amd")
n = int(sys.argv[1]) if len(sys.argv)>1 else 100000
desc = trt.scenes.sphere_grid(n, 384, 216) if n <= 200000 else trt.scenes.sphere_field(n, 384, 216)
w, cam = trt.world_from_description(desc)
sc = w.get_bvh()
bbox, prim, skip = sc.cull_nodes()
N = len(prim)
# depth of every node (pre-order, skip links)
depth = np.zeros(N, np.int32); ends=[]
for i in range(N):
    while ends and ends[-1] <= i: ends.pop()
    depth[i] = len(ends)
    if prim[i] < 0: ends.append(skip[i])
# level-order rank for "top K nodes = whole levels"
per_level = np.bincount(depth)
cum = np.cumsum(per_level)
print("nodes", N, "levels", len(per_level), "cum nodes by level", cum[:16])
rng = np.random.default_rng(1)
# rays: primary rays from the camera through random pixels + secondary rays from random ground points in random upward directions
c = desc["camera"]; pos = np.array(c["position"], np.float64)
def walk(o, d):
    inv = 1.0/d
    t_best = np.inf
    i = 0; steps_by_depth = np.zeros(len(per_level), np.int64)
    while i < N:
        steps_by_depth[depth[i]] += 1
        lo = bbox[i,:3]; hi = bbox[i,3:]
        t0 = (lo-o)*inv; t1 = (hi-o)*inv
        tn = max(np.minimum(t0,t1).max(), 0.001); tf = min(np.maximum(t0,t1).min(), t_best)
        if not (tf <= tn):
            if prim[i] >= 0:
                # approximate the hit: box centre distance (good enough for culling statistics)
                t_best = min(t_best, max(tn, 0.001)) if prim[i] != 0 or True else t_best
                i += 1
            else:
                i += 1
        else:
            i = skip[i]
    return steps_by_depth
tot = np.zeros(len(per_level), np.int64); nr = 0
half = np.sqrt(n)/2
for k in range(300):
    # primary
    tgt = np.array([rng.uniform(-half*0.6, half*0.6), 0.2, rng.uniform(-half*0.6, half*0.6)])
    d = tgt - pos; d /= np.linalg.norm(d)
    tot += walk(pos, d); nr += 1
    # secondary: from a ground point, cosine-ish upward direction
    o = np.array([rng.uniform(-half*0.6, half*0.6), 0.0005, rng.uniform(-half*0.6, half*0.6)])
    v = rng.normal(size=3); v /= np.linalg.norm(v); v[1] = abs(v[1])
    tot += walk(o, v); nr += 1
print("steps per ray", tot.sum()/nr)
share = np.cumsum(tot)/tot.sum()
for L in range(len(per_level)):
    print("levels 0..%d: %8d nodes (%8.1f KB as 16-byte nodes)  share of steps %.3f" % (L, cum[L], cum[L]*16/1024, share[L]))
