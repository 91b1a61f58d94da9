#!/usr/bin/env python3
"""Prototype (CPU, numpy): box tests per ray for culling trees built over the reference's LEAF ORDER.
Any conservative hierarchy over the same leaf sequence yields the same primitive tests (leaf box passes imply
ancestor passes), so only the number of inner tests changes."""
import math, sys, time
import numpy as np
sys.path.insert(0, ".")
import tinyrt_amd as t

def leaf_order_and_boxes(desc):
    w, cam = t.world_from_description(desc)
    bbox, prim, skip = w.get_bvh().nodes()
    leaves = [i for i in range(len(prim)) if prim[i] >= 0]
    return bbox, prim, skip, bbox[leaves].astype(np.float64), [prim[i] for i in leaves]

def sa(lo, hi):
    d = np.maximum(hi - lo, 0); return 2*(d[...,0]*d[...,1] + d[...,1]*d[...,2] + d[...,2]*d[...,0])

def build_sah(boxes, prune=None, max_cand=None):
    """returns preorder arrays (bbox, leafidx or -1, skip)"""
    n = len(boxes); out_b=[]; out_l=[]; out_s=[]
    lo = boxes[:, :3]; hi = boxes[:, 3:]
    sys.setrecursionlimit(100000)
    def rec(a, b, parent_sa):
        me = len(out_b)
        blo = lo[a:b].min(0); bhi = hi[a:b].max(0)
        mysa = sa(blo, bhi)
        emit = True
        if b - a > 1 and prune is not None and parent_sa is not None and mysa >= prune * parent_sa:
            emit = False
        if emit:
            out_b.append(np.concatenate([blo, bhi])); out_l.append(-1); out_s.append(0)
        if b - a == 1:
            out_l[me] = a
        else:
            m = b - a
            plo = np.minimum.accumulate(lo[a:b], 0); phi = np.maximum.accumulate(hi[a:b], 0)
            slo = np.minimum.accumulate(lo[a:b][::-1], 0)[::-1]; shi = np.maximum.accumulate(hi[a:b][::-1], 0)[::-1]
            k = np.arange(1, m)
            cost = sa(plo[k-1], phi[k-1]) * k + sa(slo[k], shi[k]) * (m - k)
            split = a + 1 + int(np.argmin(cost))
            rec(a, split, mysa if emit else parent_sa); rec(split, b, mysa if emit else parent_sa)
        if emit: out_s[me] = len(out_b)
    rec(0, n, None)
    return np.array(out_b), np.array(out_l), np.array(out_s)

def count_tests(nb, nl, ns, prims_fn, rays):
    n = len(nl); tot = 0; leafpass = 0
    for o, d in rays:
        inv = 1.0/d; i = 0; tbest = np.inf
        while i < n:
            tot += 1
            b = nb[i]; t0 = (b[:3]-o)*inv; t1 = (b[3:]-o)*inv
            tn = max(np.minimum(t0,t1).max(), 0.001); tf = min(np.maximum(t0,t1).min(), tbest)
            if tf > tn:
                if nl[i] >= 0:
                    leafpass += 1
                    tt = prims_fn(nl[i], o, d, tbest)
                    if tt is not None: tbest = tt
                i += 1
            else: i = ns[i]
    return tot/len(rays), leafpass/len(rays)

def make_prim_fn(desc, leaf_prims):
    geos = desc["geometries"]
    def f(li, o, d, tbest):
        g = geos[leaf_prims[li]]
        if g[0] == "sphere":
            c = np.array(g[1]); r = g[2]; oc = o-c; a = d@d; hb = oc@d; cc = oc@oc-r*r; disc = hb*hb-a*cc
            if disc < 0: return None
            s = math.sqrt(disc); tt = (-hb-s)/a
            if not (0.001 <= tt < tbest): tt = (-hb+s)/a
            return tt if 0.001 <= tt < tbest else None
        corner=np.array(g[1]); u=np.array(g[2]); v=np.array(g[3]); nrm=np.cross(u,v); w=nrm/(nrm@nrm)
        den = d@nrm
        if den == 0: return None
        tt = (nrm@corner - o@nrm)/den
        if not (0.001 <= tt < tbest): return None
        p = o+tt*d-corner; px=np.cross(p,v)@w; py=np.cross(u,p)@w
        return tt if (0<=px<1 and 0<=py<1) else None
    return f

def gen_rays(desc, n, rng, scale):
    cp = np.array(desc["camera"]["position"], float); la = np.array(desc["camera"]["look_at"], float)
    rays = []
    for k in range(n):
        tgt = la + rng.normal(0, scale, 3); d = tgt-cp; d /= np.linalg.norm(d); rays.append((cp, d))
    return rays

def secondary(rays, nb, nl, ns, pf, rng):
    out=[]
    for o,d in rays:
        inv=1.0/d;i=0;tb=np.inf;n=len(nl)
        while i<n:
            b=nb[i];t0=(b[:3]-o)*inv;t1=(b[3:]-o)*inv
            tn=max(np.minimum(t0,t1).max(),0.001);tf=min(np.maximum(t0,t1).min(),tb)
            if tf>tn:
                if nl[i]>=0:
                    tt=pf(nl[i],o,d,tb)
                    if tt is not None: tb=tt
                i+=1
            else:i=ns[i]
        if np.isfinite(tb):
            p=o+tb*d; d2=rng.normal(size=3); d2/=np.linalg.norm(d2)
            if d2@d>0: d2=-d2
            out.append((p+1e-3*d2,d2))
    return out

rng = np.random.default_rng(0)
for name, desc, nr, scale in [("cornell", t.scenes.cornell(), 300, 30.0), ("random_spheres", t.scenes.random_spheres(), 200, 3.0), ("grid100k", t.scenes.sphere_grid(100000), 120, 25.0)]:
    bbox, prim, skip, lboxes, lprims = leaf_order_and_boxes(desc)
    pf = make_prim_fn(desc, lprims)
    # reference tree arrays in the same format
    leafidx = np.full(len(prim), -1); c = 0
    for i in range(len(prim)):
        if prim[i] >= 0: leafidx[i] = c; c += 1
    ref = (bbox.astype(np.float64), leafidx, skip)
    rays = gen_rays(desc, nr, rng, scale)
    rays = rays + secondary(rays, *ref, pf, rng)
    t0 = time.time(); r = count_tests(*ref, pf, rays); print(name, "rays", len(rays), "reference tree: tests/ray %.1f leafpass %.2f nodes %d" % (r[0], r[1], len(prim)))
    for prune in (None, 0.9, 0.75):
        tb = time.time(); tr = build_sah(lboxes, prune); bt = time.time()-tb
        r = count_tests(*tr, pf, rays)
        print("   SAH over fixed leaf order prune=%s: tests/ray %.1f leafpass %.2f nodes %d (build %.1fs)" % (prune, r[0], r[1], len(tr[1]), bt))
