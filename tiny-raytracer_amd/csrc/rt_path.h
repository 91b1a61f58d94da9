// rt_path.h — device building blocks shared by both backends (megakernel and wavefront): scene
// access, the reference-order closest-hit traversal, hit shading / scattering, primary rays.
//
// What is restated here is the reference's hot path, raytracer/src/renderer/sampler/cpu.rs:39-65
// and its callees, plus primary-ray generation (renderer/pointgen.rs:37-52, camera.rs:58-66).
#pragma once

#include "kernels.h"
#include "rt_device.h"

namespace trt {

extern __shared__ float4 g_lds[];          // dynamic LDS: [scene copy (LDS variants)] [backend-private area]

// ------------------------------------------------------------------------------------------------
// Scene access: LDS copy or global blob, same element offsets (scene.h).
// ------------------------------------------------------------------------------------------------
enum { MODE_GLOBAL = 0,      // scene read through L1/L2
       MODE_LDS = 1 };       // the whole hot part of the blob (culling tree, primitives, materials) copied into LDS
// (A third mode - the upper part of a large scene's tree in LDS - was built twice and lost twice: round 4's level-order top levels of the 32-byte
// tree, "slower at every size", and round 5's split of the 16-byte tree with a hand-written exec-mask loop, 7-10 % slower one step per trip and twice
// as slow with LDS lanes running ahead.  The walk of a scene in global memory is bound by VALU issue; LDS hits spare it nothing and the split
// costs instructions.  DESIGN_HISTORY.md B.5, profiles/r05_top_in_lds_walk_ab*.txt, tools/archive/top_in_lds_walk/.)

template <int MODE>
struct SceneAcc {
    const float4* blob;
    SceneLayout L;
    TRT_DEV float4 f4(uint32_t idx) const { return MODE == MODE_LDS ? g_lds[idx] : blob[idx]; }
    TRT_DEV uint32_t u32(uint32_t idx) const {
        return MODE == MODE_LDS ? reinterpret_cast<const uint32_t*>(g_lds)[idx] : reinterpret_cast<const uint32_t*>(blob)[idx];
    }
    // culling tree: the two halves of a node are adjacent (one 32-byte sector per visit)
    TRT_DEV void node(uint32_t i, float4& a, float4& b) const { a = f4(2u * i); b = f4(2u * i + 1u); }
    // reference tree: always read from HBM/L2 (counting kernels and NaN-prone rays only)
    TRT_DEV void ref_node(uint32_t i, float4& a, float4& b) const { a = blob[L.off_ref_nodes + 2u * i]; b = blob[L.off_ref_nodes + 2u * i + 1u]; }
    TRT_DEV float4 sphere(uint32_t i) const { return f4(L.off_sphere + i); }
    // a quad is five consecutive elements (scene.h): one address per quad, the planes at immediate offsets
    TRT_DEV float4 quad(uint32_t plane, uint32_t i) const { return f4(L.off_quad + 5u * i + plane); }
    TRT_DEV float4 material(uint32_t i) const { return f4(L.off_material + i); }
    TRT_DEV uint32_t sphere_material(uint32_t i) const { return u32(L.off_sphere_mat + i); }
    TRT_DEV uint32_t material_kind(uint32_t i) const { return u32(L.off_material_kind + i); }
    // bytes of dynamic LDS the scene copy occupies (the backend-private area starts there)
    TRT_DEV uint32_t lds_bytes() const { return MODE == MODE_LDS ? L.hot_bytes : 0u; }
};

// Cooperative copy into the front of dynamic LDS (whole workgroup; barrier inside).
template <int MODE>
TRT_DEV void stage_scene_to_lds(const SceneDev& sc) {
    if constexpr (MODE != MODE_GLOBAL) {
        const uint32_t n16 = sc.L.hot_bytes >> 4;
        for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x) g_lds[k] = sc.blob[k];
        __syncthreads();
    }
}

// Diagnostic build only (-DTRT_PHASE_CLOCK, tools/phase_clock.sh): shader-clock time per phase of the streamed kernels, per wave
// (s_memtime at the phase boundaries, summed into counters[12..15]: 0 = fetch / generate, 1 = box steps, 2 = leaf phase,
// 3 = shade).  The instrumentation itself costs ~10 % (guides/MI355X_MICROARCH.md); the split is what is read off it.
#ifdef TRT_PHASE_CLOCK
struct PhaseClock {
    uint32_t t[4] = {0u, 0u, 0u, 0u};
    uint64_t last = 0;
    TRT_DEV void start() { last = __builtin_amdgcn_s_memtime(); }
    TRT_DEV void mark(int phase) { const uint64_t now = __builtin_amdgcn_s_memtime(); t[phase] += (uint32_t)(now - last); last = now; }
};
#define TRT_CLK_START(c) (c).clk.start()
#define TRT_CLK(c, phase) (c).clk.mark(phase)
#else
#define TRT_CLK_START(c) ((void)0)
#define TRT_CLK(c, phase) ((void)0)
#endif

template <bool STATS>
struct Counters {
    uint32_t node = 0, sphere = 0, quad_plane = 0, quad_inside = 0, shade = 0;
    uint32_t pend = 0;                                             // leaves put aside by the lock-step walk (diagnostic: tools/leaf_phase_budget.py)
    uint32_t shade_lambertian = 0, shade_metal = 0, shade_dielectric = 0, shade_light = 0;     // `shade` by material kind (bench.py useful_frac)
    uint32_t w_rounds = 0, w_steps = 0, w_leaf = 0, w_gen = 0;     // wave-level trips, counted by the first active lane
#ifdef TRT_PHASE_CLOCK
    PhaseClock clk;
#endif
};
template <>
struct Counters<false> {
#ifdef TRT_PHASE_CLOCK
    PhaseClock clk;
#endif
};

TRT_DEV bool first_active_lane() {
    return (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
}

// ------------------------------------------------------------------------------------------------
// Closest hit: BVH::hit / Node::hit (hittable/bvh.rs:24-27,88-107) with t_range = 0.001..inf
// (cpu.rs:48).
//
// Node::hit tests the box with the interval it was handed; an inner node hands its left child the
// same interval and its right child [t_min, t_left) if the left child hit.  Walking the pre-order
// array with one running t_best, used as the exclusive end for boxes and primitives alike, is that
// recursion unrolled: a primitive is accepted only if t < t_best, so on equal t the primitive that
// comes first in left-first order wins, as in bvh.rs:96-101.
// ------------------------------------------------------------------------------------------------
constexpr float kTMin = 0.001f;

struct Trav {
    V3 inv;                 // 1/d per axis (aabb.rs:42), hoisted out of the node loop
    float t_best;
    uint32_t prim_best;
    uint32_t i;             // pre-order cursor
    uint32_t n;             // node count of the tree this lane walks; i >= n means the walk is over
    bool fast;              // slab_fast is exact for this ray (see rt_device.h)
    bool ref;               // walk the reference tree instead of the culling tree
};

// ref_tree: walk the reference tree whatever the ray (counting kernels: their counters then equal the oracle's).
// A ray whose slab arithmetic can produce NaN (zero / non-finite direction component, non-finite origin) always
// walks the reference tree with the reference's compare-and-assign slab test.
// COMPACT_DOMAIN / compact_domain (every walk that can reach walk_compact's hand-written loop - known at compile time in the specialised kernels,
// at run time in the general ones): the fast path also needs the ray inside the domain in which that loop's fused slab arithmetic is
// conservative - |o| <= SceneLayout::compact_origin_limit per axis and |1/d| <= 2^60 (box_loop_compact).
template <int MODE, bool COMPACT_DOMAIN = false>
TRT_DEV Trav trav_begin(const SceneAcc<MODE>& sc, const Ray& ray, bool ref_tree, bool compact_domain = false) {
    Trav tr;
    tr.inv = v3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);     // (the short division of rt_device.h gains nothing here: measured)
    tr.fast = sc.L.all_finite && finite_f(tr.inv.x) && finite_f(tr.inv.y) && finite_f(tr.inv.z) && finite_f(ray.o.x) &&
              finite_f(ray.o.y) && finite_f(ray.o.z);
    if (COMPACT_DOMAIN || compact_domain) {
        const float big = 1152921504606846976.0f;                    // 2^60
        tr.fast = tr.fast && __builtin_fabsf(ray.o.x) <= sc.L.compact_origin_limit[0] && __builtin_fabsf(ray.o.y) <= sc.L.compact_origin_limit[1] &&
                  __builtin_fabsf(ray.o.z) <= sc.L.compact_origin_limit[2] && __builtin_fabsf(tr.inv.x) <= big && __builtin_fabsf(tr.inv.y) <= big &&
                  __builtin_fabsf(tr.inv.z) <= big;
    }
    tr.ref = ref_tree || !tr.fast;
    tr.n = tr.ref ? sc.L.n_nodes : sc.L.n_cull_nodes;
    tr.t_best = __builtin_inff();
    tr.prim_best = PRIM_NONE;
    tr.i = 0;
    return tr;
}

// One box test at the cursor.  Returns the leaf's primitive reference if the cursor stood on a leaf
// whose box the ray hits (the caller must then run trav_leaf before the next step), else PRIM_NONE.
template <int MODE, bool STATS>
TRT_DEV uint32_t trav_box_step(const SceneAcc<MODE>& sc, const Ray& ray, Trav& tr, Counters<STATS>& ctr) {
    float4 na, nb;
    bool pass;
    if constexpr (STATS) { ctr.node++; if (first_active_lane()) ctr.w_steps++; }
    if (__builtin_expect(!tr.ref, 1)) {                       // common case: culling tree, finite ray (ref == false implies fast)
        sc.node(tr.i, na, nb);
        pass = slab_fast(na, nb, ray.o, tr.inv, kTMin, tr.t_best);
    } else {
        sc.ref_node(tr.i, na, nb);
        if (tr.fast) pass = slab_fast(na, nb, ray.o, tr.inv, kTMin, tr.t_best);
        else pass = slab_exact(na, nb, ray.o, tr.inv, kTMin, tr.t_best);
    }
    const uint32_t link = __float_as_uint(nb.w);
    const bool inner = (link & NODE_INNER_BIT) != 0u;
    tr.i = (pass && inner) ? (link & ~NODE_INNER_BIT) : __float_as_uint(nb.z);     // descend, or skip (a leaf's skip is its successor)
    return (pass && !inner) ? link : PRIM_NONE;
}

// Primitive test with the interval the leaf's box was tested with (bvh.rs:93-94).
template <int MODE, bool STATS>
TRT_DEV void trav_leaf(const SceneAcc<MODE>& sc, const Ray& ray, Trav& tr, uint32_t leaf, Counters<STATS>& ctr) {
    const uint32_t idx = leaf & PRIM_INDEX_MASK;
    if (leaf & PRIM_QUAD_BIT) {                                        // Quad::hit, quad.rs:33-54
        if constexpr (STATS) ctr.quad_plane++;
        // Branch-free: all five planes are requested at once and the inside test is evaluated whatever the plane stage says.
        // Nearly every quad whose box still passes also passes the plane stage (Cornell: 1.08 plane tests and 0.97 inside
        // tests per ray), so the branch saved next to nothing and cost a second LDS round trip in a phase that is
        // latency-bound.  A t outside the range (or inf / NaN from a ray parallel to the plane) only feeds values that the
        // final conjunction discards: the same quads are accepted as with quad.rs:37,41's early returns.
        const float4 q0 = sc.quad(0, idx), q1 = sc.quad(1, idx), q2 = sc.quad(2, idx), q3 = sc.quad(3, idx), q4 = sc.quad(4, idx);
        const V3 nrm = v3(q0.x, q0.y, q0.z);
        const float dir_norm = dot(ray.d, nrm);
        const float t = (q0.w - dot(ray.o, nrm)) / dir_norm;
        const bool in_range = (kTMin <= t) & (t < tr.t_best);
        if constexpr (STATS) { if (in_range) ctr.quad_inside++; }
        const V3 p = ray_at(ray, t) - v3(q1.x, q1.y, q1.z);
        const V3 vv = v3(q2.x, q2.y, q2.z), ww = v3(q2.w, q3.x, q3.y), uu = v3(q3.z, q3.w, q4.x);
        const float planar_x = dot(cross(p, vv), ww);
        const float planar_y = dot(cross(uu, p), ww);
        if (in_range & (0.0f <= planar_x) & (planar_x < 1.0f) & (0.0f <= planar_y) & (planar_y < 1.0f)) {
            tr.t_best = t;
            tr.prim_best = leaf;
        }
    } else {                                                           // Sphere::hit, sphere.rs:29-54
        if constexpr (STATS) ctr.sphere++;
        float t;
        if (sphere_test(sc.sphere(idx), ray, kTMin, tr.t_best, t)) {
            tr.t_best = t;
            tr.prim_best = leaf;
        }
    }
}

// A ray with a NaN in its origin or direction hits NOTHING in the reference, whatever the scene: Sphere::hit forms `a = d.d`, `half_b = oc.d`
// (sphere.rs:31-34) - one of them is NaN, so is the discriminant, `disc < 0` is false, both roots are NaN and `Range::contains` (sphere.rs:40,42)
// is false for NaN; Quad::hit's `t = (d - o.n) / (dir.n)` (quad.rs:34-37) is NaN and fails the same `contains`.  Meanwhile every slab the NaN
// reaches PASSES (aabb.rs:36-61: comparisons with NaN are false, so start / end keep their values), so the reference walks the tree - ALL of it
// when the whole direction is NaN, which is the usual case: `normal + random_unit_vector()` with u3 = 0 normalises a zero vector (vec3extend.rs:
// 16-34; one Lambertian scatter in 2^23) - to find nothing: 2N - 1 box tests and N primitive tests by ONE lane while its wave's other 63 wait.
// Harmless at 18 quads; at 100 k spheres ~70 such rays per 16-spp launch of 3840x2160 each held a wave for tens of milliseconds, and at 4 M
// spheres they WERE the launch (4.8 s of a 4.8 s launch: 2.8 % of the wave slots occupied, profiles/r05_nan_rays_tail.txt).  The production
// kernels therefore answer "miss" at once; the counting kernels (STATS) still walk, because their counters are compared with the oracle's.
TRT_DEV bool ray_has_nan(const Ray& r) {
    return !(r.o.x == r.o.x) | !(r.o.y == r.o.y) | !(r.o.z == r.o.z) | !(r.d.x == r.d.x) | !(r.d.y == r.d.y) | !(r.d.z == r.d.z);
}

// The rare walks: reference tree (counting kernels) and rays whose slab arithmetic needs the reference's
// compare-and-assign form.  Runs the walk `tr` to its end.
template <int MODE, bool STATS>
TRT_DEV void closest_hit_ref(const SceneAcc<MODE>& sc, const Ray& ray, Trav& tr, Counters<STATS>& ctr) {
    if constexpr (!STATS) {
        if (ray_has_nan(ray)) { tr.i = tr.n; return; }              // nothing can be hit (see ray_has_nan): t_best = inf, prim_best = PRIM_NONE
    }
    for (;;) {
        uint32_t leaf = PRIM_NONE;
        while (tr.i < tr.n) {
            leaf = trav_box_step<MODE, STATS>(sc, ray, tr, ctr);
            if (leaf != PRIM_NONE) break;
        }
        if (leaf == PRIM_NONE) break;
        if constexpr (STATS) { if (first_active_lane()) ctr.w_leaf++; }
        trav_leaf<MODE, STATS>(sc, ray, tr, leaf, ctr);
    }
}

// The common walk (finite ray, culling tree, min/max slab test): speculative "while-while" (Aila & Laine 2009).  The lanes
// of a wave step boxes together, then run primitive tests together; a lane that stands on a leaf whose box it hit
// remembers the leaf and keeps stepping boxes until it holds SLOTS of them (or its walk is over); then its primitive
// tests run, in walk order.  With one slot the wave leaves the box loop at every leaf any lane finds and a lane mostly
// waits for its neighbours' longer searches (Cornell: 31 % of the lanes busy in a box step); with four it leaves a
// quarter as often.
//
// Scheduling only - a lane runs the reference's primitive tests, no more and no fewer, and gets the same hit:
//  * the boxes stepped while tests are postponed see the t_best of before those tests, a LARGER interval, so no leaf
//    the reference reaches is skipped;
//  * a postponed leaf is tested only if its box still passes with the CURRENT t_best.  No arithmetic is needed for
//    that: slab_fast passes iff min(t_best, t_far) > start with start = max(t_min, t_near), it did pass with the
//    older, larger t_best (so t_far > start), hence it passes now iff t_best > start - one comparison with the
//    `start` kept beside the postponed leaf.  That is bit for bit the reference's box test at the moment the
//    reference stands on this leaf (same operands, same t_best), and the leaf's ancestors pass whenever the leaf does
//    (DESIGN.md 4.1), so the primitive is tested exactly when the reference tests it.
template <int MODE, bool STATS, int SLOTS>
TRT_DEV void walk_fast(const SceneAcc<MODE>& sc, const Ray& ray, Trav& tr, Counters<STATS>& ctr) {
    const uint32_t n = sc.L.n_cull_nodes;
    for (;;) {
        uint32_t pend[SLOTS];
        float entry[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; k++) { pend[k] = PRIM_NONE; entry[k] = 0.0f; }
        while (tr.i < n && pend[SLOTS - 1] == PRIM_NONE) {
            float4 na, nb;
            sc.node(tr.i, na, nb);
            if constexpr (STATS) { ctr.node++; if (first_active_lane()) ctr.w_steps++; }
            float start;
            const bool pass = slab_fast_entry(na, nb, ray.o, tr.inv, kTMin, tr.t_best, start);
            const uint32_t link = __float_as_uint(nb.w);
            const bool inner = (link & NODE_INNER_BIT) != 0u;
            tr.i = (pass && inner) ? (link & ~NODE_INNER_BIT) : __float_as_uint(nb.z);     // descend, or skip (a leaf's skip is its successor)
            if (pass && !inner) {                                                          // into the first free slot
                bool placed = false;
#pragma unroll
                for (int k = 0; k < SLOTS; k++) {
                    const bool here = !placed && pend[k] == PRIM_NONE;
                    pend[k] = here ? link : pend[k];
                    entry[k] = here ? start : entry[k];
                    placed = placed || here;
                }
            }
        }
        if (pend[0] == PRIM_NONE) break;
        while (pend[0] != PRIM_NONE) {
            if (tr.t_best > entry[0]) {                                                    // the leaf's box test with the current t_best
                if constexpr (STATS) { if (first_active_lane()) ctr.w_leaf++; }
                trav_leaf<MODE, STATS>(sc, ray, tr, pend[0], ctr);
            }
#pragma unroll
            for (int k = 0; k + 1 < SLOTS; k++) { pend[k] = pend[k + 1]; entry[k] = entry[k + 1]; }
            pend[SLOTS - 1] = PRIM_NONE;
        }
    }
}

// Leaf phase of the LDS-stack walks: the lane's `cnt` postponed leaves (slot k at stk[64 k] = (leaf, start)) are tested in
// walk order, slot k in trip k, each only if its box still passes with the CURRENT t_best (t_best > start: see walk_fast).
// Most later slots no longer pass once slot 0's hit has shrunk t_best (Cornell: 1.08 primitive tests per ray out of 3-4
// postponed leaves, 3.4 trips at 31 % of the lanes).  Two ways of running the primitive test only as often as the busiest
// lane needs it were measured in round 2 and rejected (same-box A/B, Cornell): a per-lane forward SCAN to the next slot that
// still passes (fewer vector instructions, 9 % slower: ~20 scalar instructions of EXEC bookkeeping and an exposed LDS round
// trip per slot of the divergent loop), and a survivor bit mask built by all lanes at once after slot 0 (3.7 % slower with
// a second copy of the primitive test for slot 0, 6 % slower with a single test site: the ballot-driven loop and the
// mask bookkeeping cost more than the trips they save, which held two or three lanes each).
// The stack is a bump pointer: `top` is the lane's first free slot (slots are 64 float2 apart), so a push is one LDS write
// and one add, and nothing but the pointer is carried (a slot COUNT cost a second add and a shift-add per push: 18 pushes
// per Cornell walk).
template <int MODE, bool STATS, typename LeafTest>
TRT_DEV void leaf_phase(const float2* stk, const float2* top, Trav& tr, Counters<STATS>& ctr, LeafTest&& test) {
    // One 8-byte LDS read for (leaf, start): read as two words the compiler fetches `start`, waits, branches, fetches `leaf` and
    // waits again - two LDS round trips in a phase that is latency-bound (30 % of Cornell's wave time at 15 % of its vector
    // instructions).  (Requesting slot k+1 before slot k's primitive is tested was measured too: no gain.)
    for (const float2* slot = stk; slot != top; slot += 64) {
        const unsigned long long e = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(slot));
        const float start = __uint_as_float((uint32_t)(e >> 32));
        if (tr.t_best > start) {                                                           // the leaf's box test with the current t_best
            if constexpr (STATS) { if (first_active_lane()) ctr.w_leaf++; }
            test((uint32_t)e);
        }
    }
}

// A walk that is left unfinished (resumable walks below) parks its cursor, its t_best and its best primitive in the lane's leaf stack,
// which is empty at that moment; the caller sets the walk up again from the ray (trav_begin: the three 1/d are recomputed, in the same
// instruction stream in which the wave's other lanes start their new walks) and takes the three values back.  Nothing of the walk is then
// live in registers while the finished lanes are shaded: kept in VGPRs, those 9 registers cost the random-spheres kernel 12 %.
TRT_DEV void trav_park(float2* stk, const Trav& tr) {
    stk[0] = make_float2(__uint_as_float(tr.i), tr.t_best);
    stk[64] = make_float2(__uint_as_float(tr.prim_best), 0.0f);
}
TRT_DEV void trav_unpark(const float2* stk, Trav& tr) {
    const float2 a = stk[0], b = stk[64];
    tr.i = __float_as_uint(a.x);
    tr.t_best = a.y;
    tr.prim_best = __float_as_uint(b.x);
}

// TRT_SLAB_MED3 (round 5): `max(t_min, min(x0, x1))` and `min(t_best, max(x0, x1))` - the x axis' entry folded with t_min, its exit with
// t_best - are each ONE v_med3_f32: med3(x0, x1, c) is c clamped into [min(x0, x1), max(x0, x1)], i.e. max(min(x0, x1), c) whenever
// c <= max(x0, x1) and min(max(x0, x1), c) whenever c >= min(x0, x1).  Where that does not hold the box is rejected either way: t_min >=
// max(x0, x1) makes the reference's end <= exit_x <= t_min <= start, and here start' >= med3 = max(x0, x1) >= exit_x' >= end'; t_best <=
// min(x0, x1) makes the reference's end <= t_best <= entry_x <= start, and here end' <= med3 = min(x0, x1) <= entry_x' <= start'.  In every
// other case both medians ARE the reference's values, so `start`, `end` and the decision are the reference's bit for bit (all operands
// finite or infinite, never NaN: the fast-slab domain).  Two instructions fewer per box step in all three hand-written loops: 23 / 27 / 35
// vector instructions per box instead of 25 / 29 / 37.
//
// The box-step loop of walk_fast_lds, written by hand (round 3).  The loop the compiler builds from the C++ below spends 22 scalar
// instructions per trip on 29 vector ones - the structuriser's exec-mask bookkeeping for `while (a && b) { ...; if (c) push; if (d) break; }` -
// and the scalar unit issues one instruction per ~4.8 cycles per SIMD (tools/micro/salu_rate.hip: 24 scalar instructions ride free on 32
// vector ones, 32 do not), so the trip was bound by its scalar half.  Here a trip is 29 vector + 10 scalar instructions: the lanes that
// leave the loop (walk over, or leaf stack full) are dropped from exec for good, a leaf is put aside under a saved exec mask, and the
// stragglers exit is one s_bcnt1 + compare.  Same operations on the same values as slab_fast_entry + the C++ loop body (min / max of
// non-NaN values are exact, so their order is free); exec is restored on exit.  v48-v57 are fixed because a 128-bit LDS read needs four
// consecutive registers and inline-asm operands cannot be taken apart.  Returns the lane's new stack top.
// (Round 4 measured the LDS bank conflicts of the two ds_read_b128 - 22 % of the LDS index cycles - by keeping the nodes' two halves in two planes of
// the LDS copy: conflicts 3.09 -> 1.80 G cycles per launch, launch 0.4 % SLOWER for the one extra address instruction per trip.  The conflicts cost less
// than one vector instruction per box step: profiles/r04_lds_split_nodes_ab.txt, tools/archive/lds_split_nodes/.)
#ifndef TRT_ASM_BOX_LOOP
#define TRT_ASM_BOX_LOOP 1
#endif
constexpr bool kAsmBoxLoop = TRT_ASM_BOX_LOOP != 0;

TRT_DEV uint32_t lds_offset(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }

TRT_DEV float2* box_loop_lds(Trav& tr, const V3& o, float2* stk, float2* limit, uint32_t n, uint32_t few) {
    const uint32_t stk_off = lds_offset(stk);
    uint32_t top = stk_off;
    const uint32_t lim = lds_offset(limit), base = lds_offset(g_lds);
    unsigned long long saved, m0, m1;
    uint32_t cnt;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "1:\n"
        "v_cmp_gt_u32_e32 vcc, %[n], %[i]\n"                 // tr.i < n
        "v_cmp_ne_u32_e64 %[m0], %[top], %[lim]\n"           // top != limit
        "s_and_b64 vcc, vcc, %[m0]\n"
        "s_and_b64 exec, exec, vcc\n"                        // lanes that fail either leave the loop for good
        "s_cbranch_scc0 2f\n"
        "v_lshl_add_u32 v52, %[i], 5, %[base]\n"
        "ds_read_b128 v[48:51], v52\n"                       // lo.x lo.y lo.z hi.x
        "ds_read_b128 v[52:55], v52 offset:16\n"             // hi.y hi.z skip link
        "s_waitcnt lgkmcnt(1)\n"
        "v_sub_f32_e32 v48, v48, %[ox]\n"
        "v_sub_f32_e32 v51, v51, %[ox]\n"
        "v_sub_f32_e32 v49, v49, %[oy]\n"
        "v_sub_f32_e32 v50, v50, %[oz]\n"
        "v_mul_f32_e32 v48, v48, %[ix]\n"
        "v_mul_f32_e32 v51, v51, %[ix]\n"
        "v_mul_f32_e32 v49, v49, %[iy]\n"
        "v_mul_f32_e32 v50, v50, %[iz]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_sub_f32_e32 v52, v52, %[oy]\n"
        "v_sub_f32_e32 v53, v53, %[oz]\n"
        "v_mul_f32_e32 v52, v52, %[iy]\n"
        "v_mul_f32_e32 v53, v53, %[iz]\n"
        "v_med3_f32 v56, v48, v51, %[tmin]\n"                // max(t_min, entry x)  (TRT_SLAB_MED3 below: the median where it matters)
        "v_med3_f32 v48, v48, v51, %[tb]\n"                  // min(t_best, exit x)
        "v_min_f32_e32 v57, v49, v52\n"
        "v_max_f32_e32 v49, v49, v52\n"
        "v_min_f32_e32 v51, v50, v53\n"
        "v_max_f32_e32 v50, v50, v53\n"
        "v_max3_f32 v56, v56, v57, v51\n"                    // start = max(t_min, entries)
        "v_min3_f32 v48, v48, v49, v50\n"                    // end = min(t_best, exits)
        "v_cmp_nle_f32_e32 vcc, v48, v56\n"                  // pass = !(end <= start)
        "v_cmp_gt_i32_e64 %[m0], 0, v55\n"                   // inner node: NODE_INNER_BIT is the sign bit of the link
        "s_and_b64 %[m1], vcc, %[m0]\n"
        "v_cndmask_b32_e64 %[i], v54, |v55|, %[m1]\n"        // descend (link without the bit), or skip (a leaf's skip is its successor)
        "s_andn2_b64 %[m1], vcc, %[m0]\n"                    // a leaf whose box passes:
        "s_and_saveexec_b64 %[m0], %[m1]\n"
        "ds_write2_b32 %[top], v55, v56 offset1:1\n"         //   put (leaf, start) aside (two dwords: an odd-aligned register pair is no tuple)
        "v_add_u32_e32 %[top], 0x200, %[top]\n"
        "s_mov_b64 exec, %[m0]\n"
        "s_bcnt1_i32_b64 %[cnt], exec\n"                     // resumable walk: go on only while more than `few` lanes still step boxes
        "s_cmp_gt_u32 %[cnt], %[few]\n"
        "s_cbranch_scc1 1b\n"
        "2:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [i] "+v"(tr.i), [top] "+v"(top), [sv] "=&s"(saved), [m0] "=&s"(m0), [m1] "=&s"(m1), [cnt] "=&s"(cnt)
        : [n] "s"(n), [few] "s"(few), [base] "s"(base), [lim] "v"(lim), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(tr.inv.x),
          [iy] "v"(tr.inv.y), [iz] "v"(tr.inv.z), [tb] "v"(tr.t_best), [tmin] "s"(kTMin)
        : "vcc", "scc", "memory", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
    return stk + ((top - stk_off) >> 3);
}

// The same walk with the postponed leaves in LDS instead of registers: `stk` is this lane's slot 0, slot k lives at
// stk[64 * k] (one 8-byte (leaf, start) pair per lane and slot, lane-contiguous: conflict-free ds_write_b64 /
// ds_read_b64).  Putting a leaf aside costs one address, one LDS write and one add instead of the compare/select
// chain over SLOTS registers (13 VALU instructions per box-step trip at 4 slots: the wave pays them whenever ANY lane
// finds a leaf, which is nearly every trip), and the slots cost no VGPRs.  LDS operations of one wave complete in
// order, so a lane reads back what it wrote without a barrier.
// Resumable (round 3), like walk_compact: with `stragglers` > 0 the function returns false - walk unfinished, tr holds where it stands, the
// leaf stack is empty - as soon as at most that many lanes of the wave still walk while others have finished.  The walk lengths of a wave's
// 64 rays are spread widely (random-spheres: mean 26 box steps, 95th percentile 47, longest of 64 about 55), and a round runs as many trips
// as its longest walk: carrying the few long ones into the next round cuts the trips per ray by a third in the model
// (tools/proto/cull_tree_model.c + the walk-length replay in profiles/r03_stragglers_model.txt).
template <int MODE, bool STATS>
TRT_DEV bool walk_fast_lds(const SceneAcc<MODE>& sc, const Ray& ray, Trav& tr, Counters<STATS>& ctr, float2* stk, uint32_t slots,
                           uint32_t stragglers = 0u, uint32_t entered = 64u) {
    const uint32_t n = sc.L.n_cull_nodes;
    float2* const limit = stk + 64u * slots;
    // the walk is left once at most `few` lanes still walk: no more than `stragglers`, and fewer than came in (so that a round always
    // finishes some walk); 0 = never (at least one lane is active wherever it is compared)
    const uint32_t few = stragglers < entered ? stragglers : entered - 1u;
    for (;;) {
        float2* top = stk;
        if constexpr (kAsmBoxLoop && !STATS && MODE == MODE_LDS) {
            top = box_loop_lds(tr, ray.o, stk, limit, n, few);
        } else
        while (tr.i < n && top != limit) {
            float4 na, nb;
            sc.node(tr.i, na, nb);
            if constexpr (STATS) { ctr.node++; if (first_active_lane()) ctr.w_steps++; }
            float start;
            const bool pass = slab_fast_entry(na, nb, ray.o, tr.inv, kTMin, tr.t_best, start);
            const uint32_t link = __float_as_uint(nb.w);
            const bool inner = (link & NODE_INNER_BIT) != 0u;
            tr.i = (pass && inner) ? (link & ~NODE_INNER_BIT) : __float_as_uint(nb.z);     // descend, or skip (a leaf's skip is its successor)
            if (pass && !inner) {
                *top = make_float2(nb.w, start);
                top += 64;
            }
            // resumable: once only a few lanes are still stepping boxes (the others finished, or wait with a full leaf stack), stop
            // stepping: the pending leaves are tested and the check below decides whether the walk is left (s_bcnt1 of the exec mask
            // and one scalar compare per trip)
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) <= few) break;
        }
        TRT_CLK(ctr, 1);
        if (top == stk && tr.i >= n) return true;
        if (top != stk) leaf_phase<MODE, STATS>(stk, top, tr, ctr, [&](uint32_t leaf) { trav_leaf<MODE, STATS>(sc, ray, tr, leaf, ctr); });
        TRT_CLK(ctr, 2);
        if (tr.i >= n) return true;
        if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) <= few) { trav_park(stk, tr); return false; }
    }
}

// The box-step loop of walk_flat by hand (round 3; conventions of box_loop_lds).  The leaf pair under test lives in s[36:51] - sixteen fixed
// SGPRs, because a 64-byte scalar load needs an aligned tuple and inline-asm operands cannot be taken apart - and the next pair is requested
// as soon as the last value of the current one has been read (the link of the second leaf is copied out first); the wave's other lanes'
// work covers the rest of that load's latency.  25 vector instructions per box (23 for the slab test + the link copy and the stack-top add
// under the push mask) and 6.5 scalar ones, against 26 + 12.5 from the C++ loop below.  `i` is the next leaf to test (wave-uniform, in and out);
// the loop ends when i == n or when some lane cannot hold another pair of leaves.  Returns the lane's new stack top.
#define TRT_FLAT_BOX(LOX, LOY, LOZ, HIX, HIY, HIZ)                                                                                        \
    "v_sub_f32_e32 %[t0], " LOX ", %[ox]\n v_sub_f32_e32 %[t1], " HIX ", %[ox]\n v_sub_f32_e32 %[t2], " LOY ", %[oy]\n"                   \
    "v_sub_f32_e32 %[t3], " HIY ", %[oy]\n v_mul_f32_e32 %[t0], %[ix], %[t0]\n v_mul_f32_e32 %[t1], %[ix], %[t1]\n"                       \
    "v_mul_f32_e32 %[t2], %[iy], %[t2]\n v_mul_f32_e32 %[t3], %[iy], %[t3]\n v_sub_f32_e32 %[t4], " LOZ ", %[oz]\n"                      \
    "v_sub_f32_e32 %[t5], " HIZ ", %[oz]\n v_mul_f32_e32 %[t4], %[iz], %[t4]\n v_mul_f32_e32 %[t5], %[iz], %[t5]\n"                      \
    "v_med3_f32 %[st], %[t0], %[t1], %[tmin]\n v_med3_f32 %[t0], %[t0], %[t1], %[tb]\n v_min_f32_e32 %[t1], %[t2], %[t3]\n"              \
    "v_max_f32_e32 %[t2], %[t2], %[t3]\n v_min_f32_e32 %[t3], %[t4], %[t5]\n v_max_f32_e32 %[t4], %[t4], %[t5]\n"                        \
    "v_max3_f32 %[st], %[st], %[t1], %[t3]\n v_min3_f32 %[t0], %[t0], %[t2], %[t4]\n v_cmp_nle_f32_e32 vcc, %[t0], %[st]\n"

TRT_DEV float2* box_loop_flat(const Trav& tr, const V3& o, const float4* __restrict__ leaf_list, uint32_t& i, uint32_t n, float2* stk, float2* limit) {
    const uint32_t stk_off = lds_offset(stk);
    uint32_t top = stk_off;
    const uint32_t lim = lds_offset(limit);
    unsigned long long m;
    uint32_t off;
    float t0, t1, t2, t3, t4, t5, st, lk;
    asm volatile(
        "s_lshl_b32 %[off], %[i], 5\n"                              // a leaf is 32 bytes: (lo.x lo.y lo.z hi.x) (hi.y hi.z skip link)
        "s_load_dwordx16 s[36:51], %[list], %[off]\n"               // leaves i and i + 1 (the list is padded: scene.h kLeafListPad)
        "1:\n"
        "s_waitcnt lgkmcnt(0)\n"
        TRT_FLAT_BOX("s36", "s37", "s38", "s39", "s40", "s41")
        "s_and_saveexec_b64 %[m], vcc\n"                            // box passes: put (leaf, start) aside
        "v_mov_b32_e32 %[lk], s43\n"
        "ds_write2_b32 %[top], %[lk], %[st] offset1:1\n"
        "v_add_u32_e32 %[top], 0x200, %[top]\n"
        "s_mov_b64 exec, %[m]\n"
        "s_add_u32 %[i], %[i], 1\n"
        "s_cmp_ge_u32 %[i], %[n]\n"
        "s_cbranch_scc1 2f\n"                                       // an odd last leaf
        TRT_FLAT_BOX("s44", "s45", "s46", "s47", "s48", "s49")
        "v_mov_b32_e32 %[lk], s51\n"                                // last value of this pair: the next one may come
        "s_add_u32 %[off], %[off], 64\n"
        "s_load_dwordx16 s[36:51], %[list], %[off]\n"
        "s_and_saveexec_b64 %[m], vcc\n"
        "ds_write2_b32 %[top], %[lk], %[st] offset1:1\n"
        "v_add_u32_e32 %[top], 0x200, %[top]\n"
        "s_mov_b64 exec, %[m]\n"
        "s_add_u32 %[i], %[i], 1\n"
        "v_cmp_gt_u32_e32 vcc, %[top], %[lim]\n"                    // some lane cannot hold another pair: test what is pending
        "s_cmp_lg_u64 vcc, 0\n"
        "s_cbranch_scc1 2f\n"
        "s_cmp_lt_u32 %[i], %[n]\n"
        "s_cbranch_scc1 1b\n"
        "2:\n"
        "s_waitcnt lgkmcnt(0)\n"                                    // nothing may land in s[36:51] after the block
        : [i] "+s"(i), [top] "+v"(top), [m] "=&s"(m), [off] "=&s"(off), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [t4] "=&v"(t4), [t5] "=&v"(t5), [st] "=&v"(st), [lk] "=&v"(lk)
        : [n] "s"(n), [list] "s"(leaf_list), [lim] "v"(lim), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(tr.inv.x), [iy] "v"(tr.inv.y),
          [iz] "v"(tr.inv.z), [tb] "v"(tr.t_best), [tmin] "s"(kTMin)
        : "vcc", "scc", "memory", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
    return stk + ((top - stk_off) >> 3);
}

// Scenes with a handful of primitives (SceneLayout::flat_walk): no tree at all.  Every lane of the wave steps the SAME
// leaf box in the same trip - the list of leaves in walk order - so the node is wave-uniform: it comes through the
// scalar cache into SGPRs (`leaf_list` must be a __restrict__ kernel argument for that), there is no per-lane address,
// no LDS read and no cursor, and all lanes are busy in every box step: 18 trips of ~28 VALU instructions for
// Cornell's 18 quads against ~20 trips of ~35 at 53 % occupancy through the culling tree.  It is the culling tree
// with every inner node pruned, so the argument of walk_fast applies unchanged: leaves whose box passes are put
// aside with their `start` and tested in walk order, each only if its box still passes with the current t_best.
// All lanes that call this must enter together (they do: a wave's lanes start their walks in the same trip).
// The loop is software-pipelined by hand: a trip tests TWO leaves, and the scalar loads of the next trip's pair are
// issued before this trip's boxes are tested (the compiler otherwise waits for each node right after requesting it,
// `s_load_dwordx4; s_waitcnt lgkmcnt(0)`, once per box step, and spends 3 branches and ~10 scalar instructions per step).
template <int MODE, bool STATS>
TRT_DEV void walk_flat(const SceneAcc<MODE>& sc, const float4* __restrict__ leaf_list, const Ray& ray, Trav& tr, Counters<STATS>& ctr,
                       float2* stk, uint32_t slots) {
    const uint32_t n = sc.L.n_leaves;                    // >= 1
    uint32_t i = 0;                                      // wave-uniform
    if constexpr (kAsmBoxLoop && !STATS) {
        float2* const lim = stk + 64u * (slots - 2u);
        do {
            float2* const top = box_loop_flat(tr, ray.o, leaf_list, i, n, stk, lim);
            TRT_CLK(ctr, 1);
            leaf_phase<MODE, STATS>(stk, top, tr, ctr, [&](uint32_t leaf) { trav_leaf<MODE, STATS>(sc, ray, tr, leaf, ctr); });
            TRT_CLK(ctr, 2);
        } while (i < n);
        return;
    }
    const float4* __restrict__ pair = leaf_list;         // wave-uniform: the pair under test; the list is padded (scene.h kLeafListPad), so
                                                         // reading one pair ahead needs no clamping: ONE 64-byte scalar load per trip
    float4 a0 = pair[0], b0 = pair[1], a1 = pair[2], b1 = pair[3];
    float2* const limit = stk + 64u * (slots - 2u);      // a lane whose top is beyond it cannot hold another pair (slots >= 2)
    do {
        float2* top = stk;
        for (; i < n;) {
            // request the next pair
            pair += 4;
            const float4 na0 = pair[0], nb0 = pair[1], na1 = pair[2], nb1 = pair[3];
            if constexpr (STATS) { ctr.node++; if (first_active_lane()) ctr.w_steps++; }
            float start;
            if (slab_fast_entry(a0, b0, ray.o, tr.inv, kTMin, tr.t_best, start)) {
                *top = make_float2(b0.w, start);
                top += 64;
            }
            if (i + 1u < n) {
                if constexpr (STATS) { ctr.node++; if (first_active_lane()) ctr.w_steps++; }
                if (slab_fast_entry(a1, b1, ray.o, tr.inv, kTMin, tr.t_best, start)) {
                    *top = make_float2(b1.w, start);
                    top += 64;
                }
            }
            i += 2u;
            a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
            if (__builtin_amdgcn_ballot_w64(top > limit) != 0ull) break;         // some lane could not hold another pair: test what is pending
        }
        TRT_CLK(ctr, 1);
        if constexpr (STATS) ctr.pend += (uint32_t)(top - stk) >> 6;
        leaf_phase<MODE, STATS>(stk, top, tr, ctr, [&](uint32_t leaf) { trav_leaf<MODE, STATS>(sc, ray, tr, leaf, ctr); });
        TRT_CLK(ctr, 2);
    } while (i < n);
}

// Scenes too large for LDS are bound by the vector-memory front end, not by arithmetic: every box step of every lane is
// two divergent 16-byte loads (100 k spheres: 9.96 G wave loads per launch, one per 28 cycles per CU, VALU 35 % busy).
// walk_compact steps a 16-BYTE node instead - the culling tree's boxes rounded OUTWARD to f16, one load per step, half
// the tree's footprint (3 MB: it now fits one XCD's L2).  Legal because the hierarchy only has to be conservative
// (DESIGN.md 4.1): a box that contains the exact box passes whenever the exact box does (same slab arithmetic, monotonic
// rounding), so no leaf the reference reaches is skipped; and a leaf whose coarse box passes is put aside and, at its
// turn, tested against its EXACT f32 box with the current t_best - the reference's own leaf-box test - before its
// primitive is touched.  An inner node's first child is the next node (pre-order), so its fourth word is the skip
// link; a leaf's is LEAF | its sequence number in the leaf list (exact box + primitive reference).
constexpr uint32_t kCompactLeafBit = 0x80000000u;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// The exact-box re-test and the primitive test of a postponed leaf of walk_compact (its leaf_phase body).
template <int MODE, bool STATS>
TRT_DEV void compact_leaf_test(const SceneAcc<MODE>& sc, const float4* __restrict__ leaf_list, const Ray& ray, Trav& tr, uint32_t leaf, Counters<STATS>& ctr) {
    leaf &= ~kCompactLeafBit;                                        // (box_loop_compact puts the node's fourth word aside as it is)
    const float4 na = leaf_list[2u * leaf], nb = leaf_list[2u * leaf + 1u];
    if constexpr (STATS) ctr.node++;
    if (slab_fast(na, nb, ray.o, tr.inv, kTMin, tr.t_best)) trav_leaf<MODE, STATS>(sc, ray, tr, __float_as_uint(nb.w), ctr);
}

// The box-step loop of walk_compact by hand, like box_loop_lds (same reasons, same conventions): one 16-byte node per trip through the
// scalar-base form of global_load (a 32-bit byte offset in ONE register instead of a 64-bit address in two).  Round 3: 37 vector + 10 scalar
// instructions per trip instead of the compiler's 38 + 21; round 5: 21 vector.
// FUSED SLAB ARITHMETIC (round 5).  After the NaN-ray fix this walk is bound by VALU issue (3.25 cycles per wave-instruction per SIMD against ~2.7
// for its mix at full issue), and a third of a trip was the reference's `(x - o) * inv` on six planes: six subtractions, six multiplications.  The
// COARSE walk need not be the reference's arithmetic - only conservative: a leaf whose coarse box passes is re-tested on its exact f32 box with
// the reference's own slab test before its primitive is touched (walk_compact below), so a coarse test may accept too much, never too little.
// Here a plane's distance is g = fma(x, inv, -m), m = fl(o * inv) (one v_mul per axis and call), and the f16 -> f32 conversion of x rides in the same
// instruction (v_fma_mix_f32): six instructions per trip instead of the eighteen of rounds 3-4 (six conversions, six subtractions, six multiplications).
// Error against the real number F(x) = (x - o) inv (inv the f32 value both sides use): |g - F(x)| <= u |inv| (|x| + 2.001 |o|), u = 2^-24; the
// reference's f(x) = fl(fl(x - o) inv) on the EXACT plane x0 errs by at most 2.001 u |inv| (|x0| + |o|).  The node boxes were grown by
// eps = 2^-19 B per axis before their outward rounding to f16 (scene_host.cpp; B = largest |coordinate| on the axis): for |o| <= 4 B that is
// u (19.1 B) <= 2^-19 B / 1.67, more than both errors together, so for an entry plane g(x') <= f(x0) and for an exit plane g(x') >= f(x0): the
// coarse interval contains the reference's interval on the exact box, min / max / med3 are monotonic, hence "exact box passes" implies "coarse
// box passes" and the coarse `start` is no later than the exact one (what the leaf phase's early drop relies on).  Rays outside the domain
// (|o| > 4 B on an axis, |inv| > 2^60) are not `fast` (trav_begin COMPACT_DOMAIN) and walk the reference tree.  Overflow: |x inv| <= 1e12 * 2^60.
// The cursor tr.i of this walk is the node's BYTE offset (16 i) and the inner nodes' links are stored that way (scene_host.cpp), so that a trip
// needs no shift before its load; `end` = 16 n.
TRT_DEV float2* box_loop_compact(Trav& tr, const V3& o, const uint4* __restrict__ nodes16, float2* stk, float2* limit, uint32_t end, uint32_t few) {
    const uint32_t stk_off = lds_offset(stk);
    uint32_t top = stk_off;
    const uint32_t lim = lds_offset(limit);
    unsigned long long saved, m0, m1;
    uint32_t cnt;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "v_mul_f32_e64 v57, -%[ox], %[ix]\n"                 // -m = fl(-o * 1/d) per axis, once per call: a plane's distance is fma(x, 1/d, -m)
        "v_mul_f32_e64 v58, -%[oy], %[iy]\n"
        "v_mul_f32_e64 v59, -%[oz], %[iz]\n"
        "1:\n"
        "v_cmp_gt_u32_e32 vcc, %[n], %[i]\n"                 // tr.i < end (both in bytes)
        "v_cmp_ne_u32_e64 %[m0], %[top], %[lim]\n"           // top != limit
        "s_and_b64 vcc, vcc, %[m0]\n"
        "s_and_b64 exec, exec, vcc\n"
        "s_cbranch_scc0 2f\n"
        "global_load_dwordx4 v[48:51], %[i], %[nodes]\n"     // (lo.x lo.y) (lo.z hi.x) (hi.y hi.z) as f16 pairs, link; the cursor IS the byte offset
        "s_waitcnt vmcnt(0)\n"
        // one instruction per plane: v_fma_mix_f32 reads the f16 coordinate straight out of the low / high half of the node's word, converts it
        // exactly and fuses x / d - m with ONE rounding (the reference: two, of (x - o) / d; conservative on these boxes - they were grown for it:
        // see above the function).  Bit-identical to v_cvt_f32_f16 + v_fma_f32 for every f16 pattern: tools/micro/fma_mix_exact.hip.
        "v_fma_mix_f32 v52, v48, %[ix], v57 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"      // lo.x
        "v_fma_mix_f32 v53, v48, %[iy], v58 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"      // lo.y
        "v_fma_mix_f32 v54, v49, %[iz], v59 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"      // lo.z
        "v_fma_mix_f32 v49, v49, %[ix], v57 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"      // hi.x
        "v_fma_mix_f32 v48, v50, %[iy], v58 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"      // hi.y
        "v_fma_mix_f32 v50, v50, %[iz], v59 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"      // hi.z
        "v_med3_f32 v55, v52, v49, %[tmin]\n"                // max(t_min, entry x)   (TRT_SLAB_MED3)
        "v_med3_f32 v52, v52, v49, %[tb]\n"                  // min(t_best, exit x)
        "v_min_f32_e32 v56, v53, v48\n"
        "v_max_f32_e32 v48, v53, v48\n"
        "v_min_f32_e32 v49, v54, v50\n"
        "v_max_f32_e32 v54, v54, v50\n"
        "v_max3_f32 v55, v55, v56, v49\n"                    // start
        "v_min3_f32 v52, v52, v48, v54\n"                    // end
        "v_cmp_nle_f32_e32 vcc, v52, v55\n"                  // pass = !(end <= start)
        "v_cmp_gt_i32_e64 %[m0], 0, v51\n"                   // leaf: kCompactLeafBit is the sign bit of the fourth word
        "s_or_b64 %[m1], vcc, %[m0]\n"
        "v_add_u32_e32 v56, 16, %[i]\n"
        "v_cndmask_b32_e64 %[i], v51, v56, %[m1]\n"          // first child / a leaf's successor, or the skip link (a byte offset too)
        "s_and_b64 %[m1], vcc, %[m0]\n"                      // a leaf whose coarse box passes:
        "s_and_saveexec_b64 %[m0], %[m1]\n"
        "ds_write2_b32 %[top], v51, v55 offset1:1\n"         //   put (LEAF | leaf sequence number, coarse start) aside (compact_leaf_test strips the bit)
        "v_add_u32_e32 %[top], 0x200, %[top]\n"
        "s_mov_b64 exec, %[m0]\n"
        "s_bcnt1_i32_b64 %[cnt], exec\n"
        "s_cmp_gt_u32 %[cnt], %[few]\n"
        "s_cbranch_scc1 1b\n"
        "2:\n"
        "s_mov_b64 exec, %[sv]\n"
        : [i] "+v"(tr.i), [top] "+v"(top), [sv] "=&s"(saved), [m0] "=&s"(m0), [m1] "=&s"(m1), [cnt] "=&s"(cnt)
        : [n] "s"(end), [few] "s"(few), [nodes] "s"(nodes16), [lim] "v"(lim), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(tr.inv.x),
          [iy] "v"(tr.inv.y), [iz] "v"(tr.inv.z), [tb] "v"(tr.t_best), [tmin] "s"(kTMin)
        : "vcc", "scc", "memory", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
    return stk + ((top - stk_off) >> 3);
}

// Resumable: with `stragglers` > 0 the function returns false - walk unfinished, tr holds where it stands - as soon as at most
// that many lanes of the wave are still walking while others have finished (`entered` = lanes that came in); the leaf stack is
// empty at that point, so nothing but tr has to be kept.  The caller shades the finished lanes, gives them their next ray, and
// calls again: the stragglers go on from tr.i while the new walks start beside them, instead of 50 lanes idling through the
// last third of the trips behind a few long walks (100 k spheres: 324 trips per round for 222 box steps per ray).  Measured:
// +10 % on that scene in round 2; since round 3 (exit test inside the box-step loop, walk state parked in the leaf stack) the
// LDS-resident tree walk is resumable too (walk_fast_lds: random-spheres +10 %).
template <int MODE, bool STATS>
TRT_DEV bool walk_compact(const SceneAcc<MODE>& sc, const uint4* __restrict__ nodes16, const float4* __restrict__ leaf_list,
                          const Ray& ray, Trav& tr, Counters<STATS>& ctr, float2* stk, uint32_t slots, uint32_t stragglers = 0u,
                          uint32_t entered = 64u) {
    const uint32_t n = sc.L.n_cull_nodes << 4;                                         // tr.i counts bytes in this walk (box_loop_compact)
    float2* const limit = stk + 64u * slots;
    const uint32_t few = stragglers < entered ? stragglers : entered - 1u;             // see walk_fast_lds
    for (;;) {
        float2* top = stk;
        // (Measured and rejected in round 2: requesting nodes i and i + 1 together and stepping i + 1 from the data already
        // there when the walk goes on to it - the successor IS the next node whenever the box passes or the node is a leaf.  It
        // shortens the chain of dependent loads by a third, and is 15 % slower on the 100 k-sphere scene: twice the vector-memory
        // instructions, and the second step runs with about half of the lanes.)
        if constexpr (kAsmBoxLoop && !STATS) {
            top = box_loop_compact(tr, ray.o, nodes16, stk, limit, n, few);
        } else
        while (tr.i < n && top != limit) {
            const uint4 q = nodes16[tr.i >> 4];
            if constexpr (STATS) { ctr.node++; if (first_active_lane()) ctr.w_steps++; }
            const half2_t a = __builtin_bit_cast(half2_t, q.x), b = __builtin_bit_cast(half2_t, q.y), c = __builtin_bit_cast(half2_t, q.z);
            float start;
            const bool pass = slab_fast6_entry(v3((float)a.x, (float)a.y, (float)b.x), v3((float)b.y, (float)c.x, (float)c.y), ray.o, tr.inv,
                                               kTMin, tr.t_best, start);
            const bool is_leaf = (q.w & kCompactLeafBit) != 0u;
            const uint32_t next = tr.i + 16u;                                        // first child, or a leaf's successor
            if (pass && is_leaf) {
                *top = make_float2(__uint_as_float(q.w & ~kCompactLeafBit), start);
                top += 64;
            }
            tr.i = (pass || is_leaf) ? next : q.w;
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) <= few) break;      // see walk_fast_lds
        }
        TRT_CLK(ctr, 1);
        if (top == stk && tr.i >= n) return true;
        // The coarse box contains the exact one, so its interval starts no later: a leaf whose COARSE start is not below the
        // current t_best fails the exact test too and is dropped by the scan without touching memory; the others take the
        // reference's leaf-box test on the exact f32 box, at the leaf's turn.
        if (top != stk) leaf_phase<MODE, STATS>(stk, top, tr, ctr, [&](uint32_t leaf) { compact_leaf_test<MODE, STATS>(sc, leaf_list, ray, tr, leaf, ctr); });
        TRT_CLK(ctr, 2);
        if (tr.i >= n) return true;
        if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) <= few) { trav_park(stk, tr); return false; }      // resumable: see walk_fast_lds
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Two walks per lane (round 4; trt_tuning.dual_walk).  walk_compact keeps ONE 16-byte load in flight per wave, and a trip waits for it.  Eight is
// the most waves a SIMD holds, so more loads in flight have to come from inside a wave: every lane walks TWO rays, A and B, each exactly as
// walk_compact walks one (same node order, same tests, same t_best evolution - the two walks share nothing but the instruction stream), and a trip
// requests both rays' nodes before it waits for the first.
// MEASURED.  Round 4 (profiles/r04_dual_walk_sweep.txt) found +0..+2 % on the 100 k-sphere scene and concluded that the walk "saturates at 3.0
// Gray/s"; those launches carried NaN rays walking the whole tree (ray_has_nan above).  Round 5, without them and with box_loop_compact's fused,
// mixed-precision box step in this loop too (profiles/r05_dual_walk_fused_ab.txt): where the tree fits L2 one path per lane at 7 waves beats two
// paths at 4 / 5 / 6 / 7 waves (5.49 against 4.82 / 5.21 / 5.34 / 5.17 Gray/s), where the walk's loads really miss - sphere_field, 1 M / 4 M spheres,
// L2 hit rate 0.948 - two paths at 6 waves win 2.7 % / 4.2 %.  The launch plan therefore takes this kernel by scene size (dual_walk = 0: beyond the
// 32 MiB of L2; 1: wherever it exists; 2: never).
// box_loop_compact2 is box_loop_compact's body twice between the two requests and the two waits; the registers it works in (v44-v56) are shared by
// the two halves, the loads land in v[44:47] and v[48:51].
// A walk that is not active (no ray in that slot, or its walk is over) comes in with i == n; a slot's lanes are masked by an SGPR
// pair recomputed every trip (i < n && top != limit), and the loop ends when at most `few` RAYS of the wave (both slots counted) can
// still step - 0 when the last one is done.  A VMEM instruction issued with exec == 0 moves no data but keeps vmcnt in step, so both
// requests are always issued and vmcnt(1) / vmcnt(0) mean "A's node" / "B's node" whatever the masks; a half whose mask is empty
// skips its arithmetic (s_cbranch_execz).
#define TRT_COMPACT_STEP(D0, D1, D2, D3, I, TOP)                                                                                              \
        /* the fused, mixed-precision slab arithmetic of box_loop_compact (round 5): x / d - m in one v_fma_mix_f32 per plane */                  \
        "v_fma_mix_f32 v52, " D0 ", %[ix" I "], %[nx" I "] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"          /* lo.x */                              \
        "v_fma_mix_f32 v53, " D0 ", %[iy" I "], %[ny" I "] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"          /* lo.y */                              \
        "v_fma_mix_f32 v54, " D1 ", %[iz" I "], %[nz" I "] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"          /* lo.z */                              \
        "v_fma_mix_f32 " D1 ", " D1 ", %[ix" I "], %[nx" I "] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"       /* hi.x */                              \
        "v_fma_mix_f32 " D0 ", " D2 ", %[iy" I "], %[ny" I "] op_sel:[0,0,0] op_sel_hi:[1,0,0]\n"       /* hi.y */                              \
        "v_fma_mix_f32 " D2 ", " D2 ", %[iz" I "], %[nz" I "] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"       /* hi.z */                              \
        "v_med3_f32 v55, v52, " D1 ", %[tmin]\n v_med3_f32 v52, v52, " D1 ", %[tb" I "]\n v_min_f32_e32 v56, v53, " D0 "\n"                    \
        "v_max_f32_e32 " D0 ", v53, " D0 "\n v_min_f32_e32 " D1 ", v54, " D2 "\n v_max_f32_e32 v54, v54, " D2 "\n"                             \
        "v_max3_f32 v55, v55, v56, " D1 "\n"                                                                          /* start */           \
        "v_min3_f32 v52, v52, " D0 ", v54\n"                                                                          /* end */             \
        "v_cmp_nle_f32_e32 vcc, v52, v55\n"                                                         /* pass = !(end <= start) */            \
        "v_cmp_gt_i32_e64 %[m0], 0, " D3 "\n"                                                       /* leaf: sign bit of the link */        \
        "s_or_b64 %[m1], vcc, %[m0]\n"                                                                                                     \
        "v_add_u32_e32 v56, 16, %[i" I "]\n"                                                                                               \
        "v_cndmask_b32_e64 %[i" I "], " D3 ", v56, %[m1]\n"                                         /* next node, or the skip link */       \
        "s_and_b64 %[m1], vcc, %[m0]\n"                                                             /* a leaf whose coarse box passes: */   \
        "s_and_b64 exec, exec, %[m1]\n"                                                                                                    \
        "ds_write2_b32 %[top" I "], " D3 ", v55 offset1:1\n"                                        /*   (LEAF | leaf number, coarse start) */ \
        "v_add_u32_e32 %[top" I "], 0x200, %[top" I "]\n"

TRT_DEV void box_loop_compact2(uint32_t& iA, uint32_t& iB, const V3& nmA, const V3& nmB, const V3& invA, const V3& invB, float tbA, float tbB,
                               const uint4* __restrict__ nodes16, float2* stkA, float2* stkB, uint32_t slots, float2*& topA_out, float2*& topB_out,
                               uint32_t n, uint32_t few) {
    const uint32_t stkA_off = lds_offset(stkA), stkB_off = lds_offset(stkB);
    uint32_t topA = stkA_off, topB = stkB_off;
    const uint32_t limA = stkA_off + 512u * slots, limB = stkB_off + 512u * slots;
    unsigned long long saved, mA, mB, m0, m1;
    uint32_t cnt, cnt2;
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "1:\n"
        "v_cmp_gt_u32_e32 vcc, %[n], %[iA]\n"
        "v_cmp_ne_u32_e64 %[m0], %[topA], %[limA]\n"
        "s_and_b64 %[mA], vcc, %[m0]\n"
        "v_cmp_gt_u32_e32 vcc, %[n], %[iB]\n"
        "v_cmp_ne_u32_e64 %[m0], %[topB], %[limB]\n"
        "s_and_b64 %[mB], vcc, %[m0]\n"
        "s_bcnt1_i32_b64 %[cnt], %[mA]\n"
        "s_bcnt1_i32_b64 %[cnt2], %[mB]\n"
        "s_add_u32 %[cnt], %[cnt], %[cnt2]\n"
        "s_cmp_le_u32 %[cnt], %[few]\n"                       // at most `few` rays can still step (0: none): leave
        "s_cbranch_scc1 2f\n"
        "s_mov_b64 exec, %[mA]\n"
        "global_load_dwordx4 v[44:47], %[iA], %[nodes]\n"    // (cursors are byte offsets: box_loop_compact)
        "s_mov_b64 exec, %[mB]\n"
        "global_load_dwordx4 v[48:51], %[iB], %[nodes]\n"
        "s_mov_b64 exec, %[mA]\n"
        "s_waitcnt vmcnt(1)\n"
        "s_cbranch_execz 3f\n"
        TRT_COMPACT_STEP("v44", "v45", "v46", "v47", "A", "A")
        "3:\n"
        "s_mov_b64 exec, %[mB]\n"
        "s_waitcnt vmcnt(0)\n"
        "s_cbranch_execz 4f\n"
        TRT_COMPACT_STEP("v48", "v49", "v50", "v51", "B", "B")
        "4:\n"
        "s_mov_b64 exec, %[sv]\n"
        "s_branch 1b\n"
        "2:\n"
        : [iA] "+v"(iA), [iB] "+v"(iB), [topA] "+v"(topA), [topB] "+v"(topB), [sv] "=&s"(saved), [mA] "=&s"(mA), [mB] "=&s"(mB), [m0] "=&s"(m0),
          [m1] "=&s"(m1), [cnt] "=&s"(cnt), [cnt2] "=&s"(cnt2)
        : [n] "s"(n), [few] "s"(few), [nodes] "s"(nodes16), [limA] "v"(limA), [limB] "v"(limB), [nxA] "v"(nmA.x), [nyA] "v"(nmA.y), [nzA] "v"(nmA.z),
          [ixA] "v"(invA.x), [iyA] "v"(invA.y), [izA] "v"(invA.z), [tbA] "v"(tbA), [nxB] "v"(nmB.x), [nyB] "v"(nmB.y), [nzB] "v"(nmB.z), [ixB] "v"(invB.x),
          [iyB] "v"(invB.y), [izB] "v"(invB.z), [tbB] "v"(tbB), [tmin] "s"(kTMin)
        : "vcc", "scc", "memory", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56");
    topA_out = stkA + ((topA - stkA_off) >> 3);
    topB_out = stkB + ((topB - stkB_off) >> 3);
}

// Both walks of a lane, resumable like walk_compact: returns with doneA / doneB = "that slot's walk is complete" once at most `few` rays
// of the wave (both slots) are still walking; a slot that is not `act` is left alone.  `entered` = rays that came in.
template <int MODE>
TRT_DEV void walk_compact2(const SceneAcc<MODE>& sc, const uint4* __restrict__ nodes16, const float4* __restrict__ leaf_list, const Ray& rayA,
                           Trav& trA, bool actA, const Ray& rayB, Trav& trB, bool actB, Counters<false>& ctr, float2* stkA, float2* stkB,
                           uint32_t slots, uint32_t stragglers, uint32_t entered, bool& doneA, bool& doneB) {
    const uint32_t n = sc.L.n_cull_nodes << 4;                                         // cursors count bytes (box_loop_compact)
    // the rare walks (a ray whose slab arithmetic can produce NaN): reference tree, to the end, one slot after the other
    if (actA && trA.ref) { closest_hit_ref<MODE, false>(sc, rayA, trA, ctr); trA.i = n; }
    if (actB && trB.ref) { closest_hit_ref<MODE, false>(sc, rayB, trB, ctr); trB.i = n; }
    uint32_t iA = actA ? trA.i : n, iB = actB ? trB.i : n;
    const uint32_t few = stragglers < entered ? stragglers : entered - 1u;             // see walk_fast_lds
    for (;;) {
        float2 *topA, *topB;
        // -m = fl(-o * 1/d) per axis (box_loop_compact): recomputed per call from the rays, so that nothing but the rays stays live across the shades
        const V3 nmA = v3(-rayA.o.x * trA.inv.x, -rayA.o.y * trA.inv.y, -rayA.o.z * trA.inv.z), nmB = v3(-rayB.o.x * trB.inv.x, -rayB.o.y * trB.inv.y, -rayB.o.z * trB.inv.z);
        box_loop_compact2(iA, iB, nmA, nmB, trA.inv, trB.inv, trA.t_best, trB.t_best, nodes16, stkA, stkB, slots, topA, topB, n, few);
        TRT_CLK(ctr, 1);
        if (topA != stkA) leaf_phase<MODE, false>(stkA, topA, trA, ctr, [&](uint32_t leaf) { compact_leaf_test<MODE, false>(sc, leaf_list, rayA, trA, leaf, ctr); });
        if (topB != stkB) leaf_phase<MODE, false>(stkB, topB, trB, ctr, [&](uint32_t leaf) { compact_leaf_test<MODE, false>(sc, leaf_list, rayB, trB, leaf, ctr); });
        TRT_CLK(ctr, 2);
        const uint32_t still = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(iA < n)) + (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(iB < n));
        if (still <= few) break;
    }
    trA.i = iA; trB.i = iB;
    doneA = iA >= n; doneB = iB >= n;
}

constexpr uint32_t kLdsLeafSlotsMax = 16; // most slots per lane of the LDS leaf stack (8 bytes each)

// Which walk a kernel instantiation runs.  WALK_RUNTIME picks by the launch arguments (every knob combination; counting
// kernels); the others fix the walk at compile time, which is what the production launches use: the kernel then holds ONE
// walk instead of five, with the SGPRs, VGPRs and instruction-cache footprint of one (stream_pool_kernel on Cornell:
// 6849 lines of ISA, 44 spilled SGPRs and 16 spilled VGPRs with the runtime choice).
enum { WALK_RUNTIME = 0, WALK_LDS_STACK = 1, WALK_FLAT = 2, WALK_COMPACT = 3, WALK_REGS = 5 };

// Whole walk for one lane.  Returns the primitive reference (PRIM_NONE on a miss) and its t.  Postponed leaves go to
// `lds_stack` (this lane's slot 0 of a `leaf_slots`-deep LDS stack) if the kernel has one, else into registers:
// `leaf_slots` = 4 (also for 0 = default), 2 or 1 (tuning and tests; wave-uniform).
template <int MODE, bool STATS, int WALK = WALK_RUNTIME>
TRT_DEV uint32_t closest_hit(const SceneAcc<MODE>& sc, const Ray& ray, bool ref_tree, float& t_hit, Counters<STATS>& ctr,
                             uint32_t leaf_slots = 4u, float2* lds_stack = nullptr, const float4* __restrict__ leaf_list = nullptr,
                             const uint4* __restrict__ nodes16 = nullptr) {
    // (a general kernel reaches walk_compact's loop whenever it is handed the 16-byte nodes and an LDS stack: same domain as the specialised ones)
    const bool fused_loop = kAsmBoxLoop && !STATS && (WALK == WALK_COMPACT || (WALK == WALK_RUNTIME && lds_stack != nullptr && nodes16 != nullptr));
    Trav tr = trav_begin<MODE>(sc, ray, ref_tree, fused_loop);
    if (__builtin_expect(!tr.ref, 1)) {
        if constexpr (WALK == WALK_COMPACT) {
            walk_compact<MODE, STATS>(sc, nodes16, leaf_list, ray, tr, ctr, lds_stack, leaf_slots);
        } else if constexpr (WALK == WALK_FLAT) {
            walk_flat<MODE, STATS>(sc, leaf_list, ray, tr, ctr, lds_stack, leaf_slots);
        } else if constexpr (WALK == WALK_LDS_STACK) {
            walk_fast_lds<MODE, STATS>(sc, ray, tr, ctr, lds_stack, leaf_slots);
        } else if constexpr (WALK == WALK_REGS) {
            walk_fast<MODE, STATS, 4>(sc, ray, tr, ctr);
        } else {
            if (lds_stack != nullptr && nodes16 != nullptr) walk_compact<MODE, STATS>(sc, nodes16, leaf_list, ray, tr, ctr, lds_stack, leaf_slots);
            else if (lds_stack != nullptr && leaf_list != nullptr) walk_flat<MODE, STATS>(sc, leaf_list, ray, tr, ctr, lds_stack, leaf_slots);
            else if (lds_stack != nullptr) walk_fast_lds<MODE, STATS>(sc, ray, tr, ctr, lds_stack, leaf_slots);
            else if (leaf_slots >= 4u || leaf_slots == 0u) walk_fast<MODE, STATS, 4>(sc, ray, tr, ctr);
            else if (leaf_slots >= 2u) walk_fast<MODE, STATS, 2>(sc, ray, tr, ctr);
            else walk_fast<MODE, STATS, 1>(sc, ray, tr, ctr);
        }
    } else {
        closest_hit_ref<MODE, STATS>(sc, ray, tr, ctr);
    }
    t_hit = tr.t_best;
    return tr.prim_best;
}

// The per-lane tree walks in resumable form (WALK_COMPACT: 16-byte nodes from global memory; WALK_LDS_STACK: the culling tree in LDS): `tr`
// is set up by the caller (trav_begin) when the ray starts its walk and kept while the function returns false.  Returns true when the
// walk is complete (tr.t_best / tr.prim_best).
template <int MODE, bool STATS, int WALK>
TRT_DEV bool closest_hit_resume(const SceneAcc<MODE>& sc, const Ray& ray, Trav& tr, Counters<STATS>& ctr, uint32_t leaf_slots, float2* lds_stack,
                                const float4* __restrict__ leaf_list, const uint4* __restrict__ nodes16, uint32_t stragglers, uint32_t entered) {
    static_assert(WALK == WALK_COMPACT || WALK == WALK_LDS_STACK, "only the per-lane tree walks with an LDS leaf stack can be left and resumed");
    if (__builtin_expect(!tr.ref, 1)) {
        if constexpr (WALK == WALK_COMPACT) return walk_compact<MODE, STATS>(sc, nodes16, leaf_list, ray, tr, ctr, lds_stack, leaf_slots, stragglers, entered);
        else return walk_fast_lds<MODE, STATS>(sc, ray, tr, ctr, lds_stack, leaf_slots, stragglers, entered);
    }
    closest_hit_ref<MODE, STATS>(sc, ray, tr, ctr);
    return true;
}

// Material index of a primitive reference.
template <int MODE>
TRT_DEV uint32_t prim_material(const SceneAcc<MODE>& sc, uint32_t prim) {
    const uint32_t idx = prim & PRIM_INDEX_MASK;
    return (prim & PRIM_QUAD_BIT) ? __float_as_uint(sc.quad(1, idx).w) : sc.sphere_material(idx);
}

// ------------------------------------------------------------------------------------------------
// One pass of the loop body of CpuSampler::single_point_sampling after the hit query (cpu.rs:48-62):
// emission, scatter, attenuation.  Returns true when the path ended (light, miss, budget spent).
// ------------------------------------------------------------------------------------------------
struct Path {
    Ray ray;
    V3 color, atten;
    uint32_t remain;
    Rng rng;
};

// LAZY (SceneLayout::lazy_color): the colour is known to be +0 until the path ends - `color += attenuation * emission`
// (cpu.rs:49-50) adds +-0 at every hit that is not a light, and the attenuation cannot overflow - so it is written once,
// as 0 + attenuation * (light colour | background), and need not be carried from bounce to bounce.
template <int MODE, bool STATS, bool LAZY = false>
TRT_DEV bool shade_hit(const SceneAcc<MODE>& sc, Path& p, uint32_t prim, float t, V3 background, Counters<STATS>& ctr) {
    if (prim == PRIM_NONE) {                                           // cpu.rs:58-61
        p.color = (LAZY ? v3(0.0f, 0.0f, 0.0f) : p.color) + p.atten * background;
        return true;
    }
    if constexpr (STATS) ctr.shade++;
    // HitRecord::new (hittable/mod.rs:28-48), built once for the winning primitive
    const uint32_t idx = prim & PRIM_INDEX_MASK;
    V3 point = ray_at(p.ray, t);
    V3 normal;
    bool front_face;
    uint32_t mat;
    if (prim & PRIM_QUAD_BIT) {
        float4 q0 = sc.quad(0, idx), q1 = sc.quad(1, idx), q4 = sc.quad(4, idx);
        front_face = dot(p.ray.d, v3(q0.x, q0.y, q0.z)) < 0.0f;        // outward normal = n, un-normalised (quad.rs:45)
        V3 nu = v3(q4.y, q4.z, q4.w);                                  // n.normalized(), precomputed on the host
        normal = front_face ? nu : -nu;
        mat = __float_as_uint(q1.w);
    } else {
        float4 sp = sc.sphere(idx);
        V3 outward = point - v3(sp.x, sp.y, sp.z);                     // sphere.rs:47-51 (p = ray.at(t))
        front_face = dot(p.ray.d, outward) < 0.0f;
        V3 nu = normalized(outward);
        normal = front_face ? nu : -nu;
        mat = sc.sphere_material(idx);
    }
    const float4 m = sc.material(mat);
    const uint32_t kind = sc.material_kind(mat);
    const V3 albedo = v3(m.x, m.y, m.z);
    if constexpr (STATS) {
        ctr.shade_lambertian += kind == TRT_LAMBERTIAN; ctr.shade_metal += kind == TRT_METAL;
        ctr.shade_dielectric += kind == TRT_DIELECTRIC; ctr.shade_light += kind == TRT_LIGHT;
    }
    // cpu.rs:49-50: emitted() is the light's colour, None -> 0 for everything else (material/mod.rs:8-10)
    if constexpr (LAZY) {
        p.color = v3(0.0f, 0.0f, 0.0f);                                // what the sum is after any hit that is not a light
        if (kind == TRT_LIGHT) p.color = v3(0.0f, 0.0f, 0.0f) + p.atten * albedo;
    } else {
        V3 emission = (kind == TRT_LIGHT) ? albedo : v3(0.0f, 0.0f, 0.0f);
        p.color = p.color + p.atten * emission;
    }
    V3 dir;
    if (kind == TRT_LAMBERTIAN || kind == TRT_METAL) {
        // Both draw ONE point in the unit sphere (three random numbers, acos, cbrt, two sincos: the expensive part of a shade) and use
        // it differently: evaluated once for the lanes of either kind - as two branches a wave that holds both kinds (nearly every wave
        // of the random-spheres scene) ran that code twice, the second time for a handful of metal lanes.  Same draws, same operations.
        const V3 in_sphere = random_in_unit_sphere(p.rng);
        if (kind == TRT_LAMBERTIAN) {                                  // lambertian.rs:16-22: normal + random_unit_vector (vec3extend.rs:32-34)
            dir = normal + normalized(in_sphere);
            if (near_zero(dir)) dir = normal;
        } else {                                                       // metal.rs:18-25 (fuzz clamped at creation)
            V3 reflected = reflect(p.ray.d, normal);
            dir = reflected + m.w * in_sphere;
        }
    } else if (kind == TRT_DIELECTRIC) {                               // dielectric.rs:26-46
        float ri = front_face ? 1.0f / m.w : m.w;
        float cosv = __builtin_fminf(-dot(normal, p.ray.d), 1.0f);
        float sinv = __builtin_sqrtf(1.0f - cosv * cosv);
        bool total_reflection = ri * sinv > 1.0f;
        float sqrt_r0 = (1.0f - ri) / (1.0f + ri);                     // reflectance(), dielectric.rs:16-22
        float r0 = sqrt_r0 * sqrt_r0;
        float x = 1.0f - cosv;
        float x2 = x * x;
        float reflectance = r0 + (1.0f - r0) * (x * (x2 * x2));        // powi(5): x * ((x*x)*(x*x))
        bool do_reflect = total_reflection;
        if (!do_reflect) do_reflect = reflectance > rng_random(p.rng); // `||` short-circuit: no draw on TIR
        dir = do_reflect ? reflect(p.ray.d, normal) : refract(p.ray.d, normal, ri);
    } else {                                                           // Light::scatter -> None (light.rs:17-19)
        return true;
    }
    p.atten = p.atten * albedo;                                        // cpu.rs:52
    p.ray = ray_new(point, dir);                                       // Ray::new normalises (ray.rs:12-14)
    p.remain -= 1u;                                                    // cpu.rs:54
    return p.remain == 0u;
}

// SamplePointGenerator::generate body (pointgen.rs:41-43) + Camera::get_ray (camera.rs:58-66)
TRT_DEV Ray primary_ray(const CameraDev& cam, uint32_t x, uint32_t y, Rng& rng) {
    float u = ((float)x + rng_random(rng)) / (float)(cam.width - 1u);
    float v = ((float)y + rng_random(rng)) / (float)(cam.height - 1u);
    float px, py;
    random_in_unit_disk(rng, px, py);
    V3 pos = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    V3 du = v3(cam.du[0], cam.du[1], cam.du[2]), dv = v3(cam.dv[0], cam.dv[1], cam.dv[2]);
    V3 origin = (pos + px * du) + py * dv;
    V3 ul = v3(cam.upper_left[0], cam.upper_left[1], cam.upper_left[2]);
    V3 hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    V3 ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    V3 target = (ul + u * hor) - v * ver;
    return ray_new(origin, target - origin);
}

// Start sample `s` of image pixel (x, y): fresh RNG stream, primary ray, colour 0, attenuation 1 (cpu.rs:42-45).
TRT_DEV void path_begin(Path& p, const CameraDev& cam, const RenderArgs& ra, uint32_t x, uint32_t y, uint32_t s) {
    p.rng = rng_seed(ra.seed_key, y * cam.width + x, s);
    p.ray = primary_ray(cam, x, y, p.rng);
    p.color = v3(0.0f, 0.0f, 0.0f);
    p.atten = v3(1.0f, 1.0f, 1.0f);
    p.remain = ra.max_bounces;
}

// A radiance record of the streamed backend: 12 bytes at a 4-byte-aligned address, moved as ONE vector-memory instruction
// (global_store_dwordx3 / global_load_dwordx3: assigning the packed struct is what makes the compiler emit it - three float stores
// through a float* stay three instructions, which is what rounds 1-3 shipped: 3 VMEM writes per finished path and partial-sector
// writes that the L2 had to merge, WRITE_SIZE 1.6-4.3x the records' bytes in profiles/r03_*_pmc.json).
struct __attribute__((packed, aligned(4))) Radiance { float r, g, b; };
TRT_DEV void radiance_store(float* __restrict__ colors, uint32_t slot, const V3& c) {
    *reinterpret_cast<Radiance*>(colors + 3ull * slot) = Radiance{c.x, c.y, c.z};
}

// Workgroup -> tile index.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, each XCD
// has its own L2).  With xcd_aware, XCD k renders the k-th contiguous eighth of the tiles, which keeps the part of
// a large scene that an image region touches in one L2.  Placement only: any mapping gives the same frame.
// Measured on MI355X (100 k spheres, 3840x2160): 2x SLOWER, because image regions differ in cost and whole XCDs
// go idle; the round-robin default interleaves cheap and expensive tiles on every XCD.  Kept behind TRT_XCD_REMAP.
TRT_DEV uint32_t xcd_tile(uint32_t block, uint32_t n_blocks, uint32_t xcd_aware) {
    if (!xcd_aware) return block;
    const uint32_t per = n_blocks >> 3, main = per << 3;     // the first `main` blocks split evenly; the tail maps to itself
    return block < main ? (block & 7u) * per + (block >> 3) : block;
}

// local row -> image row (tinyrt.h trt_render_params)
TRT_DEV uint32_t image_row(const RenderArgs& ra, uint32_t row) {
    if (ra.band_rows == 0u) return row;
    return ((row / ra.band_rows) * ra.band_stride + ra.band_offset) * ra.band_rows + row % ra.band_rows;
}

TRT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// All 64 lanes must call this (inactive pixels pass zeros).
template <bool STATS>
TRT_DEV void flush_counters(unsigned long long* counters, uint32_t samples, uint32_t rays, const Counters<STATS>& ctr) {
    if (counters == nullptr) return;
    const bool lane0 = (threadIdx.x & 63u) == 0u;
    uint32_t s = wave_sum(samples), r = wave_sum(rays);
    if (lane0) {
        if (s) atomicAdd(&counters[CTR_SAMPLES], (unsigned long long)s);
        if (r) atomicAdd(&counters[CTR_RAYS], (unsigned long long)r);
    }
#ifdef TRT_PHASE_CLOCK
    if (lane0) { for (int k = 0; k < 4; k++) atomicAdd(&counters[12 + k], (unsigned long long)ctr.clk.t[k]); }
#endif
    if constexpr (STATS) {
#ifdef TRT_PHASE_CLOCK
        constexpr int kN = 10;                                     // counters[12..15] carry the phase clock in that build
#else
        constexpr int kN = 14;
#endif
        uint32_t v[14] = {ctr.node, ctr.sphere, ctr.quad_plane, ctr.quad_inside, ctr.shade, ctr.w_rounds, ctr.w_steps, ctr.w_leaf, ctr.w_gen, ctr.pend,
                          ctr.shade_lambertian, ctr.shade_metal, ctr.shade_dielectric, ctr.shade_light};
        const int slot[14] = {CTR_NODE, CTR_SPHERE, CTR_QUAD_PLANE, CTR_QUAD_INSIDE, CTR_SHADE, CTR_W_ROUNDS, CTR_W_STEPS, CTR_W_LEAF, CTR_W_GEN, CTR_PEND,
                              CTR_SHADE_KIND + 0, CTR_SHADE_KIND + 1, CTR_SHADE_KIND + 2, CTR_SHADE_KIND + 3};
#pragma unroll
        for (int k = 0; k < kN; k++) {
            uint32_t t = wave_sum(v[k]);
            if (lane0 && t) atomicAdd(&counters[slot[k]], (unsigned long long)t);
        }
    }
}

}  // namespace trt
