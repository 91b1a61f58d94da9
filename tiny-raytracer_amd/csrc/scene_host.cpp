// scene_host.cpp — host side of the path: World -> reference-order BVH -> packed pre-order scene,
// Camera::new, and the Imager's gamma/quantise step.  Plain C++ (no device code); compiled with
// -ffp-contract=off so the f32 values it precomputes (boxes, quad planes, camera basis) are the
// ones the reference's constructors produce.
#include "scene.h"
#include "trt_pow.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <exception>
#include <system_error>
#include <thread>

namespace trt {
namespace {

struct H3 { float x, y, z; };
inline H3 h3(trt_vec3 v) { return H3{v.x, v.y, v.z}; }
inline H3 operator+(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline H3 operator-(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 operator*(H3 a, float s) { return H3{a.x * s, a.y * s, a.z * s}; }
inline H3 operator/(H3 a, float s) { return H3{a.x / s, a.y / s, a.z / s}; }
inline float hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline H3 hnormalized(H3 a) { return a / sqrtf(hdot(a, a)); }                   // vec3.rs:45-47
inline H3 hmin(H3 a, H3 b) { return H3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }   // vec3extend.rs:59-65
inline H3 hmax(H3 a, H3 b) { return H3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }   // vec3extend.rs:67-73

struct Box { H3 lo, hi; };

// AABB::new: the two corners ordered per axis, then grown by PADDING_AMOUNT/2 (aabb.rs:13-20)
Box box_from_corners(H3 a, H3 b) {
    const float pad = 0.0001f / 2.0f;
    H3 p{pad, pad, pad};
    return Box{hmin(a, b) - p, hmax(a, b) + p};
}
Box box_union(Box a, Box b) { return Box{hmin(a.lo, b.lo), hmax(a.hi, b.hi)}; }                // aabb.rs:30-34

// AABB::longest_axis (aabb.rs:63-78): ties go to the later axis.
int box_longest_axis(const Box& b) {
    float sx = b.hi.x - b.lo.x, sy = b.hi.y - b.lo.y, sz = b.hi.z - b.lo.z;
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}
inline float axis_of(H3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// f32::total_cmp as an integer key (aabb.rs:80-82)
inline int32_t total_order_key(float f) {
    int32_t bits;
    memcpy(&bits, &f, 4);
    return bits ^ (int32_t)((uint32_t)(bits >> 31) >> 1);
}

inline float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Node::new (bvh.rs:42-84) over `objs`, nodes in pre-order.  A subtree over n primitives has exactly 2n - 1 nodes (one primitive per leaf), so
// every node's index is known before its subtree is built: node `me` over n objects has its left child at me + 1 (mid = n / 2 objects, 2 mid - 1
// nodes) and its right child at me + 2 mid.  The arrays are sized up front and the two halves of the top levels are built by different threads
// (they write disjoint index ranges).  The reference's `sort_by(total_cmp of bbox.min[axis])` is a STABLE sort of the objects in the order the
// parent handed them over; here: one u64 per object, (order-preserving key << 32) | position in the incoming order, and a plain sort of those -
// equal keys stay in incoming order, which is what a stable sort does, without an indirect comparison per step (round 5: 4 M spheres in
// seconds instead of most of a minute).
// Runs `left_half` on a new thread and `right_half` on this one.  No exception may leave a thread function (std::terminate) or this library (C ABI):
// what the other thread throws - std::bad_alloc on a scene of millions - is carried over and rethrown here, after both halves have ended; if the
// system has no thread to give, this thread does both halves.
template <typename L, typename R>
void run_halves(L&& left_half, R&& right_half) {
    std::exception_ptr left_error;
    std::thread t;
    bool spawned = false;
    try {
        t = std::thread([&] { try { left_half(); } catch (...) { left_error = std::current_exception(); } });
        spawned = true;
    } catch (const std::system_error&) {
    }
    try {
        if (!spawned) left_half();
        right_half();
    } catch (...) {
        if (spawned) t.join();
        throw;
    }
    if (spawned) t.join();
    if (left_error) std::rethrow_exception(left_error);
}

struct Builder {
    const std::vector<Box>& prim_box;
    std::vector<Box> node_box;
    std::vector<int32_t> node_prim;
    std::vector<int32_t> node_skip;
    std::atomic<uint32_t> max_depth{0};

    explicit Builder(const std::vector<Box>& pb) : prim_box(pb) {
        const size_t nn = 2 * pb.size() - 1;
        node_box.resize(nn);
        node_prim.assign(nn, -1);
        node_skip.assign(nn, 0);
    }

    void note_depth(uint32_t depth) {
        uint32_t seen = max_depth.load(std::memory_order_relaxed);
        while (depth > seen && !max_depth.compare_exchange_weak(seen, depth, std::memory_order_relaxed)) {}
    }

    // `objs`: this node's objects in the order the parent hands them over; reordered in place (the reference sorts a copy that only
    // its own subtree sees: same thing).  `threads`: how many threads this subtree may still use.
    void build(uint32_t me, uint32_t* objs, size_t n, uint32_t depth, unsigned threads) {
        if (n == 1) {
            note_depth(depth);
            node_box[me] = prim_box[objs[0]];
            node_prim[me] = (int32_t)objs[0];
        } else if (n == 2) {                     // children keep the given order: no sort (bvh.rs:58-67)
            note_depth(depth + 1);
            node_box[me + 1] = prim_box[objs[0]]; node_prim[me + 1] = (int32_t)objs[0]; node_skip[me + 1] = (int32_t)me + 2;
            node_box[me + 2] = prim_box[objs[1]]; node_prim[me + 2] = (int32_t)objs[1]; node_skip[me + 2] = (int32_t)me + 3;
            node_box[me] = box_union(node_box[me + 1], node_box[me + 2]);
        } else {
            Box all = prim_box[objs[0]];
            for (size_t i = 1; i < n; i++) all = box_union(all, prim_box[objs[i]]);
            const int axis = box_longest_axis(all);
            {
                constexpr size_t kSmall = 64;                               // most nodes are small: no heap traffic for them
                uint64_t keyed_small[kSmall];
                uint32_t in_small[kSmall];
                std::vector<uint64_t> keyed_big;
                std::vector<uint32_t> in_big;
                uint64_t* keyed = keyed_small;
                uint32_t* in = in_small;
                if (n > kSmall) { keyed_big.resize(n); in_big.resize(n); keyed = keyed_big.data(); in = in_big.data(); }
                for (size_t i = 0; i < n; i++) {
                    const uint32_t k = (uint32_t)total_order_key(axis_of(prim_box[objs[i]].lo, axis)) ^ 0x80000000u;    // signed order -> unsigned order
                    keyed[i] = (uint64_t)k << 32 | (uint64_t)i;
                    in[i] = objs[i];
                }
                std::sort(keyed, keyed + n);
                for (size_t i = 0; i < n; i++) objs[i] = in[(uint32_t)keyed[i]];
            }
            const size_t mid = n / 2;
            const uint32_t l = me + 1, r = me + 2 * (uint32_t)mid;
            if (threads > 1 && n >= 32768) {
                const unsigned tl = threads / 2, tr = threads - tl;
                run_halves([&] { build(l, objs, mid, depth + 1, tl); }, [&] { build(r, objs + mid, n - mid, depth + 1, tr); });
            } else {
                build(l, objs, mid, depth + 1, 1);
                build(r, objs + mid, n - mid, depth + 1, 1);
            }
            node_box[me] = box_union(node_box[l], node_box[r]);
        }
        node_skip[me] = (int32_t)(me + 2 * (uint32_t)n - 1);
    }
};

inline unsigned host_threads() {
    const unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1u : (hc > 16u ? 16u : hc);
}

inline bool tame(float v) { return fabsf(v) < 1e30f; }
inline bool tame3(H3 v) { return tame(v.x) && tame(v.y) && tame(v.z); }

inline double surface_area(const Box& b) {
    double dx = (double)b.hi.x - b.lo.x, dy = (double)b.hi.y - b.lo.y, dz = (double)b.hi.z - b.lo.z;
    if (!(dx > 0)) dx = 0;
    if (!(dy > 0)) dy = 0;
    if (!(dz > 0)) dz = 0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}

// Culling tree: a second hierarchy over the reference tree's LEAF SEQUENCE (scene.h explains why any such
// hierarchy gives bit-identical hits).  Contiguous leaf ranges are split where the surface-area heuristic
// SA(left)*n_left + SA(right)*n_right is smallest; an inner node whose box is at least `kPrune` of its nearest
// emitted ancestor's is not emitted at all (its children hang directly off that ancestor), since a ray that
// passed the ancestor almost surely passes it too.  Boxes are exact f32 unions of leaf boxes.
// f32 -> f16 bits rounded toward -inf (up = false) or +inf (up = true): the nearest-even conversion, stepped by one
// f16 if it landed on the wrong side.  |x| beyond the f16 range becomes +-inf or +-65504, whichever is conservative.
float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t out;
    if (e == 0u) {
        if (m == 0u) out = sign;
        else {                                              // subnormal: m * 2^-24
            float v = (float)m * 5.9604644775390625e-08f;
            uint32_t b; memcpy(&b, &v, 4); out = b | sign;
        }
    } else if (e == 31u) out = sign | 0x7F800000u | (m << 13);
    else out = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &out, 4); return f;
}
uint16_t f32_to_f16_nearest(float x) {
    uint32_t b; memcpy(&b, &x, 4);
    const uint32_t sign = (b >> 16) & 0x8000u;
    const uint32_t a = b & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (a > 0x7F800000u ? 0x200u : 0u));
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            // >= 65520 rounds to inf
    if (a < 0x33000001u) return (uint16_t)sign;                           // <= 2^-25 rounds to zero
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
    uint32_t shift = e < -14 ? (uint32_t)(13 + (-14 - e)) : 13u;          // subnormal results lose more bits
    uint32_t half = m >> shift, rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1u);
    if (rem > mid || (rem == mid && (half & 1u))) half++;
    uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t out = e < -14 ? half : ((he << 10) + (half - 0x400u));       // a mantissa carry rolls into the exponent
    return (uint16_t)(sign | out);
}
uint16_t f32_to_f16_dir(float x, bool up) {
    if (x != x) return up ? 0x7C00u : 0xFC00u;                            // NaN: the conservative infinity
    uint16_t h = f32_to_f16_nearest(x);
    const float back = f16_bits_to_f32(h);
    if (up ? back >= x : back <= x) return h;
    // step one f16 toward the wanted side
    if (up) {
        if (h & 0x8000u) return (h & 0x7FFFu) == 0u ? (uint16_t)0x0001u : (uint16_t)(h - 1u);   // negative: smaller magnitude
        return (uint16_t)(h + 1u);                                                           // positive: larger magnitude (0x7BFF -> inf)
    }
    if (h & 0x8000u) return (uint16_t)(h + 1u);
    return (h & 0x7FFFu) == 0u ? (uint16_t)0x8001u : (uint16_t)(h - 1u);
}

struct CullBuilder {
    double kPrune;                           // trt_scene_options.cull_prune: 0.5 measured best on MI355X (round 3, profiles/r03_defaults_sweep.txt: random-spheres
                                             // 0.4-0.5 beat 0.7 by 3 %, the 100 k-sphere scene is flat from 0.3 to 0.7); any value gives the same hits
    const std::vector<Box>& leaf_box;        // leaf k of the reference tree, in left-first order
    // the finished tree, pre-order; a skip link is the index of the node to go to when the subtree is done
    std::vector<Box> node_box;
    std::vector<int32_t> node_leaf;          // leaf sequence number or -1
    std::vector<int32_t> node_skip;

    CullBuilder(const std::vector<Box>& lb, double prune) : kPrune(prune), leaf_box(lb) {}

    // A subtree in its own arrays, indices relative to its first node (so that subtrees built by different threads can be concatenated)
    struct Part {
        std::vector<Box> box;
        std::vector<int32_t> leaf, skip;
        void append(const Part& p) {
            const int32_t base = (int32_t)box.size();
            box.insert(box.end(), p.box.begin(), p.box.end());
            leaf.insert(leaf.end(), p.leaf.begin(), p.leaf.end());
            skip.reserve(skip.size() + p.skip.size());
            for (int32_t s : p.skip) skip.push_back(s + base);
        }
    };
    struct Frame { uint32_t a, b; double parent_sa; int32_t node; bool close; };
    struct Split { Box all; double sa; bool emit; uint32_t k; };

    // One node: the box of leaves [a, b), whether it is emitted under an emitted ancestor of area `parent_sa`, and its best split
    // by SAH over the fixed order (prefix boxes on the fly, suffix boxes precomputed).  m = b - a >= 2 for the split.
    Split split(uint32_t a, uint32_t b, double parent_sa, std::vector<Box>& suffix) const {
        Split r;
        const uint32_t m = b - a;
        r.all = leaf_box[a];
        for (uint32_t k = a + 1; k < b; k++) r.all = box_union(r.all, leaf_box[k]);
        r.sa = surface_area(r.all);
        r.emit = m == 1 || parent_sa < 0.0 || r.sa < kPrune * parent_sa;
        r.k = 1;
        if (m == 1) return r;
        suffix.resize(m);
        suffix[m - 1] = leaf_box[b - 1];
        for (uint32_t k = m - 1; k-- > 0;) suffix[k] = box_union(leaf_box[a + k], suffix[k + 1]);
        Box prefix = leaf_box[a];
        double best = 0.0;
        for (uint32_t k = 1; k < m; k++) {           // left = [a, a+k), right = [a+k, b)
            const double c = surface_area(prefix) * k + surface_area(suffix[k]) * (m - k);
            if (k == 1 || c < best) { best = c; r.k = k; }
            prefix = box_union(prefix, leaf_box[a + k]);
        }
        return r;
    }

    // leaves [a0, b0) into `out`, one thread
    void build_serial(uint32_t a0, uint32_t b0, double parent_sa0, Part& out) const {
        std::vector<Frame> stack;
        std::vector<Box> suffix;
        stack.push_back(Frame{a0, b0, parent_sa0, -1, false});
        while (!stack.empty()) {
            Frame f = stack.back();
            stack.pop_back();
            if (f.close) { out.skip[(size_t)f.node] = (int32_t)out.box.size(); continue; }
            const uint32_t m = f.b - f.a;
            const Split sp = split(f.a, f.b, f.parent_sa, suffix);
            int32_t me = -1;
            if (sp.emit) {
                me = (int32_t)out.box.size();
                out.box.push_back(sp.all);
                out.leaf.push_back(m == 1 ? (int32_t)f.a : -1);
                out.skip.push_back(0);
            }
            if (m == 1) { out.skip[(size_t)me] = me + 1; continue; }
            const double child_parent_sa = sp.emit ? sp.sa : f.parent_sa;
            if (sp.emit) stack.push_back(Frame{0, 0, 0.0, me, true});                         // runs after both subtrees
            stack.push_back(Frame{f.a + sp.k, f.b, child_parent_sa, -1, false});              // right: popped second
            stack.push_back(Frame{f.a, f.a + sp.k, child_parent_sa, -1, false});              // left: popped first
        }
    }

    // the same tree with the two halves of the large nodes built by different threads and concatenated (pre-order: own node, left, right)
    void build_parallel(uint32_t a, uint32_t b, double parent_sa, Part& out, unsigned threads) const {
        const uint32_t m = b - a;
        if (threads <= 1 || m < 65536u) { build_serial(a, b, parent_sa, out); return; }
        std::vector<Box> suffix;
        const Split sp = split(a, b, parent_sa, suffix);
        std::vector<Box>().swap(suffix);
        int32_t me = -1;
        if (sp.emit) {
            me = (int32_t)out.box.size();
            out.box.push_back(sp.all);
            out.leaf.push_back(-1);
            out.skip.push_back(0);
        }
        const double child_parent_sa = sp.emit ? sp.sa : parent_sa;
        Part left, right;
        // threads in proportion to the halves' sizes (SAH splits are uneven)
        unsigned tl = (unsigned)(((uint64_t)threads * sp.k + m / 2) / m);
        if (tl < 1) tl = 1;
        if (tl >= threads) tl = threads - 1;
        run_halves([&] { build_parallel(a, a + sp.k, child_parent_sa, left, tl); }, [&] { build_parallel(a + sp.k, b, child_parent_sa, right, threads - tl); });
        out.append(left);
        out.append(right);
        if (sp.emit) out.skip[(size_t)me] = (int32_t)out.box.size();
    }

    void build(unsigned threads) {
        Part all;
        build_parallel(0, (uint32_t)leaf_box.size(), -1.0, all, threads);
        node_box.swap(all.box);
        node_leaf.swap(all.leaf);
        node_skip.swap(all.skip);
    }
};


void dump_tree(NodeDump& d, const std::vector<Box>& boxes, const std::vector<int32_t>& prim_geo, const std::vector<int32_t>& skip) {
    const size_t n = boxes.size();
    d.bbox6.resize(6 * n);
    d.prim_geo = prim_geo;
    d.skip = skip;
    for (size_t i = 0; i < n; i++) {
        const Box& bx = boxes[i];
        float* o = &d.bbox6[6 * i];
        o[0] = bx.lo.x; o[1] = bx.lo.y; o[2] = bx.lo.z; o[3] = bx.hi.x; o[4] = bx.hi.y; o[5] = bx.hi.z;
    }
}

}  // namespace

trt_scene_options scene_options_builtin() {
    trt_scene_options o{};
    o.cull_prune = 0.5f;
    o.flat_walk = -1;
    o.compact_nodes = -1;
    o.top_nodes = 0;                               // ignored since ABI 4
    o.scratch_cap_bytes = (uint64_t)32 << 30;      // one full-size streamed workspace is up to 16 GB: the cap must hold it, or every render re-allocates
    return o;
}

bool compile_scene(const World& w, const trt_scene_options& opt, SceneHost& out, std::string& msg) {
    const size_t ng = w.geometries.size();
    if (ng == 0) { msg = "world has no geometry"; return false; }
    if (ng > PRIM_INDEX_MASK) { msg = "too many geometries"; return false; }

    // per-primitive constructor work: Sphere::new (sphere.rs:16-26), Quad::new (quad.rs:20-29)
    std::vector<Box> prim_box(ng);
    std::vector<uint32_t> local_index(ng);
    std::vector<F4> spheres, q0, q1, q2, q3, q4;
    std::vector<uint32_t> sphere_mat;
    bool all_finite = true;
    for (size_t g = 0; g < ng; g++) {
        const Geometry& geo = w.geometries[g];
        if (geo.material >= w.materials.size()) { msg = "geometry refers to a material index that does not exist"; return false; }
        if (geo.kind == 0) {
            H3 c = h3(geo.a);
            float r = geo.b.x;
            H3 rv{r, r, r};
            prim_box[g] = box_from_corners(c - rv, c + rv);
            local_index[g] = (uint32_t)spheres.size();
            spheres.push_back(F4{c.x, c.y, c.z, r});
            sphere_mat.push_back(geo.material);
            all_finite = all_finite && tame3(c) && tame(r);
        } else {
            H3 corner = h3(geo.a), u = h3(geo.b), v = h3(geo.c);
            prim_box[g] = box_union(box_from_corners(corner, (corner + u) + v), box_from_corners(corner + u, corner + v));
            H3 n = hcross(u, v);
            H3 wv = n / hdot(n, n);
            float d = hdot(n, corner);
            H3 nu = hnormalized(n);              // HitRecord::new normalises per hit (mod.rs:35-40): same value, once
            local_index[g] = (uint32_t)q0.size();
            q0.push_back(F4{n.x, n.y, n.z, d});
            q1.push_back(F4{corner.x, corner.y, corner.z, bitsf(geo.material)});
            q2.push_back(F4{v.x, v.y, v.z, wv.x});
            q3.push_back(F4{wv.y, wv.z, u.x, u.y});
            q4.push_back(F4{u.z, nu.x, nu.y, nu.z});
            all_finite = all_finite && tame3(corner) && tame3(u) && tame3(v) && tame3(n) && tame(d);
        }
        all_finite = all_finite && tame3(prim_box[g].lo) && tame3(prim_box[g].hi);
    }

    // BVH::new (bvh.rs:12-22): objects in insertion order
    std::vector<uint32_t> order(ng);
    for (size_t g = 0; g < ng; g++) order[g] = (uint32_t)g;
    Builder b(prim_box);
    b.build(0, order.data(), ng, 1, host_threads());
    const uint32_t nn = (uint32_t)b.node_box.size();

    // culling tree over the reference tree's leaf sequence
    std::vector<Box> leaf_box;
    std::vector<int32_t> leaf_geo;
    leaf_box.reserve(ng);
    leaf_geo.reserve(ng);
    for (uint32_t i = 0; i < nn; i++) {
        if (b.node_prim[i] >= 0) { leaf_box.push_back(b.node_box[i]); leaf_geo.push_back(b.node_prim[i]); }
    }
    CullBuilder cb(leaf_box, opt.cull_prune > 0.0f ? (double)opt.cull_prune : 0.5);
    cb.build(host_threads());
    const uint32_t nc = (uint32_t)cb.node_box.size();
    std::vector<int32_t> cull_prim_geo(nc);
    for (uint32_t i = 0; i < nc; i++) cull_prim_geo[i] = cb.node_leaf[i] >= 0 ? leaf_geo[(size_t)cb.node_leaf[i]] : -1;

    const uint32_t ns = (uint32_t)spheres.size(), nq = (uint32_t)q0.size(), nm = (uint32_t)w.materials.size();
    SceneLayout& L = out.layout;
    L.n_nodes = nn; L.n_cull_nodes = nc; L.n_spheres = ns; L.n_quads = nq; L.n_materials = nm;
    L.off_sphere = 2 * nc;
    L.off_quad = L.off_sphere + ns;
    L.off_material = L.off_quad + 5 * nq;
    uint32_t n_f4 = L.off_material + nm;
    L.off_sphere_mat = n_f4 * 4;
    L.off_material_kind = L.off_sphere_mat + ns;
    uint32_t n_u32 = L.off_material_kind + nm;
    L.hot_bytes = ((n_u32 * 4u) + 15u) & ~15u;
    L.off_ref_nodes = L.hot_bytes / 16u;
    L.n_leaves = (uint32_t)leaf_box.size();
    L.off_leaf_list = L.off_ref_nodes + 2u * nn;
    L.blob_bytes = L.hot_bytes + 32u * nn + 32u * (L.n_leaves + kLeafListPad);     // the leaf list is followed by kLeafListPad copies of its last entry
    L.off_compact = 0u;
    bool want_compact = L.hot_bytes > kLdsSceneMaxBytes;                      // scenes walked from global memory
    if (opt.compact_nodes >= 0) want_compact = opt.compact_nodes != 0;                        // tuning / tests; same frames either way
    if (nc >= (1u << 27)) want_compact = false;                                               // (its links are byte offsets below 2^31)
    if (want_compact) {
        L.off_compact = L.blob_bytes / 16u;
        L.blob_bytes += 16u * nc;
    }
    {   // every offset and size of the layout is 32 bits wide: refuse scenes that do not fit instead of wrapping around
        const uint64_t prims = (uint64_t)ns + 5ull * nq + nm;
        const uint64_t total = 16ull * (2ull * nc + prims) + 4ull * ((uint64_t)ns + nm) + 16ull                 // hot part
                               + 32ull * nn + 32ull * ((uint64_t)L.n_leaves + kLeafListPad)                          // reference tree, leaf list
                               + (want_compact ? 16ull * nc : 0ull);
        if (total > 0xFFFFFFFFull) { msg = "scene too large: the packed scene would exceed 4 GiB"; return false; }
    }
    L.all_finite = all_finite ? 1u : 0u;
    // cpu.rs:49-50 adds `attenuation * emission` at EVERY hit, emission = 0 for everything but lights: the sum stays +0 as long as
    // the attenuation is finite (x * 0 = +-0, 0 + +-0 = +0), which |albedo| <= 1 guarantees (NaN fails the comparison)
    bool lazy = true;
    for (const trt_material& m : w.materials) {
        if (m.kind == TRT_LIGHT) continue;
        lazy = lazy && fabsf(m.albedo.x) <= 1.0f && fabsf(m.albedo.y) <= 1.0f && fabsf(m.albedo.z) <= 1.0f;
    }
    L.lazy_color = lazy ? 1u : 0u;
    L.flat_walk = L.n_leaves <= kFlatWalkMaxLeaves ? 1u : 0u;
    if (opt.flat_walk >= 0) L.flat_walk = opt.flat_walk ? 1u : 0u;                           // tuning / tests; same frames either way

    out.blob.assign(L.blob_bytes, 0);
    F4* f4 = reinterpret_cast<F4*>(out.blob.data());
    uint32_t* u32 = reinterpret_cast<uint32_t*>(out.blob.data());
    // Node links: a leaf carries its primitive reference, an inner node NODE_INNER_BIT | index of its first child;
    // every node carries `skip`, the node to go to when its box is missed or its subtree is finished (END = count).
    auto pack_nodes = [&](F4* dst, const std::vector<Box>& boxes, const std::vector<int32_t>& prim_geo, const std::vector<int32_t>& skip,
                          const std::vector<uint32_t>& place) {
        // place[i] = position of pre-order node i in the packed array (identity for the reference tree)
        const size_t n = boxes.size();
        auto at = [&](int32_t pre) -> uint32_t { return (size_t)pre >= n ? (uint32_t)n : place[(size_t)pre]; };
        for (size_t i = 0; i < n; i++) {
            const Box& bx = boxes[i];
            uint32_t link;
            if (prim_geo[i] >= 0) {
                const Geometry& geo = w.geometries[(size_t)prim_geo[i]];
                link = local_index[(size_t)prim_geo[i]] | (geo.kind == 1 ? PRIM_QUAD_BIT : 0u);
            } else {
                link = NODE_INNER_BIT | at((int32_t)i + 1);          // first child follows in pre-order
            }
            const size_t o = place[i];
            dst[2 * o] = F4{bx.lo.x, bx.lo.y, bx.lo.z, bx.hi.x};
            dst[2 * o + 1] = F4{bx.hi.y, bx.hi.z, bitsf(at(skip[i])), bitsf(link)};
        }
    };
    std::vector<uint32_t> cull_place(nc), ref_place(nn);                  // both trees are packed in pre-order
    for (uint32_t i = 0; i < nc; i++) cull_place[i] = i;
    for (uint32_t i = 0; i < nn; i++) ref_place[i] = i;
    L.reserved0 = 0u;
    pack_nodes(f4, cb.node_box, cull_prim_geo, cb.node_skip, cull_place);
    pack_nodes(f4 + L.off_ref_nodes, b.node_box, b.node_prim, b.node_skip, ref_place);
    {   // leaf list: the leaves alone, in walk order
        const uint32_t nl = L.n_leaves;
        std::vector<int32_t> lprim(leaf_geo.begin(), leaf_geo.end()), lskip(nl);
        std::vector<uint32_t> lplace(nl);
        for (uint32_t k = 0; k < nl; k++) { lskip[k] = (int32_t)k + 1; lplace[k] = k; }
        pack_nodes(f4 + L.off_leaf_list, leaf_box, lprim, lskip, lplace);
        // padding: the lock-step walk requests the NEXT pair of leaves before it tests the current one, without clamping the index
        // (rt_path.h walk_flat); what it reads beyond the list are copies of the last leaf, which it never tests
        for (uint32_t k = 0; k < kLeafListPad; k++) {
            F4* last = f4 + L.off_leaf_list + 2u * (size_t)(nl - 1u);
            F4* dst = f4 + L.off_leaf_list + 2u * (size_t)(nl + k);
            dst[0] = last[0]; dst[1] = last[1];
        }
    }
    L.compact_origin_limit[0] = L.compact_origin_limit[1] = L.compact_origin_limit[2] = 0.0f;
    if (L.off_compact) {   // compact culling tree: f16 boxes rounded outward (any superset box keeps the hits: DESIGN.md 4.1), pre-order
        // Round 5: the hand-written walk over these nodes evaluates a plane's distance as fma(x, 1/d, -fl(o * 1/d)) - one instruction instead of the
        // reference's fl(fl(x - o) * 1/d), two - which is NOT the reference's value: it may err by u (|x| + 2.001 |o|) |1/d|, u = 2^-24, and the
        // reference's own value by 2.001 u (|x| + |o|) |1/d| (rt_path.h box_loop_compact has the derivation).  The boxes therefore grow by
        // eps = 2^-19 B per axis (B = the largest |coordinate| of the tree on that axis) BEFORE they are rounded outward to f16: for ray origins with
        // |o| <= 4 B that is 1.67 times the two errors together, so a coarse box never rejects what the reference's arithmetic on the exact box
        // accepts.  eps is 0.002 where f16's own spacing is 0.125 (|x| in [128, 256)): the tree does not get looser in any way that shows.
        // Rays that start further out (or whose 1/d exceeds 2^60) walk the reference tree like every ray outside the fast-slab domain.
        float eps[3];
        {
            const Box& root = cb.node_box[0];
            const float b3[3] = {fmaxf(fabsf(root.lo.x), fabsf(root.hi.x)), fmaxf(fabsf(root.lo.y), fabsf(root.hi.y)), fmaxf(fabsf(root.lo.z), fabsf(root.hi.z))};
            for (int a = 0; a < 3; a++) {
                const bool ok = all_finite && b3[a] <= 1.0e12f;                           // (products with 1/d <= 2^60 and 4 B stay far from overflow)
                eps[a] = ok ? b3[a] * 1.9073486328125e-06f : 0.0f;                       // 2^-19 B
                L.compact_origin_limit[a] = ok ? 4.0f * b3[a] : 0.0f;
            }
            if (!(L.compact_origin_limit[0] > 0.0f && L.compact_origin_limit[1] > 0.0f && L.compact_origin_limit[2] > 0.0f))
                L.compact_origin_limit[0] = L.compact_origin_limit[1] = L.compact_origin_limit[2] = 0.0f;
        }
        uint32_t* c = u32 + 4u * (size_t)L.off_compact;
        std::vector<int32_t> leaf_seq(nc, -1);
        {
            int32_t k = 0;
            for (uint32_t i = 0; i < nc; i++) if (cb.node_leaf[i] >= 0) leaf_seq[i] = k++;
        }
        for (uint32_t i = 0; i < nc; i++) {
            const Box& bx = cb.node_box[i];
            const uint32_t lx = f32_to_f16_dir(bx.lo.x - eps[0], false), ly = f32_to_f16_dir(bx.lo.y - eps[1], false), lz = f32_to_f16_dir(bx.lo.z - eps[2], false);
            const uint32_t hx = f32_to_f16_dir(bx.hi.x + eps[0], true), hy = f32_to_f16_dir(bx.hi.y + eps[1], true), hz = f32_to_f16_dir(bx.hi.z + eps[2], true);
            c[4 * i + 0] = lx | ly << 16;
            c[4 * i + 1] = lz | hx << 16;
            c[4 * i + 2] = hy | hz << 16;
            // an inner node's link is its skip node's BYTE offset in this array: the walk's cursor is the offset its load takes (box_loop_compact)
            c[4 * i + 3] = cb.node_leaf[i] >= 0 ? (0x80000000u | (uint32_t)cb.node_leaf[i]) : (uint32_t)cb.node_skip[i] << 4;
        }
    }
    dump_tree(out.reference, b.node_box, b.node_prim, b.node_skip);
    dump_tree(out.culling, cb.node_box, cull_prim_geo, cb.node_skip);
    for (uint32_t i = 0; i < ns; i++) { f4[L.off_sphere + i] = spheres[i]; u32[L.off_sphere_mat + i] = sphere_mat[i]; }
    for (uint32_t i = 0; i < nq; i++) {
        F4* q = f4 + L.off_quad + 5u * (size_t)i;                 // one record of five elements per quad
        q[0] = q0[i]; q[1] = q1[i]; q[2] = q2[i]; q[3] = q3[i]; q[4] = q4[i];
    }
    for (uint32_t i = 0; i < nm; i++) {
        const trt_material& m = w.materials[i];
        f4[L.off_material + i] = F4{m.albedo.x, m.albedo.y, m.albedo.z, m.param};
        u32[L.off_material_kind + i] = m.kind;
    }
    out.max_depth = b.max_depth.load();
    return true;
}

// Camera::new (camera.rs:17-56).  f32::to_radians multiplies by the f32 constant PI/180.
void camera_init(trt_camera& out, float focus_distance, float defocus_angle_deg, trt_vec3 position, trt_vec3 look_at,
                 trt_vec3 up, float vertical_fov_deg, uint32_t width, uint32_t height) {
    const float rad_per_deg = 3.14159265358979323846f / 180.0f;
    float viewport_height = 2.0f * focus_distance * tanf((vertical_fov_deg * rad_per_deg) / 2.0f);
    float aspect_ratio = (float)width / (float)height;
    float viewport_width = aspect_ratio * viewport_height;

    H3 pos = h3(position);
    H3 w = hnormalized(pos - h3(look_at));
    H3 u = hnormalized(hcross(h3(up), w));
    H3 v = hnormalized(hcross(w, u));

    H3 forward = w * focus_distance;
    H3 horizontal = u * viewport_width;
    H3 vertical = v * viewport_height;
    H3 upper_left = ((pos - horizontal / 2.0f) + vertical / 2.0f) - forward;

    float defocus_radius = focus_distance * tanf((defocus_angle_deg * rad_per_deg) / 2.0f);
    H3 du = u * defocus_radius, dv = v * defocus_radius;

    out.position = position;
    out.viewport_upper_left = trt_vec3{upper_left.x, upper_left.y, upper_left.z};
    out.forward = trt_vec3{forward.x, forward.y, forward.z};
    out.horizontal = trt_vec3{horizontal.x, horizontal.y, horizontal.z};
    out.vertical = trt_vec3{vertical.x, vertical.y, vertical.z};
    out.defocus_disk_u = trt_vec3{du.x, du.y, du.z};
    out.defocus_disk_v = trt_vec3{dv.x, dv.y, dv.z};
    out.width = width;
    out.height = height;
}

// Imager: set_pixel applies gamma (imager.rs:52-53, image.rs:38-44,92-98); RgbImage conversion clamps
// to [0, 0.999], scales by 255 and truncates (image.rs:101-111; Rust's `as u8` saturates, NaN -> 0).
void tonemap_u8(const float* accum, uint32_t npixels, float gamma, uint8_t* rgb) {
    const float inv_gamma = 1.0f / gamma;                                   // image.rs:94-96
    for (size_t i = 0; i < (size_t)npixels * 3; i++) rgb[i] = tm_quantise_channel(accum[i], inv_gamma);
}

}  // namespace trt
