// scene_host_check.cpp — CPU-only check of the host scene compiler, meant to be built with
// -fsanitize=address,undefined (tests/test_host_sanitizers.py).  Random worlds in, invariants checked:
// reference tree = 2N-1 nodes in pre-order with consistent skip links; culling tree = same leaves in the same
// order, inner boxes = exact unions of the leaves below them.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../tiny-raytracer_amd/csrc/scene.h"

using namespace trt;

static uint32_t g_state = 12345u;
static float frand() { g_state = g_state * 1664525u + 1013904223u; return (float)((g_state >> 8) & 0xFFFFFF) / 16777216.0f; }
static float frange(float a, float b) { return a + (b - a) * frand(); }

static int check_tree(const NodeDump& d, size_t n_geo, bool binary, const char* what) {
    const size_t n = d.skip.size();
    std::vector<int> seen(n_geo, 0);
    for (size_t i = 0; i < n; i++) {
        if (d.skip[i] <= (int32_t)i || d.skip[i] > (int32_t)n) { std::printf("%s: bad skip at %zu\n", what, i); return 1; }
        if (d.prim_geo[i] >= 0) {
            if ((size_t)d.prim_geo[i] >= n_geo || d.skip[i] != (int32_t)i + 1) { std::printf("%s: bad leaf at %zu\n", what, i); return 1; }
            seen[(size_t)d.prim_geo[i]]++;
        } else {
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int32_t j = (int32_t)i + 1; j < d.skip[i]; j++) {
                if (d.skip[(size_t)j] > d.skip[i]) { std::printf("%s: subtree of %zu not nested\n", what, i); return 1; }
                if (d.prim_geo[(size_t)j] < 0) continue;
                for (int k = 0; k < 3; k++) {
                    lo[k] = fminf(lo[k], d.bbox6[6 * (size_t)j + k]);
                    hi[k] = fmaxf(hi[k], d.bbox6[6 * (size_t)j + 3 + k]);
                }
            }
            for (int k = 0; k < 3; k++) {
                if (d.bbox6[6 * i + k] != lo[k] || d.bbox6[6 * i + 3 + k] != hi[k]) { std::printf("%s: box of %zu is not the union of its leaves\n", what, i); return 1; }
            }
        }
    }
    for (size_t g = 0; g < n_geo; g++) if (seen[g] != 1) { std::printf("%s: geometry %zu appears %d times\n", what, g, seen[g]); return 1; }
    if (binary && n != 2 * n_geo - 1) { std::printf("%s: %zu nodes for %zu leaves\n", what, n, n_geo); return 1; }
    return 0;
}

// f16 bits -> f32 (the device's v_cvt_f32_f16)
static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t out;
    if (e == 0u) { float v = (float)m * 5.9604644775390625e-08f; std::memcpy(&out, &v, 4); out |= sign; }
    else if (e == 31u) out = sign | 0x7F800000u | (m << 13);
    else out = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &out, 4); return f;
}
// the slab test of the 16-byte-node walks on a decoded node (rt_device.h slab_fast6_entry)
static bool slab16(const uint32_t* w, const float o[3], const float inv[3], float t_best) {
    const float lo[3] = {half_to_float((uint16_t)w[0]), half_to_float((uint16_t)(w[0] >> 16)), half_to_float((uint16_t)w[1])};
    const float hi[3] = {half_to_float((uint16_t)(w[1] >> 16)), half_to_float((uint16_t)w[2]), half_to_float((uint16_t)(w[2] >> 16))};
    float tn = 0.001f, tf = t_best;
    for (int k = 0; k < 3; k++) {
        const float a = (lo[k] - o[k]) * inv[k], b = (hi[k] - o[k]) * inv[k];
        tn = fmaxf(tn, fminf(a, b));
        tf = fminf(tf, fmaxf(a, b));
    }
    return !(tf <= tn);
}
// The hybrid layout (scene.h; scene_host.cpp build_hybrid) against the compact tree it was cut from: for random finite rays and random
// t_best the cursor walk over top + main visits the same real nodes in the same order, pushes the same leaves, and ends; portals never pass.
static long g_hybrid_checked = 0, g_hybrid_main = 0;
static int check_hybrid(const SceneHost& s, uint32_t budget) {
    const SceneLayout& L = s.layout;
    if (L.off_compact == 0u) return 0;
    if (budget < 4u) { if (L.n_hyb_top != 0u) { std::printf("hybrid layout without a budget\n"); return 1; } return 0; }
    g_hybrid_checked++; g_hybrid_main += L.n_hyb_main;
    if (L.n_hyb_top == 0u || L.n_hyb_top > budget) { std::printf("hybrid top part: %u entries for a budget of %u\n", L.n_hyb_top, budget); return 1; }
    const uint32_t* words = reinterpret_cast<const uint32_t*>(s.blob.data());
    const uint32_t* compact = words + 4u * (size_t)L.off_compact;
    const uint32_t* top = words + 4u * (size_t)L.off_hyb_top;
    const uint32_t* main_ = words + 4u * (size_t)L.off_hyb_main;
    const uint32_t nc = L.n_cull_nodes;
    for (int r = 0; r < 200; r++) {
        float o[3] = {frange(-80, 80), frange(-80, 80), frange(-80, 80)}, d[3] = {frange(-1, 1), frange(-1, 1), frange(-1, 1)}, inv[3];
        for (int k = 0; k < 3; k++) { if (fabsf(d[k]) < 1e-3f) d[k] = 1e-3f; inv[k] = 1.0f / d[k]; }
        const float t_best = r % 3 == 0 ? INFINITY : frange(1.0f, 150.0f);
        std::vector<uint32_t> a_nodes, b_nodes;                                   // first three words' hash + link kind of every real node stepped
        std::vector<uint32_t> a_leaves, b_leaves;
        for (uint32_t i = 0; i < nc;) {
            const uint32_t* w = compact + 4u * (size_t)i;
            const bool pass = slab16(w, o, inv, t_best), leaf = (w[3] & 0x80000000u) != 0u;
            a_nodes.push_back(w[0] ^ (w[1] * 31u) ^ (w[2] * 131u));
            if (pass && leaf) a_leaves.push_back(w[3] & 0x7FFFFFFFu);
            i = (pass || leaf) ? i + 1u : w[3];
        }
        uint32_t cur = 0, steps = 0;
        while (cur != L.n_hyb_top) {
            if (++steps > 4u * nc + 16u) { std::printf("hybrid walk does not end\n"); return 1; }
            const bool in_main = (cur & kHybMainBit) != 0u;
            const uint32_t idx = cur & ~kHybMainBit;
            if (in_main ? idx >= L.n_hyb_main : idx >= L.n_hyb_top) { std::printf("hybrid cursor %08x out of range\n", cur); return 1; }
            const uint32_t* w = (in_main ? main_ : top) + 4u * (size_t)idx;
            const bool portal = w[0] == 0x7C00u && w[1] == (0x7C00u << 16) && w[2] == 0u;
            const bool pass = slab16(w, o, inv, t_best), leaf = (w[3] & 0x80000000u) != 0u;
            if (portal) { if (pass || leaf) { std::printf("a portal passed\n"); return 1; } cur = w[3]; continue; }
            b_nodes.push_back(w[0] ^ (w[1] * 31u) ^ (w[2] * 131u));
            if (pass && leaf) b_leaves.push_back(w[3] & 0x7FFFFFFFu);
            cur = (pass || leaf) ? cur + 1u : w[3];
        }
        if (a_nodes != b_nodes || a_leaves != b_leaves) { std::printf("hybrid walk differs from the compact walk (%zu vs %zu nodes, %zu vs %zu leaves)\n", a_nodes.size(), b_nodes.size(), a_leaves.size(), b_leaves.size()); return 1; }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 40;
    for (int r = 0; r < rounds; r++) {
        World w;
        const int n_mat = 1 + (int)(frand() * 5);
        for (int i = 0; i < n_mat; i++) {
            w.material_index.emplace("m" + std::to_string(i), (uint32_t)i);
            w.materials.push_back(trt_material{(uint32_t)(frand() * 4), trt_vec3{frand(), frand(), frand()}, frange(-1, 2)});
        }
        const int n_geo = r < 4 ? r + 1 : 1 + (int)(frand() * (r % 5 == 0 ? 3000 : 200));
        for (int i = 0; i < n_geo; i++) {
            Geometry g{};
            g.kind = frand() < 0.5f ? 0u : 1u;
            g.material = (uint32_t)(frand() * n_mat) % (uint32_t)n_mat;
            g.a = trt_vec3{frange(-50, 50), frange(-50, 50), frange(-50, 50)};
            if (g.kind == 0) g.b = trt_vec3{frange(0, 5), 0, 0};
            else { g.b = trt_vec3{frange(-5, 5), frange(-5, 5), frange(-5, 5)}; g.c = trt_vec3{frange(-5, 5), frange(-5, 5), frange(-5, 5)}; }
            if (r % 7 == 3 && i % 11 == 0) g.a.x = g.a.y;                       // ties on the sort key
            w.geometries.push_back(g);
        }
        SceneHost s;
        std::string msg;
        trt_scene_options opt = scene_options_builtin();                        // every third world with another placement: same leaf sequence
        if (r % 3 == 1) { opt.cull_prune = 0.2f + 0.1f * (float)(r % 8); opt.compact_nodes = r & 1; opt.flat_walk = (r >> 1) & 1; opt.top_nodes = (uint32_t)(r % 5) * 31u; }
        if (r % 3 == 2) { opt.compact_nodes = 1; opt.top_nodes = (uint32_t[]){4u, 7u, 40u, 320u, 1280u, 5000u}[r % 6]; }        // the top-in-LDS split of the 16-byte tree
        if (!compile_scene(w, opt, s, msg)) { std::printf("compile failed: %s\n", msg.c_str()); return 1; }
        if (check_tree(s.reference, (size_t)n_geo, true, "reference")) return 1;
        if (check_tree(s.culling, (size_t)n_geo, false, "culling")) return 1;
        // same leaf sequence
        std::vector<int32_t> a, b;
        for (int32_t p : s.reference.prim_geo) if (p >= 0) a.push_back(p);
        for (int32_t p : s.culling.prim_geo) if (p >= 0) b.push_back(p);
        if (a != b) { std::printf("leaf sequences differ\n"); return 1; }
        if (s.blob.size() != s.layout.blob_bytes || s.layout.hot_bytes > s.layout.blob_bytes) { std::printf("layout sizes\n"); return 1; }
        if (check_hybrid(s, opt.top_nodes > kHybridTopMax ? kHybridTopMax : opt.top_nodes)) return 1;
    }
    World empty;
    SceneHost s;
    std::string msg;
    if (compile_scene(empty, scene_options_builtin(), s, msg)) { std::printf("empty world must fail\n"); return 1; }
    trt_camera cam;
    camera_init(cam, 1.0f, 10.0f, trt_vec3{0, 0, 0}, trt_vec3{0, 0, 1}, trt_vec3{0, 1, 0}, 90.0f, 16, 9);
    std::vector<float> acc = {NAN, -1.0f, 0.5f, 2.0f, INFINITY, 0.0f};
    std::vector<uint8_t> rgb(6);
    tonemap_u8(acc.data(), 2, 2.2f, rgb.data());
    if (rgb[0] != 0 || rgb[1] != 0 || rgb[3] != 254 || rgb[4] != 254) { std::printf("tonemap\n"); return 1; }
    std::printf("ok %d worlds (%ld with a top-in-LDS split, %ld entries in their main parts)\n", rounds, g_hybrid_checked, g_hybrid_main);
    return 0;
}
