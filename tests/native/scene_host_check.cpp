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
        if (!compile_scene(w, opt, s, msg)) { std::printf("compile failed: %s\n", msg.c_str()); return 1; }
        if (check_tree(s.reference, (size_t)n_geo, true, "reference")) return 1;
        if (check_tree(s.culling, (size_t)n_geo, false, "culling")) return 1;
        // same leaf sequence
        std::vector<int32_t> a, b;
        for (int32_t p : s.reference.prim_geo) if (p >= 0) a.push_back(p);
        for (int32_t p : s.culling.prim_geo) if (p >= 0) b.push_back(p);
        if (a != b) { std::printf("leaf sequences differ\n"); return 1; }
        if (s.blob.size() != s.layout.blob_bytes || s.layout.hot_bytes > s.layout.blob_bytes) { std::printf("layout sizes\n"); return 1; }
    }
    {   // a world large enough for the THREADED builders (round 5: reference-order tree from 32 768 primitives, culling tree from 65 536 leaves): the same
        // invariants, and two builds of it - whatever the threads' interleaving - pack the same bytes
        World big;
        big.material_index.emplace("m", 0u);
        big.materials.push_back(trt_material{0u, trt_vec3{0.5f, 0.5f, 0.5f}, 0.0f});
        const int n_big = 90000;
        for (int i = 0; i < n_big; i++) {
            Geometry g{};
            g.kind = i % 5 == 0 ? 1u : 0u;
            g.a = trt_vec3{frange(-300, 300), frange(0, 3), frange(-300, 300)};
            if (i % 97 == 0) g.a.x = std::floor(g.a.x);                           // ties on the sort key
            if (g.kind == 0) g.b = trt_vec3{frange(0.05f, 0.5f), 0, 0};
            else { g.b = trt_vec3{frange(-1, 1), frange(-1, 1), frange(-1, 1)}; g.c = trt_vec3{frange(-1, 1), frange(-1, 1), frange(-1, 1)}; }
            big.geometries.push_back(g);
        }
        SceneHost s1, s2;
        std::string m1;
        if (!compile_scene(big, scene_options_builtin(), s1, m1) || !compile_scene(big, scene_options_builtin(), s2, m1)) { std::printf("big world: %s\n", m1.c_str()); return 1; }
        if (check_tree(s1.reference, (size_t)n_big, true, "big reference")) return 1;
        if (check_tree(s1.culling, (size_t)n_big, false, "big culling")) return 1;
        if (s1.blob != s2.blob || s1.reference.skip != s2.reference.skip || s1.culling.skip != s2.culling.skip) { std::printf("two builds of one world differ\n"); return 1; }
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
    std::printf("ok %d worlds\n", rounds);
    return 0;
}
