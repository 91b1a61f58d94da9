// bvh4_walk_model.c — CPU model behind DESIGN.md section 13 (round 3): how many dependent node loads per ray does a K-wide
// collapse of the culling tree need on the 100 k-sphere scene, against the binary 16-byte-node walk the kernels run now, and how
// deep does its traversal stack get?  Same leaf order, same speculative schedule (leaves postponed into 4 slots, re-checked
// against the current t_best at their turn), so the hit of every ray is identical by construction; only the counts differ.
//   tools/proto/bvh4_walk_model.py dumps the scene (culling tree + spheres + camera) and runs this.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float lo[3], hi[3]; } Box;
static int n_nodes, n_geo;
static Box* nbox; static int32_t *nprim, *nskip;
static float (*sph)[4];

typedef struct { float o[3], d[3]; } Ray;
static const float TMIN = 0.001f;

static int slab(const Box* b, const Ray* r, const float inv[3], float tbest, float* start_out) {
    float tn = -INFINITY, tf = INFINITY;
    for (int a = 0; a < 3; a++) {
        float t0 = (b->lo[a] - r->o[a]) * inv[a], t1 = (b->hi[a] - r->o[a]) * inv[a];
        float mn = fminf(t0, t1), mx = fmaxf(t0, t1);
        tn = fmaxf(tn, mn); tf = fminf(tf, mx);
    }
    float start = fmaxf(TMIN, tn), end = fminf(tbest, tf);
    *start_out = start;
    return !(end <= start);
}
static int sphere_hit(int g, const Ray* r, float t1, float* t_out) {
    float oc[3] = {r->o[0] - sph[g][0], r->o[1] - sph[g][1], r->o[2] - sph[g][2]};
    float a = r->d[0] * r->d[0] + r->d[1] * r->d[1] + r->d[2] * r->d[2];
    float hb = oc[0] * r->d[0] + oc[1] * r->d[1] + oc[2] * r->d[2];
    float c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - sph[g][3] * sph[g][3];
    float disc = hb * hb - a * c;
    if (disc < 0) return 0;
    float sq = sqrtf(disc), t = (-hb - sq) / a;
    if (!(TMIN <= t && t < t1)) { t = (-hb + sq) / a; if (!(TMIN <= t && t < t1)) return 0; }
    *t_out = t;
    return 1;
}

// ---- binary walk as the kernel runs it (walk_compact): 4 postponed-leaf slots ----
typedef struct { long steps, leaf_tests; float t; int prim; } Res;
static Res walk_binary(const Ray* r, int slots) {
    float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]};
    Res R = {0, 0, INFINITY, -1};
    int i = 0, pend[16]; float pstart[16];
    for (;;) {
        int np = 0;
        while (i < n_nodes && np < slots) {
            float st; R.steps++;
            int pass = slab(&nbox[i], r, inv, R.t, &st);
            int leaf = nprim[i] >= 0;
            if (pass && leaf) { pend[np] = i; pstart[np] = st; np++; }
            i = (pass || leaf) ? i + 1 : nskip[i];
        }
        if (!np) break;
        for (int k = 0; k < np; k++) if (R.t > pstart[k]) { float t; R.leaf_tests++; if (sphere_hit(nprim[pend[k]], r, R.t, &t)) { R.t = t; R.prim = nprim[pend[k]]; } }
    }
    return R;
}

// ---- K-wide tree over the same leaf SEQUENCE: a range [a, b) of leaves is cut into up to K contiguous sub-ranges by greedy SAH
// (always re-split the sub-range whose SA x count is largest at its best SAH position); one-leaf sub-ranges are leaf children ----
#define KMAX 8
typedef struct { int n; Box box[KMAX]; int leaf[KMAX]; /* leaf number or -1 */ int wide[KMAX]; /* wide node index if inner, else -1 */ } Wide;
static Wide* wnodes; static int n_wide;
static int n_leaves; static Box* lbox; static int* lnode;          // leaf k: box, binary node index
static double area(const Box* b) { double x = b->hi[0] - b->lo[0], y = b->hi[1] - b->lo[1], z = b->hi[2] - b->lo[2]; return 2 * (x * y + y * z + z * x); }
static Box merge(Box a, const Box* b) { for (int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], b->lo[k]); a.hi[k] = fmaxf(a.hi[k], b->hi[k]); } return a; }
static Box range_box(int a, int b) { Box x = lbox[a]; for (int k = a + 1; k < b; k++) x = merge(x, &lbox[k]); return x; }
static Box* suffix_tmp;
static int best_split(int a, int b) {                             // SAH over the fixed order: left = [a, s), right = [s, b)
    int m = b - a;
    suffix_tmp[m - 1] = lbox[b - 1];
    for (int k = m - 2; k >= 0; k--) suffix_tmp[k] = merge(suffix_tmp[k + 1], &lbox[a + k]);
    Box pre = lbox[a]; double best = 0; int bs = a + 1;
    for (int k = 1; k < m; k++) {
        double c = area(&pre) * k + area(&suffix_tmp[k]) * (m - k);
        if (k == 1 || c < best) { best = c; bs = a + k; }
        pre = merge(pre, &lbox[a + k]);
    }
    return bs;
}
static int build_wide(int a, int b, int K) {
    int me = n_wide++;
    int ra[KMAX + 1], rb[KMAX + 1], nr = 1;
    ra[0] = a; rb[0] = b;
    while (nr < K) {
        int pick = -1; double pc = -1;
        for (int k = 0; k < nr; k++) if (rb[k] - ra[k] > 1) { Box x = range_box(ra[k], rb[k]); double c = area(&x) * (rb[k] - ra[k]); if (c > pc) { pc = c; pick = k; } }
        if (pick < 0) break;
        int s = best_split(ra[pick], rb[pick]);
        memmove(&ra[pick + 2], &ra[pick + 1], sizeof(int) * (size_t)(nr - pick - 1));
        memmove(&rb[pick + 2], &rb[pick + 1], sizeof(int) * (size_t)(nr - pick - 1));
        ra[pick + 1] = s; rb[pick + 1] = rb[pick]; rb[pick] = s;
        nr++;
    }
    Wide w; w.n = nr;
    for (int k = 0; k < nr; k++) { w.box[k] = range_box(ra[k], rb[k]); w.leaf[k] = rb[k] - ra[k] == 1 ? ra[k] : -1; w.wide[k] = -1; }
    wnodes[me] = w;
    for (int k = 0; k < nr; k++) if (rb[k] - ra[k] > 1) { int id = build_wide(ra[k], rb[k], K); wnodes[me].wide[k] = id; }
    return me;
}

typedef struct { long loads, box_tests, pushes, pops_inner, leaf_tests, max_depth, dropped; float t; int prim; } WRes;
static WRes walk_wide(const Ray* r, int slots) {
    float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]};
    WRes R; memset(&R, 0, sizeof R); R.t = INFINITY; R.prim = -1;
    struct { int node; int wide; float start; } stack[256]; int sp = 0;      // node: binary index (box / leaf identity); wide: -1 for a leaf
    int pend[16]; float pstart[16]; int np = 0;
    int cur = 0;                                     // wide node to expand, -1: none
#define LEAF_PHASE() do { for (int k_ = 0; k_ < np; k_++) if (R.t > pstart[k_]) { float t_; R.leaf_tests++; if (sphere_hit(nprim[pend[k_]], r, R.t, &t_)) { R.t = t_; R.prim = nprim[pend[k_]]; } } np = 0; } while (0)
    for (;;) {
        if (cur >= 0) {
            const Wide* w = &wnodes[cur];
            R.loads++;
            int pass[KMAX]; float st[KMAX];
            for (int k = 0; k < w->n; k++) { R.box_tests++; pass[k] = slab(&w->box[k], r, inv, R.t, &st[k]); }
            int f = -1;
            for (int k = 0; k < w->n; k++) if (pass[k] && w->wide[k] >= 0) { f = k; break; }
            int lim = f < 0 ? w->n : f;
            // what follows the first passing inner child waits on the stack (reverse order: next in walk order on top)
            if (f >= 0) for (int k = w->n - 1; k > f; k--) if (pass[k]) { stack[sp].node = w->leaf[k] >= 0 ? lnode[w->leaf[k]] : -1; stack[sp].wide = w->wide[k]; stack[sp].start = st[k]; sp++; R.pushes++; if (sp > R.max_depth) R.max_depth = sp; }
            for (int k = 0; k < lim; k++) if (pass[k]) {          // leaves that come before it: next in walk order
                if (np == slots) LEAF_PHASE();
                pend[np] = lnode[w->leaf[k]]; pstart[np] = st[k]; np++;
            }
            cur = f >= 0 ? w->wide[f] : -1;
            continue;
        }
        if (sp == 0) { LEAF_PHASE(); break; }
        sp--;
        if (stack[sp].wide < 0) {                                  // a leaf whose turn has come
            if (np == slots) LEAF_PHASE();
            pend[np] = stack[sp].node; pstart[np] = stack[sp].start; np++;
        } else if (R.t > stack[sp].start) {                        // its box with the current t_best: one comparison
            R.pops_inner++;
            cur = stack[sp].wide;
        } else R.dropped++;
    }
    return R;
}

static uint32_t rs = 12345u;
static float urand(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return (float)(rs >> 8) / 16777216.0f; }

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    int W = atoi(argv[2]), H = atoi(argv[3]), depth = atoi(argv[4]);
    float cam[12];
    if (fread(&n_nodes, 4, 1, f) != 1 || fread(&n_geo, 4, 1, f) != 1) return 1;
    nbox = malloc(sizeof(Box) * (size_t)n_nodes); nprim = malloc(4 * (size_t)n_nodes); nskip = malloc(4 * (size_t)n_nodes); sph = malloc(16 * (size_t)n_geo);
    if (fread(nbox, sizeof(Box), (size_t)n_nodes, f) != (size_t)n_nodes || fread(nprim, 4, (size_t)n_nodes, f) != (size_t)n_nodes ||
        fread(nskip, 4, (size_t)n_nodes, f) != (size_t)n_nodes || fread(sph, 16, (size_t)n_geo, f) != (size_t)n_geo || fread(cam, 4, 12, f) != 12) return 1;
    fclose(f);
    for (int K = 4; K <= 8; K += 4) {
        if (!lbox) {
            lbox = malloc(sizeof(Box) * (size_t)n_nodes); lnode = malloc(4 * (size_t)n_nodes); suffix_tmp = malloc(sizeof(Box) * (size_t)n_nodes);
            for (int i = 0; i < n_nodes; i++) if (nprim[i] >= 0) { lbox[n_leaves] = nbox[i]; lnode[n_leaves] = i; n_leaves++; }
        }
        wnodes = malloc(sizeof(Wide) * (size_t)n_leaves); n_wide = 0;
        build_wide(0, n_leaves, K);
        long fill[KMAX + 1] = {0}, leaf_kids = 0, inner_kids = 0;
        for (int i = 0; i < n_wide; i++) { fill[wnodes[i].n]++; for (int k = 0; k < wnodes[i].n; k++) if (wnodes[i].wide[k] < 0) leaf_kids++; else inner_kids++; }
        printf("K=%d: %d wide nodes from %d binary nodes (%ld leaf children, %ld inner children); fill:", K, n_wide, n_nodes, leaf_kids, inner_kids);
        for (int k = 1; k <= K; k++) printf(" %d:%ld", k, fill[k]);
        printf("\n");
        rs = 12345u;
        long rays = 0, bsteps = 0, bleaf = 0, prim_rays = 0, sec_rays = 0;
        long loads = 0, tests = 0, pushes = 0, pops = 0, wleaf = 0, mism = 0, dropped = 0, depth_hist[64] = {0};
        long loads_prim = 0, loads_sec = 0, bsteps_prim = 0, bsteps_sec = 0;
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
            Ray r; float u = ((float)x + urand()) / (float)(W - 1), v = ((float)y + urand()) / (float)(H - 1);
            float dir[3], len = 0;
            for (int a = 0; a < 3; a++) { r.o[a] = cam[a]; dir[a] = cam[3 + a] + u * cam[6 + a] - v * cam[9 + a] - cam[a]; len += dir[a] * dir[a]; }
            len = sqrtf(len); for (int a = 0; a < 3; a++) r.d[a] = dir[a] / len;
            for (int b = 0; b < depth; b++) {
                Res rb = walk_binary(&r, 4);
                WRes rw = walk_wide(&r, 4);
                rays++; bsteps += rb.steps; bleaf += rb.leaf_tests;
                loads += rw.loads; tests += rw.box_tests; pushes += rw.pushes; pops += rw.pops_inner; wleaf += rw.leaf_tests; dropped += rw.dropped;
                if (b == 0) { prim_rays++; loads_prim += rw.loads; bsteps_prim += rb.steps; } else { sec_rays++; loads_sec += rw.loads; bsteps_sec += rb.steps; }
                depth_hist[rw.max_depth < 63 ? rw.max_depth : 63]++;
                if (rb.prim != rw.prim || rb.t != rw.t) mism++;
                if (rb.prim < 0) break;
                // Lambertian-like bounce: normal + random unit vector
                float p[3], n[3], nl = 0;
                for (int a = 0; a < 3; a++) { p[a] = r.o[a] + rb.t * r.d[a]; n[a] = p[a] - sph[rb.prim][a]; nl += n[a] * n[a]; }
                nl = sqrtf(nl);
                float z = 1 - 2 * urand(), ph = 6.2831853f * urand(), s = sqrtf(fmaxf(0.f, 1 - z * z));
                float rv[3] = {s * cosf(ph), s * sinf(ph), z}, dl = 0;
                for (int a = 0; a < 3; a++) { r.o[a] = p[a]; dir[a] = n[a] / nl + rv[a]; dl += dir[a] * dir[a]; }
                dl = sqrtf(dl); if (dl < 1e-6f) break;
                for (int a = 0; a < 3; a++) r.d[a] = dir[a] / dl;
                if (urand() < 0.5f) break;                                   // ~ the kernels' 2.5 rays per sample
            }
        }
        printf("  %ld rays (%ld primary, %ld secondary), hits identical: %s (%ld mismatches)\n", rays, prim_rays, sec_rays, mism ? "NO" : "yes", mism);
        printf("  binary walk: %.1f box steps (= dependent 16-byte loads) per ray (primary %.1f, secondary %.1f), %.2f sphere tests\n",
               (double)bsteps / rays, (double)bsteps_prim / prim_rays, (double)bsteps_sec / (sec_rays ? sec_rays : 1), (double)bleaf / rays);
        printf("  %d-wide walk: %.1f node loads per ray (primary %.1f, secondary %.1f), %.1f child box tests, %.2f pushes, %.2f inner pops (%.2f dropped by t_best), %.2f sphere tests\n",
               K, (double)loads / rays, (double)loads_prim / prim_rays, (double)loads_sec / (sec_rays ? sec_rays : 1), (double)tests / rays, (double)pushes / rays, (double)pops / rays,
               (double)dropped / rays, (double)wleaf / rays);
        printf("  traversal-stack depth needed, fraction of rays: ");
        long cum = 0; for (int d = 0; d < 64; d++) { cum += depth_hist[d]; if (depth_hist[d]) printf("<=%d: %.4f  ", d, (double)cum / rays); }
        printf("\n");
        free(wnodes);
    }
    return 0;
}
