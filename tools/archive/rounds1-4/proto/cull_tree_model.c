// cull_tree_model.c — CPU model for tuning the CULLING TREE (scene_host.cpp CullBuilder): any hierarchy of unions over the reference's
// leaf sequence gives the reference's hits, so its shape is free.  For a scene dump (leaf boxes in walk order + spheres + camera) this
// builds the tree with a given pruning rule and reports, over waves of 64 path-traced rays in lock-step: box steps per ray (mean), box-step
// trips per round (the longest walk of the 64) and their ratio = the lane utilisation of the box-step loop.
//   rule 0: emit an inner node iff SA(node) < f * SA(nearest emitted ancestor)                         (the product's rule)
//   rule 2: rule 0 with factor 0.5 and the split cost SA_l * n_l^a + SA_r * n_r^a, a = the given factor (1 = plain SAH)
//   rule 1: emit iff (1 - SA(node)/SA(ancestor)) * leaves(node) >= f        (expected leaves skipped per visit of the ancestor's box)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float lo[3], hi[3]; } Box;
static int n_leaves; static Box* lbox; static int32_t* lgeo; static int n_geo; static float (*sph)[4];
static int n_nodes; static Box* nbox; static int32_t *nleaf, *nskip;
static double area(const Box* b) { double x = fmax(b->hi[0] - b->lo[0], 0), y = fmax(b->hi[1] - b->lo[1], 0), z = fmax(b->hi[2] - b->lo[2], 0); return 2 * (x * y + y * z + z * x); }
static Box merge(Box a, const Box* b) { for (int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], b->lo[k]); a.hi[k] = fmaxf(a.hi[k], b->hi[k]); } return a; }
static Box* suffix;
static int rule; static double factor; static double alpha = 1.0, prune0 = 0.5;
static void build(int a, int b, double parent_sa) {
    int m = b - a; Box all = lbox[a]; for (int k = a + 1; k < b; k++) all = merge(all, &lbox[k]);
    double sa = area(&all);
    int emit = m == 1 || parent_sa < 0 || (rule == 0 ? sa < factor * parent_sa : rule == 1 ? (1.0 - sa / parent_sa) * m >= factor : sa < prune0 * parent_sa);
    int me = -1;
    if (emit) { me = n_nodes++; nbox[me] = all; nleaf[me] = m == 1 ? a : -1; nskip[me] = 0; }
    if (m == 1) { nskip[me] = me + 1; return; }
    suffix[m - 1] = lbox[b - 1]; for (int k = m - 2; k >= 0; k--) suffix[k] = merge(suffix[k + 1], &lbox[a + k]);
    Box pre = lbox[a]; double best = 0; int bk = 1;
    Box* sfx = malloc(sizeof(Box) * (size_t)m); memcpy(sfx, suffix, sizeof(Box) * (size_t)m);
    for (int k = 1; k < m; k++) { double c = area(&pre) * pow((double)k, alpha) + area(&sfx[k]) * pow((double)(m - k), alpha); if (k == 1 || c < best) { best = c; bk = k; } pre = merge(pre, &lbox[a + k]); }
    free(sfx);
    double psa = emit ? sa : parent_sa;
    build(a, a + bk, psa); build(a + bk, b, psa);
    if (emit) nskip[me] = n_nodes;
}
typedef struct { float o[3], d[3]; } Ray;
static const float TMIN = 0.001f;
static int slab(const Box* b, const Ray* r, const float inv[3], float tbest, float* st) {
    float tn = -INFINITY, tf = INFINITY;
    for (int a = 0; a < 3; a++) { float t0 = (b->lo[a] - r->o[a]) * inv[a], t1 = (b->hi[a] - r->o[a]) * inv[a]; tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1)); }
    float s = fmaxf(TMIN, tn), e = fminf(tbest, tf); *st = s; return !(e <= s);
}
static int sphere_hit(int g, const Ray* r, float t1, float* t_out) {
    float oc[3] = {r->o[0] - sph[g][0], r->o[1] - sph[g][1], r->o[2] - sph[g][2]};
    float a = r->d[0]*r->d[0] + r->d[1]*r->d[1] + r->d[2]*r->d[2], hb = oc[0]*r->d[0] + oc[1]*r->d[1] + oc[2]*r->d[2];
    float c = oc[0]*oc[0] + oc[1]*oc[1] + oc[2]*oc[2] - sph[g][3]*sph[g][3], disc = hb*hb - a*c;
    if (disc < 0) return 0;
    float sq = sqrtf(disc), t = (-hb - sq) / a;
    if (!(TMIN <= t && t < t1)) { t = (-hb + sq) / a; if (!(TMIN <= t && t < t1)) return 0; }
    *t_out = t; return 1;
}
static int walk(const Ray* r, int slots, float* t_out, int* prim_out) {          // returns box steps
    float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]}, tb = INFINITY; int prim = -1, steps = 0, i = 0, pend[16]; float ps[16];
    for (;;) {
        int np = 0;
        while (i < n_nodes && np < slots) { float st; steps++; int pass = slab(&nbox[i], r, inv, tb, &st), leaf = nleaf[i] >= 0; if (pass && leaf) { pend[np] = i; ps[np] = st; np++; } i = (pass && !leaf) ? i + 1 : nskip[i]; }
        if (!np) break;
        for (int k = 0; k < np; k++) if (tb > ps[k]) { float t; if (sphere_hit(lgeo[nleaf[pend[k]]], r, tb, &t)) { tb = t; prim = lgeo[nleaf[pend[k]]]; } }
    }
    *t_out = tb; *prim_out = prim; return steps;
}
static uint32_t rs = 4242u;
static float urand(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return (float)(rs >> 8) / 16777216.0f; }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); int W = atoi(argv[2]), H = atoi(argv[3]), slots = atoi(argv[4]); float cam[12];
    if (fread(&n_leaves, 4, 1, f) != 1 || fread(&n_geo, 4, 1, f) != 1) return 1;
    lbox = malloc(sizeof(Box) * (size_t)n_leaves); lgeo = malloc(4 * (size_t)n_leaves); sph = malloc(16 * (size_t)n_geo);
    if (fread(lbox, sizeof(Box), (size_t)n_leaves, f) != (size_t)n_leaves || fread(lgeo, 4, (size_t)n_leaves, f) != (size_t)n_leaves ||
        fread(sph, 16, (size_t)n_geo, f) != (size_t)n_geo || fread(cam, 4, 12, f) != 12) return 1;
    fclose(f);
    nbox = malloc(sizeof(Box) * 2 * (size_t)n_leaves); nleaf = malloc(8 * (size_t)n_leaves); nskip = malloc(8 * (size_t)n_leaves); suffix = malloc(sizeof(Box) * (size_t)n_leaves);
    for (int a = 5; a + 1 < argc; a += 2) {
        rule = atoi(argv[a]); factor = atof(argv[a + 1]); alpha = rule == 2 ? factor : 1.0;
        n_nodes = 0; build(0, n_leaves, -1.0);
        rs = 4242u;
        double steps = 0, trips = 0; long rays = 0, rounds = 0;
        for (int ty = 0; ty + 8 <= H; ty += 40) for (int tx = 0; tx + 8 <= W; tx += 40) {
            Ray path[64]; int live[64];
            for (int l = 0; l < 64; l++) {
                int x = tx + (l & 7), y = ty + (l >> 3); float u = ((float)x + urand()) / (float)(W - 1), v = ((float)y + urand()) / (float)(H - 1), dir[3], len = 0;
                for (int a2 = 0; a2 < 3; a2++) { path[l].o[a2] = cam[a2]; dir[a2] = cam[3 + a2] + u * cam[6 + a2] - v * cam[9 + a2] - cam[a2]; len += dir[a2] * dir[a2]; }
                len = sqrtf(len); for (int a2 = 0; a2 < 3; a2++) path[l].d[a2] = dir[a2] / len; live[l] = 1;
            }
            for (int bounce = 0; bounce < 6; bounce++) {                      // a wave's lanes keep their paths; dead lanes idle (no refill: a bound, not the kernel)
                int mx = 0, nl = 0;
                for (int l = 0; l < 64; l++) if (live[l]) {
                    float t; int prim; int s = walk(&path[l], slots, &t, &prim);
                    steps += s; rays++; nl++; if (s > mx) mx = s;
                    if (getenv("DUMP_STEPS")) { static FILE* df; if (!df) df = fopen(getenv("DUMP_STEPS"), "w"); fprintf(df, "%d %d\n", bounce, s); }
                    if (prim < 0) { live[l] = 0; continue; }
                    float p[3], n[3], nlen = 0, dir[3], dl = 0;
                    for (int a2 = 0; a2 < 3; a2++) { p[a2] = path[l].o[a2] + t * path[l].d[a2]; n[a2] = p[a2] - sph[prim][a2]; nlen += n[a2] * n[a2]; }
                    nlen = sqrtf(nlen);
                    float z = 1 - 2 * urand(), ph = 6.2831853f * urand(), sn = sqrtf(fmaxf(0.f, 1 - z * z)), rv[3] = {sn * cosf(ph), sn * sinf(ph), z};
                    for (int a2 = 0; a2 < 3; a2++) { path[l].o[a2] = p[a2]; dir[a2] = n[a2] / nlen + rv[a2]; dl += dir[a2] * dir[a2]; }
                    dl = sqrtf(dl); if (dl < 1e-6f) { live[l] = 0; continue; }
                    for (int a2 = 0; a2 < 3; a2++) path[l].d[a2] = dir[a2] / dl;
                }
                if (nl) { trips += (double)mx * nl / 64.0 * (64.0 / nl); rounds++; }
                if (!nl) break;
            }
        }
        printf("rule %d factor %-5g: %6d nodes  %6.2f steps per ray  %6.2f trips per round (longest of the wave)  utilisation %.3f\n", rule, factor, n_nodes,
               steps / rays, trips / rounds, (steps / rays) / (trips / rounds));
    }
    return 0;
}
