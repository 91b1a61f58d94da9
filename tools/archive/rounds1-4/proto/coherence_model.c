// coherence_model.c — CPU model for DESIGN.md section 13 (b): how many distinct 128-byte L1 lines does one wave load of the binary
// 16-byte-node walk touch, for (1) the primary rays of an 8x8 pixel tile, (2) their secondary rays in the lanes that traced the
// primaries (what the kernel does), (3) secondary rays of four tiles re-binned into four waves by direction octant, (4) by octant of
// the direction AND the 2-D cell of the origin.  A wave's lanes step their walks in lock-step (trip k = every live lane's k-th box step).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float lo[3], hi[3]; } Box;
static int n_nodes, n_geo; static Box* nbox; static int32_t *nprim, *nskip; static float (*sph)[4];
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
#define MAXSTEPS 4096
typedef struct { int n; int32_t node[MAXSTEPS]; float t; int prim; } Trace;
static void walk(const Ray* r, Trace* T) {
    float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]};
    T->n = 0; T->t = INFINITY; T->prim = -1;
    int i = 0, pend[4]; float ps[4];
    for (;;) {
        int np = 0;
        while (i < n_nodes && np < 4) {
            float st; if (T->n < MAXSTEPS) T->node[T->n++] = i;
            int pass = slab(&nbox[i], r, inv, T->t, &st), leaf = nprim[i] >= 0;
            if (pass && leaf) { pend[np] = i; ps[np] = st; np++; }
            i = (pass || leaf) ? i + 1 : nskip[i];
        }
        if (!np) break;
        for (int k = 0; k < np; k++) if (T->t > ps[k]) { float t; if (sphere_hit(nprim[pend[k]], r, T->t, &t)) { T->t = t; T->prim = nprim[pend[k]]; } }
    }
}
static uint32_t rs = 777u;
static float urand(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return (float)(rs >> 8) / 16777216.0f; }
static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }
// lines per wave load, lock-step over the traces of 64 rays
static void wave_lines(Trace** tr, int n, double* lines_sum, long* loads, double* lanes_sum) {
    int maxn = 0; for (int l = 0; l < n; l++) if (tr[l]->n > maxn) maxn = tr[l]->n;
    for (int k = 0; k < maxn; k++) {
        int ln[64], c = 0;
        for (int l = 0; l < n; l++) if (k < tr[l]->n) ln[c++] = tr[l]->node[k] / 8;      // 8 nodes of 16 bytes per 128-byte line
        qsort(ln, (size_t)c, sizeof(int), cmp_int);
        int d = 0; for (int j = 0; j < c; j++) if (j == 0 || ln[j] != ln[j - 1]) d++;
        *lines_sum += d; *lanes_sum += c; (*loads)++;
    }
}
typedef struct { Ray r; int key; } Sec;
static int cmp_sec(const void* a, const void* b) { return ((const Sec*)a)->key - ((const Sec*)b)->key; }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); int W = atoi(argv[2]), H = atoi(argv[3]); float cam[12];
    if (fread(&n_nodes, 4, 1, f) != 1 || fread(&n_geo, 4, 1, f) != 1) return 1;
    nbox = malloc(sizeof(Box) * (size_t)n_nodes); nprim = malloc(4 * (size_t)n_nodes); nskip = malloc(4 * (size_t)n_nodes); sph = malloc(16 * (size_t)n_geo);
    if (fread(nbox, sizeof(Box), (size_t)n_nodes, f) != (size_t)n_nodes || fread(nprim, 4, (size_t)n_nodes, f) != (size_t)n_nodes ||
        fread(nskip, 4, (size_t)n_nodes, f) != (size_t)n_nodes || fread(sph, 16, (size_t)n_geo, f) != (size_t)n_geo || fread(cam, 4, 12, f) != 12) return 1;
    fclose(f);
    Trace* T = malloc(sizeof(Trace) * 256); Trace* T2 = malloc(sizeof(Trace) * 256); Trace* tp[64];
    double lp = 0, ls = 0, lo8 = 0, loc = 0, np_ = 0, ns_ = 0, no8 = 0, noc = 0; long cp = 0, cs = 0, co8 = 0, coc = 0;
    // four neighbouring 8x8 tiles at a time = one 256-lane workgroup's worth of primary rays (full-resolution pixel spacing)
    for (int ty = 0; ty + 8 <= H; ty += 64) for (int tx = 0; tx + 32 <= W; tx += 256) {
        Sec sec[256]; int nsec = 0;
        for (int tile = 0; tile < 4; tile++) {
            for (int l = 0; l < 64; l++) {
                int x = tx + tile * 8 + (l & 7), y = ty + (l >> 3);
                Ray r; float u = ((float)x + urand()) / (float)(W - 1), v = ((float)y + urand()) / (float)(H - 1), dir[3], len = 0;
                for (int a = 0; a < 3; a++) { r.o[a] = cam[a]; dir[a] = cam[3 + a] + u * cam[6 + a] - v * cam[9 + a] - cam[a]; len += dir[a] * dir[a]; }
                len = sqrtf(len); for (int a = 0; a < 3; a++) r.d[a] = dir[a] / len;
                walk(&r, &T[tile * 64 + l]); tp[l] = &T[tile * 64 + l];
                Trace* t = &T[tile * 64 + l];
                if (t->prim >= 0) {
                    float p[3], n[3], nl = 0;
                    for (int a = 0; a < 3; a++) { p[a] = r.o[a] + t->t * r.d[a]; n[a] = p[a] - sph[t->prim][a]; nl += n[a] * n[a]; }
                    nl = sqrtf(nl);
                    float z = 1 - 2 * urand(), ph = 6.2831853f * urand(), s = sqrtf(fmaxf(0.f, 1 - z * z)), rv[3] = {s * cosf(ph), s * sinf(ph), z}, dl = 0;
                    Sec* q = &sec[nsec];
                    for (int a = 0; a < 3; a++) { q->r.o[a] = p[a]; dir[a] = n[a] / nl + rv[a]; dl += dir[a] * dir[a]; }
                    dl = sqrtf(dl); if (dl < 1e-6f) continue;
                    for (int a = 0; a < 3; a++) q->r.d[a] = dir[a] / dl;
                    nsec++;
                }
            }
            wave_lines(tp, 64, &lp, &cp, &np_);
        }
        // (2) secondary rays in the lanes that traced the primaries: waves of consecutive survivors
        for (int k = 0; k < nsec; k++) walk(&sec[k].r, &T2[k]);
        for (int w0 = 0; w0 < nsec; w0 += 64) { int n = nsec - w0 < 64 ? nsec - w0 : 64; for (int l = 0; l < n; l++) tp[l] = &T2[w0 + l]; wave_lines(tp, n, &ls, &cs, &ns_); }
        // (3) re-binned by direction octant across the workgroup's four waves
        for (int k = 0; k < nsec; k++) sec[k].key = ((sec[k].r.d[0] < 0) | (sec[k].r.d[1] < 0) << 1 | (sec[k].r.d[2] < 0) << 2) * 1024 + k;
        { Sec tmp[256]; memcpy(tmp, sec, sizeof(Sec) * (size_t)nsec); qsort(tmp, (size_t)nsec, sizeof(Sec), cmp_sec);
          for (int k = 0; k < nsec; k++) walk(&tmp[k].r, &T2[k]);
          for (int w0 = 0; w0 < nsec; w0 += 64) { int n = nsec - w0 < 64 ? nsec - w0 : 64; for (int l = 0; l < n; l++) tp[l] = &T2[w0 + l]; wave_lines(tp, n, &lo8, &co8, &no8); } }
        // (4) by octant, then by a 2-D cell of the origin
        for (int k = 0; k < nsec; k++) {
            int oct = (sec[k].r.d[0] < 0) | (sec[k].r.d[1] < 0) << 1 | (sec[k].r.d[2] < 0) << 2;
            int cx = (int)floorf(sec[k].r.o[0] * 0.5f) & 63, cz = (int)floorf(sec[k].r.o[2] * 0.5f) & 63;
            sec[k].key = (oct * 4096 + cx * 64 + cz) * 256 + (k & 255);
        }
        { Sec tmp[256]; memcpy(tmp, sec, sizeof(Sec) * (size_t)nsec); qsort(tmp, (size_t)nsec, sizeof(Sec), cmp_sec);
          for (int k = 0; k < nsec; k++) walk(&tmp[k].r, &T2[k]);
          for (int w0 = 0; w0 < nsec; w0 += 64) { int n = nsec - w0 < 64 ? nsec - w0 : 64; for (int l = 0; l < n; l++) tp[l] = &T2[w0 + l]; wave_lines(tp, n, &loc, &coc, &noc); } }
    }
    // (5) a GLOBAL sort: the first-bounce rays of a whole (coarse) frame sorted by origin cell (2 x 2 world units) and direction octant,
    //     cut into waves of 64 - what a wavefront design with ray queues in HBM would feed the walk
    {
        int gw = W / 8, gh = H / 8, cap = gw * gh, ng = 0;                        // one primary ray per 8x8 pixels
        Sec* all = malloc(sizeof(Sec) * (size_t)cap);
        Trace* t1 = malloc(sizeof(Trace));
        for (int gy = 0; gy < gh; gy++) for (int gx = 0; gx < gw; gx++) {
            Ray r; float u = ((float)(gx * 8) + urand() * 8) / (float)(W - 1), v = ((float)(gy * 8) + urand() * 8) / (float)(H - 1), dir[3], len = 0;
            for (int a = 0; a < 3; a++) { r.o[a] = cam[a]; dir[a] = cam[3 + a] + u * cam[6 + a] - v * cam[9 + a] - cam[a]; len += dir[a] * dir[a]; }
            len = sqrtf(len); for (int a = 0; a < 3; a++) r.d[a] = dir[a] / len;
            walk(&r, t1);
            if (t1->prim < 0) continue;
            float p[3], n[3], nl = 0;
            for (int a = 0; a < 3; a++) { p[a] = r.o[a] + t1->t * r.d[a]; n[a] = p[a] - sph[t1->prim][a]; nl += n[a] * n[a]; }
            nl = sqrtf(nl);
            float z = 1 - 2 * urand(), ph = 6.2831853f * urand(), s = sqrtf(fmaxf(0.f, 1 - z * z)), rv[3] = {s * cosf(ph), s * sinf(ph), z}, dl = 0;
            Sec* q = &all[ng];
            for (int a = 0; a < 3; a++) { q->r.o[a] = p[a]; dir[a] = n[a] / nl + rv[a]; dl += dir[a] * dir[a]; }
            dl = sqrtf(dl); if (dl < 1e-6f) continue;
            for (int a = 0; a < 3; a++) q->r.d[a] = dir[a] / dl;
            int oct = (q->r.d[0] < 0) | (q->r.d[1] < 0) << 1 | (q->r.d[2] < 0) << 2;
            int cx = ((int)floorf(q->r.o[0] * 0.5f) + 512) & 1023, cz = ((int)floorf(q->r.o[2] * 0.5f) + 512) & 1023;
            q->key = (cx * 1024 + cz) * 8 + oct;
            ng++;
        }
        double lu = 0, nu = 0, lso = 0, nso = 0; long cu = 0, cso = 0;
        Trace* TT = malloc(sizeof(Trace) * 64);
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1) qsort(all, (size_t)ng, sizeof(Sec), cmp_sec);
            for (int w0 = 0; w0 + 64 <= ng; w0 += 64 * 7) {                       // every 7th wave is enough for the average
                for (int l = 0; l < 64; l++) { walk(&all[w0 + l].r, &TT[l]); tp[l] = &TT[l]; }
                if (pass == 0) wave_lines(tp, 64, &lu, &cu, &nu); else wave_lines(tp, 64, &lso, &cso, &nso);
            }
        }
        printf("first-bounce rays of the whole frame (%d rays), 64 per wave:\n", ng);
        printf("  in image order (8x8-pixel spacing):                   %.1f lines  (%.1f lanes, %ld loads)\n", lu / cu, nu / cu, cu);
        printf("  globally sorted by origin cell and octant:            %.1f lines  (%.1f lanes, %ld loads)\n", lso / cso, nso / cso, cso);
    }
    printf("distinct 128-byte lines per wave load (and live lanes per load), lock-step model of the binary 16-byte-node walk:\n");
    printf("  primary rays of an 8x8 tile:                         %.1f lines  (%.1f lanes, %ld loads)\n", lp / cp, np_ / cp, cp);
    printf("  secondary rays, lanes as they fall (what runs now):   %.1f lines  (%.1f lanes, %ld loads)\n", ls / cs, ns_ / cs, cs);
    printf("  secondary rays of 4 waves re-binned by octant:        %.1f lines  (%.1f lanes, %ld loads)\n", lo8 / co8, no8 / co8, co8);
    printf("  ... by octant and origin cell:                        %.1f lines  (%.1f lanes, %ld loads)\n", loc / coc, noc / coc, coc);
    return 0;
}
