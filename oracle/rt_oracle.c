/*
 * rt_oracle.c — CPU ORACLE (test infrastructure, never shipped, never on the product path).
 * See rt_oracle.h for scope and parity status.  Every function cites the reference
 * file:line it restates (paths relative to /root/reference/raytracer/src unless noted).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  All arithmetic is
 * f32 (`pub type Float = f32`, lib.rs:4) in the reference's expression order; Rust never
 * contracts a*b+c, hence -ffp-contract=off.
 */
#define _POSIX_C_SOURCE 200809L
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Vec3  (math/vec3.rs:18-164)
 * ---------------------------------------------------------------------------------------- */
typedef orc_vec3 vec3;

static inline vec3 v3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 v3_diag(float v) { return v3(v, v, v); }                         /* vec3.rs:23-25 */
static inline vec3 v3_neg(vec3 a) { return v3(-a.x, -a.y, -a.z); }                  /* :63-68 */
static inline vec3 v3_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :70-75 */
static inline vec3 v3_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :85-90 */
static inline vec3 v3_muls(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }       /* :100-105 */
static inline vec3 v3_smul(float s, vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }       /* :107-112 */
static inline vec3 v3_mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   /* :122-127 */
static inline vec3 v3_divs(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }       /* :137-142 */
static inline vec3 v3_div(vec3 a, vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }   /* :152-157 */
static inline float v3_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    /* :49-51 */
static inline float v3_sqlen(vec3 a) { return v3_dot(a, a); }                       /* :41-43 */
static inline float v3_len(vec3 a) { return sqrtf(v3_sqlen(a)); }                   /* :37-39 */
static inline vec3 v3_normalized(vec3 a) { return v3_divs(a, v3_len(a)); }          /* :45-47 */
static inline vec3 v3_cross(vec3 a, vec3 b) {                                       /* :53-59 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline int v3_near_zero(vec3 a) {                                            /* :32-35 */
    const float eps = 1e-7f;
    return fabsf(a.x) < eps && fabsf(a.y) < eps && fabsf(a.z) < eps;
}
static inline float v3_idx(vec3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }  /* :166-175 */
/* Rust f32::min/max ignore a NaN operand, like C fminf/fmaxf.  vec3extend.rs:59-73 */
static inline vec3 v3_min(vec3 a, vec3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline vec3 v3_max(vec3 a, vec3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

/* vec3extend.rs:75-77: self - 2.0 * self.dot(normal) * normal  ==  v - ((2*dot) * n) */
static inline vec3 v3_reflect(vec3 v, vec3 n) {
    return v3_sub(v, v3_smul(2.0f * v3_dot(v, n), n));
}
/* vec3extend.rs:79-84 */
static inline vec3 v3_refract(vec3 v, vec3 n, float etai_over_etat) {
    float c = fminf(-v3_dot(n, v), 1.0f);
    vec3 perp = v3_smul(etai_over_etat, v3_add(v, v3_muls(n, c)));
    vec3 parallel = v3_smul(-sqrtf(fabsf(1.0f - v3_sqlen(perp))), n);
    return v3_add(parallel, perp);
}

/* tolerant PartialEq, vec3.rs:189-205 */
static int v3_eq(vec3 a, vec3 b) {
    float av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    for (int i = 0; i < 3; i++) {
        if (av[i] == INFINITY && av[i] == bv[i]) { av[i] = 0.0f; bv[i] = 0.0f; }
    }
    return v3_near_zero(v3(av[0] - bv[0], av[1] - bv[1], av[2] - bv[2]));
}

/* ------------------------------------------------------------------------------------------
 * trt-math v2: fixed f32 algorithms standing in for the platform libm the Rust std calls
 * (acos/cbrt/sin/cos at vec3extend.rs:21-27).  Only + - * sqrt, fused multiply-add (fmaf),
 * compares and bit moves, each correctly rounded on x86-64 (SSE + FMA3) and on gfx950, so both
 * sides agree bit for bit.  v2 differs from v1 (rounds 1-3) in evaluating every polynomial as an
 * fma Horner chain and in a cube root without divisions; the polynomials and the argument
 * reductions are v1's.  The build passes -ffp-contract=off: only the fmaf() written here fuses.
 * ---------------------------------------------------------------------------------------- */
static int g_use_libm = 0;
void orc_set_use_libm(int on) { g_use_libm = on; }

/* The build passes -mfma (fmaf as one instruction): refuse to load on a CPU without it instead of dying on an illegal instruction. */
__attribute__((constructor)) static void orc_require_fma(void) {
#if defined(__x86_64__) && defined(__FMA__)
    __builtin_cpu_init();
    if (!__builtin_cpu_supports("fma")) {
        fprintf(stderr, "liboracle: this build uses FMA3 instructions (-mfma) and the CPU has none; rebuild oracle/ with CFLAGS without -mfma\n");
        abort();
    }
#endif
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* sin and cos of x, |x| < 8192: octant reduction by a 3-term split of pi/4, degree-7/8
 * minimax polynomials on [-pi/4, pi/4]. */
static void m_sincos(float x, float *sn, float *cs) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    float ax = fabsf(x);
    if (!(ax < 8192.0f)) { *sn = NAN; *cs = NAN; return; }
    uint32_t j = (uint32_t)(ax * FOPI);
    j = (j + 1u) & ~1u;
    float y = (float)j;
    float r = fmaf(-y, DP3, fmaf(-y, DP2, fmaf(-y, DP1, ax)));
    float z = r * r;
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                    fmaf(-0.5f, z, 1.0f));
    float s, c;
    switch ((j >> 1) & 3u) {
        case 0: s = ps; c = pc; break;
        case 1: s = pc; c = -ps; break;
        case 2: s = -ps; c = -pc; break;
        default: s = -pc; c = ps; break;
    }
    *sn = (x < 0.0f) ? -s : s;
    *cs = c;
}

static float m_asin_poly(float z) {
    return fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                1.6666752422e-1f);
}

static float m_acos(float x) {
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.57079632679489661923f;
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        float s = sqrtf(z);
        float r = fmaf(m_asin_poly(z) * z, s, s);
        return r + r;
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        float s = sqrtf(z);
        float r = fmaf(m_asin_poly(z) * z, s, s);
        return PI_F - (r + r);
    }
    float z = x * x;
    float r = fmaf(m_asin_poly(z) * z, x, x);
    return PIO2_F - r;
}

/* cube root without a division: exponent/3 bit guess of r ~ a^(-1/3), two Newton steps
 * r <- r (4/3 - a/3 r^3), y = a r^2 ~ a^(1/3), then one Newton step on y whose residual y^3 - a is
 * formed to working precision (y^2 as an exact hi + lo pair): at most 0.52 ulp over every f32 in
 * [2^-24, 2) against the f64 cbrt, and exact on the perfect cubes (tests/test_oracle_properties.py). */
static float m_cbrt(float x) {
    const float THIRD = 0.333333343267440796f, FOUR_THIRDS = 1.33333337306976318f;
    uint32_t ux = f2u(x);
    uint32_t sign = ux & 0x80000000u;
    uint32_t ua = ux & 0x7fffffffu;
    if (ua == 0u || ua >= 0x7f800000u) return x;       /* +-0, inf, nan */
    float a = u2f(ua);
    float scale = 1.0f;
    if (ua < 0x00800000u) { a = a * 16777216.0f; scale = 0.00390625f; ua = f2u(a); }  /* 2^24, 2^-8 */
    float r = u2f(0x54a21d2au - ua / 3u);
    float a3 = a * THIRD;
    r = r * fmaf(-a3, r * r * r, FOUR_THIRDS);
    r = r * fmaf(-a3, r * r * r, FOUR_THIRDS);
    float r2 = r * r;
    float y = a * r2;
    float hi = y * y, lo = fmaf(y, y, -hi);
    float e = fmaf(hi, y, -a) + lo * y;
    y = fmaf(-THIRD * e, r2, y);
    y = y * scale;
    return u2f(f2u(y) | sign);
}

float orc_sinf(float x) { if (g_use_libm) return sinf(x); float s, c; m_sincos(x, &s, &c); return s; }
float orc_cosf(float x) { if (g_use_libm) return cosf(x); float s, c; m_sincos(x, &s, &c); return c; }
float orc_acosf(float x) { return g_use_libm ? acosf(x) : m_acos(x); }
float orc_cbrtf(float x) { return g_use_libm ? cbrtf(x) : m_cbrt(x); }

/* ------------------------------------------------------------------------------------------
 * trt-rng v1.  The reference draws from rand::thread_rng() (utils/random.rs:15-18): OS-seeded
 * ChaCha12, unreproducible by construction.  Defined here instead: one xoroshiro64* stream
 * per (seed, pixel, sample), consumed in the reference's draw order.  u32 -> f32 follows
 * rand 0.8.5's UniformFloat::sample_single (23 mantissa bits into [1,2), minus 1, times
 * (hi-lo), plus lo), a dependency pinned in raytracer/Cargo.lock but absent from /root/reference.
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

void orc_rng_seed(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t rng[2]) {
    uint32_t a = mix32(seed + 0x9E3779B9u);
    uint32_t s0 = mix32(sample + mix32(pixel ^ a));
    uint32_t s1 = mix32(pixel + mix32(sample ^ ~a));
    if ((s0 | s1) == 0u) s1 = 0x6C078965u;
    rng[0] = s0; rng[1] = s1;
}
uint32_t orc_rng_next_u32(uint32_t rng[2]) {
    uint32_t s0 = rng[0], s1 = rng[1];
    uint32_t r = s0 * 0x9E3779BBu;
    s1 ^= s0;
    rng[0] = rotl32(s0, 26) ^ s1 ^ (s1 << 9);
    rng[1] = rotl32(s1, 13);
    return r;
}
/* random::<f32>() = random_range(0.0..1.0), utils/random.rs:11-13 */
float orc_rng_random(uint32_t rng[2]) {
    float v12 = u2f(0x3f800000u | (orc_rng_next_u32(rng) >> 9));
    return v12 - 1.0f;
}
/* random_range(lo..hi), utils/random.rs:15-18 */
float orc_rng_random_range(uint32_t rng[2], float lo, float hi) {
    float v01 = u2f(0x3f800000u | (orc_rng_next_u32(rng) >> 9)) - 1.0f;
    return v01 * (hi - lo) + lo;
}

/* vec3extend.rs:15-30 */
orc_vec3 orc_random_in_unit_sphere(uint32_t rng[2]) {
    float u1 = orc_rng_random(rng);
    float u2 = orc_rng_random(rng);
    float u3 = orc_rng_random(rng);
    float theta = (2.0f * 3.14159265358979323846f) * u1;
    float phi = orc_acosf(1.0f - 2.0f * u2);
    float r = orc_cbrtf(u3);
    float sin_phi = orc_sinf(phi);
    float x = r * sin_phi * orc_cosf(theta);
    float y = r * sin_phi * orc_sinf(theta);
    float z = r * orc_cosf(phi);
    return v3(x, y, z);
}
/* vec3extend.rs:32-34 */
orc_vec3 orc_random_unit_vector(uint32_t rng[2]) { return v3_normalized(orc_random_in_unit_sphere(rng)); }
/* vec3extend.rs:45-53 */
orc_vec3 orc_random_in_unit_disk(uint32_t rng[2]) {
    for (;;) {
        float px = orc_rng_random_range(rng, -1.0f, 1.0f);
        float py = orc_rng_random_range(rng, -1.0f, 1.0f);
        vec3 p = v3(px, py, 0.0f);
        if (v3_sqlen(p) < 1.0f) return p;
    }
}

/* ------------------------------------------------------------------------------------------
 * Ray (ray.rs:12-26)
 * ---------------------------------------------------------------------------------------- */
orc_ray orc_ray_new(orc_vec3 origin, orc_vec3 direction) {
    orc_ray r; r.origin = origin; r.direction = v3_normalized(direction); return r;
}
static inline vec3 ray_at(const orc_ray *r, float t) { return v3_add(r->origin, v3_smul(t, r->direction)); }
orc_vec3 orc_ray_at(const orc_ray *r, float t) { return ray_at(r, t); }

/* Range<f32>::contains: start <= t && t < end (half open) */
static inline int range_contains(float t0, float t1, float t) { return t0 <= t && t < t1; }

/* ------------------------------------------------------------------------------------------
 * AABB (hittable/aabb.rs)
 * ---------------------------------------------------------------------------------------- */
static orc_aabb aabb_new(vec3 a, vec3 b) {                                         /* aabb.rs:13-20 */
    const float PADDING_AMOUNT = 0.0001f;
    vec3 padding = v3_diag(PADDING_AMOUNT / 2.0f);
    orc_aabb r;
    r.min = v3_sub(v3_min(a, b), padding);
    r.max = v3_add(v3_max(a, b), padding);
    return r;
}
static orc_aabb aabb_merge(orc_aabb a, orc_aabb b) {                               /* aabb.rs:30-34 */
    orc_aabb r; r.min = v3_min(a.min, b.min); r.max = v3_max(a.max, b.max); return r;
}
static int aabb_intersect(const orc_aabb *bx, const orc_ray *ray, float start, float end) {   /* aabb.rs:36-61 */
    vec3 o = ray->origin, d = ray->direction;
    for (int i = 0; i < 3; i++) {
        float mn = v3_idx(bx->min, i), mx = v3_idx(bx->max, i);
        float inv_d = 1.0f / v3_idx(d, i);
        float t0 = (mn - v3_idx(o, i)) * inv_d;
        float t1 = (mx - v3_idx(o, i)) * inv_d;
        if (t1 < t0) { float tmp = t0; t0 = t1; t1 = tmp; }
        if (start < t0) start = t0;
        if (t1 < end) end = t1;
        if (end <= start) return 0;
    }
    return 1;
}
static int aabb_longest_axis(const orc_aabb *b) {                                   /* aabb.rs:63-78 */
    float s0 = b->max.x - b->min.x, s1 = b->max.y - b->min.y, s2 = b->max.z - b->min.z;
    if (s0 > s1) return (s0 > s2) ? 0 : 2;
    return (s1 > s2) ? 1 : 2;
}
/* f32::total_cmp key (aabb.rs:80-82) */
static inline int32_t total_key(float f) {
    int32_t i = (int32_t)f2u(f);
    i ^= (int32_t)(((uint32_t)(i >> 31)) >> 1);
    return i;
}
int orc_aabb_intersect(const orc_aabb *b, const orc_ray *r, float t0, float t1) { return aabb_intersect(b, r, t0, t1); }

/* ------------------------------------------------------------------------------------------
 * Geometry + HitRecord (hittable/mod.rs, sphere.rs, quad.rs)
 * ---------------------------------------------------------------------------------------- */
enum { GEO_SPHERE = 0, GEO_QUAD = 1 };

typedef struct {
    int kind;
    int material;
    orc_aabb bbox;
    /* sphere.rs:8-13 */
    vec3 center; float radius;
    /* quad.rs:8-17 */
    vec3 corner, u, v, n, w; float d;
} geometry;

/* hittable/mod.rs:28-48 */
static void hit_record_new(orc_hit_record *rec, const orc_ray *ray, float t, vec3 outward_normal, int material) {
    rec->point = ray_at(ray, t);
    rec->front_face = v3_dot(ray->direction, outward_normal) < 0.0f;
    rec->normal = rec->front_face ? v3_normalized(outward_normal) : v3_neg(v3_normalized(outward_normal));
    rec->t = t;
    rec->material = material;
}

static void sphere_init(geometry *g, vec3 center, float radius, int material) {    /* sphere.rs:16-26 */
    memset(g, 0, sizeof *g);
    g->kind = GEO_SPHERE; g->material = material;
    g->center = center; g->radius = radius;
    vec3 rv = v3_diag(radius);
    g->bbox = aabb_new(v3_sub(center, rv), v3_add(center, rv));
}
static int sphere_hit(const geometry *g, const orc_ray *ray, float t0, float t1, orc_hit_record *rec) {  /* sphere.rs:29-54 */
    vec3 oc = v3_sub(ray->origin, g->center);
    float a = v3_sqlen(ray->direction);
    float half_b = v3_dot(oc, ray->direction);
    float c = v3_sqlen(oc) - g->radius * g->radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return 0;
    float sqrtd = sqrtf(discriminant);
    float t = (-half_b - sqrtd) / a;
    if (!range_contains(t0, t1, t)) {
        t = (-half_b + sqrtd) / a;
        if (!range_contains(t0, t1, t)) return 0;
    }
    vec3 p = ray_at(ray, t);
    hit_record_new(rec, ray, t, v3_sub(p, g->center), g->material);
    return 1;
}

static void quad_init(geometry *g, vec3 corner, vec3 u, vec3 v, int material) {     /* quad.rs:20-29 */
    memset(g, 0, sizeof *g);
    g->kind = GEO_QUAD; g->material = material;
    g->corner = corner; g->u = u; g->v = v;
    g->bbox = aabb_merge(aabb_new(corner, v3_add(v3_add(corner, u), v)),
                         aabb_new(v3_add(corner, u), v3_add(corner, v)));
    g->n = v3_cross(u, v);
    g->w = v3_divs(g->n, v3_dot(g->n, g->n));
    g->d = v3_dot(g->n, corner);
}
static int quad_hit(const geometry *g, const orc_ray *ray, float t0, float t1, orc_hit_record *rec, orc_stats *st) {  /* quad.rs:33-54 */
    float dir_norm = v3_dot(ray->direction, g->n);
    float t = (g->d - v3_dot(ray->origin, g->n)) / dir_norm;
    if (range_contains(t0, t1, t)) {
        if (st) st->quad_inside_tests++;
        vec3 p = v3_sub(ray_at(ray, t), g->corner);
        float planar_x = v3_dot(v3_cross(p, g->v), g->w);
        float planar_y = v3_dot(v3_cross(g->u, p), g->w);
        if (range_contains(0.0f, 1.0f, planar_x) && range_contains(0.0f, 1.0f, planar_y)) {
            hit_record_new(rec, ray, t, g->n, g->material);
            return 1;
        }
        return 0;
    }
    return 0;
}
static int geometry_hit(const geometry *g, const orc_ray *ray, float t0, float t1, orc_hit_record *rec, orc_stats *st) {
    if (g->kind == GEO_SPHERE) { if (st) st->sphere_tests++; return sphere_hit(g, ray, t0, t1, rec); }
    if (st) st->quad_plane_tests++;
    return quad_hit(g, ray, t0, t1, rec, st);
}

/* ------------------------------------------------------------------------------------------
 * Materials (material/{lambertian,metal,dielectric,light}.rs)
 * ---------------------------------------------------------------------------------------- */
typedef struct { int kind; vec3 albedo; float param; char *name; } material;

/* dielectric.rs:16-22.  powi(5) = compiler-rt __powisf2 square-and-multiply: x * ((x*x)*(x*x)). */
static float dielectric_reflectance(float cosv, float refraction_index) {
    float one = 1.0f;
    float sqrt_r0 = (one - refraction_index) / (one + refraction_index);
    float r0 = sqrt_r0 * sqrt_r0;
    float x = one - cosv;
    float x2 = x * x;
    float p5 = x * (x2 * x2);
    return r0 + (one - r0) * p5;
}

int orc_material_scatter(int kind, orc_vec3 albedo, float param, const orc_ray *ray, const orc_hit_record *rec,
                         uint32_t rng[2], orc_ray *scattered, orc_vec3 *attenuation) {
    switch (kind) {
    case ORC_LAMBERTIAN: {                                                          /* lambertian.rs:16-22 */
        vec3 dir = v3_add(rec->normal, orc_random_unit_vector(rng));
        if (v3_near_zero(dir)) dir = rec->normal;
        *scattered = orc_ray_new(rec->point, dir);
        *attenuation = albedo;
        return 1;
    }
    case ORC_METAL: {                                                               /* metal.rs:18-25 */
        vec3 reflected = v3_reflect(ray->direction, rec->normal);
        *scattered = orc_ray_new(rec->point, v3_add(reflected, v3_smul(param, orc_random_in_unit_sphere(rng))));
        *attenuation = albedo;
        return 1;
    }
    case ORC_DIELECTRIC: {                                                          /* dielectric.rs:26-46 */
        float refraction_index = rec->front_face ? 1.0f / param : param;
        float cosv = fminf(-v3_dot(rec->normal, ray->direction), 1.0f);
        float sinv = sqrtf(1.0f - cosv * cosv);
        int total_reflection = refraction_index * sinv > 1.0f;
        float reflectance = dielectric_reflectance(cosv, refraction_index);
        vec3 direction;
        if (total_reflection || reflectance > orc_rng_random(rng))                  /* short-circuit: no draw on TIR */
            direction = v3_reflect(ray->direction, rec->normal);
        else
            direction = v3_refract(ray->direction, rec->normal, refraction_index);
        *scattered = orc_ray_new(rec->point, direction);
        *attenuation = albedo;
        return 1;
    }
    default:                                                                        /* light.rs:17-19 */
        return 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * BVH (hittable/bvh.rs)
 * ---------------------------------------------------------------------------------------- */
typedef struct node {
    struct node *left, *right;
    int hittable;                 /* geometry index or -1 */
    orc_aabb bbox;
} node;

struct orc_world {
    geometry *geos; int ngeo, capgeo;
    material *mats; int nmat, capmat;
    int *name_slots; int nslots;      /* open-addressing index over mats[].name (HashMap, world.rs:12) */
    node *root;
};

static void stable_sort_by_axis(const geometry *geos, int *idx, int n, int axis, int *tmp) {
    /* Rust slice::sort_by is a stable merge sort; comparator AABB::compare (aabb.rs:80-82). */
    if (n < 2) return;
    int mid = n / 2;
    stable_sort_by_axis(geos, idx, mid, axis, tmp);
    stable_sort_by_axis(geos, idx + mid, n - mid, axis, tmp);
    int i = 0, j = mid, k = 0;
    while (i < mid && j < n) {
        int32_t ki = total_key(v3_idx(geos[idx[i]].bbox.min, axis));
        int32_t kj = total_key(v3_idx(geos[idx[j]].bbox.min, axis));
        if (kj < ki) tmp[k++] = idx[j++]; else tmp[k++] = idx[i++];
    }
    while (i < mid) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, (size_t)n * sizeof(int));
}

static node *node_new(const geometry *geos, int *objects, int n) {                  /* bvh.rs:42-84 */
    orc_aabb bbox = geos[objects[0]].bbox;
    for (int i = 1; i < n; i++) bbox = aabb_merge(bbox, geos[objects[i]].bbox);
    int axis = aabb_longest_axis(&bbox);
    node *nd = (node *)calloc(1, sizeof(node));
    if (n == 1) {
        nd->hittable = objects[0];
        nd->bbox = geos[objects[0]].bbox;
    } else if (n == 2) {
        nd->hittable = -1;
        nd->left = node_new(geos, objects, 1);
        nd->right = node_new(geos, objects + 1, 1);
        nd->bbox = aabb_merge(nd->left->bbox, nd->right->bbox);
    } else {
        int *sorted = (int *)malloc((size_t)n * sizeof(int));
        int *tmp = (int *)malloc((size_t)n * sizeof(int));
        memcpy(sorted, objects, (size_t)n * sizeof(int));
        stable_sort_by_axis(geos, sorted, n, axis, tmp);
        int mid = n / 2;
        nd->hittable = -1;
        nd->left = node_new(geos, sorted, mid);
        nd->right = node_new(geos, sorted + mid, n - mid);
        nd->bbox = aabb_merge(nd->left->bbox, nd->right->bbox);
        free(tmp); free(sorted);
    }
    return nd;
}
static void node_free(node *n) { if (!n) return; node_free(n->left); node_free(n->right); free(n); }

static int node_hit(const orc_world *w, const node *nd, const orc_ray *ray, float t0, float t1,
                    orc_hit_record *rec, orc_stats *st) {                           /* bvh.rs:88-107 */
    if (st) st->node_tests++;
    if (!aabb_intersect(&nd->bbox, ray, t0, t1)) return 0;
    if (nd->hittable >= 0) return geometry_hit(&w->geos[nd->hittable], ray, t0, t1, rec, st);
    orc_hit_record left_rec;
    if (node_hit(w, nd->left, ray, t0, t1, &left_rec, st)) {
        orc_hit_record right_rec;
        if (node_hit(w, nd->right, ray, t0, left_rec.t, &right_rec, st)) *rec = right_rec;
        else *rec = left_rec;
        return 1;
    }
    return node_hit(w, nd->right, ray, t0, t1, rec, st);
}

/* ------------------------------------------------------------------------------------------
 * World (hittable/world.rs:16-45)
 * ---------------------------------------------------------------------------------------- */
orc_world *orc_world_new(void) { return (orc_world *)calloc(1, sizeof(orc_world)); }
void orc_world_free(orc_world *w) {
    if (!w) return;
    node_free(w->root);
    for (int i = 0; i < w->nmat; i++) free(w->mats[i].name);
    free(w->name_slots); free(w->mats); free(w->geos); free(w);
}
static uint32_t name_hash(const char *s) {
    uint32_t h = 2166136261u;
    for (; *s; s++) { h ^= (uint8_t)*s; h *= 16777619u; }
    return h;
}
static void name_index_insert(orc_world *w, int idx) {
    uint32_t m = (uint32_t)w->nslots - 1u, k = name_hash(w->mats[idx].name) & m;
    while (w->name_slots[k] >= 0) k = (k + 1u) & m;
    w->name_slots[k] = idx;
}
int orc_world_get_material(const orc_world *w, const char *name) {
    if (w->nslots == 0) return -1;
    uint32_t m = (uint32_t)w->nslots - 1u, k = name_hash(name) & m;
    while (w->name_slots[k] >= 0) {
        if (strcmp(w->mats[w->name_slots[k]].name, name) == 0) return w->name_slots[k];
        k = (k + 1u) & m;
    }
    return -1;
}
int orc_world_add_material(orc_world *w, const char *name, int kind, orc_vec3 albedo, float param) {
    if (orc_world_get_material(w, name) >= 0) return -1;                            /* world.rs:29-31 panics */
    if (w->nmat == w->capmat) { w->capmat = w->capmat ? 2 * w->capmat : 8; w->mats = (material *)realloc(w->mats, (size_t)w->capmat * sizeof(material)); }
    material *m = &w->mats[w->nmat];
    m->kind = kind; m->albedo = albedo;
    /* Metal::new clamps fuzz to [0,1], metal.rs:12-14 */
    m->param = (kind == ORC_METAL) ? fminf(fmaxf(param, 0.0f), 1.0f) : param;
    m->name = strdup(name);
    w->nmat++;
    if (2 * w->nmat > w->nslots) {
        w->nslots = w->nslots ? 2 * w->nslots : 16;
        w->name_slots = (int *)realloc(w->name_slots, (size_t)w->nslots * sizeof(int));
        for (int i = 0; i < w->nslots; i++) w->name_slots[i] = -1;
        for (int i = 0; i < w->nmat; i++) name_index_insert(w, i);
    } else {
        name_index_insert(w, w->nmat - 1);
    }
    return w->nmat - 1;
}
static geometry *world_push_geo(orc_world *w) {
    if (w->ngeo == w->capgeo) { w->capgeo = w->capgeo ? 2 * w->capgeo : 64; w->geos = (geometry *)realloc(w->geos, (size_t)w->capgeo * sizeof(geometry)); }
    node_free(w->root); w->root = NULL;
    return &w->geos[w->ngeo++];
}
int orc_world_add_sphere(orc_world *w, orc_vec3 c, float r, int material) { sphere_init(world_push_geo(w), c, r, material); return w->ngeo - 1; }
int orc_world_add_quad(orc_world *w, orc_vec3 corner, orc_vec3 u, orc_vec3 v, int material) { quad_init(world_push_geo(w), corner, u, v, material); return w->ngeo - 1; }
/* n x World::add_geometry(Sphere::new(..)) in array order (world.rs:23-25): what a loop of orc_world_add_sphere does, for scenes of millions */
int orc_world_add_spheres(orc_world *w, int n, const float *xyzr, const int32_t *material) {
    for (int i = 0; i < n; i++) {
        orc_vec3 c = {xyzr[4 * i], xyzr[4 * i + 1], xyzr[4 * i + 2]};
        sphere_init(world_push_geo(w), c, xyzr[4 * i + 3], material[i]);
    }
    return w->ngeo;
}
int orc_world_num_geometries(const orc_world *w) { return w->ngeo; }

void orc_world_build(orc_world *w) {                                                /* world.rs:43-45, bvh.rs:12-22 */
    if (w->root || w->ngeo == 0) return;
    int *objects = (int *)malloc((size_t)w->ngeo * sizeof(int));
    for (int i = 0; i < w->ngeo; i++) objects[i] = i;
    w->root = node_new(w->geos, objects, w->ngeo);
    free(objects);
}

static int dump_rec(const node *nd, float *bbox6, int32_t *prim, int32_t *subtree, int cap, int *pos) {
    int me = (*pos)++;
    int size = 1;
    if (nd->left) size += dump_rec(nd->left, bbox6, prim, subtree, cap, pos);
    if (nd->right) size += dump_rec(nd->right, bbox6, prim, subtree, cap, pos);
    if (me < cap) {
        bbox6[6 * me + 0] = nd->bbox.min.x; bbox6[6 * me + 1] = nd->bbox.min.y; bbox6[6 * me + 2] = nd->bbox.min.z;
        bbox6[6 * me + 3] = nd->bbox.max.x; bbox6[6 * me + 4] = nd->bbox.max.y; bbox6[6 * me + 5] = nd->bbox.max.z;
        prim[me] = nd->hittable; subtree[me] = size;
    }
    return size;
}
int orc_world_bvh_dump(orc_world *w, float *bbox6, int32_t *prim, int32_t *subtree, int cap) {
    orc_world_build(w);
    if (!w->root) return 0;
    int pos = 0;
    dump_rec(w->root, bbox6, prim, subtree, cap, &pos);
    return pos;
}

int orc_world_hit(orc_world *w, const orc_ray *ray, float t0, float t1, orc_hit_record *out, orc_stats *st) {
    orc_world_build(w);
    if (!w->root) return 0;
    if (st) st->rays++;
    return node_hit(w, w->root, ray, t0, t1, out, st);                              /* bvh.rs:25-27 */
}
int orc_world_hit_bruteforce(orc_world *w, const orc_ray *ray, float t0, float t1, orc_hit_record *out) {
    int hit = 0;
    for (int i = 0; i < w->ngeo; i++) {
        orc_hit_record rec;
        if (geometry_hit(&w->geos[i], ray, t0, t1, &rec, NULL)) { *out = rec; t1 = rec.t; hit = 1; }
    }
    return hit;
}

/* ------------------------------------------------------------------------------------------
 * Camera (camera.rs:17-66)
 * ---------------------------------------------------------------------------------------- */
static inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); }   /* f32::to_radians */

void orc_camera_new(orc_camera *cam, float focus_distance, float defocus_angle, orc_vec3 position, orc_vec3 look_at,
                    orc_vec3 up, float vertical_fov, uint32_t width, uint32_t height) {
    float viewport_height = 2.0f * focus_distance * tanf(to_radians(vertical_fov) / 2.0f);
    float aspect_ratio = (float)width / (float)height;
    float viewport_width = aspect_ratio * viewport_height;
    vec3 w = v3_normalized(v3_sub(position, look_at));
    vec3 u = v3_normalized(v3_cross(up, w));
    vec3 v = v3_normalized(v3_cross(w, u));
    vec3 forward = v3_muls(w, focus_distance);
    vec3 horizontal = v3_muls(u, viewport_width);
    vec3 vertical = v3_muls(v, viewport_height);
    vec3 upper_left = v3_sub(v3_add(v3_sub(position, v3_divs(horizontal, 2.0f)), v3_divs(vertical, 2.0f)), forward);
    float defocus_radius = focus_distance * tanf(to_radians(defocus_angle) / 2.0f);
    cam->position = position;
    cam->viewport_upper_left = upper_left;
    cam->forward = forward;
    cam->horizontal = horizontal;
    cam->vertical = vertical;
    cam->defocus_disk_u = v3_muls(u, defocus_radius);
    cam->defocus_disk_v = v3_muls(v, defocus_radius);
    cam->width = width; cam->height = height;
}
static orc_ray camera_get_ray(const orc_camera *cam, float u, float v, uint32_t rng[2]) {       /* camera.rs:58-66 */
    vec3 p = orc_random_in_unit_disk(rng);
    vec3 origin = v3_add(v3_add(cam->position, v3_smul(p.x, cam->defocus_disk_u)), v3_smul(p.y, cam->defocus_disk_v));
    vec3 target = v3_sub(v3_add(cam->viewport_upper_left, v3_smul(u, cam->horizontal)), v3_smul(v, cam->vertical));
    return orc_ray_new(origin, v3_sub(target, origin));
}

/* ------------------------------------------------------------------------------------------
 * CpuSampler::single_point_sampling (renderer/sampler/cpu.rs:39-65)
 * ---------------------------------------------------------------------------------------- */
static vec3 single_point_sampling(orc_world *w, orc_ray ray, uint32_t max_bounces, vec3 background,
                                  uint32_t rng[2], orc_stats *st) {
    uint32_t remain_bounces = max_bounces;
    vec3 color = v3(0.0f, 0.0f, 0.0f);
    vec3 cumulated_attenuation = v3_diag(1.0f);
    st->samples++;
    while (remain_bounces > 0) {
        orc_hit_record rec;
        st->rays++;
        if (node_hit(w, w->root, &ray, 0.001f, INFINITY, &rec, st)) {
            const material *m = &w->mats[rec.material];
            st->shades++;
            vec3 emission = (m->kind == ORC_LIGHT) ? m->albedo : v3(0.0f, 0.0f, 0.0f);      /* light.rs:21-23, mod.rs:8-10 */
            color = v3_add(color, v3_mul(cumulated_attenuation, emission));
            orc_ray new_ray; vec3 attenuation;
            if (orc_material_scatter(m->kind, m->albedo, m->param, &ray, &rec, rng, &new_ray, &attenuation)) {
                cumulated_attenuation = v3_mul(cumulated_attenuation, attenuation);
                ray = new_ray;
                remain_bounces -= 1;
            } else {
                break;
            }
        } else {
            color = v3_add(color, v3_mul(cumulated_attenuation, background));
            break;
        }
    }
    return color;
}

/* ------------------------------------------------------------------------------------------
 * Renderer::render = SamplePointGenerator::generate (pointgen.rs:37-52) → sampler →
 * Imager::collect accumulation (imager.rs:34-60), fused; samples accumulate in s order.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    orc_world *w; const orc_camera *cam; const orc_render_params *p; float *accum;
    atomic_uint next_row; orc_stats stats; pthread_mutex_t mu;
} render_job;

static void render_row(render_job *job, uint32_t y, orc_stats *st) {
    const orc_render_params *p = job->p; const orc_camera *cam = job->cam;
    uint32_t W = cam->width, H = cam->height;
    float color_multiplier = 1.0f / (float)p->spp;                                  /* imager.rs:35 */
    for (uint32_t x = 0; x < W; x++) {
        float *px = job->accum + 3 * ((size_t)y * W + x);
        vec3 acc = p->accumulate ? v3(px[0], px[1], px[2]) : v3(0.0f, 0.0f, 0.0f);
        for (uint32_t s = p->sample_begin; s < p->sample_end; s++) {
            uint32_t rng[2];
            orc_rng_seed(p->seed, y * W + x, s, rng);
            float u = ((float)x + orc_rng_random(rng)) / (float)(W - 1);            /* pointgen.rs:41 */
            float v = ((float)y + orc_rng_random(rng)) / (float)(H - 1);            /* pointgen.rs:42 */
            orc_ray ray = camera_get_ray(cam, u, v, rng);
            vec3 c = single_point_sampling(job->w, ray, p->max_bounces, p->background, rng, st);
            acc = v3_add(acc, v3_muls(c, color_multiplier));                        /* imager.rs:50 */
        }
        px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
    }
}
static void *render_worker(void *arg) {
    render_job *job = (render_job *)arg;
    orc_stats st; memset(&st, 0, sizeof st);
    for (;;) {
        uint32_t y = atomic_fetch_add(&job->next_row, 1u);
        if (y >= job->p->row_end) break;
        render_row(job, y, &st);
    }
    pthread_mutex_lock(&job->mu);
    job->stats.samples += st.samples; job->stats.rays += st.rays; job->stats.node_tests += st.node_tests;
    job->stats.sphere_tests += st.sphere_tests; job->stats.quad_plane_tests += st.quad_plane_tests;
    job->stats.quad_inside_tests += st.quad_inside_tests; job->stats.shades += st.shades;
    pthread_mutex_unlock(&job->mu);
    return NULL;
}
void orc_render(orc_world *w, const orc_camera *cam, const orc_render_params *p, float *accum, orc_stats *stats, int nthreads) {
    orc_world_build(w);
    render_job job;
    job.w = w; job.cam = cam; job.p = p; job.accum = accum;
    atomic_init(&job.next_row, p->row_begin);
    memset(&job.stats, 0, sizeof job.stats);
    pthread_mutex_init(&job.mu, NULL);
    if (w->root) {
        if (nthreads <= 1) {
            render_worker(&job);
        } else {
            pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
            for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, render_worker, &job);
            for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
            free(th);
        }
    }
    pthread_mutex_destroy(&job.mu);
    if (stats) *stats = job.stats;
}

void orc_sample_batch(orc_world *w, const orc_sample_point *in, uint32_t n, orc_sampled_color *out,
                      uint32_t max_bounces, orc_vec3 background, uint32_t seed, orc_stats *stats) {
    orc_world_build(w);
    orc_stats st; memset(&st, 0, sizeof st);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t rng[2];
        orc_rng_seed(seed, i, 0u, rng);
        out[i].x = in[i].x; out[i].y = in[i].y;
        out[i].color = w->root ? single_point_sampling(w, in[i].ray, max_bounces, background, rng, &st)
                               : v3(0.0f, 0.0f, 0.0f);
    }
    if (stats) *stats = st;
}

/* ------------------------------------------------------------------------------------------
 * Imager finalisation + Color (imager.rs:52-53; utils/image.rs:92-111)
 * ---------------------------------------------------------------------------------------- */
/* trt-math v2 powf (the product's tiny-raytracer_amd/csrc/trt_pow.h states the same algorithm): the platform libm behind
 * `powf` (image.rs:94-96) is pinned by nothing, so one algorithm is fixed: x^y = 2^k exp(r) with y log x = k ln2 + r,
 * in f64 with + - * / only, rounded once to f32.  log x: x = m 2^e, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m-1)/(m+1))
 * (odd series to s^23); exp r: Taylor to r^13.  Special cases as C99 pow. */
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static float m_pow_core(float x, float y) {            /* finite x > 0, finite y */
    uint32_t ux = f2u(x);
    int e = 0;
    if (ux < 0x00800000u) { x = x * 16777216.0f; ux = f2u(x); e = -24; }
    e += (int)(ux >> 23) - 127;
    uint32_t um = (ux & 0x007fffffu) | 0x3f800000u;
    if (um >= 0x3fb504f3u) { um -= 0x00800000u; e += 1; }
    const double m = (double)u2f(um);
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    static const double L[11] = {0x1.642c8590b2164p-5, 0x1.8618618618618p-5, 0x1.af286bca1af28p-5, 0x1.e1e1e1e1e1e1ep-5,
                                 0x1.1111111111111p-4, 0x1.3b13b13b13b14p-4, 0x1.745d1745d1746p-4, 0x1.c71c71c71c71cp-4,
                                 0x1.2492492492492p-3, 0x1.999999999999ap-3, 0x1.5555555555555p-2};   /* 1/23 ... 1/3 */
    double p = L[0];
    for (int i = 1; i < 11; i++) p = p * z + L[i];
    const double log_m = (s + s) + (s + s) * (z * p);
    const double log_x = (double)e * 0x1.62e42fefa39efp-1 + log_m;
    const double t = (double)y * log_x;
    if (t > 89.0) return INFINITY;
    if (t < -104.0) return 0.0f;
    const double q = t * 0x1.71547652b82fep+0;
    const int k = (int)(q + (q < 0.0 ? -0.5 : 0.5));
    const double kd = (double)k;
    const double r = (t - kd * 0x1.62e42fee00000p-1) - kd * 0x1.a39ef35793c76p-33;
    static const double E[14] = {0x1.6124613a86d09p-33, 0x1.1eed8eff8d898p-29, 0x1.ae64567f544e4p-26, 0x1.27e4fb7789f5cp-22,
                                 0x1.71de3a556c734p-19, 0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-13, 0x1.6c16c16c16c17p-10,
                                 0x1.1111111111111p-7, 0x1.5555555555555p-5, 0x1.5555555555555p-3, 0.5, 1.0, 1.0};   /* 1/13! ... 1/0! */
    double ex = E[0];
    for (int i = 1; i < 14; i++) ex = ex * r + E[i];
    return (float)(ex * u2d((uint64_t)(k + 1023) << 52));
}
static float m_powf(float x, float y) {
    const uint32_t ux = f2u(x), uy = f2u(y);
    const uint32_t ax = ux & 0x7fffffffu, ay = uy & 0x7fffffffu;
    if (ay == 0u || ux == 0x3f800000u) return 1.0f;
    if (ax > 0x7f800000u || ay > 0x7f800000u) return x + y;
    if (uy == 0x3f800000u) return x;
    const int y_neg = (int)(uy >> 31);
    if (ay == 0x7f800000u) {
        if (ax == 0x3f800000u) return 1.0f;
        return ((ax > 0x3f800000u) != y_neg) ? INFINITY : 0.0f;
    }
    int y_int = 0, y_odd = 0;
    if (ay >= 0x4b800000u) y_int = 1;
    else if (ay >= 0x3f800000u) { int yi = (int)y; y_int = (float)yi == y; y_odd = y_int && (yi & 1); }
    const int x_neg = (int)(ux >> 31);
    float mag;
    if (ax == 0u) mag = y_neg ? INFINITY : 0.0f;
    else if (ax == 0x7f800000u) mag = y_neg ? 0.0f : INFINITY;
    else if (x_neg && !y_int) return NAN;
    else mag = m_pow_core(u2f(ax), y);
    return (x_neg && y_odd) ? -mag : mag;
}
float orc_powf(float x, float y) { return g_use_libm ? powf(x, y) : m_powf(x, y); }

float orc_gamma_correct(float c, float gamma) { return orc_powf(c, 1.0f / gamma); }     /* image.rs:92-98 */
static uint8_t color_to_u8(float c) {                                               /* image.rs:101-111 */
    const float INTENSITY_MIN = 0.000f, INTENSITY_MAX = 0.999f;
    /* f32::clamp: NaN stays NaN; `as u8` saturates and maps NaN to 0 */
    float v = c;
    if (v < INTENSITY_MIN) v = INTENSITY_MIN;
    if (v > INTENSITY_MAX) v = INTENSITY_MAX;
    v = v * 255.0f;
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
void orc_tonemap_u8(const float *accum, uint32_t npixels, float gamma, uint8_t *rgb) {
    for (size_t i = 0; i < (size_t)npixels * 3; i++) rgb[i] = color_to_u8(orc_gamma_correct(accum[i], gamma));
}

/* ------------------------------------------------------------------------------------------
 * unit entry points for known-answer tests
 * ---------------------------------------------------------------------------------------- */
int orc_sphere_hit(orc_vec3 center, float radius, const orc_ray *ray, float t0, float t1, orc_hit_record *out) {
    geometry g; sphere_init(&g, center, radius, 0); return sphere_hit(&g, ray, t0, t1, out);
}
int orc_quad_hit(orc_vec3 corner, orc_vec3 u, orc_vec3 v, const orc_ray *ray, float t0, float t1, orc_hit_record *out) {
    geometry g; quad_init(&g, corner, u, v, 0); return quad_hit(&g, ray, t0, t1, out, NULL);
}
orc_aabb orc_sphere_bbox(orc_vec3 c, float r) { geometry g; sphere_init(&g, c, r, 0); return g.bbox; }
orc_aabb orc_quad_bbox(orc_vec3 corner, orc_vec3 u, orc_vec3 v) { geometry g; quad_init(&g, corner, u, v, 0); return g.bbox; }
orc_vec3 orc_vec3_binop(int op, orc_vec3 a, orc_vec3 b) {
    switch (op) { case 0: return v3_add(a, b); case 1: return v3_sub(a, b); case 2: return v3_mul(a, b);
                  case 3: return v3_div(a, b); default: return v3_cross(a, b); }
}
orc_vec3 orc_vec3_scale(int op, orc_vec3 a, float s) { return op == 0 ? v3_muls(a, s) : v3_divs(a, s); }
float orc_vec3_dot(orc_vec3 a, orc_vec3 b) { return v3_dot(a, b); }
float orc_vec3_length(orc_vec3 a) { return v3_len(a); }
int orc_vec3_eq(orc_vec3 a, orc_vec3 b) { return v3_eq(a, b); }
orc_vec3 orc_vec3_reflect(orc_vec3 v, orc_vec3 n) { return v3_reflect(v, n); }
orc_vec3 orc_vec3_refract(orc_vec3 v, orc_vec3 n, float eta) { return v3_refract(v, n, eta); }
