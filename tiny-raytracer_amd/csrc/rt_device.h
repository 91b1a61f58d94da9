// rt_device.h — device-side arithmetic of the path-tracing sampler (gfx950 only).
//
// f32 throughout (reference: `pub type Float = f32`, raytracer/src/lib.rs:4), written in the
// reference's expression order and compiled with -ffp-contract=off: Rust never fuses a*b+c,
// and a sample whose path flips at a hit/miss edge moves a pixel by more than the parity
// tolerance, so every value that feeds a branch has to be the same float the CPU path gets.
// Division and sqrt are hipcc's correctly rounded forms (the default for f32 on ROCm 7.2).
//
// Two pieces have no counterpart that could be matched bit for bit in the reference and are
// specified by this project: "trt-rng v1" (xoroshiro64* streams keyed by seed,
// pixel and sample, standing in for the unseeded rand::thread_rng of utils/random.rs:15-18)
// and "trt-math v2" (fixed polynomial sin/cos/acos/cbrt standing in for the platform libm
// behind vec3extend.rs:21-27).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define TRT_DEV __device__ __forceinline__
// The plain-IEEE fallbacks of the short division / sqrt sequences below run for operands outside [2^-40, 2^40] only - never in a bench
// scene.  They are inlined like everything else; tools/micro/event_costs.hip alone defines TRT_EVENT_COSTS to keep them OUT of line, so
// that its static instruction counts (profiles/isa_event_costs.json: what one lane needs per event) are those of the path really taken.
#ifdef TRT_EVENT_COSTS
#define TRT_COLD __device__ __noinline__
#else
#define TRT_COLD __device__ __forceinline__
#endif

namespace trt {

// ------------------------------------------------------------------------------------------------
// Vec3 (math/vec3.rs:18-164): operators keep the reference's association order.
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };

TRT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
TRT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
TRT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
TRT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
TRT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
TRT_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
TRT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
TRT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
TRT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // vec3.rs:49-51
TRT_DEV float sqlen(V3 a) { return dot(a, a); }                                      // vec3.rs:41-43
TRT_DEV float length(V3 a) { return __builtin_sqrtf(sqlen(a)); }                     // vec3.rs:37-39
// Vec3::normalized (vec3.rs:45-47) is three IEEE divisions by one denominator.  hipcc expands each a / b into v_div_scale x2,
// v_rcp, two FMAs refining the reciprocal, a multiply, four FMAs and v_div_fixup.  Where v_div_scale does not scale - every
// operand comfortably inside the exponent range, so that no residual (2^-24 of its operand) goes subnormal - scale and fixup
// are identities and the refined reciprocal depends on the denominator alone: it is computed once and the three quotients
// take five operations each (24 instead of 33 instructions, one transcendental instead of three).  Same arithmetic, same
// bits: tools/micro/div_exact.hip compares the sequence with `/` on 1.4e11 operand pairs with exponents in [-48, 48], edge
// mantissas included - no mismatch (with exponents up to +-120 it finds 3e-6: the scaling is there for a reason).  Operands
// outside [2^-40, 2^40] (zeros, subnormals, huge values, infinities) take the plain divisions.
TRT_DEV float div_by_refined_rcp(float a, float b, float r1) {
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}
// sqrtf likewise: hipcc scales a tiny argument by 2^32, takes v_sqrt_f32, steps the estimate one ulp down / up with an FMA residual
// each, scales back and passes 0 / inf through by a class test (16 instructions).  For x in [2^-80, 2^81) the scaling and the
// class test are identities; the 9 instructions that remain give sqrtf's bits for EVERY float in that range
// (tools/micro/sqrt_exact.hip, exhaustive: 1.35e9 values, 0 mismatches).
TRT_DEV float sqrt_in_range(float x) {                                                   // requires 2^-80 <= x < 2^81
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}
// Vec3::normalized (vec3.rs:45-47): a / sqrt(a.a).  Components in [2^-39, 2^39] put the squared length into [2^-78, 2^80) and the
// length into [2^-39, 2^40): both short forms apply; anything else (a zero component - the 0/0 of vec3extend.rs:32-34 when u3 = 0
// included -, subnormals, huge values, infinities) takes the plain sqrtf and the three plain divisions.
TRT_COLD V3 normalized_plain(V3 a, float sq) { return a / __builtin_sqrtf(sq); }
TRT_DEV V3 normalized(V3 a) {
    const float lo = 1.8189894035458565e-12f, hi = 549755813888.0f;                     // 2^-39, 2^39
    const float ax = __builtin_fabsf(a.x), ay = __builtin_fabsf(a.y), az = __builtin_fabsf(a.z);
    const float mn = __builtin_fminf(__builtin_fminf(ax, ay), az), mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    const float sq = sqlen(a);
    if (__builtin_expect(mn >= lo && mx <= hi, 1)) {
        const float b = sqrt_in_range(sq);
        const float r0 = __builtin_amdgcn_rcpf(b);
        const float e = __builtin_fmaf(-b, r0, 1.0f);
        const float r1 = __builtin_fmaf(e, r0, r0);
        return v3(div_by_refined_rcp(a.x, b, r1), div_by_refined_rcp(a.y, b, r1), div_by_refined_rcp(a.z, b, r1));
    }
    return normalized_plain(a, sq);
}
TRT_DEV V3 cross(V3 a, V3 b) {                                                       // vec3.rs:53-59
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
TRT_DEV bool near_zero(V3 a) {                                                       // vec3.rs:32-35
    const float eps = 1e-7f;
    return __builtin_fabsf(a.x) < eps && __builtin_fabsf(a.y) < eps && __builtin_fabsf(a.z) < eps;
}
// vec3extend.rs:75-77: v - (2*dot(v,n)) * n
TRT_DEV V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }
// vec3extend.rs:79-84
TRT_DEV V3 refract(V3 v, V3 n, float eta) {
    float c = __builtin_fminf(-dot(n, v), 1.0f);
    V3 perp = eta * (v + n * c);
    V3 parallel = -__builtin_sqrtf(__builtin_fabsf(1.0f - sqlen(perp))) * n;
    return parallel + perp;
}

// ------------------------------------------------------------------------------------------------
// trt-math v2: fma Horner chains over v1's reductions and polynomials, a cube root without
// divisions.  The operation order is the specification - the CPU checker states the same sequence - and
// -ffp-contract=off means only the fma written here fuses.
// ------------------------------------------------------------------------------------------------
// sin and cos of x, |x| < 8192: octant reduction by a three-term split of pi/4, then
// degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].
TRT_DEV void dm_sincos(float x, float& sn, float& cs) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    float ax = __builtin_fabsf(x);
    if (!(ax < 8192.0f)) { sn = __builtin_nanf(""); cs = sn; return; }
    uint32_t j = (uint32_t)(ax * FOPI);
    j = (j + 1u) & ~1u;
    float y = (float)j;
    float r = __builtin_fmaf(-y, DP3, __builtin_fmaf(-y, DP2, __builtin_fmaf(-y, DP1, ax)));
    float z = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                              __builtin_fmaf(-0.5f, z, 1.0f));
    uint32_t q = (j >> 1) & 3u;
    float s = (q & 1u) ? pc : ps;
    float c = (q & 1u) ? ps : pc;
    s = (q & 2u) ? -s : s;                  // q: 0 (ps,pc) 1 (pc,-ps) 2 (-ps,-pc) 3 (-pc,ps)
    c = ((q + 1u) & 2u) ? -c : c;
    sn = (x < 0.0f) ? -s : s;
    cs = c;
}

TRT_DEV float dm_asin_poly(float z) {
    return __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z,
                                         7.4953002686e-2f), z, 1.6666752422e-1f);
}

// Branch-free form of the three-range evaluation (|x| <= 0.5: pi/2 - asin(x); x > 0.5: 2 asin(sqrt((1-x)/2));
// x < -0.5: pi - 2 asin(sqrt((1+x)/2))): every lane performs exactly the operations of its own range (1 + x for a
// negative x IS 1 - |x|), selected instead of branched to - the lanes of a wave fall into all three ranges
// (x = 1 - 2u), so branches made the wave issue all three bodies.
TRT_COLD float sqrt_plain(float z) { return __builtin_sqrtf(z); }
TRT_DEV float dm_acos(float x) {
    const float PI_F = 3.14159265358979323846f, PIO2_F = 1.57079632679489661923f;
    const float ax = __builtin_fabsf(x);
    const bool big = ax > 0.5f;
    const float z = big ? 0.5f * (1.0f - ax) : x * x;
    // z <= 0.25; it is 0 for x = +-1 and tiny next to it: the short sqrt applies from 2^-80 up
    const float w = big ? (__builtin_expect(z >= 8.271806125530277e-25f, 1) ? sqrt_in_range(z) : sqrt_plain(z)) : x;
    const float r = __builtin_fmaf(dm_asin_poly(z) * z, w, w);
    const float two_r = r + r;
    return big ? (x > 0.0f ? two_r : PI_F - two_r) : PIO2_F - r;
}

// cube root without a division: bit guess of r ~ a^(-1/3), two Newton steps on r, y = a r^2, one Newton step on y with
// the residual y^3 - a formed to working precision (<= 0.52 ulp, exact on perfect cubes).
TRT_DEV float dm_cbrt(float x) {
    const float THIRD = 0.333333343267440796f, FOUR_THIRDS = 1.33333337306976318f;
    uint32_t ux = __float_as_uint(x);
    uint32_t sign = ux & 0x80000000u;
    uint32_t ua = ux & 0x7fffffffu;
    if (ua == 0u || ua >= 0x7f800000u) return x;
    float a = __uint_as_float(ua);
    float scale = 1.0f;
    if (ua < 0x00800000u) { a = a * 16777216.0f; scale = 0.00390625f; ua = __float_as_uint(a); }
    float r = __uint_as_float(0x54a21d2au - ua / 3u);
    const float a3 = a * THIRD;
    r = r * __builtin_fmaf(-a3, r * r * r, FOUR_THIRDS);
    r = r * __builtin_fmaf(-a3, r * r * r, FOUR_THIRDS);
    const float r2 = r * r;
    float y = a * r2;
    const float hi = y * y, lo = __builtin_fmaf(y, y, -hi);
    const float e = __builtin_fmaf(hi, y, -a) + lo * y;
    y = __builtin_fmaf(-THIRD * e, r2, y);
    y = y * scale;
    return __uint_as_float(__float_as_uint(y) | sign);
}

// ------------------------------------------------------------------------------------------------
// trt-rng v1: one xoroshiro64* stream per (seed, pixel, sample), drawn in the reference's order.
// ------------------------------------------------------------------------------------------------
struct Rng { uint32_t s0, s1; };

TRT_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// seed_key = mix32(seed + 0x9E3779B9), the same for every lane of a launch (computed on the host).
TRT_DEV Rng rng_seed(uint32_t seed_key, uint32_t pixel, uint32_t sample) {
    Rng r;
    r.s0 = mix32(sample + mix32(pixel ^ seed_key));
    r.s1 = mix32(pixel + mix32(sample ^ ~seed_key));
    if ((r.s0 | r.s1) == 0u) r.s1 = 0x6C078965u;
    return r;
}
TRT_DEV uint32_t rng_next(Rng& g) {
    uint32_t s0 = g.s0, s1 = g.s1;
    uint32_t r = s0 * 0x9E3779BBu;
    s1 ^= s0;
    g.s0 = __builtin_rotateleft32(s0, 26) ^ s1 ^ (s1 << 9);
    g.s1 = __builtin_rotateleft32(s1, 13);
    return r;
}
// random::<f32>() (utils/random.rs:11-13): 23 mantissa bits into [1,2), minus 1.
TRT_DEV float rng_random(Rng& g) { return __uint_as_float(0x3f800000u | (rng_next(g) >> 9)) - 1.0f; }
// random_range(lo..hi) (utils/random.rs:15-18)
TRT_DEV float rng_range(Rng& g, float lo, float hi) {
    float v01 = __uint_as_float(0x3f800000u | (rng_next(g) >> 9)) - 1.0f;
    return v01 * (hi - lo) + lo;
}

// vec3extend.rs:15-30
TRT_DEV V3 random_in_unit_sphere(Rng& g) {
    float u1 = rng_random(g);
    float u2 = rng_random(g);
    float u3 = rng_random(g);
    float theta = (2.0f * 3.14159265358979323846f) * u1;
    float phi = dm_acos(1.0f - 2.0f * u2);
    float r = dm_cbrt(u3);
    float sin_phi, cos_phi, sin_theta, cos_theta;
    dm_sincos(phi, sin_phi, cos_phi);
    dm_sincos(theta, sin_theta, cos_theta);
    float x = r * sin_phi * cos_theta;
    float y = r * sin_phi * sin_theta;
    float z = r * cos_phi;
    return v3(x, y, z);
}
// vec3extend.rs:32-34
TRT_DEV V3 random_unit_vector(Rng& g) { return normalized(random_in_unit_sphere(g)); }
// vec3extend.rs:45-53 (rejection loop; z stays 0)
TRT_DEV void random_in_unit_disk(Rng& g, float& px, float& py) {
    for (;;) {
        px = rng_range(g, -1.0f, 1.0f);
        py = rng_range(g, -1.0f, 1.0f);
        if (px * px + py * py < 1.0f) return;          // squared_length with z = 0
    }
}

// ------------------------------------------------------------------------------------------------
// Ray (ray.rs): direction is normalised on construction (ray.rs:12-14).
// ------------------------------------------------------------------------------------------------
struct Ray { V3 o, d; };
TRT_DEV Ray ray_new(V3 o, V3 d) { return Ray{o, normalized(d)}; }
TRT_DEV V3 ray_at(const Ray& r, float t) { return r.o + t * r.d; }                   // ray.rs:24-26

// ------------------------------------------------------------------------------------------------
// AABB slab test (hittable/aabb.rs:36-61).
//
// slab_exact is the reference's compare-and-assign sequence, NaN behaviour included (a zero
// direction component on a box face gives 0*inf = NaN, which the reference's `<` tests leave
// untouched).  The per-axis early return is dropped: start only grows and end only shrinks, and
// neither can become NaN, so `end <= start` after the third axis decides the same way.
//
// slab_fast needs 1/d and the origin finite (checked once per ray): then no t is NaN, the
// reference's swap is min/max of the two plane distances, and the assignments are max/min.
// ------------------------------------------------------------------------------------------------
TRT_DEV bool slab_exact(float4 na, float4 nb, V3 o, V3 inv, float start, float end) {
    float t0, t1, tmp;
    t0 = (na.x - o.x) * inv.x; t1 = (na.w - o.x) * inv.x;
    if (t1 < t0) { tmp = t0; t0 = t1; t1 = tmp; }
    if (start < t0) start = t0;
    if (t1 < end) end = t1;
    t0 = (na.y - o.y) * inv.y; t1 = (nb.x - o.y) * inv.y;
    if (t1 < t0) { tmp = t0; t0 = t1; t1 = tmp; }
    if (start < t0) start = t0;
    if (t1 < end) end = t1;
    t0 = (na.z - o.z) * inv.z; t1 = (nb.y - o.z) * inv.z;
    if (t1 < t0) { tmp = t0; t0 = t1; t1 = tmp; }
    if (start < t0) start = t0;
    if (t1 < end) end = t1;
    return !(end <= start);
}

TRT_DEV bool slab_fast(float4 na, float4 nb, V3 o, V3 inv, float start, float end) {
    float x0 = (na.x - o.x) * inv.x, x1 = (na.w - o.x) * inv.x;
    float y0 = (na.y - o.y) * inv.y, y1 = (nb.x - o.y) * inv.y;
    float z0 = (na.z - o.z) * inv.z, z1 = (nb.y - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    start = __builtin_fmaxf(start, tn);
    end = __builtin_fminf(end, tf);
    return !(end <= start);
}

// slab_fast that also returns the interval's start, max(start, nearest entry): with it the same box can be re-tested
// against a smaller `end` later by one comparison (pass' = pass && end' > start_out; see rt_path.h walk_fast).
TRT_DEV bool slab_fast_entry(float4 na, float4 nb, V3 o, V3 inv, float start, float end, float& start_out) {
    float x0 = (na.x - o.x) * inv.x, x1 = (na.w - o.x) * inv.x;
    float y0 = (na.y - o.y) * inv.y, y1 = (nb.x - o.y) * inv.y;
    float z0 = (na.z - o.z) * inv.z, z1 = (nb.y - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    start = __builtin_fmaxf(start, tn);
    end = __builtin_fminf(end, tf);
    start_out = start;
    return !(end <= start);
}

// slab_fast on a box given as six floats (compact culling nodes, decoded from f16).
TRT_DEV bool slab_fast6(V3 lo, V3 hi, V3 o, V3 inv, float start, float end) {
    float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
    float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
    float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    start = __builtin_fmaxf(start, tn);
    end = __builtin_fminf(end, tf);
    return !(end <= start);
}

// ... and with the interval's start returned (see slab_fast_entry)
TRT_DEV bool slab_fast6_entry(V3 lo, V3 hi, V3 o, V3 inv, float start, float end, float& start_out) {
    float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
    float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
    float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    start = __builtin_fmaxf(start, tn);
    end = __builtin_fminf(end, tf);
    start_out = start;
    return !(end <= start);
}

TRT_DEV bool finite_f(float v) { return __builtin_fabsf(v) < __builtin_inff(); }

// ------------------------------------------------------------------------------------------------
// Primitive tests.  Range<f32>::contains is half open: t0 <= t < t1.
// ------------------------------------------------------------------------------------------------
// Sphere::hit (hittable/sphere.rs:29-54).  sp = (center.xyz, radius).
TRT_DEV bool sphere_test(float4 sp, const Ray& ray, float t0, float t1, float& t_out) {
    V3 oc = ray.o - v3(sp.x, sp.y, sp.z);
    float a = sqlen(ray.d);
    float half_b = dot(oc, ray.d);
    float c = sqlen(oc) - sp.w * sp.w;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sqrtd = __builtin_sqrtf(disc);
    float t = (-half_b - sqrtd) / a;
    if (!(t0 <= t && t < t1)) {
        t = (-half_b + sqrtd) / a;
        if (!(t0 <= t && t < t1)) return false;
    }
    t_out = t;
    return true;
}

}  // namespace trt
