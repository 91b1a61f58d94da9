// trt_pow.h — trt-math v2 `powf`, the one function the Imager's finalisation needs
// (Color::gamma_correction, utils/image.rs:92-98: `c.powf(1.0 / gamma)`).
//
// The reference calls the platform libm through Rust's std, which no test of the reference pins and which the GPU
// does not have; like sin/cos/acos/cbrt (rt_device.h) this project fixes ONE algorithm and uses it everywhere - host
// tonemap (scene_host.cpp), device tonemap (kernels.hip) and, restated in C, the test suite's CPU checker (its
// m_powf) - so the three produce the same float for every input and the quantised frames are compared byte for byte.
//
// Algorithm: x^y = 2^k * exp(r), y * log(x) = k ln2 + r, evaluated in f64 with + - * / and bit moves only (each
// correctly rounded on x86-64 SSE2 and on gfx950; no contraction: -ffp-contract=off), then rounded once to f32:
//   log:  x = m 2^e with m in [sqrt(1/2), sqrt(2));  log m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^23
//         (|s| <= 0.1716, remainder < 1e-18)
//   exp:  k = round(t / ln2), r = t - k ln2 in two pieces (|r| <= 0.3466), Taylor series to r^13 (remainder < 4e-18)
// The f64 result carries ~1e-14 relative error, so the f32 it rounds to is the correctly rounded power except for
// about one input in 10^6 - the same can be said of glibc's powf, but not that the two err on the same inputs.
// Special cases follow C99 pow (Rust's powf is the C one): y = 0 or x = 1 -> 1; NaN operands; +-0, +-inf; negative x
// with a non-integer y -> NaN (`as u8` then maps it to 0, image.rs:101-111).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TRT_HD __host__ __device__ inline
#else
#define TRT_HD inline
#endif

namespace trt {

TRT_HD uint32_t pw_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
TRT_HD float pw_float(uint32_t u) { return __builtin_bit_cast(float, u); }
TRT_HD double pw_double(uint64_t u) { return __builtin_bit_cast(double, u); }

// x^y for finite x > 0 and finite y
TRT_HD float pw_core(float x, float y) {
    uint32_t ux = pw_bits(x);
    int e = 0;
    if (ux < 0x00800000u) { x = x * 16777216.0f; ux = pw_bits(x); e = -24; }          // subnormal: scale by 2^24 (exact)
    e += (int)(ux >> 23) - 127;
    uint32_t um = (ux & 0x007fffffu) | 0x3f800000u;                                    // mantissa in [1, 2)
    if (um >= 0x3fb504f3u) { um -= 0x00800000u; e += 1; }                              // m >= sqrt(2): halve it
    const double m = (double)pw_float(um);
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 0x1.642c8590b2164p-5;                                                   // 1/23
    p = p * z + 0x1.8618618618618p-5;                                                  // 1/21
    p = p * z + 0x1.af286bca1af28p-5;                                                  // 1/19
    p = p * z + 0x1.e1e1e1e1e1e1ep-5;                                                  // 1/17
    p = p * z + 0x1.1111111111111p-4;                                                  // 1/15
    p = p * z + 0x1.3b13b13b13b14p-4;                                                  // 1/13
    p = p * z + 0x1.745d1745d1746p-4;                                                  // 1/11
    p = p * z + 0x1.c71c71c71c71cp-4;                                                  // 1/9
    p = p * z + 0x1.2492492492492p-3;                                                  // 1/7
    p = p * z + 0x1.999999999999ap-3;                                                  // 1/5
    p = p * z + 0x1.5555555555555p-2;                                                  // 1/3
    const double log_m = (s + s) + (s + s) * (z * p);
    const double log_x = (double)e * 0x1.62e42fefa39efp-1 + log_m;
    const double t = (double)y * log_x;
    if (t > 89.0) return __builtin_inff();                                             // above ln(FLT_MAX) = 88.72
    if (t < -104.0) return 0.0f;                                                       // below ln(2^-150) = -103.97
    const double q = t * 0x1.71547652b82fep+0;                                         // t / ln2
    const int k = (int)(q + (q < 0.0 ? -0.5 : 0.5));                                   // nearest integer (truncating conversion)
    const double kd = (double)k;
    const double r = (t - kd * 0x1.62e42fee00000p-1) - kd * 0x1.a39ef35793c76p-33;     // ln2 in two pieces
    double ex = 0x1.6124613a86d09p-33;                                                 // 1/13!
    ex = ex * r + 0x1.1eed8eff8d898p-29;                                               // 1/12!
    ex = ex * r + 0x1.ae64567f544e4p-26;                                               // 1/11!
    ex = ex * r + 0x1.27e4fb7789f5cp-22;                                               // 1/10!
    ex = ex * r + 0x1.71de3a556c734p-19;                                               // 1/9!
    ex = ex * r + 0x1.a01a01a01a01ap-16;                                               // 1/8!
    ex = ex * r + 0x1.a01a01a01a01ap-13;                                               // 1/7!
    ex = ex * r + 0x1.6c16c16c16c17p-10;                                               // 1/6!
    ex = ex * r + 0x1.1111111111111p-7;                                                // 1/5!
    ex = ex * r + 0x1.5555555555555p-5;                                                // 1/4!
    ex = ex * r + 0x1.5555555555555p-3;                                                // 1/3!
    ex = ex * r + 0.5;
    ex = ex * r + 1.0;
    ex = ex * r + 1.0;
    const double two_k = pw_double((uint64_t)(k + 1023) << 52);                        // k in [-151, 129]: a normal double
    return (float)(ex * two_k);                                                        // the one rounding to f32
}

TRT_HD float tm_powf(float x, float y) {
    const uint32_t ux = pw_bits(x), uy = pw_bits(y);
    const uint32_t ax = ux & 0x7fffffffu, ay = uy & 0x7fffffffu;
    if (ay == 0u || ux == 0x3f800000u) return 1.0f;                                    // pow(x, +-0) = pow(1, y) = 1, NaN included
    if (ax > 0x7f800000u || ay > 0x7f800000u) return x + y;                            // NaN
    if (uy == 0x3f800000u) return x;                                                   // pow(x, 1) = x
    const bool y_neg = (uy >> 31) != 0u;
    if (ay == 0x7f800000u) {                                                           // y = +-inf
        if (ax == 0x3f800000u) return 1.0f;                                            // pow(-1, +-inf) = 1
        return ((ax > 0x3f800000u) != y_neg) ? __builtin_inff() : 0.0f;
    }
    // y: integer? odd?
    bool y_int = false, y_odd = false;
    if (ay >= 0x4b800000u) y_int = true;                                               // |y| >= 2^24: an even integer
    else if (ay >= 0x3f800000u) {
        const int yi = (int)y;
        y_int = (float)yi == y;
        y_odd = y_int && (yi & 1) != 0;
    }
    const bool x_neg = (ux >> 31) != 0u;
    const bool negate = x_neg && y_odd;
    float mag;
    if (ax == 0u) mag = y_neg ? __builtin_inff() : 0.0f;                               // pow(+-0, y)
    else if (ax == 0x7f800000u) mag = y_neg ? 0.0f : __builtin_inff();                 // pow(+-inf, y)
    else if (x_neg && !y_int) return __builtin_nanf("");                               // negative base, fractional power
    else mag = pw_core(pw_float(ax), y);
    return negate ? -mag : mag;
}

// Color::gamma_correction + From<Color> for Rgb<u8>, one channel (utils/image.rs:92-111): c^(1/gamma), f32::clamp to
// [0, 0.999] (NaN stays NaN), * 255, `as u8` (truncates, saturates, NaN -> 0).  inv_gamma = 1.0f / gamma, rounded once
// as in the reference.
TRT_HD uint8_t tm_quantise_channel(float c, float inv_gamma) {
    float g = tm_powf(c, inv_gamma);
    if (g < 0.000f) g = 0.000f;
    if (g > 0.999f) g = 0.999f;
    const float s = g * 255.0f;
    if (!(s == s) || s <= 0.0f) return 0u;
    return s >= 255.0f ? (uint8_t)255u : (uint8_t)s;
}

}  // namespace trt
