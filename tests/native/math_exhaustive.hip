// Test program (tests/test_gpu_fuzz.py builds and runs it on the GPU box): the device's trt-math v2 (rt_device.h dm_sincos / dm_acos /
// dm_cbrt) against the CPU checker's statement of the same functions (liboracle: orc_sinf / orc_cosf / orc_acosf / orc_cbrtf) on EVERY
// input the path can produce.  random::<f32>() has 2^23 values u = k / 2^23; vec3extend.rs:15-30 feeds theta = 2 pi u1, phi =
// acos(1 - 2 u2) and r = cbrt(u3) - so 2^23 arguments per function, plus sin / cos of the 2^23 possible phi.  Seven floats per k, compared
// bit for bit; also the composed random_in_unit_sphere on 2^22 generator states.  Exit status 0 and "0 mismatches" on success.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rt_device.h"
extern "C" {
#include "rt_oracle.h"
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void eval_all(float* out, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float u = __uint_as_float(0x3f800000u | k) - 1.0f;                 // rng_random's mapping, all 2^23 values
    const float theta = (2.0f * 3.14159265358979323846f) * u;
    const float phi = trt::dm_acos(1.0f - 2.0f * u);
    float st, ct, sp, cp;
    trt::dm_sincos(theta, st, ct);
    trt::dm_sincos(phi, sp, cp);
    float* o = out + 7ull * k;
    o[0] = st; o[1] = ct; o[2] = phi; o[3] = sp; o[4] = cp; o[5] = trt::dm_cbrt(u); o[6] = theta;
}

__global__ void eval_ball(float* out, uint32_t n) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    trt::Rng g = trt::rng_seed(trt::mix32(7u + 0x9E3779B9u), k, k >> 7);
    const trt::V3 p = trt::random_in_unit_sphere(g);
    const trt::V3 q = trt::random_unit_vector(g);
    float* o = out + 6ull * k;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z;
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    const uint32_t n = 1u << 23, nb = 1u << 22;
    float* d; CHECK(hipMalloc(&d, 7ull * n * sizeof(float)));
    eval_all<<<(n + 255) / 256, 256>>>(d, n);
    std::vector<float> h(7ull * n);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    unsigned long long bad = 0;
    for (uint32_t k = 0; k < n; k++) {
        union { uint32_t u; float f; } c; c.u = 0x3f800000u | k;
        const float u = c.f - 1.0f;
        const float theta = (2.0f * 3.14159265358979323846f) * u;
        const float phi = orc_acosf(1.0f - 2.0f * u);
        const float want[7] = {orc_sinf(theta), orc_cosf(theta), phi, orc_sinf(phi), orc_cosf(phi), orc_cbrtf(u), theta};
        for (int j = 0; j < 7; j++)
            if (bits(want[j]) != bits(h[7ull * k + j])) { if (bad++ < 5) printf("k %u value %d: device %a oracle %a\n", k, j, h[7ull * k + j], want[j]); }
    }
    eval_ball<<<(nb + 255) / 256, 256>>>(d, nb);
    CHECK(hipMemcpy(h.data(), d, 6ull * nb * sizeof(float), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < nb; k++) {
        uint32_t st[2];
        orc_rng_seed(7u, k, k >> 7, st);
        const orc_vec3 p = orc_random_in_unit_sphere(st), q = orc_random_unit_vector(st);
        const float want[6] = {p.x, p.y, p.z, q.x, q.y, q.z};
        for (int j = 0; j < 6; j++)
            if (bits(want[j]) != bits(h[6ull * k + j])) { if (bad++ < 10) printf("state %u value %d: device %a oracle %a\n", k, j, h[6ull * k + j], want[j]); }
    }
    printf("%u inputs x 7 values + %u generator states x 6 values: %llu mismatches\n", n, nb, bad);
    return bad ? 1 : 0;
}
