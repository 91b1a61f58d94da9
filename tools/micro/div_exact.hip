// Is the shared-reciprocal division bit-identical to IEEE `/` on gfx950?  (rt_device.h div_shared)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/div_exact tools/micro/div_exact.hip && ./build/div_exact
// The compiler expands a / b into v_div_scale x2, v_rcp, two FMAs that refine the reciprocal, a multiply, four FMAs
// (v_div_fmas last) and v_div_fixup.  Where v_div_scale does not scale - operands and quotient far from the ends of the
// exponent range - the scale and fixup steps are identities, and what remains is the sequence below; the reciprocal part
// depends on b alone, so the three divisions of Vec3::normalized (vec3.rs:45-47) can share it.  This program compares the two
// bit for bit over random operands (exponents of a and of b in [-RANGE, RANGE], random mantissas plus the edge mantissas) and
// reports every mismatch; it also checks 1 / b.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float div_fast(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}

__device__ __forceinline__ uint32_t next(uint32_t& s0, uint32_t& s1) {
    uint32_t r = s0 * 0x9E3779BBu;
    s1 ^= s0; s0 = __builtin_rotateleft32(s0, 26) ^ s1 ^ (s1 << 9); s1 = __builtin_rotateleft32(s1, 13);
    return r;
}

__device__ __forceinline__ uint32_t mantissa(uint32_t r, uint32_t sel) {
    // mostly random, sometimes an edge pattern: 0, all ones, one bit, all ones but one
    switch (sel & 15u) {
        case 0: return 0u;
        case 1: return 0x7fffffu;
        case 2: return 1u << (r % 23u);
        case 3: return 0x7fffffu ^ (1u << (r % 23u));
        default: return r & 0x7fffffu;
    }
}

#ifndef RANGE
#define RANGE 40          // exponents of a and of b in [-RANGE, RANGE]: every intermediate (residuals are ~2^-24 of a) stays normal
#endif
__global__ void compare(unsigned long long* mismatches, unsigned long long* first_bad, int iters, uint32_t seed) {
    uint32_t s0 = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed, s1 = s0 ^ 0x6C078965u;
    unsigned long long bad = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t r1 = next(s0, s1), r2 = next(s0, s1), r3 = next(s0, s1), r4 = next(s0, s1);
        const int eb = (int)(r3 % (2u * RANGE + 1u)) - RANGE, ea = (int)((r3 >> 8) % (2u * RANGE + 1u)) - RANGE;
        const uint32_t ub = ((uint32_t)(eb + 127) << 23) | mantissa(r1, r4) | (r4 & 0x80000000u);
        const uint32_t ua = ((uint32_t)(ea + 127) << 23) | mantissa(r2, r4 >> 4) | ((r4 << 1) & 0x80000000u);
        const float a = __uint_as_float(ua), b = __uint_as_float(ub);
        const float want = a / b, got = div_fast(a, b);
        const float want1 = 1.0f / b, got1 = div_fast(1.0f, b);
        if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(want1) != __float_as_uint(got1)) {
            if (bad == 0) { first_bad[0] = ((unsigned long long)ua << 32) | ub; }
            bad++;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    unsigned long long *d, *d2, h = 0, h2 = 0;
    CHECK(hipMalloc(&d, 8)); CHECK(hipMalloc(&d2, 8));
    CHECK(hipMemset(d, 0, 8)); CHECK(hipMemset(d2, 0, 8));
    const int blocks = 256 * 32, threads = 256;
    for (uint32_t rep = 0; rep < 8; rep++) {
        compare<<<blocks, threads>>>(d, d2, iters, rep * 7919u + 1u);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&h2, d2, 8, hipMemcpyDeviceToHost));
    const double pairs = 8.0 * blocks * threads * (double)iters;
    printf("%.3g operand pairs (a / b and 1 / b each): %llu mismatches against IEEE division", pairs, h);
    if (h) printf(" (one of them: a = 0x%08x, b = 0x%08x)", (unsigned)(h2 >> 32), (unsigned)h2);
    printf("\n");
    return h ? 1 : 0;
}
