// Microbenchmark: what a select costs on gfx950, by where its mask lives (follow-up of valu_forms.hip, where 32 back-to-back
// v_cndmask_b32_e32 ..., vcc took 23.5 cycles each against 4.4 for the VOP3 form reading an SGPR pair).
//   hipcc --offload-arch=gfx950 -O3 -o build/vcc_select tools/micro/vcc_select.hip && ./build/vcc_select
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int FORM>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3f + k;
    const float m = 1.0000001f, c = 1e-7f;
    asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" : : "v"(m), "v"(a[0]) : "vcc", "s20", "s21");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float &x0 = a[(k & 1) * 4 + 0], &x1 = a[(k & 1) * 4 + 1], &x2 = a[(k & 1) * 4 + 2], &x3 = a[(k & 1) * 4 + 3];
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m), "v"(c) : "vcc", "s20", "s21", "s22", "s23"
            if (FORM == 0) asm volatile("v_cndmask_b32_e32 %0, %4, %0, vcc\n v_cndmask_b32_e32 %1, %4, %1, vcc\n v_cndmask_b32_e32 %2, %4, %2, vcc\n v_cndmask_b32_e32 %3, %4, %3, vcc" OPS);
            if (FORM == 1) asm volatile("v_cndmask_b32_e64 %0, %4, %0, vcc\n v_cndmask_b32_e64 %1, %4, %1, vcc\n v_cndmask_b32_e64 %2, %4, %2, vcc\n v_cndmask_b32_e64 %3, %4, %3, vcc" OPS);
            if (FORM == 2) asm volatile("v_cndmask_b32_e64 %0, %4, %0, s[20:21]\n v_cndmask_b32_e64 %1, %4, %1, s[20:21]\n v_cndmask_b32_e64 %2, %4, %2, s[20:21]\n v_cndmask_b32_e64 %3, %4, %3, s[20:21]" OPS);
            // the compiler's usual pair: compare into vcc, select on vcc
            if (FORM == 3) asm volatile("v_cmp_lt_f32_e32 vcc, %4, %0\n v_cndmask_b32_e32 %0, %4, %0, vcc\n v_cmp_lt_f32_e32 vcc, %4, %1\n v_cndmask_b32_e32 %1, %4, %1, vcc" OPS);
            // compare into an SGPR pair, select on it
            if (FORM == 4) asm volatile("v_cmp_lt_f32_e64 s[20:21], %4, %0\n v_cndmask_b32_e64 %0, %4, %0, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %4, %1\n v_cndmask_b32_e64 %1, %4, %1, s[22:23]" OPS);
            // select on vcc between other work
            if (FORM == 5) asm volatile("v_cndmask_b32_e32 %0, %4, %0, vcc\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_mul_f32_e32 %3, %4, %3" OPS);
            if (FORM == 6) asm volatile("v_cmp_lt_f32_e32 vcc, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_cndmask_b32_e32 %3, %4, %3, vcc" OPS);
            // carry chain through vcc
            if (FORM == 7) asm volatile("v_add_co_u32_e32 %0, vcc, %4, %0\n v_addc_co_u32_e32 %1, vcc, %4, %1, vcc\n v_add_co_u32_e32 %2, vcc, %4, %2\n v_addc_co_u32_e32 %3, vcc, %4, %3, vcc" OPS);
            // the select as arithmetic on a 0/1 mask kept in a VGPR: x = x + mask * (m - x) is NOT exact; a bitwise blend is: (x & ~k) | (m & k)
            if (FORM == 8) asm volatile("v_bfi_b32 %0, %5, %4, %0\n v_bfi_b32 %1, %5, %4, %1\n v_bfi_b32 %2, %5, %4, %2\n v_bfi_b32 %3, %5, %4, %3" OPS);
        }
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM>
static void run(const char* name, int waves_per_simd, float* d_out, double ghz) {
    const int cus = 256, iters = 20000;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    spin<FORM><<<grid, block>>>(d_out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    spin<FORM><<<grid, block>>>(d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-64s waves/SIMD %d  %6.2f ms  %6.2f cycles per group of 4 per SIMD\n", name, waves_per_simd, ms, cycles / ((double)iters * waves_per_simd * 8.0));
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    int khz = 0; CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock %.3f GHz\n", ghz);
    for (int w : {1, 6}) {
        run<0>("4 x v_cndmask_b32_e32 .., vcc", w, d_out, ghz);
        run<1>("4 x v_cndmask_b32_e64 .., vcc", w, d_out, ghz);
        run<2>("4 x v_cndmask_b32_e64 .., s[20:21]", w, d_out, ghz);
        run<3>("2 x (v_cmp_e32 vcc ; v_cndmask_e32 vcc)", w, d_out, ghz);
        run<4>("2 x (v_cmp_e64 s[n:n+1] ; v_cndmask_e64 s[n:n+1])", w, d_out, ghz);
        run<5>("v_cndmask_e32 vcc ; 3 x v_mul", w, d_out, ghz);
        run<6>("v_cmp_e32 vcc ; 2 x v_mul ; v_cndmask_e32 vcc", w, d_out, ghz);
        run<7>("2 x (v_add_co vcc ; v_addc_co vcc)", w, d_out, ghz);
        run<8>("4 x v_bfi_b32 (bitwise blend on a VGPR mask)", w, d_out, ghz);
    }
    return 0;
}
