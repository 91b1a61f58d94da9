// Microbenchmark: how much scalar work a loop of vector work can carry on gfx950 before the scalar unit is the limit.
//   hipcc --offload-arch=gfx950 -O3 -o build/salu_rate tools/micro/salu_rate.hip && ./build/salu_rate
// A trip = 32 independent v_fma_f32 + M scalar instructions spread between them; W waves per SIMD.  The walk loops of rt_path.h
// run 30 VALU + ~22 SALU per trip (loop conditions, exec-mask bookkeeping of the structuriser, the stragglers check): is that free?
// KIND 0: s_add_u32 on private SGPRs; 1: s_or_b64 exec, exec, exec (an exec write in front of VALU work); 2: s_cmp + untaken s_cbranch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND, int PER4>        // PER4 scalar instructions after every 4 vector instructions: M = 8 * PER4 per trip
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-7f;
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // ONE asm statement per group (the compiler pads separate statements with s_nop, which would be counted as issue slots)
#define V4 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
#define SA "s_add_u32 %4, %4, 1\n"
#define SB "s_add_u32 %5, %5, 1\n"
#define SC "s_add_u32 %6, %6, 1\n"
#define SD "s_add_u32 %7, %7, 1\n"
#define EX "s_or_b64 exec, exec, exec\n"
#define BR(n) "s_cmp_eq_u32 %4, -1\n s_cbranch_scc1 " #n "f\n" #n ":\n"
#define OPS : "+v"(k & 1 ? a4 : a0), "+v"(k & 1 ? a5 : a1), "+v"(k & 1 ? a6 : a2), "+v"(k & 1 ? a7 : a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(m), "v"(c) : "scc"
            if (KIND == 0 && PER4 == 0) asm volatile(V4 OPS);
            if (KIND == 0 && PER4 == 1) asm volatile(V4 SA OPS);
            if (KIND == 0 && PER4 == 2) asm volatile(V4 SA SB OPS);
            if (KIND == 0 && PER4 == 3) asm volatile(V4 SA SB SC OPS);
            if (KIND == 0 && PER4 == 4) asm volatile(V4 SA SB SC SD OPS);
            if (KIND == 1 && PER4 == 1) asm volatile(V4 EX OPS);
            if (KIND == 1 && PER4 == 2) asm volatile(V4 EX EX OPS);
            if (KIND == 1 && PER4 == 3) asm volatile(V4 EX EX EX OPS);
            if (KIND == 2 && PER4 == 1) asm volatile(V4 BR(1) OPS);
            if (KIND == 2 && PER4 == 2) asm volatile(V4 BR(1) BR(2) OPS);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
}

template <int KIND, int PER4>
static void run(int waves_per_simd, float* d_out, double ghz) {
    const int cus = 256, iters = 20000;
    dim3 grid(cus * waves_per_simd), block(256);          // 4 waves per block -> one per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    spin<KIND, PER4><<<grid, block>>>(d_out, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    spin<KIND, PER4><<<grid, block>>>(d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double trips_per_simd = (double)iters * waves_per_simd;
    const double cycles = ms * 1e-3 * ghz * 1e9;
    const int scalar = KIND == 2 ? 8 * PER4 * 2 : 8 * PER4;
    printf("kind %d  waves/SIMD %d  32 VALU + %2d scalar per trip: %7.2f ms  %6.1f cycles per trip per SIMD  (%.2f per VALU)\n", KIND, waves_per_simd, scalar, ms,
           cycles / trips_per_simd, cycles / trips_per_simd / 32.0);
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    int khz = 0; CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock %.3f GHz\n", ghz);
    for (int w : {6, 8}) {
        run<0, 0>(w, d_out, ghz); run<0, 1>(w, d_out, ghz); run<0, 2>(w, d_out, ghz); run<0, 3>(w, d_out, ghz); run<0, 4>(w, d_out, ghz);
        run<1, 1>(w, d_out, ghz); run<1, 2>(w, d_out, ghz); run<1, 3>(w, d_out, ghz);
        run<2, 1>(w, d_out, ghz); run<2, 2>(w, d_out, ghz);
    }
    return 0;
}
