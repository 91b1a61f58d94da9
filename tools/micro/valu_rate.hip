// Microbenchmark: f32 VALU issue rate per SIMD on gfx950, with full and partial EXEC masks.
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_rate tools/micro/valu_rate.hip && ./build/valu_rate
// Answers two questions the traversal design depends on: how many cycles a wave64 f32 instruction occupies a SIMD,
// and whether a wave whose EXEC mask has whole 16-lane quarters empty issues faster (it does not on GCN; measured here).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, int iters, unsigned long long mask) {
    const unsigned lane = threadIdx.x & 63u;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-7f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (KIND == 0) {            // independent FMAs
                    a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                    a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
                } else if (KIND == 1) {     // min/max/mul/add mix, like the slab test
                    a0 = fminf(a0 * m, a1); a1 = fmaxf(a1 + c, a2); a2 = fminf(a2 * m, a3); a3 = fmaxf(a3 + c, a4);
                    a4 = fminf(a4 * m, a5); a5 = fmaxf(a5 + c, a6); a6 = fminf(a6 * m, a7); a7 = fmaxf(a7 + c, a0);
                } else if (KIND == 3) {     // packed f32: 8 v_pk_mul_f32 + 8 v_pk_add_f32 worth of work counted as 8 instructions of 2 results
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
                    const f2 mm = {m, m}, cc = {c, c};
                    p0 = p0 * mm; p1 = p1 + cc; p2 = p2 * mm; p3 = p3 + cc;
                    p0 = p0 + cc; p1 = p1 * mm; p2 = p2 + cc; p3 = p3 * mm;
                    a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
                } else if (KIND == 4) {     // the same arithmetic unpacked: 16 instructions
                    a0 = a0 * m; a1 = a1 * m; a2 = a2 + c; a3 = a3 + c; a4 = a4 * m; a5 = a5 * m; a6 = a6 + c; a7 = a7 + c;
                    a0 = a0 + c; a1 = a1 + c; a2 = a2 * m; a3 = a3 * m; a4 = a4 + c; a5 = a5 + c; a6 = a6 * m; a7 = a7 * m;
                } else {                    // dependent chain (latency)
                    a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
                    a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, unsigned long long mask, float* d_out, double ghz) {
    const int cus = 256, iters = 20000;
    dim3 grid(cus * waves_per_simd), block(256);          // 4 waves per block -> one per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    spin<KIND><<<grid, block>>>(d_out, 100, mask);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    spin<KIND><<<grid, block>>>(d_out, iters, mask);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)iters * 16 * 8;
    const double per_simd = instr_per_wave * waves_per_simd;           // wave-instructions issued by one SIMD
    const double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-10s waves/SIMD %d  mask %016llx  %.2f ms  -> %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, mask, ms,
           cycles / per_simd);
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    int khz = 0; CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock %.3f GHz\n", ghz);
    const unsigned long long full = ~0ull, half = 0xFFFFFFFFull, quarter = 0xFFFFull, sparse = 0x1111111111111111ull, one = 1ull;
    for (int w : {1, 2, 4, 7}) {
        run<0>("fma", w, full, d_out, ghz);
    }
    run<0>("fma", 4, half, d_out, ghz);
    run<0>("fma", 4, quarter, d_out, ghz);
    run<0>("fma", 4, sparse, d_out, ghz);
    run<0>("fma", 4, one, d_out, ghz);
    run<1>("minmax", 4, full, d_out, ghz);
    run<1>("minmax", 4, quarter, d_out, ghz);
    run<3>("pk 8/trip", 4, full, d_out, ghz);           // printed per 8 instructions: x1 if a pk op costs one slot
    run<4>("unpk16/trip", 4, full, d_out, ghz);          // printed per 8: expect 2x the single-op figure
    run<2>("chain", 1, full, d_out, ghz);
    run<2>("chain", 4, full, d_out, ghz);
    return 0;
}
