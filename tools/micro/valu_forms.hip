// Microbenchmark: issue cost of one wave64 VALU instruction on gfx950 BY INSTRUCTION FORM (encoding, number of VGPR sources).
//   hipcc --offload-arch=gfx950 -O3 -o build/valu_forms tools/micro/valu_forms.hip && ./build/valu_forms
// tools/micro/valu_rate.hip let the compiler choose the instructions (it packed the FMAs into v_pk_fma_f32, two per instruction); here a
// trip is 32 copies of ONE hand-written instruction on 8 independent accumulators (4 per asm statement), W waves per SIMD.
// (generated table: see the list at the top of main)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int FORM>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3f + k;
    const float m = 1.0000001f, c = 1e-7f;
    float sm;
    asm volatile("s_mov_b32 %0, 0x3f800001" : "=s"(sm));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4]; for (int k = 0; k < 4; ++k) p[k] = f2{a[2 * k], a[2 * k + 1]};
    const f2 pm = {m, m}, pc = {c, c};
    asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" : : "v"(m), "v"(a[0]) : "vcc", "s20", "s21");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float &x0 = a[(k & 1) * 4 + 0], &x1 = a[(k & 1) * 4 + 1], &x2 = a[(k & 1) * 4 + 2], &x3 = a[(k & 1) * 4 + 3];
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m), "v"(c), "s"(sm) : "vcc", "s20", "s21"
            if (FORM == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" OPS);
            if (FORM == 1) asm volatile("v_fma_f32 %0, %0, %6, %5\n v_fma_f32 %1, %1, %6, %5\n v_fma_f32 %2, %2, %6, %5\n v_fma_f32 %3, %3, %6, %5" OPS);
            if (FORM == 2) asm volatile("v_fmac_f32_e32 %0, %4, %5\n v_fmac_f32_e32 %1, %4, %5\n v_fmac_f32_e32 %2, %4, %5\n v_fmac_f32_e32 %3, %4, %5" OPS);
            if (FORM == 3) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_mul_f32_e32 %3, %4, %3" OPS);
            if (FORM == 4) asm volatile("v_mul_f32_e32 %0, %6, %0\n v_mul_f32_e32 %1, %6, %1\n v_mul_f32_e32 %2, %6, %2\n v_mul_f32_e32 %3, %6, %3" OPS);
            if (FORM == 5) asm volatile("v_mul_f32_e32 %0, 0x3f800001, %0\n v_mul_f32_e32 %1, 0x3f800001, %1\n v_mul_f32_e32 %2, 0x3f800001, %2\n v_mul_f32_e32 %3, 0x3f800001, %3" OPS);
            if (FORM == 6) asm volatile("v_mul_f32_e32 %0, 2.0, %0\n v_mul_f32_e32 %1, 2.0, %1\n v_mul_f32_e32 %2, 2.0, %2\n v_mul_f32_e32 %3, 2.0, %3" OPS);
            if (FORM == 7) asm volatile("v_add_f32_e32 %0, %4, %0\n v_add_f32_e32 %1, %4, %1\n v_add_f32_e32 %2, %4, %2\n v_add_f32_e32 %3, %4, %3" OPS);
            if (FORM == 8) asm volatile("v_sub_f32_e32 %0, %0, %4\n v_sub_f32_e32 %1, %1, %4\n v_sub_f32_e32 %2, %2, %4\n v_sub_f32_e32 %3, %3, %4" OPS);
            if (FORM == 9) asm volatile("v_sub_f32_e32 %0, %6, %0\n v_sub_f32_e32 %1, %6, %1\n v_sub_f32_e32 %2, %6, %2\n v_sub_f32_e32 %3, %6, %3" OPS);
            if (FORM == 10) asm volatile("v_mov_b32_e32 %0, %4\n v_mov_b32_e32 %1, %4\n v_mov_b32_e32 %2, %4\n v_mov_b32_e32 %3, %4" OPS);
            if (FORM == 11) asm volatile("v_mov_b32_e32 %0, %6\n v_mov_b32_e32 %1, %6\n v_mov_b32_e32 %2, %6\n v_mov_b32_e32 %3, %6" OPS);
            if (FORM == 12) asm volatile("v_max_f32_e32 %0, %4, %0\n v_max_f32_e32 %1, %4, %1\n v_max_f32_e32 %2, %4, %2\n v_max_f32_e32 %3, %4, %3" OPS);
            if (FORM == 13) asm volatile("v_min_f32_e32 %0, %4, %0\n v_min_f32_e32 %1, %4, %1\n v_min_f32_e32 %2, %4, %2\n v_min_f32_e32 %3, %4, %3" OPS);
            if (FORM == 14) asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" OPS);
            if (FORM == 15) asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" OPS);
            if (FORM == 16) asm volatile("v_cndmask_b32_e32 %0, %4, %0, vcc\n v_cndmask_b32_e32 %1, %4, %1, vcc\n v_cndmask_b32_e32 %2, %4, %2, vcc\n v_cndmask_b32_e32 %3, %4, %3, vcc" OPS);
            if (FORM == 17) asm volatile("v_cndmask_b32_e64 %0, %4, %0, s[20:21]\n v_cndmask_b32_e64 %1, %4, %1, s[20:21]\n v_cndmask_b32_e64 %2, %4, %2, s[20:21]\n v_cndmask_b32_e64 %3, %4, %3, s[20:21]" OPS);
            if (FORM == 18) asm volatile("v_cmp_lt_f32_e32 vcc, %4, %0\n v_cmp_lt_f32_e32 vcc, %4, %1\n v_cmp_lt_f32_e32 vcc, %4, %2\n v_cmp_lt_f32_e32 vcc, %4, %3" OPS);
            if (FORM == 19) asm volatile("v_cmp_lt_f32_e64 s[20:21], %4, %0\n v_cmp_lt_f32_e64 s[20:21], %4, %1\n v_cmp_lt_f32_e64 s[20:21], %4, %2\n v_cmp_lt_f32_e64 s[20:21], %4, %3" OPS);
            if (FORM == 20) asm volatile("v_cmp_lt_u32_e32 vcc, %4, %0\n v_cmp_lt_u32_e32 vcc, %4, %1\n v_cmp_lt_u32_e32 vcc, %4, %2\n v_cmp_lt_u32_e32 vcc, %4, %3" OPS);
            if (FORM == 21) asm volatile("v_cvt_f32_f16_e32 %0, %0\n v_cvt_f32_f16_e32 %1, %1\n v_cvt_f32_f16_e32 %2, %2\n v_cvt_f32_f16_e32 %3, %3" OPS);
            if (FORM == 22) asm volatile("v_cvt_f32_u32_e32 %0, %0\n v_cvt_f32_u32_e32 %1, %1\n v_cvt_f32_u32_e32 %2, %2\n v_cvt_f32_u32_e32 %3, %3" OPS);
            if (FORM == 23) asm volatile("v_cvt_u32_f32_e32 %0, %0\n v_cvt_u32_f32_e32 %1, %1\n v_cvt_u32_f32_e32 %2, %2\n v_cvt_u32_f32_e32 %3, %3" OPS);
            if (FORM == 24) asm volatile("v_lshl_add_u32 %0, %0, 5, %4\n v_lshl_add_u32 %1, %1, 5, %4\n v_lshl_add_u32 %2, %2, 5, %4\n v_lshl_add_u32 %3, %3, 5, %4" OPS);
            if (FORM == 25) asm volatile("v_lshlrev_b32_e32 %0, 1, %0\n v_lshlrev_b32_e32 %1, 1, %1\n v_lshlrev_b32_e32 %2, 1, %2\n v_lshlrev_b32_e32 %3, 1, %3" OPS);
            if (FORM == 26) asm volatile("v_lshrrev_b32_e32 %0, 1, %0\n v_lshrrev_b32_e32 %1, 1, %1\n v_lshrrev_b32_e32 %2, 1, %2\n v_lshrrev_b32_e32 %3, 1, %3" OPS);
            if (FORM == 27) asm volatile("v_and_b32_e32 %0, %4, %0\n v_and_b32_e32 %1, %4, %1\n v_and_b32_e32 %2, %4, %2\n v_and_b32_e32 %3, %4, %3" OPS);
            if (FORM == 28) asm volatile("v_or_b32_e32 %0, %4, %0\n v_or_b32_e32 %1, %4, %1\n v_or_b32_e32 %2, %4, %2\n v_or_b32_e32 %3, %4, %3" OPS);
            if (FORM == 29) asm volatile("v_xor_b32_e32 %0, %4, %0\n v_xor_b32_e32 %1, %4, %1\n v_xor_b32_e32 %2, %4, %2\n v_xor_b32_e32 %3, %4, %3" OPS);
            if (FORM == 30) asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" OPS);
            if (FORM == 31) asm volatile("v_alignbit_b32 %0, %0, %0, 6\n v_alignbit_b32 %1, %1, %1, 6\n v_alignbit_b32 %2, %2, %2, 6\n v_alignbit_b32 %3, %3, %3, 6" OPS);
            if (FORM == 32) asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8" OPS);
            if (FORM == 33) asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" OPS);
            if (FORM == 34) asm volatile("v_add_u32_e32 %0, %4, %0\n v_add_u32_e32 %1, %4, %1\n v_add_u32_e32 %2, %4, %2\n v_add_u32_e32 %3, %4, %3" OPS);
            if (FORM == 35) asm volatile("v_sub_u32_e32 %0, %0, %4\n v_sub_u32_e32 %1, %1, %4\n v_sub_u32_e32 %2, %2, %4\n v_sub_u32_e32 %3, %3, %4" OPS);
            if (FORM == 36) asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" OPS);
            if (FORM == 37) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" OPS);
            if (FORM == 38) asm volatile("v_mul_u32_u24_e32 %0, %4, %0\n v_mul_u32_u24_e32 %1, %4, %1\n v_mul_u32_u24_e32 %2, %4, %2\n v_mul_u32_u24_e32 %3, %4, %3" OPS);
            if (FORM == 39) asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" OPS);
            if (FORM == 40) asm volatile("v_rcp_f32_e32 %0, %0\n v_rcp_f32_e32 %1, %1\n v_rcp_f32_e32 %2, %2\n v_rcp_f32_e32 %3, %3" OPS);
            if (FORM == 41) asm volatile("v_fma_mix_f32 %0, %0, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %4, %5 op_sel_hi:[1,0,0]" OPS);
            if (FORM == 42) asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %5\n v_div_scale_f32 %1, vcc, %1, %4, %5\n v_div_scale_f32 %2, vcc, %2, %4, %5\n v_div_scale_f32 %3, vcc, %3, %4, %5" OPS);
            if (FORM == 43) asm volatile("v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %4, %5" OPS);
            if (FORM == 44) asm volatile("v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5" OPS);
            if (FORM == 45) asm volatile("v_cmp_class_f32_e64 s[20:21], %0, %6\n v_cmp_class_f32_e64 s[20:21], %1, %6\n v_cmp_class_f32_e64 s[20:21], %2, %6\n v_cmp_class_f32_e64 s[20:21], %3, %6" OPS);
            if (FORM == 46) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_max_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_max_f32_e32 %3, %4, %3" OPS);
            if (FORM == 47) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_max_f32_e32 %3, %4, %3" OPS);
            if (FORM == 48) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_max_f32_e32 %2, %4, %2\n v_max_f32_e32 %3, %4, %3" OPS);
            if (FORM == 49) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_cmp_lt_f32_e64 s[20:21], %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_cmp_lt_f32_e64 s[20:21], %4, %3" OPS);
            if (FORM == 50) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_sub_f32_e32 %1, %6, %1\n v_mul_f32_e32 %2, %4, %2\n v_sub_f32_e32 %3, %6, %3" OPS);
            if (FORM == 51) asm volatile("v_mul_f32_e32 %0, %4, %0\n v_mul_f32_e32 %1, %4, %1\n v_mul_f32_e32 %2, %4, %2\n v_rcp_f32_e32 %3, %3" OPS);
            if (FORM == 52) asm volatile("v_max_f32_e32 %0, %4, %0\n v_max_f32_e32 %1, %4, %1\n v_max_f32_e32 %2, %4, %2\n v_rcp_f32_e32 %3, %3" OPS);
            if (FORM == 53) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(pm), "v"(pc));
            if (FORM == 54) asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(pm), "v"(pc));
        }
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    for (int k = 0; k < 4; ++k) s += p[k].x + p[k].y;
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
    printf("%-52s waves/SIMD %d  %6.2f ms  %5.2f cycles per instruction per SIMD\n", name, waves_per_simd, ms, cycles / ((double)iters * waves_per_simd * 32.0));
}

int main() {
    float* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float)));
    int khz = 0; CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock %.3f GHz\n", ghz);
    for (int w : {6}) {
        run<0>("v_fma_f32 v,v,v,v   (VOP3, 3 VGPR)", w, d_out, ghz);
        run<1>("v_fma_f32 v,v,s,v   (VOP3, 2 VGPR + SGPR)", w, d_out, ghz);
        run<2>("v_fmac_f32_e32 v,v,v", w, d_out, ghz);
        run<3>("v_mul_f32_e32 v,v,v", w, d_out, ghz);
        run<4>("v_mul_f32_e32 v,s,v (SGPR operand)", w, d_out, ghz);
        run<5>("v_mul_f32_e32 v,lit,v (literal)", w, d_out, ghz);
        run<6>("v_mul_f32_e32 v,2.0,v (inline constant)", w, d_out, ghz);
        run<7>("v_add_f32_e32 v,v,v", w, d_out, ghz);
        run<8>("v_sub_f32_e32 v,v,v", w, d_out, ghz);
        run<9>("v_sub_f32_e32 v,s,v (SGPR operand)", w, d_out, ghz);
        run<10>("v_mov_b32_e32 v,v", w, d_out, ghz);
        run<11>("v_mov_b32_e32 v,s", w, d_out, ghz);
        run<12>("v_max_f32_e32", w, d_out, ghz);
        run<13>("v_min_f32_e32", w, d_out, ghz);
        run<14>("v_max3_f32", w, d_out, ghz);
        run<15>("v_med3_f32", w, d_out, ghz);
        run<16>("v_cndmask_b32_e32 (vcc)", w, d_out, ghz);
        run<17>("v_cndmask_b32_e64 (s[20:21])", w, d_out, ghz);
        run<18>("v_cmp_lt_f32_e32 vcc", w, d_out, ghz);
        run<19>("v_cmp_lt_f32_e64 s[20:21]", w, d_out, ghz);
        run<20>("v_cmp_lt_u32_e32 vcc", w, d_out, ghz);
        run<21>("v_cvt_f32_f16_e32", w, d_out, ghz);
        run<22>("v_cvt_f32_u32_e32", w, d_out, ghz);
        run<23>("v_cvt_u32_f32_e32", w, d_out, ghz);
        run<24>("v_lshl_add_u32", w, d_out, ghz);
        run<25>("v_lshlrev_b32_e32", w, d_out, ghz);
        run<26>("v_lshrrev_b32_e32", w, d_out, ghz);
        run<27>("v_and_b32_e32", w, d_out, ghz);
        run<28>("v_or_b32_e32", w, d_out, ghz);
        run<29>("v_xor_b32_e32", w, d_out, ghz);
        run<30>("v_and_or_b32", w, d_out, ghz);
        run<31>("v_alignbit_b32 (rotate)", w, d_out, ghz);
        run<32>("v_bfe_u32", w, d_out, ghz);
        run<33>("v_perm_b32", w, d_out, ghz);
        run<34>("v_add_u32_e32", w, d_out, ghz);
        run<35>("v_sub_u32_e32", w, d_out, ghz);
        run<36>("v_add3_u32", w, d_out, ghz);
        run<37>("v_mul_lo_u32", w, d_out, ghz);
        run<38>("v_mul_u32_u24_e32", w, d_out, ghz);
        run<39>("v_mad_u32_u24", w, d_out, ghz);
        run<40>("v_rcp_f32_e32", w, d_out, ghz);
        run<41>("v_fma_mix_f32 (f16 in, f32 out)", w, d_out, ghz);
        run<42>("v_div_scale_f32", w, d_out, ghz);
        run<43>("v_div_fmas_f32", w, d_out, ghz);
        run<44>("v_div_fixup_f32", w, d_out, ghz);
        run<45>("v_cmp_class_f32_e64", w, d_out, ghz);
        run<46>("mix: mul,max,mul,max", w, d_out, ghz);
        run<47>("mix: mul,mul,mul,max", w, d_out, ghz);
        run<48>("mix: mul,mul,max,max", w, d_out, ghz);
        run<49>("mix: mul,cmp,mul,cmp", w, d_out, ghz);
        run<50>("mix: mul,sub-with-SGPR,mul,sub-with-SGPR", w, d_out, ghz);
        run<51>("mix: mul,mul,mul,rcp", w, d_out, ghz);
        run<52>("mix: max,max,max,rcp", w, d_out, ghz);
        run<53>("v_pk_fma_f32 (2 FMAs)", w, d_out, ghz);
        run<54>("v_pk_mul_f32 (2 muls)", w, d_out, ghz);
    }
    return 0;
}
