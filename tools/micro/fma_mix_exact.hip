// Is v_fma_mix_f32 with an f16 first operand bit-identical to v_cvt_f32_f16 followed by v_fma_f32 on gfx950?  (rt_path.h box_loop_compact, round 5)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/fma_mix_exact tools/micro/fma_mix_exact.hip && ./build/fma_mix_exact
// The 16-byte-node walk evaluates a plane's distance as fma(float(x_f16), inv, nm) in ONE instruction - v_fma_mix_f32 reads x straight out of the
// low or the high half of the node's word - instead of a conversion and an FMA.  Compared here for EVERY f16 bit pattern (subnormals, zeros,
// infinities and NaNs included: an f16 denormal that the mixed instruction flushed would move a plane of a millimetre-sized scene by more than the
// slack its box was grown by) in both halves of the word, against 4096 (inv, nm) pairs each: magnitudes 2^-20..2^60, both signs, products that
// cancel, overflow and underflow.  Bit equality, NaN payloads aside (any NaN equals any NaN: the walk only compares).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void compare(unsigned long long* mismatches, unsigned long long* first_bad) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;             // one f16 bit pattern per lane
    if (h >= 65536u) return;
    const uint32_t word_lo = h | (mix32(h) << 16), word_hi = (h << 16) | (mix32(h + 77u) & 0xFFFFu);
    unsigned long long bad = 0;
    for (uint32_t k = 0; k < 4096u; ++k) {
        const uint32_t r0 = mix32(k * 2u + 1u), r1 = mix32(k * 2u + 2u + h * 8192u);
        // inv: sign, exponent 2^-20 .. 2^60, random mantissa
        const float inv = __uint_as_float((r0 & 0x80000000u) | ((107u + (r0 >> 8) % 81u) << 23) | (r1 & 0x7FFFFFu));
        float nm;
        if (k % 3u == 0u) {                                               // near cancellation: nm ~ -x * inv
            float x;
            asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(x) : "v"(h));
            nm = -(x * inv) * (1.0f + (float)((int)(r1 % 17u) - 8) * 1.1920929e-7f);
        } else {
            nm = __uint_as_float((r1 & 0x80000000u) | ((87u + (r1 >> 8) % 120u) << 23) | (r0 & 0x7FFFFFu));
        }
        float want_lo, want_hi, got_lo, got_hi, c;
        asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(c) : "v"(word_lo));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_lo) : "v"(c), "v"(inv), "v"(nm));
        asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(c) : "v"(word_hi));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_hi) : "v"(c), "v"(inv), "v"(nm));
        asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(got_lo) : "v"(word_lo), "v"(inv), "v"(nm));
        asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(got_hi) : "v"(word_hi), "v"(inv), "v"(nm));
        const bool ok_lo = __float_as_uint(want_lo) == __float_as_uint(got_lo) || (want_lo != want_lo && got_lo != got_lo);
        const bool ok_hi = __float_as_uint(want_hi) == __float_as_uint(got_hi) || (want_hi != want_hi && got_hi != got_hi);
        if (!ok_lo || !ok_hi) { if (!bad) first_bad[0] = ((unsigned long long)h << 32) | k; bad++; }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int main() {
    unsigned long long *d, *d2, h = 0, h2 = 0;
    CHECK(hipMalloc(&d, 8)); CHECK(hipMalloc(&d2, 8));
    CHECK(hipMemset(d, 0, 8)); CHECK(hipMemset(d2, 0, 8));
    compare<<<65536 / 256, 256>>>(d, d2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&h2, d2, 8, hipMemcpyDeviceToHost));
    printf("every f16 bit pattern in both halves of a word x 4096 (inv, nm) pairs (%.3g fused operations): %llu mismatches against v_cvt_f32_f16 + v_fma_f32",
           2.0 * 65536.0 * 4096.0, h);
    if (h) printf(" (one of them: f16 0x%04x, pair %u)", (unsigned)(h2 >> 32), (unsigned)h2);
    printf("\n");
    return h ? 1 : 0;
}
