// Microbenchmark (round 4): what would it cost to hand the TAIL of Cornell's leaf phase to idle lanes?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I tiny-raytracer_amd/csrc -o build/leaf_exchange tools/micro/leaf_exchange.hip && ./build/leaf_exchange
//
// The lock-step walk's leaf phase (rt_path.h leaf_phase) runs as many trips as the wave's BUSIEST lane has postponed leaves: 3.43 per wave
// round for 1.21 leaves per ray (tools/leaf_phase_budget.py, profiles/r04_leaf_phase_budget.txt).  Trip 0 is full (every lane has its first leaf);
// the tail - slots 1.. of the 17 % of lanes that hold more than one leaf - is ~13 tests per round spread over 2.43 trips at 8 % of the lanes.
// Those tests could run in ONE trip on any 13 lanes: the tests of one lane are ordered, but the acceptance rule is monotonic in t_best, so
// they can be evaluated against the t_best after slot 0 and folded in order afterwards (start_j < t_best and t_j < t_best, re-applied with the
// running t_best: exact).  What that needs: the items enumerated (ballot + mbcnt per level) into an LDS table, the owner's ray (origin,
// direction, t_best: 7 values) fetched by the executing lane with ds_bpermute, the owner's stack entry read, the result sent back per level
// with ds_permute, and the in-order fold.  Both variants below run the REAL test (rt_path.h trav_leaf, quads in LDS) on the same synthetic
// rounds - k leaves per lane drawn with the measured distribution - at the product's occupancy (256 lanes, 6 workgroups per CU):
//   variant 0: the tail as the product runs it (slot j in trip j, each lane its own);
//   variant 1: the tail redistributed (enumerate, gather, one trip, scatter, fold).
// Output: cycles per round per SIMD of either, i.e. what the exchange costs against what it saves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rt_path.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace trt;

constexpr uint32_t kQuads = 18, kSlots = 7;

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int VARIANT>
__global__ __launch_bounds__(256, 6) void tail(const float4* __restrict__ quads, float* out, int rounds) {
    // LDS as in the product's Cornell plan: scene | leaf stack (7 slots x 8 B per lane) | (pool area, here: the item table of variant 1)
    for (uint32_t k = threadIdx.x; k < 5u * kQuads; k += blockDim.x) g_lds[k] = quads[k];
    __syncthreads();
    SceneLayout L{};
    L.off_quad = 0;
    const SceneAcc<MODE_LDS> sc{quads, L};
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    char* tailp = reinterpret_cast<char*>(g_lds) + 5u * kQuads * 16u;
    float2* const stk = reinterpret_cast<float2*>(tailp) + wave * (64u * kSlots) + lane;
    uint32_t* const table = reinterpret_cast<uint32_t*>(tailp + 256u * kSlots * 8u) + wave * 64u;       // variant 1: item e -> (owner lane | slot << 8)
    Counters<false> ctr;
    float acc = 0.0f;
    uint32_t seed = hash32(blockIdx.x * 256u + threadIdx.x + 1u);
    for (int r = 0; r < rounds; r++) {
        seed = hash32(seed + (uint32_t)r);
        // this round's ray and leaves: k = 1 (83 %), 2 (14 %), 3 (2.5 %), 4 (0.5 %)
        Ray ray;
        ray.o = v3(50.0f + (float)(seed & 31u), 40.0f + (float)((seed >> 5) & 31u), 10.0f + (float)((seed >> 10) & 31u));
        ray.d = normalized(v3((float)((seed >> 15) & 15u) - 7.5f, (float)((seed >> 19) & 15u) - 7.5f, (float)((seed >> 23) & 15u) - 7.5f));
        const uint32_t u = seed % 1000u;
        const uint32_t k = u < 830u ? 1u : (u < 970u ? 2u : (u < 995u ? 3u : 4u));
        for (uint32_t j = 0; j < k; j++) stk[64u * j] = make_float2(__uint_as_float(PRIM_QUAD_BIT | ((seed >> (3u * j)) % kQuads)), kTMin);
        Trav tr{};
        tr.t_best = 1.0e3f;                               // t_best after slot 0 (the common trip is the same in both variants and not run here)
        tr.prim_best = PRIM_NONE;
        if constexpr (VARIANT == 2) {
            // round set-up only: what both variants pay before their tail
        } else if constexpr (VARIANT == 0) {
            for (uint32_t j = 1; j < k; j++) {            // the product's loop: slot j in trip j
                const float2 e = stk[64u * j];
                if (tr.t_best > e.y) trav_leaf<MODE_LDS, false>(sc, ray, tr, __float_as_uint(e.x), ctr);
            }
        } else {
            // 1. enumerate the tail items level by level: item index = items of lower levels + rank within the level
            uint32_t base = 0;
            for (uint32_t j = 1; j < 4u; j++) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(k > j);
                if (m == 0ull) break;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (k > j && base + rank < 64u) table[base + rank] = lane | (j << 8);
                base += (uint32_t)__builtin_popcountll(m);
            }
            const uint32_t items = base < 64u ? base : 64u;
            // 2. executing lanes: fetch the owner's ray and t_best, its stack entry, test
            float t1 = __builtin_inff(), t2 = t1, t3 = t1;                       // results by level, at the OWNER
            const bool exec_item = lane < items;
            const uint32_t it = exec_item ? table[lane] : lane;
            const uint32_t owner = it & 63u, slot = exec_item ? (it >> 8) : 0u;
            const int addr = (int)(owner << 2);
            Ray r2;
            r2.o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ray.o.x)));
            r2.o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ray.o.y)));
            r2.o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ray.o.z)));
            r2.d.x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ray.d.x)));
            r2.d.y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ray.d.y)));
            r2.d.z = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(ray.d.z)));
            const float tb0 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(tr.t_best)));
            float t_res = __builtin_inff();
            if (exec_item) {
                const float2 e = (stk - lane + owner)[64u * slot];
                Trav t2r{};
                t2r.t_best = tb0;
                t2r.prim_best = PRIM_NONE;
                if (tb0 > e.y) trav_leaf<MODE_LDS, false>(sc, r2, t2r, __float_as_uint(e.x), ctr);
                if (t2r.prim_best != PRIM_NONE) t_res = t2r.t_best;
            }
            // 3. results back to the owners, one forward permute per level (a lane owns at most one item per level)
            const int r1 = __builtin_amdgcn_ds_permute(addr, slot == 1u ? __float_as_int(t_res) : 0x7f800000);
            const int r2i = __builtin_amdgcn_ds_permute(addr, slot == 2u ? __float_as_int(t_res) : 0x7f800000);
            const int r3 = __builtin_amdgcn_ds_permute(addr, slot == 3u ? __float_as_int(t_res) : 0x7f800000);
            // (a forward permute delivers only where some lane wrote; lanes without an item at a level keep +inf)
            const uint64_t got1 = __builtin_amdgcn_ballot_w64(exec_item && slot == 1u), got2 = __builtin_amdgcn_ballot_w64(exec_item && slot == 2u),
                           got3 = __builtin_amdgcn_ballot_w64(exec_item && slot == 3u);
            (void)got1; (void)got2; (void)got3;
            t1 = k > 1u ? __int_as_float(r1) : t1;
            t2 = k > 2u ? __int_as_float(r2i) : t2;
            t3 = k > 3u ? __int_as_float(r3) : t3;
            // 4. fold in walk order with the running t_best: start_j < t_best and t_j < t_best
            const float ts[3] = {t1, t2, t3};
#pragma unroll
            for (uint32_t j = 1; j < 4u; j++) {
                if (k > j) {
                    const float2 e = stk[64u * j];
                    if (tr.t_best > e.y && ts[j - 1] < tr.t_best) { tr.t_best = ts[j - 1]; tr.prim_best = __float_as_uint(e.x); }
                }
            }
        }
        acc += tr.t_best + (float)(tr.prim_best & 31u);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int cus = 256, wg_per_cu = 6, rounds = 4000;
    std::vector<float> q(5 * kQuads * 4);
    for (size_t i = 0; i < q.size(); i++) q[i] = (float)((i * 2654435761u) % 97u) * 1.03f + 1.0f;
    float4* d_q; float* d_out;
    CHECK(hipMalloc((void**)&d_q, q.size() * 4)); CHECK(hipMemcpy(d_q, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&d_out, (size_t)cus * wg_per_cu * 256 * 4));
    const size_t lds = 5 * kQuads * 16 + 256 * kSlots * 8 + 4 * 64 * 4 + 9 * 1024;          // + the product's ray pool, to keep 6 workgroups per CU and no more
    const double ghz = 2.4;
    float ms[3] = {0, 0, 0};
    const char* what[3] = {"tail as shipped: slot j in trip j", "tail redistributed over idle lanes", "round set-up only"};
    for (int rep = 0; rep < 3; rep++) for (int v = 0; v < 3; v++) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto go = [&](int n) {
            if (v == 0) hipLaunchKernelGGL(tail<0>, dim3(cus * wg_per_cu), dim3(256), lds, 0, d_q, d_out, n);
            else if (v == 1) hipLaunchKernelGGL(tail<1>, dim3(cus * wg_per_cu), dim3(256), lds, 0, d_q, d_out, n);
            else hipLaunchKernelGGL(tail<2>, dim3(cus * wg_per_cu), dim3(256), lds, 0, d_q, d_out, n);
        };
        go(50);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        go(rounds);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[v], e0, e1));
        printf("rep %d variant %d (%s): %8.3f ms  = %7.1f cycles per round per SIMD (6 waves per SIMD)\n", rep, v, what[v], ms[v],
               ms[v] * 1e-3 * ghz * 1e9 / ((double)rounds * wg_per_cu));
    }
    const double c = 1e-3 * ghz * 1e9 / ((double)rounds * wg_per_cu);
    printf("tail as shipped: %.1f cycles per round per SIMD; redistributed: %.1f (enumerate + gather + one trip + scatter + fold); ratio %.2f\n",
           (ms[0] - ms[2]) * c, (ms[1] - ms[2]) * c, (ms[1] - ms[2]) / (ms[0] - ms[2]));
    return 0;
}
