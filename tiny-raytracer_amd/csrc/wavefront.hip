// wavefront.hip — wavefront backend (TRT_BACKEND_WAVEFRONT), written for gfx950 (CDNA4) only.
//
// The classic wavefront path tracer keeps ray queues in device memory and runs generate / extend /
// shade / compact as separate kernels joined by global atomics and a host loop.  On MI355X that
// shape pays for (a) one returning global atomic per wave per queue operation (one counter word
// saturates near 88 operations/us), (b) a kernel boundary per stage per bounce (1.5-2 us each,
// hundreds of bounces per pass) and (c) a host read-back per bounce to learn that the queues are
// empty.  This backend keeps the wavefront *algorithm* and moves its plumbing on chip
// (DESIGN.md §5):
//
//  * a workgroup owns a 64x32 pixel tile for the whole launch and runs the stages as phases of
//    one persistent kernel, separated by workgroup barriers instead of kernel boundaries;
//  * path state (origin, direction, throughput, radiance, RNG, bounce budget, sample index, hit)
//    is SoA in HBM, one slot per pixel: planes of 16-byte elements, so a lane moves its ray with
//    two 16-byte accesses and a wave's accesses stay inside the tile's 32 KiB window per plane;
//  * the ray queues are 16-bit slot indices in LDS.  EXTEND lanes pull rays from the queue with
//    one LDS atomic per refill and are refilled as soon as enough of a wave's lanes have finished
//    (ballot + mbcnt prefix), so a wave does not idle behind its longest traversal;
//  * before SHADE the tile's hits are binned by outcome (miss / light / Lambertian / metal /
//    dielectric) with 64-bit ballots, an LDS histogram and a prefix scan, so the 64 lanes of a
//    wave shade one material; paths that continue or start their next sample are compacted into
//    the next EXTEND queue with ballot + mbcnt + one LDS atomic per wave.
//
// One slot per pixel and samples walked in order keep the reference's accumulation order
// (`pixels[idx] += color * (1/spp)`, imager.rs:50): the frame is bit-identical to the megakernel's
// and to the CPU oracle's.
#include <stdlib.h>

#include "kernels.h"
#include "rt_path.h"

namespace trt {

constexpr uint32_t kWfThreads = 256;
constexpr uint32_t kWfTileW = 64, kWfTileH = 32;
constexpr uint32_t kWfSlots = kWfTileW * kWfTileH;            // 2048 paths in flight per workgroup
constexpr uint32_t kNone = 0xFFFFFFFFu;
enum { BIN_MISS = 0, BIN_LIGHT, BIN_LAMBERTIAN, BIN_METAL, BIN_DIELECTRIC, BIN_COUNT };

struct WfLds {                                                 // backend-private LDS, after the scene copy
    uint16_t queue[2][kWfSlots];                               // EXTEND queues (double buffered): local slot ids
    uint16_t sorted[kWfSlots];                                 // SHADE order: the current queue binned by outcome
    uint8_t bin_of[kWfSlots];                                  // outcome bin of each slot's last hit query
    uint32_t count[2];                                         // entries in queue[0], queue[1]
    uint32_t head;                                             // EXTEND pull cursor
    uint32_t hist[BIN_COUNT], base[BIN_COUNT], cursor[BIN_COUNT];
};

TRT_DEV uint32_t lane_id() { return threadIdx.x & 63u; }
TRT_DEV uint32_t rank_in(uint64_t mask) {                      // number of set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
TRT_DEV uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }

// Append this lane's slot to an LDS queue if `push`: ballot, one LDS atomic per wave, prefix rank.
TRT_DEV void wave_push(uint16_t* q, uint32_t* count, bool push, uint32_t slot) {
    const uint64_t m = __builtin_amdgcn_ballot_w64(push);
    if (m == 0ull) return;
    uint32_t base = 0;
    if (lane_id() == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(count, popc64(m));
    base = __shfl(base, __builtin_ctzll(m), 64);
    if (push) q[base + rank_in(m)] = (uint16_t)slot;
}

// local slot -> pixel of the tile: 8x8 sub-tiles so that the 64 slots a wave starts with are neighbours
TRT_DEV void slot_xy(uint32_t ls, uint32_t& lx, uint32_t& ly) {
    const uint32_t st = ls >> 6, l = ls & 63u;
    lx = (st & 7u) * 8u + (l & 7u);
    ly = (st >> 3) * 8u + (l >> 3);
}

struct WfTile {
    uint32_t x0, row0;          // tile origin (local rows)
    unsigned long long slot0;   // first global slot of this workgroup
};

TRT_DEV void store_path(const WfState& st, unsigned long long g, const Path& p, uint32_t sample) {
    st.s0[g] = make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, 0.0f);
    st.s1[g] = make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, __uint_as_float(PRIM_NONE));
    st.s2[g] = make_float4(p.atten.x, p.atten.y, p.atten.z, __uint_as_float(p.remain));
    st.s3[g] = make_float4(p.color.x, p.color.y, p.color.z, __uint_as_float(sample));
    st.rng[g] = make_uint2(p.rng.s0, p.rng.s1);
}

template <int MODE, bool STATS, int MINW = 1>
__global__ __launch_bounds__(kWfThreads, MINW) void wavefront_kernel(SceneDev scd, CameraDev cam, RenderArgs ra, WfState st,
                                                               float* __restrict__ accum,
                                                               unsigned long long* __restrict__ counters, uint32_t tiles_x,
                                                               uint32_t serve_min) {
    stage_scene_to_lds<MODE>(scd);
    const SceneAcc<MODE> sc{scd.blob, scd.L};
    WfLds& lds = *reinterpret_cast<WfLds*>(reinterpret_cast<char*>(g_lds) + sc.lds_bytes());

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    WfTile tile;
    const uint32_t tile_id = xcd_tile(blockIdx.x, gridDim.x, ra.xcd_aware);
    tile.x0 = (tile_id % tiles_x) * kWfTileW;
    tile.row0 = (tile_id / tiles_x) * kWfTileH;
    tile.slot0 = (unsigned long long)tile_id * kWfSlots;
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);
    uint32_t n_samples = 0, n_rays = 0;
    Counters<STATS> ctr;

    // ---- GENERATE (first sample of every pixel of the tile) ----------------------------------
    if (tid == 0) { lds.count[0] = 0; lds.count[1] = 0; lds.head = 0; }
    __syncthreads();
    for (uint32_t ls = tid; ls < kWfSlots; ls += kWfThreads) {
        uint32_t lx, ly;
        slot_xy(ls, lx, ly);
        const uint32_t x = tile.x0 + lx, row = tile.row0 + ly;
        const bool valid = x < cam.width && row < ra.rows_local && ra.sample_begin < ra.sample_end;
        if (valid) {
            if (!ra.accumulate) {
                float* out = accum + 3ull * ((unsigned long long)row * cam.width + x);
                out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f;
            }
            Path p;
            path_begin(p, cam, ra, x, image_row(ra, row), ra.sample_begin);
            store_path(st, tile.slot0 + ls, p, ra.sample_begin);
            n_samples++;
        }
        if constexpr (STATS) { if (first_active_lane()) ctr.w_gen++; }
        wave_push(lds.queue[0], &lds.count[0], valid, ls);
    }
    __syncthreads();

    uint32_t cur = 0;
    for (;;) {
        const uint32_t n = lds.count[cur];
        if (n == 0) break;                                   // uniform: every path of the tile has finished
        const uint16_t* q = lds.queue[cur];

        // ---- EXTEND: closest hit for every queued path; lanes refill from the LDS queue --------
        {
            uint32_t slot = kNone, leaf = PRIM_NONE;
            Ray ray;
            Trav tr;
            tr.i = 0;
            tr.n = 0;
            bool exhausted = false;
            for (;;) {
                if constexpr (STATS) { if (first_active_lane()) ctr.w_rounds++; }
                uint64_t free_m = __builtin_amdgcn_ballot_w64(slot == kNone);
                uint32_t n_free = popc64(free_m);
                if (!exhausted && n_free != 0u) {
                    uint32_t b = 0;
                    if (lane == (uint32_t)__builtin_ctzll(free_m)) b = atomicAdd(&lds.head, n_free);
                    b = __shfl(b, __builtin_ctzll(free_m), 64);
                    const uint32_t mine = b + rank_in(free_m);
                    if (slot == kNone && mine < n) {
                        slot = q[mine];
                        const float4 a = st.s0[tile.slot0 + slot], d = st.s1[tile.slot0 + slot];
                        ray.o = v3(a.x, a.y, a.z);
                        ray.d = v3(d.x, d.y, d.z);
                        tr = trav_begin(sc, ray, ra.ref_tree != 0u);
                        leaf = PRIM_NONE;
                    }
                    exhausted = b + n_free >= n;
                    free_m = __builtin_amdgcn_ballot_w64(slot == kNone);
                    n_free = popc64(free_m);
                }
                if (n_free == 64u) break;                    // nothing in flight and nothing left to pull
                // box tests while enough lanes take part; lanes on a leaf or at the end of their walk wait
                for (;;) {
                    const bool in_box = slot != kNone && leaf == PRIM_NONE && tr.i < tr.n;
                    const uint32_t n_box = popc64(__builtin_amdgcn_ballot_w64(in_box));
                    const uint32_t n_wait = 64u - n_free - n_box;
                    if (n_box == 0u || n_wait >= serve_min) break;
                    if (in_box) leaf = trav_box_step<MODE, STATS>(sc, ray, tr, ctr);
                }
                // primitive tests for lanes standing on a leaf
                if (slot != kNone && leaf != PRIM_NONE) {
                    if constexpr (STATS) { if (first_active_lane()) ctr.w_leaf++; }
                    trav_leaf<MODE, STATS>(sc, ray, tr, leaf, ctr);
                    leaf = PRIM_NONE;
                }
                // retire finished walks: hit record to HBM, outcome bin to LDS
                if (slot != kNone && tr.i >= tr.n) {
                    const unsigned long long g = tile.slot0 + slot;
                    reinterpret_cast<float*>(&st.s0[g])[3] = tr.t_best;
                    reinterpret_cast<uint32_t*>(&st.s1[g])[3] = tr.prim_best;
                    uint32_t bin = BIN_MISS;
                    if (tr.prim_best != PRIM_NONE) {
                        const uint32_t kind = sc.material_kind(prim_material(sc, tr.prim_best));
                        bin = kind == TRT_LIGHT ? BIN_LIGHT : (kind == TRT_LAMBERTIAN ? BIN_LAMBERTIAN : (kind == TRT_METAL ? BIN_METAL : BIN_DIELECTRIC));
                    }
                    lds.bin_of[slot] = (uint8_t)bin;
                    n_rays++;
                    slot = kNone;
                }
            }
        }
        __syncthreads();

        // ---- SORT: bin the tile's hits by outcome (ballot histogram + prefix scan in LDS) -------
        if (tid < BIN_COUNT) { lds.hist[tid] = 0; lds.cursor[tid] = 0; }
        if (tid == 0) { lds.count[cur ^ 1u] = 0; lds.head = 0; }
        __syncthreads();
        for (uint32_t j0 = 0; j0 < n; j0 += kWfThreads) {
            const uint32_t j = j0 + tid;
            const uint32_t bin = j < n ? lds.bin_of[q[j]] : kNone;
#pragma unroll
            for (uint32_t b = 0; b < BIN_COUNT; b++) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(bin == b);
                if (m != 0ull && lane == (uint32_t)__builtin_ctzll(m)) atomicAdd(&lds.hist[b], popc64(m));
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (uint32_t b = 0; b < BIN_COUNT; b++) { lds.base[b] = run; run += lds.hist[b]; }
        }
        __syncthreads();
        for (uint32_t j0 = 0; j0 < n; j0 += kWfThreads) {
            const uint32_t j = j0 + tid;
            const uint32_t slot = j < n ? q[j] : 0u;
            const uint32_t bin = j < n ? lds.bin_of[slot] : kNone;
#pragma unroll
            for (uint32_t b = 0; b < BIN_COUNT; b++) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(bin == b);
                if (m == 0ull) continue;
                uint32_t off = 0;
                if (lane == (uint32_t)__builtin_ctzll(m)) off = atomicAdd(&lds.cursor[b], popc64(m));
                off = __shfl(off, __builtin_ctzll(m), 64);
                if (bin == b) lds.sorted[lds.base[b] + off + rank_in(m)] = (uint16_t)slot;
            }
        }
        __syncthreads();

        // ---- SHADE: emission + scatter per material; finished samples fold into the pixel and the
        //      next sample's primary ray is generated in place; survivors are compacted for EXTEND ----
        uint16_t* qn = lds.queue[cur ^ 1u];
        for (uint32_t j0 = 0; j0 < n; j0 += kWfThreads) {
            const uint32_t j = j0 + tid;
            bool go_on = false;
            uint32_t slot = 0;
            if (j < n) {
                slot = lds.sorted[j];
                const unsigned long long g = tile.slot0 + slot;
                const float4 a = st.s0[g], d = st.s1[g], th = st.s2[g], co = st.s3[g];
                const uint2 rs = st.rng[g];
                Path p;
                p.ray.o = v3(a.x, a.y, a.z);
                p.ray.d = v3(d.x, d.y, d.z);
                p.atten = v3(th.x, th.y, th.z);
                p.remain = __float_as_uint(th.w);
                p.color = v3(co.x, co.y, co.z);
                p.rng = Rng{rs.x, rs.y};
                uint32_t sample = __float_as_uint(co.w);
                go_on = true;
                if (shade_hit<MODE, STATS>(sc, p, __float_as_uint(d.w), a.w, background, ctr)) {
                    uint32_t lx, ly;
                    slot_xy(slot, lx, ly);
                    const uint32_t x = tile.x0 + lx, row = tile.row0 + ly;
                    float* out = accum + 3ull * ((unsigned long long)row * cam.width + x);
                    const V3 acc = v3(out[0], out[1], out[2]) + p.color * ra.inv_spp;       // imager.rs:50
                    out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
                    sample++;
                    go_on = sample < ra.sample_end;
                    if (go_on) {
                        path_begin(p, cam, ra, x, image_row(ra, row), sample);
                        n_samples++;
                    }
                }
                if (go_on) store_path(st, g, p, sample);
            }
            wave_push(qn, &lds.count[cur ^ 1u], go_on, slot);
        }
        __syncthreads();
        cur ^= 1u;
    }
    flush_counters<STATS>(counters, n_samples, n_rays, ctr);
}

size_t wavefront_workspace_bytes(uint32_t width, uint32_t rows) {
    const size_t tiles = (size_t)((width + kWfTileW - 1) / kWfTileW) * ((rows + kWfTileH - 1) / kWfTileH);
    return tiles * kWfSlots * (4 * sizeof(float4) + sizeof(uint2));
}

hipError_t launch_wavefront(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, const trt_tuning& tn, void* workspace, float* d_accum,
                            unsigned long long* d_counters, bool stats, hipStream_t stream) {
    const uint32_t tiles_x = (cam.width + kWfTileW - 1) / kWfTileW, tiles_y = (ra.rows_local + kWfTileH - 1) / kWfTileH;
    if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
    const size_t n_slots = (size_t)tiles_x * tiles_y * kWfSlots;
    WfState st;
    char* base = static_cast<char*>(workspace);
    st.s0 = reinterpret_cast<float4*>(base);
    st.s1 = st.s0 + n_slots;
    st.s2 = st.s1 + n_slots;
    st.s3 = st.s2 + n_slots;
    st.rng = reinterpret_cast<uint2*>(st.s3 + n_slots);
    const size_t lds_bytes = scene_lds_bytes(sc.L) + sizeof(WfLds);
    const dim3 grid(tiles_x * tiles_y), block(kWfThreads);
    const uint32_t serve_min = tn.wf_serve_min ? tn.wf_serve_min : 12u;        // flat optimum 8..24 on the 100 k-sphere scene
    auto go = [&](auto kernel) -> hipError_t {
        if (lds_bytes > 48u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        timing_mark(stream, true);
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, sc, cam, ra, st, d_accum, d_counters, tiles_x, serve_min);
        const hipError_t le = hipGetLastError();
        timing_mark(stream, false);
        return le;
    };
    switch (scene_mode(sc.L)) {
        case MODE_LDS: return stats ? go(wavefront_kernel<MODE_LDS, true>) : go(wavefront_kernel<MODE_LDS, false>);
        default: {
            int w = 6;                              // 80 VGPRs, 12 B of scratch: +2 % over the 87-VGPR / 5-wave allocation; 7, 8: slower
            if (tn.wf_waves_per_simd) w = (int)tn.wf_waves_per_simd;
            if (w >= 8) return stats ? go(wavefront_kernel<MODE_GLOBAL, true, 8>) : go(wavefront_kernel<MODE_GLOBAL, false, 8>);
            if (w == 7) return stats ? go(wavefront_kernel<MODE_GLOBAL, true, 7>) : go(wavefront_kernel<MODE_GLOBAL, false, 7>);
            if (w == 6) return stats ? go(wavefront_kernel<MODE_GLOBAL, true, 6>) : go(wavefront_kernel<MODE_GLOBAL, false, 6>);
            return stats ? go(wavefront_kernel<MODE_GLOBAL, true>) : go(wavefront_kernel<MODE_GLOBAL, false>);
        }
    }
}

}  // namespace trt
