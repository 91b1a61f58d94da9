// pooled.hip — "pooled" megakernel (TRT_BACKEND_POOLED), written for gfx950 (CDNA4) only.
//
// The plain megakernel binds one path to one lane, so in every bounce the 64 lanes of a wave wait
// for the longest of 64 BVH walks: measured on the Cornell box only ~26 % of the lanes are doing a
// box test in an average box-test instruction.  This variant keeps the megakernel's ownership rule
// (a lane owns its pixels and walks their samples in order, hence the same bit-exact accumulation)
// but unbinds the *rays* from the lanes while they are being traced:
//
//  * every lane owns TWO pixels (two 8x8 sub-tiles per wave), so a wave has up to 128 paths alive;
//  * each bounce, the wave writes the rays of all its live paths into a ray pool in LDS
//    (origin, direction | hit t, hit primitive: 32 B per ray), compacts their slot numbers into a
//    queue with 64-bit ballots + mbcnt prefix ranks, and then traces the queue with ALL lanes:
//    a lane that finishes a walk retires it (hit record back to the pool) and pulls the next ray
//    from the queue (wave-local cursor, no atomics), so lanes idle only at the very end of the queue;
//  * shading then runs per owner lane, path A then path B, each pass with (almost) every lane active;
//  * the next sample's primary ray of every path is generated ahead of time into an LDS stock
//    (as in kernels.hip), so ray generation also runs with most lanes active.
//
// The pool, the stock and the queue are private to a wave: the phases need no workgroup barrier.
#include <stdlib.h>

#include "kernels.h"
#include "rt_path.h"

namespace trt {

constexpr uint32_t kPoolThreads = 256;
constexpr uint32_t kPoolTileW = 32, kPoolTileH = 16;          // pixels per workgroup: 8 sub-tiles of 8x8, two per wave
constexpr uint32_t kPoolNone = 0xFFFFFFFFu;

struct PoolWaveLds {                                           // per wave
    float4 ray_o[128];                                         // origin.xyz, hit t
    float4 ray_d[128];                                         // direction.xyz, bits(hit primitive)
    float4 stock_o[128];                                       // next sample's primary ray: origin.xyz, bits(rng.s0)
    float4 stock_d[128];                                       //                              direction.xyz, bits(rng.s1)
    uint8_t queue[128];
};

struct Owned {                                                 // what a lane keeps in registers per owned pixel
    V3 atten, color, acc;
    Rng rng;
    uint32_t remain, s, x, row;
    bool alive, fresh, stocked;
};

TRT_DEV uint32_t pl_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
TRT_DEV uint32_t pl_popc(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }

template <int MODE, bool STATS, int MINW = 1>
__global__ __launch_bounds__(kPoolThreads, MINW) void pooled_kernel(SceneDev scd, CameraDev cam, RenderArgs ra,
                                                                    float* __restrict__ accum,
                                                                    unsigned long long* __restrict__ counters,
                                                                    uint32_t tiles_x, uint32_t serve_min) {
    stage_scene_to_lds<MODE>(scd);
    const SceneAcc<MODE> sc{scd.blob, scd.L};
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    PoolWaveLds& lds = reinterpret_cast<PoolWaveLds*>(reinterpret_cast<char*>(g_lds) + sc.lds_bytes())[wave];

    const uint32_t tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);
    uint32_t n_samples = 0, n_rays = 0;
    Counters<STATS> ctr;

    Owned own[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t st = wave * 2u + (uint32_t)k;                                   // sub-tile 0..7 of the 32x16 tile
        own[k].x = tile_x * kPoolTileW + (st & 3u) * 8u + (lane & 7u);
        own[k].row = tile_y * kPoolTileH + (st >> 2) * 8u + (lane >> 3);
        own[k].s = ra.sample_begin;
        own[k].alive = own[k].x < cam.width && own[k].row < ra.rows_local && ra.sample_begin < ra.sample_end;
        own[k].fresh = true;
        own[k].stocked = false;
        own[k].remain = 0u;
        own[k].acc = v3(0.0f, 0.0f, 0.0f);
        if (own[k].alive && ra.accumulate) {
            const float* out = accum + 3ull * ((unsigned long long)own[k].row * cam.width + own[k].x);
            own[k].acc = v3(out[0], out[1], out[2]);
        }
    }
    const bool in_image[2] = {own[0].alive, own[1].alive};

    for (;;) {
        if (__builtin_amdgcn_ballot_w64(own[0].alive || own[1].alive) == 0ull) break;

        // ---- STOCK: primary rays one sample ahead, generated for every path whose stock is empty ----
        if (__builtin_amdgcn_ballot_w64((own[0].alive && own[0].fresh && !own[0].stocked) ||
                                        (own[1].alive && own[1].fresh && !own[1].stocked)) != 0ull) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                Owned& o = own[k];
                const uint32_t s_next = o.fresh ? o.s : o.s + 1u;
                if (o.alive && !o.stocked && s_next < ra.sample_end) {
                    if constexpr (STATS) { if (first_active_lane()) ctr.w_gen++; }
                    const uint32_t y = image_row(ra, o.row);
                    Rng g = rng_seed(ra.seed_key, y * cam.width + o.x, s_next);
                    const Ray r = primary_ray(cam, o.x, y, g);
                    lds.stock_o[k * 64 + lane] = make_float4(r.o.x, r.o.y, r.o.z, __uint_as_float(g.s0));
                    lds.stock_d[k * 64 + lane] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(g.s1));
                    o.stocked = true;
                }
            }
        }

        // ---- START fresh paths from the stock; QUEUE every live path's ray ----
        uint32_t n_queue = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            Owned& o = own[k];
            if (o.alive && o.fresh) {                                                  // cpu.rs:42-45
                const float4 so = lds.stock_o[k * 64 + lane], sd = lds.stock_d[k * 64 + lane];
                lds.ray_o[k * 64 + lane] = make_float4(so.x, so.y, so.z, 0.0f);
                lds.ray_d[k * 64 + lane] = make_float4(sd.x, sd.y, sd.z, __uint_as_float(PRIM_NONE));
                o.rng = Rng{__float_as_uint(so.w), __float_as_uint(sd.w)};
                o.color = v3(0.0f, 0.0f, 0.0f);
                o.atten = v3(1.0f, 1.0f, 1.0f);
                o.remain = ra.max_bounces;
                o.stocked = false;
                o.fresh = false;
                n_samples++;
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(o.alive);
            if (o.alive) lds.queue[n_queue + pl_rank(m)] = (uint8_t)(k * 64 + lane);
            n_queue += pl_popc(m);
        }

        // ---- EXTEND: all lanes trace the queue; a lane that finishes pulls the next ray ----
        {
            uint32_t slot = kPoolNone, leaf = PRIM_NONE, cursor = 0;
            Ray ray;
            Trav tr;
            tr.i = 0;
            tr.n = 0;
            for (;;) {
                if constexpr (STATS) { if (first_active_lane()) ctr.w_rounds++; }
                uint64_t free_m = __builtin_amdgcn_ballot_w64(slot == kPoolNone);
                uint32_t n_free = pl_popc(free_m);
                if (cursor < n_queue && n_free != 0u) {
                    const uint32_t mine = cursor + pl_rank(free_m);
                    if (slot == kPoolNone && mine < n_queue) {
                        slot = lds.queue[mine];
                        const float4 a = lds.ray_o[slot], d = lds.ray_d[slot];
                        ray.o = v3(a.x, a.y, a.z);
                        ray.d = v3(d.x, d.y, d.z);
                        tr = trav_begin(sc, ray, ra.ref_tree != 0u);
                        leaf = PRIM_NONE;
                    }
                    cursor += n_free;
                    free_m = __builtin_amdgcn_ballot_w64(slot == kPoolNone);
                    n_free = pl_popc(free_m);
                }
                if (n_free == 64u) break;
                for (;;) {
                    const bool in_box = slot != kPoolNone && leaf == PRIM_NONE && tr.i < tr.n;
                    const uint32_t n_box = pl_popc(__builtin_amdgcn_ballot_w64(in_box));
                    const uint32_t n_wait = 64u - n_free - n_box;
                    if (n_box == 0u || n_wait >= serve_min) break;
                    if (in_box) leaf = trav_box_step<MODE, STATS>(sc, ray, tr, ctr);
                }
                if (slot != kPoolNone && leaf != PRIM_NONE) {
                    if constexpr (STATS) { if (first_active_lane()) ctr.w_leaf++; }
                    trav_leaf<MODE, STATS>(sc, ray, tr, leaf, ctr);
                    leaf = PRIM_NONE;
                }
                if (slot != kPoolNone && tr.i >= tr.n) {
                    reinterpret_cast<float*>(&lds.ray_o[slot])[3] = tr.t_best;
                    reinterpret_cast<uint32_t*>(&lds.ray_d[slot])[3] = tr.prim_best;
                    n_rays++;
                    slot = kPoolNone;
                }
            }
        }

        // ---- SHADE: each owner lane, path A then path B ----
#pragma unroll
        for (int k = 0; k < 2; k++) {
            Owned& o = own[k];
            if (o.alive) {
                const float4 a = lds.ray_o[k * 64 + lane], d = lds.ray_d[k * 64 + lane];
                Path p;
                p.ray.o = v3(a.x, a.y, a.z);
                p.ray.d = v3(d.x, d.y, d.z);
                p.color = o.color; p.atten = o.atten; p.remain = o.remain; p.rng = o.rng;
                if (shade_hit<MODE, STATS>(sc, p, __float_as_uint(d.w), a.w, background, ctr)) {
                    o.acc = o.acc + p.color * ra.inv_spp;                               // imager.rs:50
                    o.s++;
                    o.fresh = true;
                    o.alive = o.s < ra.sample_end;
                } else {
                    lds.ray_o[k * 64 + lane] = make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, 0.0f);
                    lds.ray_d[k * 64 + lane] = make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, __uint_as_float(PRIM_NONE));
                }
                o.color = p.color; o.atten = p.atten; o.remain = p.remain; o.rng = p.rng;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (in_image[k]) {
            float* out = accum + 3ull * ((unsigned long long)own[k].row * cam.width + own[k].x);
            out[0] = own[k].acc.x; out[1] = own[k].acc.y; out[2] = own[k].acc.z;
        }
    }
    flush_counters<STATS>(counters, n_samples, n_rays, ctr);
}

hipError_t launch_pooled(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, float* d_accum,
                         unsigned long long* d_counters, bool stats, uint32_t serve_min, hipStream_t stream) {
    const uint32_t tiles_x = (cam.width + kPoolTileW - 1) / kPoolTileW, tiles_y = (ra.rows_local + kPoolTileH - 1) / kPoolTileH;
    if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
    const size_t lds_bytes = scene_lds_bytes(sc.L) + 4 * sizeof(PoolWaveLds);
    const dim3 grid(tiles_x * tiles_y), block(kPoolThreads);
    if (serve_min == 0) serve_min = 16;
    auto go = [&](auto kernel) -> hipError_t {
        if (lds_bytes > 48u * 1024u) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, sc, cam, ra, d_accum, d_counters, tiles_x, serve_min);
        return hipGetLastError();
    };
    int w = 1;
    if (const char* e = getenv("TRT_POOL_MINW")) w = atoi(e);
    switch (scene_mode(sc.L)) {
        case MODE_LDS:
            if (w >= 5) return stats ? go(pooled_kernel<MODE_LDS, true, 5>) : go(pooled_kernel<MODE_LDS, false, 5>);
            if (w == 4) return stats ? go(pooled_kernel<MODE_LDS, true, 4>) : go(pooled_kernel<MODE_LDS, false, 4>);
            return stats ? go(pooled_kernel<MODE_LDS, true>) : go(pooled_kernel<MODE_LDS, false>);
        case MODE_HYBRID: return stats ? go(pooled_kernel<MODE_HYBRID, true>) : go(pooled_kernel<MODE_HYBRID, false>);
        default: return stats ? go(pooled_kernel<MODE_GLOBAL, true>) : go(pooled_kernel<MODE_GLOBAL, false>);
    }
}

}  // namespace trt
