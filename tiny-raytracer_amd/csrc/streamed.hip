// streamed.hip — "streamed" backend (TRT_BACKEND_STREAMED), written for gfx950 (CDNA4) only.
//
// The megakernel binds a lane to one pixel for a whole launch.  That ties the length of a launch to the most
// expensive pixel (in the Cornell box the per-pixel cost varies ~2x across the image) and to the longest of a
// wave's 64 sample chains; both hurt exactly when the launch is small: a short progressive pass, or one of 8 GPUs'
// share of the frame (2048 workgroups for 1792 resident ones: measured 16.5 Gray/s per GPU against 25.0 on the
// whole frame).  Here a SAMPLE, not a pixel, is the unit of work:
//
//  * SAMPLE kernel: a persistent grid of waves pulls batches (one 8x8 pixel tile x 8 consecutive samples = 512 items,
//    ~3.6 k rays: small against a wave's share even of one GPU's eighth of a frame) from a global counter, one
//    returning atomic per batch (<10 per us, the counter word saturates near 88); inside a wave every lane takes the next item of
//    the batch whenever its path ends (ballot + mbcnt ranks on a wave-uniform cursor, no atomics), so no lane waits
//    for a neighbour's longer chain or a more expensive pixel.  The finished sample's radiance (12 B, one store) goes to an HBM
//    buffer laid out [tile][sample][lane of the tile] (stream_fold_kernel).
//  * FOLD kernel: one lane per pixel adds that pixel's radiances in sample order,
//    `pixels[idx] += color * (1/spp)` (imager.rs:50): the same sequence of f32 additions as the reference, so the
//    frame is bit-identical to the megakernel's and the oracle's whatever lane computed which sample.
//
// Cost: 12 B written + 12 B read per sample (a sample is ~7 rays, ~4.6 KB algorithmic), i.e. <1 % extra traffic, and
// two launches per chunk of samples (sized to a radiance buffer of at most 16 GB: streamed_chunk_spp) instead of one.  stream_sample_kernel keeps the
// primary-ray stock of kernels.hip (items are taken ahead); stream_pool_kernel, used for small LDS scenes, generates the
// primary rays of a whole wave at once into an LDS pool.
#include "kernels.h"
#include "rt_path.h"

namespace trt {

constexpr uint32_t kBatchSpp = 8;            // samples per pixel in one batch

TRT_DEV uint32_t st_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <int MODE, bool STATS, int MINW = 1, int THREADS = 256, int WALK = WALK_RUNTIME, bool LAZY = false>
__global__ __launch_bounds__(THREADS, MINW) void stream_sample_kernel(SceneDev scd, CameraDev cam, RenderArgs ra,
                                                                             float* __restrict__ colors,
                                                                             uint32_t* __restrict__ batch_counter,
                                                                             unsigned long long* __restrict__ counters,
                                                                             uint32_t tiles_x, uint32_t n_tiles, uint32_t n_batches, uint32_t batch_spp,
                                                                             const float4* __restrict__ leaf_list,
                                                                             const uint4* __restrict__ nodes16) {
    stage_scene_to_lds<MODE>(scd);
    const SceneAcc<MODE> sc{scd.blob, scd.L};
    const uint32_t lane = threadIdx.x & 63u;
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);
    const uint32_t n_spp = ra.sample_end - ra.sample_begin;
    // postponed-leaf stack (rt_path.h walk_fast_lds): behind the scene copy, leaf_slots x 64 x 8 bytes per wave
    float2* const leaf_stack = (WALK != WALK_REGS && (WALK != WALK_RUNTIME || ra.lds_leaf_stack))
        ? reinterpret_cast<float2*>(reinterpret_cast<char*>(g_lds) + ((sc.lds_bytes() + 15u) & ~15u)) + (threadIdx.x >> 6) * (64u * ra.leaf_slots) + lane
        : nullptr;

    // wave-uniform work cursor
    uint32_t tile_x0 = 0, tile_row0 = 0, tile_slot0 = 0, ds0 = 0, items_per_batch = 0, cursor = 0;    // cursor == items_per_batch: batch used up
    bool exhausted = false;

    Path p;
    p.remain = 0u;
    bool has_path = false, stocked = false;
    uint32_t out_idx = 0, stock_idx = 0;                                  // radiance record of the sample (radiance_slot below): < 2^32 (see streamed_chunk_spp)
    Ray stock_ray;
    Rng stock_rng;
    uint32_t n_samples = 0, n_rays = 0;
    Counters<STATS> ctr;
    // resumable walks (rt_path.h walk_compact, walk_fast_lds): a round's walk phase ends once only a few lanes still walk; they carry
    // their walk into the next round (100 k spheres: 324 trips per round for 222 box steps per ray, +10 %; random-spheres: see DESIGN 13)
    constexpr bool kResumable = !STATS && (WALK == WALK_COMPACT || WALK == WALK_LDS_STACK);
    bool walking = false;                                                         // this lane's walk is parked in its leaf stack (kResumable only)

    TRT_CLK_START(ctr);
    for (;;) {
        // ---- take items ahead: whenever a lane has neither a path nor a stocked ray, every lane without stock takes one ----
        if (!exhausted && __builtin_amdgcn_ballot_w64(!has_path && !stocked) != 0ull) {
            if (cursor >= items_per_batch) {
                uint32_t b = 0;
                if (lane == 0u) b = atomicAdd(batch_counter, 1u);
                b = __builtin_amdgcn_readfirstlane(b);
                if (b >= n_batches) {
                    exhausted = true;
                } else {
                    const uint32_t tile = b % n_tiles;                                // consecutive batches: neighbouring tiles, same samples
                    ds0 = (b / n_tiles) * batch_spp;
                    tile_x0 = (tile % tiles_x) * 8u;
                    tile_row0 = (tile / tiles_x) * 8u;
                    tile_slot0 = tile * n_spp * 64u;                                  // the tile's radiance records (radiance_slot)
                    items_per_batch = 64u * (n_spp - ds0 < batch_spp ? n_spp - ds0 : batch_spp);
                    cursor = 0;
                }
            }
            if (!exhausted) {
                const uint64_t want = __builtin_amdgcn_ballot_w64(!stocked);
                const uint32_t item = cursor + st_rank(want);
                cursor += (uint32_t)__builtin_popcountll(want);
                if (!stocked && item < items_per_batch) {
                    const uint32_t l = item & 63u, ds = ds0 + (item >> 6);         // neighbouring items = neighbouring pixels, same sample
                    const uint32_t x = tile_x0 + (l & 7u), row = tile_row0 + (l >> 3);
                    if (x < cam.width && row < ra.rows_local) {
                        if constexpr (STATS) { if (first_active_lane()) ctr.w_gen++; }
                        const uint32_t y = image_row(ra, row);
                        stock_rng = rng_seed(ra.seed_key, y * cam.width + x, ra.sample_begin + ds);
                        stock_ray = primary_ray(cam, x, y, stock_rng);
                        stock_idx = tile_slot0 + ds * 64u + l;
                        stocked = true;
                    }
                }
            }
        }
        if (!has_path && stocked) {                                               // cpu.rs:42-45
            p.ray = stock_ray;
            p.rng = stock_rng;
            p.color = v3(0.0f, 0.0f, 0.0f);
            p.atten = v3(1.0f, 1.0f, 1.0f);
            p.remain = ra.max_bounces;
            out_idx = stock_idx;
            has_path = true;
            stocked = false;
            n_samples++;
        }
        if (__builtin_amdgcn_ballot_w64(has_path) == 0ull) {
            if (exhausted) break;
            continue;                                                             // a batch of off-image items: fetch the next
        }
        TRT_CLK(ctr, 0);
        if (has_path) {
            if constexpr (STATS) { if (first_active_lane()) ctr.w_rounds++; }
            if constexpr (kResumable) {
                // per-lane tree walk: a lane whose walk is still under way when most of the wave has finished carries it into the
                // next round (rt_path.h walk_compact) and is not shaded in this one
                Trav tr = trav_begin<MODE, WALK == WALK_COMPACT>(sc, p.ray, false);                              // every lane: a new walk, or the frame of a parked one
                if (walking) trav_unpark(leaf_stack, tr); else n_rays++;
                const uint32_t entered = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
                walking = !closest_hit_resume<MODE, STATS, WALK>(sc, p.ray, tr, ctr, ra.leaf_slots, leaf_stack, leaf_list, nodes16, ra.stragglers, entered);
                if (!walking) {
                    if (shade_hit<MODE, STATS, LAZY>(sc, p, tr.prim_best, tr.t_best, background, ctr)) {
                        radiance_store(colors, out_idx, p.color);
                        has_path = false;
                    }
                }
            } else {
                n_rays++;
                float t;
                const uint32_t prim = closest_hit<MODE, STATS, WALK>(sc, p.ray, STATS && ra.ref_tree != 0u, t, ctr, ra.leaf_slots, leaf_stack, leaf_list, nodes16);
                if (shade_hit<MODE, STATS, LAZY>(sc, p, prim, t, background, ctr)) {
                    radiance_store(colors, out_idx, p.color);
                    has_path = false;
                }
            }
        }
        TRT_CLK(ctr, 3);
    }
    flush_counters<STATS>(counters, n_samples, n_rays, ctr);
}


// ------------------------------------------------------------------------------------------------------------------
// Same kernel with the primary rays of a whole wave generated at once.  In the kernel above a lane keeps one primary
// ray in stock (9 VGPRs) and the generation code runs whenever some lane is out of stock: 0.45 times per bounce
// round with 30 % of the lanes busy.  Here an empty per-wave pool in LDS (64 entries x 36 bytes behind the leaf
// stack) is refilled by ALL 64 lanes - the next 64 items of the batch, off-image ones dropped, entries compacted by
// ballot rank - and lanes whose path ended take entries by rank: generation runs once per ~7 rounds with every lane
// busy, and the stock registers are gone.  Which lane traces which sample changes; every sample's radiance does not
// (RNG keyed by pixel and sample, radiance stored per sample, folded in order).  Used where the pool costs no resident
// workgroup (small LDS scene copies, scenes in global memory).
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kPoolDwords = 9u;         // origin, direction, rng state, radiance slot

template <int MODE, bool STATS, int MINW = 1, int THREADS = 256, int WALK = WALK_RUNTIME, bool LAZY = false>
__global__ __launch_bounds__(THREADS, MINW) void stream_pool_kernel(SceneDev scd, CameraDev cam, RenderArgs ra,
                                                                           float* __restrict__ colors,
                                                                           uint32_t* __restrict__ batch_counter,
                                                                           unsigned long long* __restrict__ counters,
                                                                           uint32_t tiles_x, uint32_t n_tiles, uint32_t n_batches, uint32_t batch_spp,
                                                                           const float4* __restrict__ leaf_list,
                                                                           const uint4* __restrict__ nodes16) {
    stage_scene_to_lds<MODE>(scd);
    const SceneAcc<MODE> sc{scd.blob, scd.L};
    const uint32_t lane = threadIdx.x & 63u;
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);
    const uint32_t n_spp = ra.sample_end - ra.sample_begin;
    char* const lds_tail = reinterpret_cast<char*>(g_lds) + ((sc.lds_bytes() + 15u) & ~15u);
    float2* const leaf_stack = reinterpret_cast<float2*>(lds_tail) + (threadIdx.x >> 6) * (64u * ra.leaf_slots) + lane;
    // the pool: field f of entry e at pool[f * 64 + e]
    uint32_t* const pool = reinterpret_cast<uint32_t*>(lds_tail + (size_t)THREADS * ra.leaf_slots * sizeof(float2)) + (threadIdx.x >> 6) * (64u * kPoolDwords);

    uint32_t tile_x0 = 0, tile_row0 = 0, tile_slot0 = 0, ds0 = 0, items_per_batch = 0, cursor = 0;    // wave-uniform work cursor
    uint32_t pool_head = 0, pool_count = 0;                                            // wave-uniform
    bool exhausted = false;

    Path p;
    p.remain = 0u;
    bool has_path = false;
    uint32_t out_idx = 0;
    uint32_t n_samples = 0, n_rays = 0;
    Counters<STATS> ctr;
    // resumable walks (rt_path.h walk_compact, walk_fast_lds): a round's walk phase ends once only a few lanes still walk; they carry
    // their walk into the next round (100 k spheres: 324 trips per round for 222 box steps per ray, +10 %; random-spheres: see DESIGN 13)
    constexpr bool kResumable = !STATS && (WALK == WALK_COMPACT || WALK == WALK_LDS_STACK);
    bool walking = false;                                                         // this lane's walk is parked in its leaf stack (kResumable only)

    TRT_CLK_START(ctr);
    for (;;) {
        // ---- lanes without a path take pool entries; an empty pool is refilled by the whole wave ----
        for (;;) {
            const uint64_t need = __builtin_amdgcn_ballot_w64(!has_path);
            if (need == 0ull) break;
            if (pool_count == 0u) {
                if (exhausted) break;
                if (cursor >= items_per_batch) {
                    uint32_t b = 0;
                    if (lane == 0u) b = atomicAdd(batch_counter, 1u);
                    b = __builtin_amdgcn_readfirstlane(b);
                    if (b >= n_batches) { exhausted = true; break; }
                    const uint32_t tile = b % n_tiles;                                    // consecutive batches: neighbouring tiles, same samples
                    ds0 = (b / n_tiles) * batch_spp;
                    tile_x0 = (tile % tiles_x) * 8u;
                    tile_row0 = (tile / tiles_x) * 8u;
                    tile_slot0 = tile * n_spp * 64u;
                    items_per_batch = 64u * (n_spp - ds0 < batch_spp ? n_spp - ds0 : batch_spp);
                    cursor = 0;
                }
                // the next 64 items: one 8x8 tile at one sample index
                const uint32_t item = cursor + lane;
                cursor += 64u;
                const uint32_t ds = ds0 + (item >> 6);
                const uint32_t x = tile_x0 + (lane & 7u), row = tile_row0 + (lane >> 3);
                const bool valid = x < cam.width && row < ra.rows_local;                  // off-image items of an edge tile are dropped
                const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
                if (valid) {
                    if constexpr (STATS) { if (first_active_lane()) ctr.w_gen++; }
                    const uint32_t y = image_row(ra, row);
                    Rng rng = rng_seed(ra.seed_key, y * cam.width + x, ra.sample_begin + ds);        // cpu.rs:42-45
                    const Ray ray = primary_ray(cam, x, y, rng);
                    const uint32_t e = st_rank(vmask);
                    pool[0u * 64u + e] = __float_as_uint(ray.o.x); pool[1u * 64u + e] = __float_as_uint(ray.o.y); pool[2u * 64u + e] = __float_as_uint(ray.o.z);
                    pool[3u * 64u + e] = __float_as_uint(ray.d.x); pool[4u * 64u + e] = __float_as_uint(ray.d.y); pool[5u * 64u + e] = __float_as_uint(ray.d.z);
                    pool[6u * 64u + e] = rng.s0; pool[7u * 64u + e] = rng.s1;
                    pool[8u * 64u + e] = tile_slot0 + ds * 64u + lane;
                }
                pool_head = 0u;
                pool_count = (uint32_t)__builtin_popcountll(vmask);
                if (pool_count == 0u) continue;                                           // a tile row entirely off the image
            }
            const uint32_t rank = st_rank(need);
            if (!has_path && rank < pool_count) {
                const uint32_t e = pool_head + rank;
                p.ray.o = v3(__uint_as_float(pool[0u * 64u + e]), __uint_as_float(pool[1u * 64u + e]), __uint_as_float(pool[2u * 64u + e]));
                p.ray.d = v3(__uint_as_float(pool[3u * 64u + e]), __uint_as_float(pool[4u * 64u + e]), __uint_as_float(pool[5u * 64u + e]));
                p.rng.s0 = pool[6u * 64u + e]; p.rng.s1 = pool[7u * 64u + e];
                out_idx = pool[8u * 64u + e];
                p.color = v3(0.0f, 0.0f, 0.0f);
                p.atten = v3(1.0f, 1.0f, 1.0f);
                p.remain = ra.max_bounces;
                has_path = true;
                n_samples++;
            }
            const uint32_t wanted = (uint32_t)__builtin_popcountll(need);
            const uint32_t taken = wanted < pool_count ? wanted : pool_count;
            pool_head += taken;
            pool_count -= taken;
        }
        if (__builtin_amdgcn_ballot_w64(has_path) == 0ull) break;                         // nothing left to trace: the batches are used up
        TRT_CLK(ctr, 0);
        if (has_path) {
            if constexpr (STATS) { if (first_active_lane()) ctr.w_rounds++; }
            if constexpr (kResumable) {
                Trav tr = trav_begin<MODE, WALK == WALK_COMPACT>(sc, p.ray, false);                              // see stream_sample_kernel
                if (walking) trav_unpark(leaf_stack, tr); else n_rays++;
                const uint32_t entered = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
                walking = !closest_hit_resume<MODE, STATS, WALK>(sc, p.ray, tr, ctr, ra.leaf_slots, leaf_stack, leaf_list, nodes16, ra.stragglers, entered);
                if (!walking) {
                    if (shade_hit<MODE, STATS, LAZY>(sc, p, tr.prim_best, tr.t_best, background, ctr)) {
                        radiance_store(colors, out_idx, p.color);
                        has_path = false;
                    }
                }
            } else {
                n_rays++;
                float t;
                const uint32_t prim = closest_hit<MODE, STATS, WALK>(sc, p.ray, STATS && ra.ref_tree != 0u, t, ctr, ra.leaf_slots, leaf_stack, leaf_list, nodes16);
                if (shade_hit<MODE, STATS, LAZY>(sc, p, prim, t, background, ctr)) {
                    radiance_store(colors, out_idx, p.color);
                    has_path = false;
                }
            }
        }
        TRT_CLK(ctr, 3);
    }
    flush_counters<STATS>(counters, n_samples, n_rays, ctr);
}

// ------------------------------------------------------------------------------------------------------------------
// The pool kernel for scenes in global memory with TWO paths per lane (rt_path.h walk_compact2): slot A and slot B of a lane are two
// independent paths - each takes its primary rays from the wave's pool, walks, is shaded and stores its radiance exactly as the one
// path of stream_pool_kernel does - and a round steps both walks in one loop that keeps two node loads in flight per wave.  LDS per
// wave: two postponed-leaf stacks (slot A's, then slot B's) and the ray pool.  The launch plan takes it for scenes beyond L2 (trt_tuning.dual_walk = 0: by
// scene): -2.6 % where the tree fits L2, +2.7 % / +4.2 % on sphere_field 1 M / 4 M (rt_path.h box_loop_compact2 has the finding).  Which lane and which slot
// traces which sample changes; no sample's radiance does (RNG keyed by pixel and sample, radiance stored per sample, folded in order).
// ------------------------------------------------------------------------------------------------------------------
struct DualSlot {
    Path p;
    uint32_t out_idx = 0;
    bool has_path = false;
    bool walking = false;          // the slot's walk is parked in its leaf stack
};

template <int MINW, int THREADS = 256>
__global__ __launch_bounds__(THREADS, MINW) void stream_dual_kernel(SceneDev scd, CameraDev cam, RenderArgs ra,
                                                                           float* __restrict__ colors,
                                                                           uint32_t* __restrict__ batch_counter,
                                                                           unsigned long long* __restrict__ counters,
                                                                           uint32_t tiles_x, uint32_t n_tiles, uint32_t n_batches, uint32_t batch_spp,
                                                                           const float4* __restrict__ leaf_list,
                                                                           const uint4* __restrict__ nodes16) {
    constexpr int MODE = MODE_GLOBAL;
    const SceneAcc<MODE> sc{scd.blob, scd.L};
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);
    const uint32_t n_spp = ra.sample_end - ra.sample_begin;
    char* const lds_tail = reinterpret_cast<char*>(g_lds);
    float2* const stkA = reinterpret_cast<float2*>(lds_tail) + (2u * wave) * (64u * ra.leaf_slots) + lane;
    float2* const stkB = stkA + 64u * ra.leaf_slots;
    uint32_t* const pool = reinterpret_cast<uint32_t*>(lds_tail + (size_t)THREADS * 2u * ra.leaf_slots * sizeof(float2)) + wave * (64u * kPoolDwords);

    uint32_t tile_x0 = 0, tile_row0 = 0, tile_slot0 = 0, ds0 = 0, items_per_batch = 0, cursor = 0;    // wave-uniform work cursor
    uint32_t pool_head = 0, pool_count = 0;                                            // wave-uniform
    bool exhausted = false;
    uint32_t n_samples = 0, n_rays = 0;
    Counters<false> ctr;
    DualSlot A, B;
    A.p.remain = 0u; B.p.remain = 0u;

    // lanes whose slot has no path take pool entries; an empty pool is refilled by the whole wave (stream_pool_kernel)
    auto take = [&](DualSlot& S) {
        for (;;) {
            const uint64_t need = __builtin_amdgcn_ballot_w64(!S.has_path);
            if (need == 0ull) break;
            if (pool_count == 0u) {
                if (exhausted) break;
                if (cursor >= items_per_batch) {
                    uint32_t b = 0;
                    if (lane == 0u) b = atomicAdd(batch_counter, 1u);
                    b = __builtin_amdgcn_readfirstlane(b);
                    if (b >= n_batches) { exhausted = true; break; }
                    const uint32_t tile = b % n_tiles;
                    ds0 = (b / n_tiles) * batch_spp;
                    tile_x0 = (tile % tiles_x) * 8u;
                    tile_row0 = (tile / tiles_x) * 8u;
                    tile_slot0 = tile * n_spp * 64u;
                    items_per_batch = 64u * (n_spp - ds0 < batch_spp ? n_spp - ds0 : batch_spp);
                    cursor = 0;
                }
                const uint32_t item = cursor + lane;
                cursor += 64u;
                const uint32_t ds = ds0 + (item >> 6);
                const uint32_t x = tile_x0 + (lane & 7u), row = tile_row0 + (lane >> 3);
                const bool valid = x < cam.width && row < ra.rows_local;
                const uint64_t vmask = __builtin_amdgcn_ballot_w64(valid);
                if (valid) {
                    const uint32_t y = image_row(ra, row);
                    Rng rng = rng_seed(ra.seed_key, y * cam.width + x, ra.sample_begin + ds);        // cpu.rs:42-45
                    const Ray ray = primary_ray(cam, x, y, rng);
                    const uint32_t e = st_rank(vmask);
                    pool[0u * 64u + e] = __float_as_uint(ray.o.x); pool[1u * 64u + e] = __float_as_uint(ray.o.y); pool[2u * 64u + e] = __float_as_uint(ray.o.z);
                    pool[3u * 64u + e] = __float_as_uint(ray.d.x); pool[4u * 64u + e] = __float_as_uint(ray.d.y); pool[5u * 64u + e] = __float_as_uint(ray.d.z);
                    pool[6u * 64u + e] = rng.s0; pool[7u * 64u + e] = rng.s1;
                    pool[8u * 64u + e] = tile_slot0 + ds * 64u + lane;
                }
                pool_head = 0u;
                pool_count = (uint32_t)__builtin_popcountll(vmask);
                if (pool_count == 0u) continue;
            }
            const uint32_t rank = st_rank(need);
            if (!S.has_path && rank < pool_count) {
                const uint32_t e = pool_head + rank;
                S.p.ray.o = v3(__uint_as_float(pool[0u * 64u + e]), __uint_as_float(pool[1u * 64u + e]), __uint_as_float(pool[2u * 64u + e]));
                S.p.ray.d = v3(__uint_as_float(pool[3u * 64u + e]), __uint_as_float(pool[4u * 64u + e]), __uint_as_float(pool[5u * 64u + e]));
                S.p.rng.s0 = pool[6u * 64u + e]; S.p.rng.s1 = pool[7u * 64u + e];
                S.out_idx = pool[8u * 64u + e];
                S.p.color = v3(0.0f, 0.0f, 0.0f);
                S.p.atten = v3(1.0f, 1.0f, 1.0f);
                S.p.remain = ra.max_bounces;
                S.has_path = true;
                n_samples++;
            }
            const uint32_t wanted = (uint32_t)__builtin_popcountll(need);
            const uint32_t taken = wanted < pool_count ? wanted : pool_count;
            pool_head += taken;
            pool_count -= taken;
        }
    };
    // a finished walk: shade, store the radiance of a finished path; an unfinished one: park it in the slot's leaf stack
    auto settle = [&](DualSlot& S, Trav& tr, bool done, float2* stk) {
        if (!S.has_path) return;
        if (!done) { trav_park(stk, tr); S.walking = true; return; }
        S.walking = false;
        if (shade_hit<MODE, false, true>(sc, S.p, tr.prim_best, tr.t_best, background, ctr)) {
            radiance_store(colors, S.out_idx, S.p.color);
            S.has_path = false;
        }
    };

    TRT_CLK_START(ctr);
    for (;;) {
        take(A);
        take(B);
        if (__builtin_amdgcn_ballot_w64(A.has_path || B.has_path) == 0ull) break;          // the batches are used up
        TRT_CLK(ctr, 0);
        Trav trA = trav_begin<MODE, true>(sc, A.p.ray, false), trB = trav_begin<MODE, true>(sc, B.p.ray, false);   // a new walk, or the frame of a parked one (fused-slab domain: rt_path.h)
        if (A.has_path) { if (A.walking) trav_unpark(stkA, trA); else n_rays++; }
        if (B.has_path) { if (B.walking) trav_unpark(stkB, trB); else n_rays++; }
        const uint32_t entered = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(A.has_path)) +
                                 (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(B.has_path));
        bool doneA = false, doneB = false;
        walk_compact2<MODE>(sc, nodes16, leaf_list, A.p.ray, trA, A.has_path, B.p.ray, trB, B.has_path, ctr, stkA, stkB, ra.leaf_slots, ra.stragglers,
                            entered, doneA, doneB);
        settle(A, trA, doneA, stkA);
        settle(B, trB, doneB, stkB);
        TRT_CLK(ctr, 3);
    }
    flush_counters<false>(counters, n_samples, n_rays, ctr);
}

// pixels[idx] += color * (1/spp), samples in order (imager.rs:35,50).
// Radiance records are TILE-MAJOR (round 4): record (tile, sample, lane) at ((tile * n_spp + sample) * 64 + lane) x 12 bytes, tile = the 8x8-pixel
// tile of the work batches, lane = (row % 8) * 8 + x % 8.  The 64 records of a tile and sample - what one wave finishes within one batch - are 768
// contiguous bytes, twelve whole 64-byte lines that no other wave writes; in rounds 1-3 ([sample][pixel]) they were eight runs of 96 bytes, each
// sharing its lines with neighbouring tiles that other waves finish at other times, and the L2 wrote the partial lines back: WRITE_SIZE 1.8x the
// records' bytes on Cornell, 4.3x on the 100 k-sphere scene (profiles/r03_*_pmc.json).  The fold reads a tile-sample as one 768-byte wave load.
__global__ __launch_bounds__(256) void stream_fold_kernel(const float* __restrict__ colors, float* __restrict__ accum, uint32_t width, uint32_t rows,
                                                          uint32_t tiles_x, uint32_t n_tiles, uint32_t n_spp, float inv_spp, uint32_t accumulate) {
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (tile >= n_tiles) return;
    const uint32_t x = (tile % tiles_x) * 8u + (lane & 7u), row = (tile / tiles_x) * 8u + (lane >> 3);
    if (x >= width || row >= rows) return;                                      // an edge tile's off-image lanes: records never written, never read
    float* out = accum + 3ull * ((unsigned long long)row * width + x);
    V3 acc = v3(0.0f, 0.0f, 0.0f);
    if (accumulate) acc = v3(out[0], out[1], out[2]);
    const float* rec = colors + 3ull * ((unsigned long long)tile * n_spp * 64ull + lane);
    for (uint32_t s = 0; s < n_spp; s++) {
        const Radiance c = *reinterpret_cast<const Radiance*>(rec + 3ull * 64ull * s);      // one global_load_dwordx3
        acc = acc + v3(c.r, c.g, c.b) * inv_spp;
    }
    out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
}

// Samples per pixel per tracing / fold launch pair: as many as keep the radiance buffer within `radiance_gb` GiB (default 16; 16..256 spp).
// Sized for 288 GB of HBM: a launch is a persistent grid that drains a batch queue, and its tail - the last waves finishing their longest
// paths while the rest of the chip idles - is paid once per launch; the Cornell bench frame went from four 64-spp launches per 256-spp
// step (4 GB) to one (+1.2 %), random-spheres 1080p from two to one (+2.6 %; profiles/r03_radiance_budget_sweep.txt).
// Pixels of the 8x8 tiles that cover the image: the radiance buffer holds a record per tile lane (tile-major layout, stream_fold_kernel).
static unsigned long long padded_pixels(uint32_t width, uint32_t rows) { return 64ull * ((width + 7u) / 8u) * ((rows + 7u) / 8u); }

uint32_t streamed_chunk_spp(uint32_t width, uint32_t rows, uint32_t radiance_gb) {
    const unsigned long long px = padded_pixels(width, rows);
    const unsigned long long budget = (unsigned long long)(radiance_gb >= 1u && radiance_gb <= 64u ? radiance_gb : 16u) << 30;
    const unsigned long long fit = px ? budget / (px * 12ull) : 256ull;
    uint32_t c = 16;
    while (c < 256u && 2ull * c <= fit) c *= 2u;                                  // power of two in 16..256
    while (c > 1u && px * c >= (1ull << 32)) c /= 2u;                             // radiance slots are indexed with 32 bits
    return c;
}
// Device scratch of one render: [batch counter, 256 bytes] [radiance records: 12 bytes per pixel and sample of a launch].  A render of fewer
// samples than a full launch holds takes only what it needs.
constexpr size_t kWorkspaceHeader = 256;
size_t streamed_workspace_bytes(uint32_t width, uint32_t rows, uint32_t samples, uint32_t radiance_gb) {
    const uint32_t chunk = streamed_chunk_spp(width, rows, radiance_gb);
    return kWorkspaceHeader + (size_t)padded_pixels(width, rows) * (samples < chunk ? (samples ? samples : 1u) : chunk) * 3 * sizeof(float);
}
// ... and the other way round: the samples per pixel a granted workspace holds (at most 256, and few enough for 32-bit record indices).
// A render that was granted less than it asked for (device memory short: capi.hip halves the request) runs more, shorter launches.
uint32_t streamed_chunk_that_fits(uint32_t width, uint32_t rows, size_t bytes) {
    const unsigned long long px = padded_pixels(width, rows);
    if (px == 0 || bytes <= kWorkspaceHeader) return 0;
    unsigned long long c = (bytes - kWorkspaceHeader) / (px * 12ull);
    if (c > 256ull) c = 256ull;
    while (c > 1ull && px * c >= (1ull << 32)) c--;
    return (uint32_t)c;
}

namespace {

constexpr size_t kLdsPerCu = 160u * 1024u;
inline size_t align16(size_t b) { return (b + 15u) & ~(size_t)15u; }

// One kernel instantiation the launcher may pick.  Every instantiation has the same parameter list, so a launch is
// hipLaunchKernel(fn, ...) with one argument array whichever entry is chosen.
struct KernelEntry {
    const void* fn;
    int mode, threads, minw;        // template arguments: scene mode, lanes per workgroup, launch bound (waves per SIMD)
    int walk;                       // WALK_RUNTIME: the kernel picks the walk from its arguments (every knob, counting variants)
    bool pool, stats, lazy;
    bool dual = false;              // stream_dual_kernel: two paths per lane (MODE_GLOBAL, 16-byte nodes, ray pool)
};
#define TRT_POOL(MODE, STATS, MINW, THREADS, WALK, LAZY) \
    KernelEntry{reinterpret_cast<const void*>(&stream_pool_kernel<MODE, STATS, MINW, THREADS, WALK, LAZY>), MODE, THREADS, MINW, WALK, true, STATS, LAZY}
#define TRT_SAMPLE(MODE, STATS, MINW, THREADS, WALK, LAZY) \
    KernelEntry{reinterpret_cast<const void*>(&stream_sample_kernel<MODE, STATS, MINW, THREADS, WALK, LAZY>), MODE, THREADS, MINW, WALK, false, STATS, LAZY}

#define TRT_DUAL(MINW) \
    KernelEntry{reinterpret_cast<const void*>(&stream_dual_kernel<MINW, 256>), MODE_GLOBAL, 256, MINW, WALK_COMPACT, true, false, true, true}

// production launches: walk fixed at compile time, lazy colour, no counters
const KernelEntry kSpecialised[] = {
    TRT_POOL(MODE_LDS, false, 6, 256, WALK_FLAT, true),
    TRT_POOL(MODE_LDS, false, 7, 256, WALK_FLAT, true),
    TRT_POOL(MODE_LDS, false, 8, 256, WALK_FLAT, true),
    TRT_POOL(MODE_LDS, false, 6, 256, WALK_LDS_STACK, true),
    TRT_SAMPLE(MODE_LDS, false, 6, 512, WALK_REGS, true),
    TRT_SAMPLE(MODE_LDS, false, 6, 768, WALK_LDS_STACK, true),
    TRT_POOL(MODE_GLOBAL, false, 8, 256, WALK_COMPACT, true),
    TRT_POOL(MODE_GLOBAL, false, 7, 256, WALK_COMPACT, true),     // 72 VGPRs: nothing spilled (at 8 waves the fused slab arithmetic spills ten registers)
    TRT_DUAL(4), TRT_DUAL(5), TRT_DUAL(6), TRT_DUAL(7), TRT_DUAL(8),
};
// every other knob combination and the counting variants: runtime choice of the walk
const KernelEntry kGeneral[] = {
    TRT_SAMPLE(MODE_LDS, false, 6, 768, WALK_RUNTIME, false),    TRT_SAMPLE(MODE_LDS, true, 6, 768, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_LDS, false, 6, 512, WALK_RUNTIME, false),    TRT_SAMPLE(MODE_LDS, true, 6, 512, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_LDS, false, 5, 512, WALK_RUNTIME, false),    TRT_SAMPLE(MODE_LDS, true, 5, 512, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_LDS, false, 7, 256, WALK_RUNTIME, false),    TRT_SAMPLE(MODE_LDS, true, 7, 256, WALK_RUNTIME, false),
    TRT_POOL(MODE_LDS, false, 6, 256, WALK_RUNTIME, false),      TRT_POOL(MODE_LDS, true, 6, 256, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_LDS, false, 6, 256, WALK_RUNTIME, false),    TRT_SAMPLE(MODE_LDS, true, 6, 256, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_LDS, false, 5, 256, WALK_RUNTIME, false),    TRT_SAMPLE(MODE_LDS, true, 5, 256, WALK_RUNTIME, false),
    TRT_POOL(MODE_GLOBAL, false, 8, 256, WALK_RUNTIME, false),   TRT_POOL(MODE_GLOBAL, true, 8, 256, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_GLOBAL, false, 8, 256, WALK_RUNTIME, false), TRT_SAMPLE(MODE_GLOBAL, true, 8, 256, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_GLOBAL, false, 7, 256, WALK_RUNTIME, false), TRT_SAMPLE(MODE_GLOBAL, true, 7, 256, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_GLOBAL, false, 6, 256, WALK_RUNTIME, false), TRT_SAMPLE(MODE_GLOBAL, true, 6, 256, WALK_RUNTIME, false),
    TRT_SAMPLE(MODE_GLOBAL, false, 1, 256, WALK_RUNTIME, false), TRT_SAMPLE(MODE_GLOBAL, true, 1, 256, WALK_RUNTIME, false),
};
#undef TRT_POOL
#undef TRT_SAMPLE
#undef TRT_DUAL

const KernelEntry* find_general(int mode, int threads, int minw, bool pool, bool stats) {
    for (const KernelEntry& k : kGeneral)
        if (k.mode == mode && k.threads == threads && k.minw == minw && k.pool == pool && k.stats == stats && !k.dual) return &k;
    return nullptr;
}

}  // namespace

// How a scene is launched: workgroup shape, waves per SIMD, where the postponed leaves live, which walk, which kernel.
// Everything the kernel assumes about its LDS (scene copy | leaf stack: threads x slots x 8 bytes | ray pool: 36 bytes per lane)
// is decided HERE and nowhere else; trt_streamed_launch_plan exposes the result, and the CPU tests check its invariants for
// every scene size and every knob (tests/test_host_boundary.py).
StreamLaunchPlan streamed_launch_plan(const SceneLayout& L, const RenderArgs& ra_all, const trt_tuning& tn, bool stats) {
    StreamLaunchPlan pl{};
    const size_t scene_bytes = scene_lds_bytes(L);
    const int mode = scene_mode(L);
    // waves per SIMD / lanes per workgroup: 256-lane workgroups for small LDS copies; a big LDS copy (> 20 KB) is shared by
    // more waves: 768 lanes (12 waves, two workgroups per CU = 6 waves per SIMD) if the copy and a 4-slot LDS leaf stack
    // fit twice into the CU's 160 KB (random-spheres: 49.6 + 24 KB; 19.0 Gray/s against 18.0 for 512 lanes with register
    // slots and 18.7 for 1024 lanes at 8 waves per SIMD, one box: tools/sweep_rs.sh), else 512 lanes with the slots in registers;
    // 8 waves per SIMD for scenes read from global memory
    int threads = 256;
    if (mode == MODE_LDS && L.hot_bytes > 20u * 1024u) {
        threads = (align16(scene_bytes) + 768u * 4u * sizeof(float2)) * 2u <= kLdsPerCu && ra_all.lds_leaf_stack != 0u ? 768 : 512;
        if (tn.stream_big_threads == 512u || tn.stream_big_threads == 768u) threads = (int)tn.stream_big_threads;
    }
    // LDS scenes: 6 waves per SIMD.  Scenes in global memory: rounds 1-4 ran them at 8 (100 k spheres: 2.03 Gray/s at 5 waves, 2.25 at 6, 2.29 at 8 -
    // measured with NaN rays walking the whole tree in every launch: rt_path.h ray_has_nan)
    // (round 5, without the NaN-ray tail and with the fused slab arithmetic: a tree that fits the chip's 32 MiB of L2 runs 5 % faster at 7 waves
    // with no spilled register than at 8 with ten; a scene whose walk waits for memory - sphere_field, 1 M spheres and up - wants the eighth wave:
    // profiles/r05_waves7_ab.txt)
    // Two paths per lane (stream_dual_kernel; trt_tuning.dual_walk: 0 by scene, 1 wherever the kernel exists, 2 never): with the fused box step in its loop
    // it LOSES 2.6 % where the tree fits L2 and WINS 2.7 % / 4.2 % on sphere_field 1 M / 4 M at 6 waves - where the walk's loads really miss, a second
    // node load in flight per lane pays (profiles/r05_dual_walk_fused_ab.txt): by scene = beyond L2, when nothing else about the launch was asked for.
    const bool beyond_l2 = mode == MODE_GLOBAL && L.hot_bytes > (32u << 20);
    const bool want_dual = mode == MODE_GLOBAL && (tn.dual_walk == 1u || (tn.dual_walk == 0u && beyond_l2 && tn.stream_waves_per_simd == 0u && !stats && !ra_all.ref_tree &&
                                                                           L.lazy_color && tn.runtime_walk == 0u && tn.ray_pool != 0u && ra_all.lds_leaf_stack != 0u &&
                                                                           L.off_compact != 0u));
    int w = mode == MODE_LDS ? 6 : (beyond_l2 ? (want_dual ? 6 : 8) : 7);
    if (tn.stream_waves_per_simd) w = (int)tn.stream_waves_per_simd;
    if (w < 5 && !(want_dual && w == 4)) w = 5;      // (4: the two-path kernel only - eight rays per SIMD like 8 x 1)
    if (w > 8) w = 8;
    if (threads == 512 && w > 6) w = 6;
    if (threads == 768) w = 6;
    uint32_t wg_per_cu = (uint32_t)(w * 4 * 64 / threads);
    // slots of the LDS stack: 4 for tree walks (5 in the 768-lane plan if they fit: random-spheres +3 %, profiles/r03_defaults_sweep.txt);
    // 7 for the lock-step leaf list, whose t_best stays stale for a whole walk (Cornell 34.0 Gray/s at 4, 35.1 at 6..12) and which
    // steps two leaves per trip, so a lane must have two free
    const bool flat = L.flat_walk && !ra_all.ref_tree;
    uint32_t slots = ra_all.leaf_slots == 0u ? (flat ? 7u : (threads == 768 ? 5u : 4u)) : (ra_all.leaf_slots > kLdsLeafSlotsMax ? kLdsLeafSlotsMax : ra_all.leaf_slots);
    if (flat && slots < 2u) slots = 2u;                                         // walk_flat pushes up to two leaves per trip
    if (flat && ra_all.leaf_slots == 0u && mode == MODE_LDS && threads == 256) {
        // the default depth gives way to occupancy: the deepest stack (<= 7, >= 4) with which stack + ray pool + scene copy of
        // all the CU's workgroups fit its 160 KB of LDS (Cornell: 7 slots at 6 waves per SIMD, 5 at 7, 4 at 8)
        const size_t pool_b = (size_t)threads / 64u * 64u * kPoolDwords * sizeof(uint32_t);
        while (slots > 4u && (align16(scene_bytes) + (size_t)threads * slots * sizeof(float2) + pool_b) * wg_per_cu > kLdsPerCu) slots--;
    }
    if (threads > 512 && ra_all.leaf_slots == 0u) {
        while (slots > 3u && (align16(scene_bytes) + (size_t)threads * slots * sizeof(float2)) * wg_per_cu > kLdsPerCu) slots--;
    }
    const size_t stack_bytes = (size_t)threads * slots * sizeof(float2);
    const size_t with_stack = align16(scene_bytes) + stack_bytes;
    // LDS: scene copy + the postponed-leaf stack (8 bytes per lane and slot), the latter only where it does not cost a
    // resident workgroup (random-spheres: 49.6 KB scene copy, 3 workgroups of 512 lanes per CU without it, 2 with it:
    // measured 7 % slower than register slots); otherwise the slots are registers
    bool lds_stack = ra_all.lds_leaf_stack != 0u && with_stack <= kLdsPerCu;
    if (lds_stack && ra_all.lds_leaf_stack != 2u) {
        const uint32_t fit_plain = scene_bytes ? (uint32_t)(kLdsPerCu / scene_bytes) : wg_per_cu;
        const uint32_t fit_stack = (uint32_t)(kLdsPerCu / with_stack);
        lds_stack = (fit_stack < wg_per_cu ? fit_stack : wg_per_cu) >= (fit_plain < wg_per_cu ? fit_plain : wg_per_cu);
    }
    // few primitives: lock-step leaf list (rt_path.h walk_flat); scenes read from global memory: 16-byte culling nodes
    // (walk_compact).  Both need the LDS stack.
    const bool compact = lds_stack && mode == MODE_GLOBAL && L.off_compact != 0u && !ra_all.ref_tree;
    // per-wave pool of primary rays (stream_pool_kernel): needs the LDS stack and 256-lane workgroups (LDS scenes at 6
    // waves per SIMD and more, global-memory scenes at 8), and must not cost a resident workgroup either
    const size_t pool_bytes = (size_t)threads / 64u * 64u * kPoolDwords * sizeof(uint32_t);
    bool pool = lds_stack && threads == 256 && !ra_all.ref_tree && ((mode == MODE_LDS && w >= 6) || (mode == MODE_GLOBAL && (w >= 7 || want_dual)));
    pool = pool && tn.ray_pool != 0u;
    if (pool) pool = (uint32_t)(kLdsPerCu / (with_stack + pool_bytes)) >= wg_per_cu;

    // ---- the kernel instantiation ----
    const int walk = compact ? WALK_COMPACT : (flat && lds_stack) ? WALK_FLAT : lds_stack ? WALK_LDS_STACK : WALK_REGS;
    const bool slots_ok = lds_stack || ra_all.leaf_slots == 0u || ra_all.leaf_slots >= 4u;      // WALK_REGS has 4 register slots
    const bool specialise = !stats && slots_ok && L.lazy_color && tn.runtime_walk == 0u;
    // two paths per lane (stream_dual_kernel): scenes in global memory on 16-byte nodes with the ray pool; two leaf stacks per lane, as deep
    // as fit beside the pool (2..4 slots: a parked walk needs two)
    bool dual = want_dual && specialise && compact && pool && threads == 256;
    if (dual) {
        uint32_t ds = ra_all.leaf_slots == 0u ? 4u : (ra_all.leaf_slots < 2u ? 2u : (ra_all.leaf_slots > kLdsLeafSlotsMax ? kLdsLeafSlotsMax : ra_all.leaf_slots));
        while (ds > 2u && (align16(scene_bytes) + 2u * (size_t)threads * ds * sizeof(float2) + pool_bytes) * wg_per_cu > kLdsPerCu) ds--;
        if ((align16(scene_bytes) + 2u * (size_t)threads * ds * sizeof(float2) + pool_bytes) * wg_per_cu > kLdsPerCu) dual = false;
        else slots = ds;
    }
    const size_t stack_total = dual ? 2u * (size_t)threads * slots * sizeof(float2) : stack_bytes;
    const KernelEntry* k = nullptr;
    if (specialise) {
        for (const KernelEntry& e : kSpecialised)
            if (e.mode == mode && e.threads == threads && e.minw == w && e.pool == pool && e.walk == walk && e.dual == dual) { k = &e; break; }
    }
    if (!k) dual = false;
    if (!dual && w < 5) { w = 5; wg_per_cu = (uint32_t)(w * 4 * 64 / threads); }             // four waves per SIMD exist for the two-path kernel only
    if (!dual && want_dual && tn.dual_walk == 0u) { w = 8; wg_per_cu = (uint32_t)(w * 4 * 64 / threads); }      // the plan's own wish fell through: the one-path shape of a scene beyond L2
    pl.specialised = k != nullptr;
    if (!k) {
        // the general instantiations exist for fewer (waves, pool) combinations than the knobs can ask for: take the nearest one
        // BELOW the request and make the plan follow the kernel, so that grid and LDS are sized for what really runs
        bool kpool = pool;
        int kw = w;
        if (mode == MODE_LDS && threads == 256) {
            if (w >= 7) { kw = 7; kpool = false; }                  // 7 waves: sample kernel only
            else if (w == 6) kw = 6;
            else { kw = 5; kpool = false; }
        } else if (mode == MODE_LDS && threads == 512) { kw = w >= 6 ? 6 : 5; kpool = false; }
        else if (mode == MODE_LDS) { kw = 6; kpool = false; }
        else {                                                      // MODE_GLOBAL
            if (w >= 8) kw = 8;
            else if (w == 7) { kw = 7; kpool = false; }
            else if (w == 6) { kw = 6; kpool = false; }
            else { kw = 1; kpool = false; }
        }
        k = find_general(mode, threads, kw, kpool, stats);
        pool = kpool;
        if (kw >= 5 && kw < w) { w = kw; wg_per_cu = (uint32_t)(w * 4 * 64 / threads); }
    }
    const size_t lds_bytes = lds_stack ? align16(scene_bytes) + (dual ? stack_total : stack_bytes) + (pool ? pool_bytes : 0u) : scene_bytes;
    if (lds_bytes) { const uint32_t by_lds = (uint32_t)(kLdsPerCu / lds_bytes); if (by_lds < wg_per_cu) wg_per_cu = by_lds ? by_lds : 1u; }
    pl.mode = mode; pl.threads = threads; pl.waves_per_simd = w; pl.wg_per_cu = wg_per_cu; pl.slots = slots;
    pl.lds_stack = lds_stack; pl.flat = flat && lds_stack; pl.compact = compact; pl.pool = pool; pl.walk = walk; pl.dual = dual;
    pl.lds_bytes = lds_bytes; pl.scene_lds_bytes = scene_bytes;
    pl.kernel = k ? k->fn : nullptr;
    pl.kernel_minw = k ? k->minw : 0; pl.kernel_threads = k ? k->threads : 0; pl.kernel_walk = k ? k->walk : 0; pl.kernel_pool = k ? k->pool : false;
    pl.kernel_stats = k ? k->stats : false;
    pl.kernel_name = dual ? "trt::stream_dual_kernel" : pool ? "trt::stream_pool_kernel" : "trt::stream_sample_kernel";
    return pl;
}

// Name of the kernel that dominates a streamed render of this scene (for profiles and bench.py's roofline line).
const char* streamed_kernel_name(const SceneLayout& L, const RenderArgs& ra, const trt_tuning& tn) { return streamed_launch_plan(L, ra, tn, false).kernel_name; }

hipError_t launch_streamed(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra_all, const trt_tuning& tn, void* workspace, size_t workspace_bytes,
                           float* d_accum, unsigned long long* d_counters, bool stats, hipStream_t stream) {
    if (ra_all.rows_local == 0 || cam.width == 0) return hipSuccess;
    uint32_t* batch_counter = static_cast<uint32_t*>(workspace);                                     // layout: streamed_workspace_bytes
    float* colors = reinterpret_cast<float*>(static_cast<char*>(workspace) + kWorkspaceHeader);
    const uint32_t samples_all = ra_all.sample_end - ra_all.sample_begin;
    uint32_t chunk_full = streamed_chunk_spp(cam.width, ra_all.rows_local, tn.radiance_gb);
    const uint32_t granted = streamed_chunk_that_fits(cam.width, ra_all.rows_local, workspace_bytes);      // the scratch in hand decides, not the wish
    if (granted == 0u) return hipErrorInvalidValue;
    if (chunk_full > granted) chunk_full = granted;
    const uint32_t chunk = samples_all < chunk_full ? (samples_all ? samples_all : 1u) : chunk_full;
    uint32_t tiles_x = (cam.width + 7u) / 8u;
    const uint32_t tiles_y = (ra_all.rows_local + 7u) / 8u;
    uint32_t n_tiles = tiles_x * tiles_y;
    int dev = 0, cus = 256;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const StreamLaunchPlan pl = streamed_launch_plan(sc.L, ra_all, tn, stats);
    if (pl.kernel == nullptr) return hipErrorInvalidDeviceFunction;               // no instantiation for this plan: a bug, never a fallback
    // the kernel's assumptions about its dynamic LDS, checked where the launch is made (scene copy | leaf stack | ray pool)
    {
        size_t need = pl.scene_lds_bytes;
        if (pl.lds_stack) need = align16(need) + (pl.dual ? 2u : 1u) * (size_t)pl.threads * pl.slots * sizeof(float2) + (pl.pool ? (size_t)pl.threads * kPoolDwords * sizeof(uint32_t) : 0u);
        if (need != pl.lds_bytes || pl.lds_bytes > kLdsPerCu || pl.kernel_threads != pl.threads || (pl.pool && !pl.lds_stack) ||
            (pl.flat && pl.slots < 2u) || (pl.kernel_pool != pl.pool)) return hipErrorInvalidConfiguration;
    }
    SceneDev scd = sc;
    CameraDev camd = cam;
    const float4* leaf_list = (pl.flat || pl.compact) ? sc.blob + sc.L.off_leaf_list : nullptr;
    const uint4* nodes16 = pl.compact ? reinterpret_cast<const uint4*>(sc.blob + sc.L.off_compact) : nullptr;
    const int walk_of_plan = pl.walk;
    const uint32_t resident = (uint32_t)cus * pl.wg_per_cu;
    const uint32_t waves_per_wg = (uint32_t)pl.threads / 64u;
    if (pl.lds_bytes > 48u * 1024u) {
        e = hipFuncSetAttribute(pl.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
    }
    bool first = true;
    for (uint32_t s0 = ra_all.sample_begin; s0 < ra_all.sample_end; s0 += chunk) {
        RenderArgs ra = ra_all;
        ra.lds_leaf_stack = pl.lds_stack ? 1u : 0u;
        if (pl.lds_stack) ra.leaf_slots = pl.slots;
        if (walk_of_plan == WALK_LDS_STACK) {                                   // the LDS tree walk's own straggler threshold
            ra.stragglers = tn.lds_stragglers;
        }
        if (!pl.lds_stack || pl.slots < 2u) ra.stragglers = 0u;                 // a parked walk occupies two slots of the lane's LDS leaf stack (rt_path.h trav_park)
        ra.sample_begin = s0;
        ra.sample_end = s0 + chunk < ra_all.sample_end ? s0 + chunk : ra_all.sample_end;
        uint32_t batch_spp = tn.stream_batch_spp ? tn.stream_batch_spp : kBatchSpp;
        uint32_t n_batches = n_tiles * ((ra.sample_end - ra.sample_begin + batch_spp - 1u) / batch_spp);
        uint32_t grid_x = resident;
        const uint32_t max_useful = (n_batches + waves_per_wg - 1u) / waves_per_wg;   // one batch per wave at least
        if (grid_x > max_useful) grid_x = max_useful ? max_useful : 1u;
        e = hipMemsetAsync(batch_counter, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        void* args[] = {&scd, &camd, &ra, &colors, &batch_counter, &d_counters, &tiles_x, &n_tiles, &n_batches, &batch_spp, &leaf_list, &nodes16};
        timing_mark(stream, true);
        e = hipLaunchKernel(pl.kernel, dim3(grid_x), dim3((uint32_t)pl.threads), args, pl.lds_bytes, stream);
        timing_mark(stream, false);
        if (e != hipSuccess) return e;
        const uint32_t fold_blocks = (n_tiles + 3u) / 4u;                           // four tiles (waves) per workgroup
        hipLaunchKernelGGL(stream_fold_kernel, dim3(fold_blocks), dim3(256), 0, stream, colors, d_accum, cam.width, ra_all.rows_local, tiles_x, n_tiles,
                           ra.sample_end - ra.sample_begin, ra_all.inv_spp, (first && !ra_all.accumulate) ? 0u : 1u);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        first = false;
    }
    return hipSuccess;
}

}  // namespace trt
