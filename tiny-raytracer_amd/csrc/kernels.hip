// kernels.hip — megakernel backend and the batch-sampler kernel, written for gfx950 (CDNA4) only.
//
// Design:
//  * one lane owns one pixel and walks its samples in order, so the running sum per pixel is the
//    reference's `pixels[idx] += color * (1/spp)` (imager.rs:50) in sample order: results are
//    bit-reproducible and independent of launch geometry, tiling or the number of GPUs;
//  * lanes are persistent: a lane whose path ended starts its next sample in the same loop trip in
//    which its neighbours trace their next bounce, so a wave never idles behind its longest path;
//  * traversal is stackless: the reference always descends left first (bvh.rs:96-106), so the
//    pre-order node array plus one skip link per node reproduces its visit order and its running
//    [t_min, t_best) interval exactly (rt_path.h);
//  * scenes up to kLdsSceneMaxBytes are copied into LDS once per workgroup and traversed from
//    there (ds_read_b128 per 16-byte plane element); larger ones are read through L1/L2.
#include <stdlib.h>

#include "kernels.h"
#include "rt_path.h"
#include "trt_pow.h"

namespace trt {

// ------------------------------------------------------------------------------------------------
// Megakernel.  Workgroup = THREADS lanes = a 16 x (THREADS/16) pixel tile; each wave owns an 8x8 sub-tile so the
// 64 paths of a wave see similar geometry.  A lane is done when its pixel has had samples
// [sample_begin, sample_end); the wave leaves the loop when every lane is done.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTileW = 16;          // tile height = THREADS / 16: 16 rows for 256 lanes, 32 for 512

template <int MODE, bool STATS, int MINW = 1, int THREADS = 256>
__global__ __launch_bounds__(THREADS, MINW) void megakernel(SceneDev scd, CameraDev cam, RenderArgs ra, float* __restrict__ accum,
                                                  unsigned long long* __restrict__ counters, uint32_t tiles_x) {
    stage_scene_to_lds<MODE>(scd);
    const SceneAcc<MODE> sc{scd.blob, scd.L};

    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t tile = xcd_tile(blockIdx.x, gridDim.x, ra.xcd_aware);
    const uint32_t tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const uint32_t x = tile_x * kTileW + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t row = tile_y * (uint32_t)(THREADS / 16) + (wave >> 1) * 8u + (lane >> 3);    // local row
    const bool in_image = x < cam.width && row < ra.rows_local;
    const uint32_t y = image_row(ra, row);                                          // image row: keys the RNG
    float* out = accum + 3ull * ((unsigned long long)row * cam.width + x);

    V3 acc = v3(0.0f, 0.0f, 0.0f);
    if (in_image && ra.accumulate) acc = v3(out[0], out[1], out[2]);
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);

    uint32_t s = ra.sample_begin;
    bool alive = in_image && s < ra.sample_end;
    bool fresh = true;                       // lane needs a new primary ray
    Path p;
    p.remain = 0u;
    // One primary ray kept ahead per lane.  A path ends every ~7 bounces, so in any one trip only a few lanes need a
    // new primary ray and generating it on demand runs that code at ~10 % lane occupancy.  Instead, whenever some lane
    // needs a ray and has none in stock, EVERY lane without stock generates the ray of the next sample it will need
    // (its stream depends on (seed, pixel, sample) only), so the generator runs less often and fuller.
    Ray next_ray;
    Rng next_rng;
    bool stocked = false;
    uint32_t n_samples = 0, n_rays = 0;
    Counters<STATS> ctr;

    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        if (__builtin_amdgcn_ballot_w64(alive && fresh && !stocked) != 0ull) {
            const uint32_t s_next = fresh ? s : s + 1u;                             // the sample this lane starts next
            if (alive && !stocked && s_next < ra.sample_end) {
                if constexpr (STATS) { if (first_active_lane()) ctr.w_gen++; }
                next_rng = rng_seed(ra.seed_key, y * cam.width + x, s_next);
                next_ray = primary_ray(cam, x, y, next_rng);
                stocked = true;
            }
        }
        if (alive) {
            if constexpr (STATS) { if (first_active_lane()) ctr.w_rounds++; }
            if (fresh) {                                                            // cpu.rs:42-45
                p.ray = next_ray;
                p.rng = next_rng;
                p.color = v3(0.0f, 0.0f, 0.0f);
                p.atten = v3(1.0f, 1.0f, 1.0f);
                p.remain = ra.max_bounces;
                stocked = false;
                fresh = false;
                n_samples++;
            }
            n_rays++;
            float t;
            const uint32_t prim = closest_hit<MODE, STATS>(sc, p.ray, ra.ref_tree != 0u, t, ctr, ra.leaf_slots);
            if (shade_hit<MODE, STATS>(sc, p, prim, t, background, ctr)) {
                acc = acc + p.color * ra.inv_spp;                                   // imager.rs:50
                s++;
                fresh = true;
                alive = s < ra.sample_end;
            }
        }
    }
    if (in_image) { out[0] = acc.x; out[1] = acc.y; out[2] = acc.z; }
    flush_counters<STATS>(counters, n_samples, n_rays, ctr);
}

// ------------------------------------------------------------------------------------------------
// Sampler plug-in form (sampler/mod.rs:10-17): one lane per caller-supplied SamplePoint.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool STATS>
__global__ __launch_bounds__(256) void sample_batch_kernel(SceneDev scd, const trt_sample_point* __restrict__ in, uint32_t n,
                                                           trt_sampled_color* __restrict__ out, RenderArgs ra,
                                                           unsigned long long* __restrict__ counters) {
    stage_scene_to_lds<MODE>(scd);
    const SceneAcc<MODE> sc{scd.blob, scd.L};
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);
    uint32_t n_rays = 0;
    Counters<STATS> ctr;
    bool alive = i < n;
    Path p;
    p.color = v3(0.0f, 0.0f, 0.0f);
    p.atten = v3(1.0f, 1.0f, 1.0f);
    p.remain = ra.max_bounces;
    if (alive) {
        const trt_sample_point sp = in[i];
        p.ray.o = v3(sp.ray.origin.x, sp.ray.origin.y, sp.ray.origin.z);           // a SamplePoint carries a built Ray
        p.ray.d = v3(sp.ray.direction.x, sp.ray.direction.y, sp.ray.direction.z);
        p.rng = rng_seed(ra.seed_key, i, 0u);
    }
    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        if (alive) {
            n_rays++;
            float t;
            const uint32_t prim = closest_hit<MODE, STATS>(sc, p.ray, ra.ref_tree != 0u, t, ctr, ra.leaf_slots);
            if (shade_hit<MODE, STATS>(sc, p, prim, t, background, ctr)) alive = false;
        }
    }
    if (i < n) {
        trt_sampled_color c;
        c.x = in[i].x; c.y = in[i].y;
        c.color.x = p.color.x; c.color.y = p.color.y; c.color.z = p.color.z;
        out[i] = c;
    }
    flush_counters<STATS>(counters, (i < n) ? 1u : 0u, n_rays, ctr);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <typename K, typename... Args>
static hipError_t launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    timing_mark(stream, true);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    const hipError_t le = hipGetLastError();
    timing_mark(stream, false);
    return le;
}

hipError_t launch_megakernel(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, const trt_tuning& tn, float* d_accum,
                             unsigned long long* d_counters, bool stats, hipStream_t stream) {
    // Workgroup size and register budget.  The kernel wants 98 VGPRs (4 waves/SIMD); capping it by launch bound buys
    // occupancy for a few spilled dwords.  Measured on Cornell 2048^2 (Gray/s): 4 waves 20.7, 5: 22.5, 6: 23.3, 7: 23.9,
    // 8: 23.5.  A workgroup's LDS scene copy is shared by its waves, so scenes with a big copy (random-spheres: 50 KB,
    // three copies per CU) use 512-lane workgroups: 3 x 8 waves per CU instead of 3 x 4.
    const size_t lds_bytes = scene_lds_bytes(sc.L);
    const int mode = scene_mode(sc.L);
    int threads = (mode == MODE_LDS && sc.L.hot_bytes > 20u * 1024u) ? 512 : 256;
    int w = mode != MODE_LDS ? 1 : (threads == 512 ? 6 : 7);
    if (tn.mega_threads) threads = (mode == MODE_LDS && tn.mega_threads == 512u) ? 512 : 256;   // only MODE_LDS has 512-lane instances
    if (tn.mega_waves_per_simd) w = (int)tn.mega_waves_per_simd;
    const uint32_t tile_h = (uint32_t)threads / 16u;
    const uint32_t tiles_x = (cam.width + kTileW - 1) / kTileW, tiles_y = (ra.rows_local + tile_h - 1) / tile_h;
    if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
    const dim3 grid(tiles_x * tiles_y), block((uint32_t)threads);
    auto go = [&](auto kernel) { return launch(kernel, grid, block, lds_bytes, stream, sc, cam, ra, d_accum, d_counters, tiles_x); };
    switch (mode) {
        case MODE_LDS:
            if (threads == 512) {
                if (w >= 6) return stats ? go(megakernel<MODE_LDS, true, 6, 512>) : go(megakernel<MODE_LDS, false, 6, 512>);
                return stats ? go(megakernel<MODE_LDS, true, 5, 512>) : go(megakernel<MODE_LDS, false, 5, 512>);
            }
            if (w >= 8) return stats ? go(megakernel<MODE_LDS, true, 8>) : go(megakernel<MODE_LDS, false, 8>);
            if (w == 7) return stats ? go(megakernel<MODE_LDS, true, 7>) : go(megakernel<MODE_LDS, false, 7>);
            if (w == 6) return stats ? go(megakernel<MODE_LDS, true, 6>) : go(megakernel<MODE_LDS, false, 6>);
            return stats ? go(megakernel<MODE_LDS, true, 5>) : go(megakernel<MODE_LDS, false, 5>);
        default:
            if (tn.mega_global_waves8) return stats ? go(megakernel<MODE_GLOBAL, true, 8>) : go(megakernel<MODE_GLOBAL, false, 8>);
            return stats ? go(megakernel<MODE_GLOBAL, true>) : go(megakernel<MODE_GLOBAL, false>);
    }
}

hipError_t launch_sample_batch(const SceneDev& sc, const trt_sample_point* d_in, uint32_t n, trt_sampled_color* d_out,
                               const RenderArgs& ra, unsigned long long* d_counters, bool stats, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const dim3 grid((n + 255u) / 256u), block(256);
    const size_t lds_bytes = scene_lds_bytes(sc.L);
    auto go = [&](auto kernel) { return launch(kernel, grid, block, lds_bytes, stream, sc, d_in, n, d_out, ra, d_counters); };
    switch (scene_mode(sc.L)) {
        case MODE_LDS: return stats ? go(sample_batch_kernel<MODE_LDS, true>) : go(sample_batch_kernel<MODE_LDS, false>);
        default: return stats ? go(sample_batch_kernel<MODE_GLOBAL, true>) : go(sample_batch_kernel<MODE_GLOBAL, false>);
    }
}

// ------------------------------------------------------------------------------------------------
// Imager finalisation on the device: Color::gamma_correction + From<Color> for Rgb<u8>
// (utils/image.rs:92-111): c^(1/gamma), clamp to [0, 0.999], * 255, truncate; NaN -> 0.  Elementwise and HBM-bound:
// 12 bytes read + 3 written per pixel, four channels per lane (one 16-byte load, one 4-byte store).
// c^(1/gamma) is trt-math v2's powf (trt_pow.h): the same function the host form (scene_host.cpp tonemap_u8) and the CPU
// oracle evaluate, so the three u8 frames are equal byte for byte (tests compare them exactly).
// ------------------------------------------------------------------------------------------------
TRT_DEV uint32_t quantise_channel(float c, float inv_gamma) { return (uint32_t)tm_quantise_channel(c, inv_gamma); }

__global__ __launch_bounds__(256) void tonemap_u8_kernel(const float* __restrict__ accum, unsigned long long n_channels,
                                                         float inv_gamma, uint8_t* __restrict__ rgb, uint32_t vectorised) {
    const float ig = inv_gamma;
    const unsigned long long i4 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 4ull;
    if (i4 >= n_channels) return;
    if (vectorised && i4 + 4ull <= n_channels) {
        const float4 v = *reinterpret_cast<const float4*>(accum + i4);
        const uint32_t q = quantise_channel(v.x, ig) | quantise_channel(v.y, ig) << 8 | quantise_channel(v.z, ig) << 16 |
                           quantise_channel(v.w, ig) << 24;
        *reinterpret_cast<uint32_t*>(rgb + i4) = q;
    } else {
        for (unsigned long long i = i4; i < i4 + 4ull && i < n_channels; i++) rgb[i] = (uint8_t)quantise_channel(accum[i], ig);
    }
}

hipError_t launch_tonemap_u8(const float* d_accum, unsigned long long npixels, float gamma, uint8_t* d_rgb, hipStream_t stream) {
    const unsigned long long n_channels = npixels * 3ull;
    if (n_channels == 0ull) return hipSuccess;
    const uint32_t vectorised = (reinterpret_cast<uintptr_t>(d_accum) % 16u == 0u && reinterpret_cast<uintptr_t>(d_rgb) % 4u == 0u) ? 1u : 0u;
    const unsigned long long lanes = (n_channels + 3ull) / 4ull;
    const dim3 grid((uint32_t)((lanes + 255ull) / 256ull)), block(256);
    hipLaunchKernelGGL(tonemap_u8_kernel, grid, block, 0, stream, d_accum, n_channels, 1.0f / gamma, d_rgb, vectorised);
    return hipGetLastError();
}

}  // namespace trt
