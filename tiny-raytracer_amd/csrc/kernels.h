// kernels.h — launch interface between the C ABI (capi.hip) and the HIP kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scene.h"

namespace trt {

struct SceneDev {
    const float4* blob;          // packed scene in HBM (scene.h layout)
    SceneLayout L;
};

struct CameraDev {               // camera.rs:4-14, the fields get_ray reads
    float pos[3], upper_left[3], horizontal[3], vertical[3], du[3], dv[3];
    uint32_t width, height;
};

struct RenderArgs {
    float background[3];
    float inv_spp;               // 1/spp: Imager's color_multiplier (imager.rs:35)
    uint32_t max_bounces;
    uint32_t seed_key;           // mix32(seed + 0x9E3779B9)
    uint32_t sample_begin, sample_end;
    uint32_t accumulate;
    uint32_t band_rows, band_stride, band_offset;   // local row -> image row (tinyrt.h)
    uint32_t rows_local;
    uint32_t xcd_aware;          // 1: remap workgroups so that each XCD renders a contiguous image region
    uint32_t lds_leaf_stack;     // streamed backend: keep the postponed leaves in LDS (rt_path.h walk_fast_lds) instead of registers: 0 no, 1 where it costs no occupancy, 2 always
    uint32_t leaf_slots;         // postponed-leaf slots per lane in the walk (rt_path.h walk_fast): 4, 2 or 1
    uint32_t ref_tree;           // 1: walk the reference tree (counting kernels: counters comparable with the oracle)
    uint32_t stragglers;         // streamed walks of scenes in global memory: a round's walk phase ends once at most this many lanes of the wave still walk
                                 // (they carry their walk into the next round); 0 = every walk runs to its end (rt_path.h walk_compact)
};

// The built-in scheduling defaults (tinyrt.h trt_tuning; each a measured optimum, DESIGN.md "Tuning").  They live in THIS header -
// which bench.py's kernel-source digest covers - because they decide which kernel runs and how: a stored PMC profile is valid
// for one set of defaults only.
inline trt_tuning tuning_builtin() {
    trt_tuning t{};
    t.stream_waves_per_simd = 0;      // by scene: 6 (LDS), 8 (global memory)
    t.stream_big_threads = 0;         // auto: 768 where the LDS leaf stack then fits, else 512
    t.stream_batch_spp = 8;
    t.radiance_gb = 16;
    t.leaf_slots = 0;                 // by launch plan
    t.lds_leaf_stack = 1;
    t.ray_pool = 1;
    t.stragglers = 8;                 // profiles/r03_stragglers_sweep.txt
    t.lds_stragglers = 8;
    t.dual_walk = 0;                  // by scene: two paths per lane beyond L2 (+2.7 % / +4.2 % on sphere_field 1 M / 4 M, -2.6 % on the 100 k-sphere scene: profiles/r05_dual_walk_fused_ab.txt)
    t.runtime_walk = 0;
    t.xcd_remap = 0;
    t.mega_waves_per_simd = 0; t.mega_threads = 0; t.mega_global_waves8 = 0;
    t.wf_waves_per_simd = 0; t.wf_serve_min = 0;
    return t;
}

// trt-rng v1 per-launch key: mix32(seed + golden ratio), evaluated once on the host.
inline uint32_t rng_seed_key(uint32_t seed) {
    uint32_t x = seed + 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// counters[0..6] = samples, rays, node_tests, sphere_tests, quad_plane_tests, quad_inside_tests, shades
// counters[8..11] (collect_stats only) = wave-level loop trips: bounce rounds, box-test steps, leaf phases, ray generations
// counters[7] (collect_stats, lock-step walk only) = leaves put aside, summed over lanes (a diagnostic read through trt_render_device's counters)
// counters[12..15] (collect_stats only) = `shades` by material kind (trt_material_kind order: Lambertian, Metal, Dielectric, Light); they sum
//                  to counters[6] (bench.py useful_frac; read through trt_render_device's counters; the phase-clock build keeps its clock there)
enum { CTR_SAMPLES = 0, CTR_RAYS, CTR_NODE, CTR_SPHERE, CTR_QUAD_PLANE, CTR_QUAD_INSIDE, CTR_SHADE, CTR_PEND = 7,
       CTR_W_ROUNDS = 8, CTR_W_STEPS, CTR_W_LEAF, CTR_W_GEN, CTR_SHADE_KIND = 12, CTR_COUNT = 16 };

// scene.h kLdsSceneMaxBytes: hot blobs up to that size are copied whole into LDS; larger scenes are read from global memory.
inline int scene_mode(const SceneLayout& L) { return L.hot_bytes <= kLdsSceneMaxBytes ? 1 : 0; }
inline uint32_t scene_lds_bytes(const SceneLayout& L) { return scene_mode(L) == 1 ? L.hot_bytes : 0u; }

// Optional device timing of the dominant kernel (trt_kernel_timing_begin / _end, capi.hip): while enabled, a launcher
// brackets every launch of its dominant kernel with timing_mark(stream, true) / timing_mark(stream, false), which record
// HIP events on the launch stream.  No-ops while disabled.
void timing_mark(hipStream_t stream, bool begin);

// Megakernel: whole bounce loop for every pixel of the local rows in one launch.
hipError_t launch_megakernel(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, const trt_tuning& tn, float* d_accum,
                             unsigned long long* d_counters, bool stats, hipStream_t stream);

// Wavefront backend (wavefront.hip): path state lives SoA in a caller-provided HBM workspace.
struct WfState {
    float4* s0;     // origin.xyz, hit t
    float4* s1;     // direction.xyz, bits(hit primitive)
    float4* s2;     // throughput.xyz, bits(remaining bounces)
    float4* s3;     // radiance.xyz, bits(sample index)
    uint2* rng;     // trt-rng v1 stream state
};
size_t wavefront_workspace_bytes(uint32_t width, uint32_t rows);
hipError_t launch_wavefront(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, const trt_tuning& tn, void* workspace, float* d_accum,
                            unsigned long long* d_counters, bool stats, hipStream_t stream);

// Streamed backend (streamed.hip): samples are work items; radiances go to an HBM buffer and are folded in order.
uint32_t streamed_chunk_spp(uint32_t width, uint32_t rows, uint32_t radiance_gb);   // samples per pixel per tracing / fold launch pair (bounds the radiance buffer)
size_t streamed_workspace_bytes(uint32_t width, uint32_t rows, uint32_t samples, uint32_t radiance_gb);   // samples = sample_end - sample_begin of the render
// Samples per pixel a workspace of `bytes` holds for this image (what launch_streamed may trace per launch with the scratch it was granted).
uint32_t streamed_chunk_that_fits(uint32_t width, uint32_t rows, size_t bytes);
const char* streamed_kernel_name(const SceneLayout& L, const RenderArgs& ra, const trt_tuning& tn);
// How launch_streamed runs a scene with these settings (streamed.hip; also what trt_streamed_launch_plan reports).  The
// kernel's view of its dynamic LDS - scene copy | leaf stack (threads x slots x 8 B) | ray pool (36 B per lane) - is fixed
// here and nowhere else; launch_streamed refuses a plan whose parts do not add up (hipErrorInvalidConfiguration).
struct StreamLaunchPlan {
    int mode, threads, waves_per_simd;      // scene mode, lanes per workgroup, waves per SIMD the grid is sized for
    uint32_t wg_per_cu, slots;              // resident workgroups per CU; postponed-leaf slots per lane
    bool lds_stack, flat, compact, pool, specialised;
    bool dual;                              // two paths per lane (stream_dual_kernel): two leaf stacks per lane
    int walk;                               // WALK_* the kernel will run
    size_t lds_bytes, scene_lds_bytes;      // dynamic LDS per workgroup; the scene copy's share of it
    const void* kernel;                     // the instantiation (nullptr: none - a bug, launch_streamed fails)
    int kernel_minw, kernel_threads, kernel_walk;      // its template arguments (launch bound, lanes, WALK_RUNTIME = chooses at run time)
    bool kernel_pool, kernel_stats;
    const char* kernel_name;
};
StreamLaunchPlan streamed_launch_plan(const SceneLayout& L, const RenderArgs& ra, const trt_tuning& tn, bool stats);
// `workspace_bytes`: what the caller was granted (at least streamed_workspace_bytes for one sample): the launches are sized to it.
hipError_t launch_streamed(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, const trt_tuning& tn, void* workspace, size_t workspace_bytes,
                           float* d_accum, unsigned long long* d_counters, bool stats, hipStream_t stream);

// Sampler plug-in form: n caller-supplied rays.
// Imager finalisation on buffers in HBM (kernels.hip).
hipError_t launch_tonemap_u8(const float* d_accum, unsigned long long npixels, float gamma, uint8_t* d_rgb, hipStream_t stream);

hipError_t launch_sample_batch(const SceneDev& sc, const trt_sample_point* d_in, uint32_t n, trt_sampled_color* d_out,
                               const RenderArgs& ra, unsigned long long* d_counters, bool stats, hipStream_t stream);

}  // namespace trt
