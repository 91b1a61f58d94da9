// kernels.hip — HIP kernels of the path-tracing sampler, written for gfx950 (CDNA4) only.
//
// What runs here is the reference's hot path, raytracer/src/renderer/sampler/cpu.rs:39-65 and its
// callees, plus the two edges that have to live on the device so that no per-sample message ever
// crosses PCIe: primary-ray generation (renderer/pointgen.rs:37-52, camera.rs:58-66) and the f32
// accumulation (renderer/imager.rs:35,50).
//
// Design:
//  * one lane owns one pixel and walks its samples in order, so the running sum per pixel is the
//    reference's `pixels[idx] += color * (1/spp)` in sample order: results are bit-reproducible and
//    independent of launch geometry, tiling or the number of GPUs;
//  * lanes are persistent: a lane whose path ended starts its next sample in the same loop trip in
//    which its neighbours trace their next bounce, so a wave never idles behind its longest path;
//  * traversal is stackless: the reference always descends left first (bvh.rs:96-106), so the
//    pre-order node array plus one skip link per node reproduces its visit order and its running
//    [t_min, t_best) interval exactly; leaf primitives are tested in a second phase of the loop so
//    the 64 lanes of a wave run box tests together and primitive tests together ("while-while");
//  * scenes up to kLdsSceneMaxBytes are copied into LDS once per workgroup and traversed from
//    there (ds_read_b128 per 16-byte plane element); larger ones are read through L1/L2.
#include "kernels.h"
#include "rt_device.h"

namespace trt {

extern __shared__ float4 g_lds_scene[];

// ------------------------------------------------------------------------------------------------
// Scene access: LDS copy or global blob, same element offsets (scene.h).
// ------------------------------------------------------------------------------------------------
template <bool LDS>
struct SceneAcc {
    const float4* blob;
    SceneLayout L;
    TRT_DEV float4 f4(uint32_t idx) const { return LDS ? g_lds_scene[idx] : blob[idx]; }
    TRT_DEV uint32_t u32(uint32_t idx) const {
        return LDS ? reinterpret_cast<const uint32_t*>(g_lds_scene)[idx] : reinterpret_cast<const uint32_t*>(blob)[idx];
    }
    TRT_DEV float4 node_a(uint32_t i) const { return f4(i); }
    TRT_DEV float4 node_b(uint32_t i) const { return f4(L.off_node_b + i); }
    TRT_DEV float4 sphere(uint32_t i) const { return f4(L.off_sphere + i); }
    TRT_DEV float4 quad(uint32_t plane, uint32_t i) const { return f4(L.off_quad + plane * L.n_quads + i); }
    TRT_DEV float4 material(uint32_t i) const { return f4(L.off_material + i); }
    TRT_DEV uint32_t sphere_material(uint32_t i) const { return u32(L.off_sphere_mat + i); }
    TRT_DEV uint32_t material_kind(uint32_t i) const { return u32(L.off_material_kind + i); }
};

template <bool STATS>
struct Counters {
    uint32_t node = 0, sphere = 0, quad_plane = 0, quad_inside = 0, shade = 0;
};
template <>
struct Counters<false> {};

// ------------------------------------------------------------------------------------------------
// Closest hit: BVH::hit / Node::hit (hittable/bvh.rs:24-27,88-107) with t_range = 0.001..inf
// (cpu.rs:48).  Returns the primitive reference (PRIM_NONE on a miss) and its t.
//
// Node::hit tests the box with the interval it was handed; an inner node hands its left child the
// same interval and its right child [t_min, t_left) if the left child hit.  Walking the pre-order
// array with one running t_best, used as the exclusive end for boxes and primitives alike, is that
// recursion unrolled: a primitive is accepted only if t < t_best, so on equal t the primitive that
// comes first in left-first order wins, as in bvh.rs:96-101.
// ------------------------------------------------------------------------------------------------
template <bool LDS, bool STATS>
TRT_DEV uint32_t closest_hit(const SceneAcc<LDS>& sc, const Ray& ray, float& t_hit, Counters<STATS>& ctr) {
    const float t_min = 0.001f;
    V3 inv = v3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);          // aabb.rs:42, hoisted out of the node loop
    const bool fast = sc.L.all_finite && finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z) &&
                      finite_f(ray.o.x) && finite_f(ray.o.y) && finite_f(ray.o.z);
    float t_best = __builtin_inff();
    uint32_t prim_best = PRIM_NONE;
    const uint32_t n = sc.L.n_nodes;
    uint32_t i = 0;
    for (;;) {
        // phase 1: box tests until this lane stands on a leaf whose box it hits, or runs off the end
        uint32_t leaf = PRIM_NONE;
        while (i < n) {
            float4 na = sc.node_a(i), nb = sc.node_b(i);
            if constexpr (STATS) ctr.node++;
            bool pass;
            if (__builtin_expect(fast, 1)) pass = slab_fast(na, nb, ray.o, inv, t_min, t_best);
            else pass = slab_exact(na, nb, ray.o, inv, t_min, t_best);
            uint32_t prim = __float_as_uint(nb.w);
            i = pass ? i + 1u : __float_as_uint(nb.z);
            if (pass && prim != PRIM_NONE) { leaf = prim; break; }
        }
        if (leaf == PRIM_NONE) break;
        // phase 2: primitive test with the same interval the leaf's box was tested with (bvh.rs:93-94)
        const uint32_t idx = leaf & PRIM_INDEX_MASK;
        if (leaf & PRIM_QUAD_BIT) {                                    // Quad::hit, quad.rs:33-54
            if constexpr (STATS) ctr.quad_plane++;
            float4 q0 = sc.quad(0, idx);
            V3 nrm = v3(q0.x, q0.y, q0.z);
            float dir_norm = dot(ray.d, nrm);
            float t = (q0.w - dot(ray.o, nrm)) / dir_norm;
            if (t_min <= t && t < t_best) {
                if constexpr (STATS) ctr.quad_inside++;
                float4 q1 = sc.quad(1, idx), q2 = sc.quad(2, idx), q3 = sc.quad(3, idx), q4 = sc.quad(4, idx);
                V3 p = ray_at(ray, t) - v3(q1.x, q1.y, q1.z);
                V3 vv = v3(q2.x, q2.y, q2.z), ww = v3(q2.w, q3.x, q3.y), uu = v3(q3.z, q3.w, q4.x);
                float planar_x = dot(cross(p, vv), ww);
                float planar_y = dot(cross(uu, p), ww);
                if (0.0f <= planar_x && planar_x < 1.0f && 0.0f <= planar_y && planar_y < 1.0f) {
                    t_best = t;
                    prim_best = leaf;
                }
            }
        } else {                                                       // Sphere::hit, sphere.rs:29-54
            if constexpr (STATS) ctr.sphere++;
            float t;
            if (sphere_test(sc.sphere(idx), ray, t_min, t_best, t)) {
                t_best = t;
                prim_best = leaf;
            }
        }
    }
    t_hit = t_best;
    return prim_best;
}

// ------------------------------------------------------------------------------------------------
// One bounce of CpuSampler::single_point_sampling (cpu.rs:47-62): closest hit, emission, scatter.
// Returns true when the path ended (light, miss, or budget spent).
// ------------------------------------------------------------------------------------------------
struct Path {
    Ray ray;
    V3 color, atten;
    uint32_t remain;
    Rng rng;
};

template <bool LDS, bool STATS>
TRT_DEV bool bounce(const SceneAcc<LDS>& sc, Path& p, V3 background, Counters<STATS>& ctr) {
    float t;
    const uint32_t prim = closest_hit<LDS, STATS>(sc, p.ray, t, ctr);
    if (prim == PRIM_NONE) {                                           // cpu.rs:58-61
        p.color = p.color + p.atten * background;
        return true;
    }
    if constexpr (STATS) ctr.shade++;
    // HitRecord::new (hittable/mod.rs:28-48), built once for the winning primitive
    const uint32_t idx = prim & PRIM_INDEX_MASK;
    V3 point = ray_at(p.ray, t);
    V3 normal;
    bool front_face;
    uint32_t mat;
    if (prim & PRIM_QUAD_BIT) {
        float4 q0 = sc.quad(0, idx), q1 = sc.quad(1, idx), q4 = sc.quad(4, idx);
        front_face = dot(p.ray.d, v3(q0.x, q0.y, q0.z)) < 0.0f;        // outward normal = n, un-normalised (quad.rs:45)
        V3 nu = v3(q4.y, q4.z, q4.w);                                  // n.normalized(), precomputed on the host
        normal = front_face ? nu : -nu;
        mat = __float_as_uint(q1.w);
    } else {
        float4 sp = sc.sphere(idx);
        V3 outward = point - v3(sp.x, sp.y, sp.z);                     // sphere.rs:47-51 (p = ray.at(t))
        front_face = dot(p.ray.d, outward) < 0.0f;
        V3 nu = normalized(outward);
        normal = front_face ? nu : -nu;
        mat = sc.sphere_material(idx);
    }
    const float4 m = sc.material(mat);
    const uint32_t kind = sc.material_kind(mat);
    const V3 albedo = v3(m.x, m.y, m.z);
    // cpu.rs:49-50: emitted() is the light's colour, None -> 0 for everything else (material/mod.rs:8-10)
    V3 emission = (kind == TRT_LIGHT) ? albedo : v3(0.0f, 0.0f, 0.0f);
    p.color = p.color + p.atten * emission;
    V3 dir;
    if (kind == TRT_LAMBERTIAN) {                                      // lambertian.rs:16-22
        dir = normal + random_unit_vector(p.rng);
        if (near_zero(dir)) dir = normal;
    } else if (kind == TRT_METAL) {                                    // metal.rs:18-25 (fuzz clamped at creation)
        V3 reflected = reflect(p.ray.d, normal);
        dir = reflected + m.w * random_in_unit_sphere(p.rng);
    } else if (kind == TRT_DIELECTRIC) {                               // dielectric.rs:26-46
        float ri = front_face ? 1.0f / m.w : m.w;
        float cosv = __builtin_fminf(-dot(normal, p.ray.d), 1.0f);
        float sinv = __builtin_sqrtf(1.0f - cosv * cosv);
        bool total_reflection = ri * sinv > 1.0f;
        float sqrt_r0 = (1.0f - ri) / (1.0f + ri);                     // reflectance(), dielectric.rs:16-22
        float r0 = sqrt_r0 * sqrt_r0;
        float x = 1.0f - cosv;
        float x2 = x * x;
        float reflectance = r0 + (1.0f - r0) * (x * (x2 * x2));        // powi(5): x * ((x*x)*(x*x))
        bool do_reflect = total_reflection;
        if (!do_reflect) do_reflect = reflectance > rng_random(p.rng); // `||` short-circuit: no draw on TIR
        dir = do_reflect ? reflect(p.ray.d, normal) : refract(p.ray.d, normal, ri);
    } else {                                                           // Light::scatter -> None (light.rs:17-19)
        return true;
    }
    p.atten = p.atten * albedo;                                        // cpu.rs:52
    p.ray = ray_new(point, dir);                                       // Ray::new normalises (ray.rs:12-14)
    p.remain -= 1u;                                                    // cpu.rs:54
    return p.remain == 0u;
}

// SamplePointGenerator::generate body (pointgen.rs:41-43) + Camera::get_ray (camera.rs:58-66)
TRT_DEV Ray primary_ray(const CameraDev& cam, uint32_t x, uint32_t y, Rng& rng) {
    float u = ((float)x + rng_random(rng)) / (float)(cam.width - 1u);
    float v = ((float)y + rng_random(rng)) / (float)(cam.height - 1u);
    float px, py;
    random_in_unit_disk(rng, px, py);
    V3 pos = v3(cam.pos[0], cam.pos[1], cam.pos[2]);
    V3 du = v3(cam.du[0], cam.du[1], cam.du[2]), dv = v3(cam.dv[0], cam.dv[1], cam.dv[2]);
    V3 origin = (pos + px * du) + py * dv;
    V3 ul = v3(cam.upper_left[0], cam.upper_left[1], cam.upper_left[2]);
    V3 hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    V3 ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    V3 target = (ul + u * hor) - v * ver;
    return ray_new(origin, target - origin);
}

template <bool LDS>
TRT_DEV void stage_scene_to_lds(const SceneDev& sc) {
    if constexpr (LDS) {
        const uint32_t n16 = sc.L.blob_bytes >> 4;
        for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x) g_lds_scene[k] = sc.blob[k];
        __syncthreads();
    }
}

TRT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool STATS>
TRT_DEV void flush_counters(unsigned long long* counters, uint32_t samples, uint32_t rays, const Counters<STATS>& ctr) {
    if (counters == nullptr) return;
    const bool lane0 = (threadIdx.x & 63u) == 0u;
    uint32_t s = wave_sum(samples), r = wave_sum(rays);
    if (lane0) { atomicAdd(&counters[CTR_SAMPLES], (unsigned long long)s); atomicAdd(&counters[CTR_RAYS], (unsigned long long)r); }
    if constexpr (STATS) {
        uint32_t a = wave_sum(ctr.node), b = wave_sum(ctr.sphere), c = wave_sum(ctr.quad_plane), d = wave_sum(ctr.quad_inside),
                 e = wave_sum(ctr.shade);
        if (lane0) {
            atomicAdd(&counters[CTR_NODE], (unsigned long long)a);
            atomicAdd(&counters[CTR_SPHERE], (unsigned long long)b);
            atomicAdd(&counters[CTR_QUAD_PLANE], (unsigned long long)c);
            atomicAdd(&counters[CTR_QUAD_INSIDE], (unsigned long long)d);
            atomicAdd(&counters[CTR_SHADE], (unsigned long long)e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Megakernel.  Workgroup = 256 lanes = a 16x16 pixel tile; each wave owns an 8x8 sub-tile so the
// 64 paths of a wave see similar geometry.  A lane is done when its pixel has had samples
// [sample_begin, sample_end); the wave leaves the loop when every lane is done.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTile = 16;

template <bool LDS, bool STATS>
__global__ __launch_bounds__(256) void megakernel(SceneDev scd, CameraDev cam, RenderArgs ra, float* __restrict__ accum,
                                                  unsigned long long* __restrict__ counters, uint32_t tiles_x) {
    stage_scene_to_lds<LDS>(scd);
    const SceneAcc<LDS> sc{scd.blob, scd.L};

    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const uint32_t x = tile_x * kTile + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t row = tile_y * kTile + (wave >> 1) * 8u + (lane >> 3);          // local row
    const bool in_image = x < cam.width && row < ra.rows_local;
    uint32_t y = row;                                                                // image row
    if (ra.band_rows != 0u) y = ((row / ra.band_rows) * ra.band_stride + ra.band_offset) * ra.band_rows + row % ra.band_rows;
    const uint32_t pixel = y * cam.width + x;                                        // RNG key: global pixel index
    float* out = accum + 3ull * ((unsigned long long)row * cam.width + x);

    V3 acc = v3(0.0f, 0.0f, 0.0f);
    if (in_image && ra.accumulate) acc = v3(out[0], out[1], out[2]);
    const V3 background = v3(ra.background[0], ra.background[1], ra.background[2]);

    uint32_t s = ra.sample_begin;
    bool alive = in_image && s < ra.sample_end;
    bool fresh = true;                       // lane needs a new primary ray
    Path p;
    p.remain = 0u;
    uint32_t n_samples = 0, n_rays = 0;
    Counters<STATS> ctr;

    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        if (alive) {
            if (fresh) {
                p.rng = rng_seed(ra.seed_key, pixel, s);
                p.ray = primary_ray(cam, x, y, p.rng);
                p.color = v3(0.0f, 0.0f, 0.0f);
                p.atten = v3(1.0f, 1.0f, 1.0f);
                p.remain = ra.max_bounces;
                fresh = false;
                n_samples++;
            }
            n_rays++;
            if (bounce<LDS, STATS>(sc, p, background, ctr)) {
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
template <bool LDS, bool STATS>
__global__ __launch_bounds__(256) void sample_batch_kernel(SceneDev scd, const trt_sample_point* __restrict__ in, uint32_t n,
                                                           trt_sampled_color* __restrict__ out, RenderArgs ra,
                                                           unsigned long long* __restrict__ counters) {
    stage_scene_to_lds<LDS>(scd);
    const SceneAcc<LDS> sc{scd.blob, scd.L};
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
            if (bounce<LDS, STATS>(sc, p, background, ctr)) alive = false;
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
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    return hipGetLastError();
}

hipError_t launch_megakernel(const SceneDev& sc, const CameraDev& cam, const RenderArgs& ra, float* d_accum,
                             unsigned long long* d_counters, bool stats, hipStream_t stream) {
    const uint32_t tiles_x = (cam.width + kTile - 1) / kTile, tiles_y = (ra.rows_local + kTile - 1) / kTile;
    if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
    const dim3 grid(tiles_x * tiles_y), block(256);
    const bool lds = sc.L.blob_bytes <= kLdsSceneMaxBytes;
    const size_t lds_bytes = lds ? sc.L.blob_bytes : 0;
    if (lds) {
        return stats ? launch(megakernel<true, true>, grid, block, lds_bytes, stream, sc, cam, ra, d_accum, d_counters, tiles_x)
                     : launch(megakernel<true, false>, grid, block, lds_bytes, stream, sc, cam, ra, d_accum, d_counters, tiles_x);
    }
    return stats ? launch(megakernel<false, true>, grid, block, lds_bytes, stream, sc, cam, ra, d_accum, d_counters, tiles_x)
                 : launch(megakernel<false, false>, grid, block, lds_bytes, stream, sc, cam, ra, d_accum, d_counters, tiles_x);
}

hipError_t launch_sample_batch(const SceneDev& sc, const trt_sample_point* d_in, uint32_t n, trt_sampled_color* d_out,
                               const RenderArgs& ra, unsigned long long* d_counters, bool stats, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const dim3 grid((n + 255u) / 256u), block(256);
    const bool lds = sc.L.blob_bytes <= kLdsSceneMaxBytes;
    const size_t lds_bytes = lds ? sc.L.blob_bytes : 0;
    if (lds) {
        return stats ? launch(sample_batch_kernel<true, true>, grid, block, lds_bytes, stream, sc, d_in, n, d_out, ra, d_counters)
                     : launch(sample_batch_kernel<true, false>, grid, block, lds_bytes, stream, sc, d_in, n, d_out, ra, d_counters);
    }
    return stats ? launch(sample_batch_kernel<false, true>, grid, block, lds_bytes, stream, sc, d_in, n, d_out, ra, d_counters)
                 : launch(sample_batch_kernel<false, false>, grid, block, lds_bytes, stream, sc, d_in, n, d_out, ra, d_counters);
}

}  // namespace trt
