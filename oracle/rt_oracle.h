/*
 * rt_oracle.h — CPU ORACLE for the tiny-raytracer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference crate's per-sample Monte-Carlo
 * bounce loop (cheolwanpark/tiny-raytracer, `raytracer/src/renderer/sampler/cpu.rs:39-65`
 * and everything it calls).  It keeps the reference's *structure*: pointer BVH with one
 * primitive per leaf, recursive left-first traversal, AoS primitives, per-sample loop.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (tiny-raytracer_amd/) never includes, links or calls it.
 *
 * Parity status: PINNED against every known-answer test the reference's own suite holds
 * for this path (tests/golden/reference_kats.json) and statistically against the
 * reference's shipped render output/output.png (tests/golden/cornell_ref_blocks.json).
 * NOT pinned by the reference (it has no such tests, and its RNG is unseeded
 * `rand::thread_rng`): BVH build/traversal, slab test, scatter functions, libm calls.
 * For those, two pieces are *defined here* and restated independently by the product:
 *   (1) "trt-rng v1": xoroshiro64* streams keyed by (seed, pixel, sample);
 *   (2) "trt-math v2": sin/cos/acos/cbrt as fixed f32 polynomial algorithms
 *       (explicit fmaf only, no contraction) so that host and device agree bit for bit.
 * `orc_set_use_libm(1)` switches (2) to the platform libm, as the Rust original would
 * use, for statistical cross-checks.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_vec3;                 /* math/vec3.rs:9-15   (12 B) */
typedef struct { orc_vec3 origin, direction; } orc_ray;      /* ray.rs:4-9          (24 B) */
typedef struct { orc_vec3 min, max; } orc_aabb;              /* hittable/aabb.rs:5-10 (24 B) */
typedef struct { uint32_t x, y; orc_ray ray; } orc_sample_point;      /* pointgen.rs:7-13 (32 B) */
typedef struct { uint32_t x, y; orc_vec3 color; } orc_sampled_color;  /* imager.rs:9-15   (20 B) */

enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_DIELECTRIC = 2, ORC_LIGHT = 3 };

typedef struct {
    float t;
    orc_vec3 point, normal;
    int32_t front_face;
    int32_t material;          /* index into the world's material table */
} orc_hit_record;              /* hittable/mod.rs:19-25 */

typedef struct {
    orc_vec3 position, viewport_upper_left, forward, horizontal, vertical;
    orc_vec3 defocus_disk_u, defocus_disk_v;
    uint32_t width, height;
} orc_camera;                  /* camera.rs:4-14 */

typedef struct {
    uint64_t samples;          /* single_point_sampling calls */
    uint64_t rays;             /* world.hit calls, cpu.rs:48 */
    uint64_t node_tests;       /* AABB::intersect calls, bvh.rs:89 */
    uint64_t sphere_tests;     /* Sphere::hit calls */
    uint64_t quad_plane_tests; /* Quad::hit calls */
    uint64_t quad_inside_tests;/* Quad::hit calls whose t passed the range check, quad.rs:37 */
    uint64_t shades;           /* hits (material evaluated), cpu.rs:49-51 */
} orc_stats;

typedef struct {
    uint32_t spp;              /* Renderer::samples_per_pixel: the 1/spp scale, imager.rs:35 */
    uint32_t max_bounces;      /* Renderer::max_bounces */
    orc_vec3 background;       /* Renderer::background_color */
    uint32_t seed;             /* trt-rng v1 seed */
    uint32_t sample_begin, sample_end;   /* render samples [begin,end) of 0..spp */
    uint32_t row_begin, row_end;         /* render image rows [begin,end) */
    uint32_t accumulate;       /* 0: start pixels at 0; 1: continue from accum's content */
} orc_render_params;

typedef struct orc_world orc_world;

/* ---- World (hittable/world.rs:16-45) ---- */
orc_world *orc_world_new(void);
void orc_world_free(orc_world *);
/* returns material index, or -1 if the name is already present (the reference panics, world.rs:29-31) */
int orc_world_add_material(orc_world *, const char *name, int kind, orc_vec3 albedo_or_color, float param);
int orc_world_get_material(const orc_world *, const char *name);       /* -1 if absent */
int orc_world_add_sphere(orc_world *, orc_vec3 center, float radius, int material);
int orc_world_add_quad(orc_world *, orc_vec3 corner, orc_vec3 u, orc_vec3 v, int material);
int orc_world_add_spheres(orc_world *, int n, const float *xyzr /* 4n */, const int32_t *material /* n */);   /* the loop of add_sphere, array order */
int orc_world_num_geometries(const orc_world *);
/* Build the BVH (World::get_bvh → BVH::new, bvh.rs:12-22,42-84).  Called lazily by render. */
void orc_world_build(orc_world *);
/* Pre-order dump of the pointer BVH: per node bbox(6 floats), prim (-1 for inner, else geometry
 * insertion index), subtree size.  Returns the node count; writes at most cap nodes. */
int orc_world_bvh_dump(orc_world *, float *bbox6, int32_t *prim, int32_t *subtree, int cap);

/* ---- closest hit through the BVH (bvh.rs:24-27,88-107); 1 = hit ---- */
int orc_world_hit(orc_world *, const orc_ray *, float t0, float t1, orc_hit_record *out, orc_stats *stats);
/* brute force over all geometries in insertion order, strict '<' narrowing (no BVH) */
int orc_world_hit_bruteforce(orc_world *, const orc_ray *, float t0, float t1, orc_hit_record *out);

/* ---- Camera (camera.rs:17-66) ---- */
void orc_camera_new(orc_camera *out, float focus_distance, float defocus_angle_deg, orc_vec3 position,
                    orc_vec3 look_at, orc_vec3 up, float vertical_fov_deg, uint32_t width, uint32_t height);

/* ---- Renderer::render (renderer.rs:37-79) = pointgen + CpuSampler + Imager accumulation.
 * accum: width*height*3 f32 linear accumulators (imager.rs:43,50), full image; only rows
 * [row_begin,row_end) are touched.  nthreads: rows are the work unit. */
void orc_render(orc_world *, const orc_camera *, const orc_render_params *, float *accum,
                orc_stats *stats, int nthreads);

/* ---- the literal Sampler plug-in form (sampler/mod.rs:10-17): n SamplePoints → n SampledColors.
 * RNG stream for point i is (seed, pixel=i, sample=0); no primary-ray draws are consumed. */
void orc_sample_batch(orc_world *, const orc_sample_point *in, uint32_t n, orc_sampled_color *out,
                      uint32_t max_bounces, orc_vec3 background, uint32_t seed, orc_stats *stats);

/* ---- Imager finalisation + Image/Color (imager.rs:52-53, utils/image.rs:92-111) ---- */
void orc_tonemap_u8(const float *accum, uint32_t npixels, float gamma, uint8_t *rgb);
float orc_powf(float x, float y);                 /* trt-math v2 powf (libm powf when orc_set_use_libm(1)) */
float orc_gamma_correct(float c, float gamma);    /* Color::gamma_correction, one channel */

/* ---- unit entry points for the reference's known-answer tests ---- */
int orc_sphere_hit(orc_vec3 center, float radius, const orc_ray *, float t0, float t1, orc_hit_record *out);
int orc_quad_hit(orc_vec3 corner, orc_vec3 u, orc_vec3 v, const orc_ray *, float t0, float t1, orc_hit_record *out);
int orc_aabb_intersect(const orc_aabb *, const orc_ray *, float t0, float t1);
orc_aabb orc_sphere_bbox(orc_vec3 center, float radius);
orc_aabb orc_quad_bbox(orc_vec3 corner, orc_vec3 u, orc_vec3 v);
orc_ray orc_ray_new(orc_vec3 origin, orc_vec3 direction);          /* normalises, ray.rs:12-14 */
orc_vec3 orc_ray_at(const orc_ray *, float t);
orc_vec3 orc_vec3_binop(int op, orc_vec3 a, orc_vec3 b);           /* 0 add 1 sub 2 mul 3 div 4 cross */
orc_vec3 orc_vec3_scale(int op, orc_vec3 a, float s);              /* 0 mul 1 div */
float orc_vec3_dot(orc_vec3 a, orc_vec3 b);
float orc_vec3_length(orc_vec3 a);
int orc_vec3_eq(orc_vec3 a, orc_vec3 b);                           /* tolerant ==, vec3.rs:189-205 */
orc_vec3 orc_vec3_reflect(orc_vec3 v, orc_vec3 n);
orc_vec3 orc_vec3_refract(orc_vec3 v, orc_vec3 n, float eta);
/* scatter one material; returns 1 if scattered.  rng = 2-word trt-rng v1 state (updated). */
int orc_material_scatter(int kind, orc_vec3 albedo, float param, const orc_ray *ray_in,
                         const orc_hit_record *rec, uint32_t rng[2], orc_ray *scattered, orc_vec3 *attenuation);

/* ---- trt-rng v1 / trt-math v2 ---- */
void orc_rng_seed(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t rng[2]);
uint32_t orc_rng_next_u32(uint32_t rng[2]);
float orc_rng_random(uint32_t rng[2]);                             /* [0,1)  utils/random.rs:11-13 */
float orc_rng_random_range(uint32_t rng[2], float lo, float hi);   /* [lo,hi) utils/random.rs:15-18 */
orc_vec3 orc_random_in_unit_sphere(uint32_t rng[2]);               /* vec3extend.rs:15-30 */
orc_vec3 orc_random_unit_vector(uint32_t rng[2]);                  /* vec3extend.rs:32-34 */
orc_vec3 orc_random_in_unit_disk(uint32_t rng[2]);                 /* vec3extend.rs:45-53 */
float orc_sinf(float), orc_cosf(float), orc_acosf(float), orc_cbrtf(float);
void orc_set_use_libm(int on);

#ifdef __cplusplus
}
#endif
#endif
