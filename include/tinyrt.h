/*
 * tinyrt.h — C ABI of the MI355X-native path-tracing sampler (libtinyrt.so).
 *
 * Drop-in boundary for ONE path of cheolwanpark/tiny-raytracer: the per-pixel Monte-Carlo
 * bounce loop (`CpuSampler::single_point_sampling`, raytracer/src/renderer/sampler/cpu.rs:39-65)
 * together with the primary-ray generation that feeds it (renderer/pointgen.rs:37-52,
 * camera.rs:58-66) and the f32 accumulation that drains it (renderer/imager.rs:34-60).
 *
 * The reference's plug-in point is `trait Sampler::sampling(world, Receiver<SamplePoint>,
 * Sender<SampledColor>)` (renderer/sampler/mod.rs:10-17), invoked by `Renderer::render`
 * (renderer/renderer.rs:45-49,68-70).  One 32-byte message in and one 20-byte message out
 * per sample cannot feed a GPU, so the boundary sits one level up, at the
 * World / Camera / Renderer::render() surface; `trt_sample_batch` keeps the literal
 * batch form (SamplePoint[] -> SampledColor[]) the reference's own GPU sampler uses
 * (renderer/sampler/metal/sampler.rs:49-65,107-130).
 *
 * Conventions: plain C, POD only, no exceptions cross this boundary.  Every function that
 * can fail returns `int`: 0 = TRT_OK, negative = error; `trt_last_error()` gives the
 * message of the calling thread's last failure.  (The reference panics instead:
 * cpu.rs:35,85; world.rs:29-31; renderer.rs:75-77.)  Handles are opaque and owned by the
 * library; every buffer is caller-allocated and caller-owned; the library keeps no caller
 * pointer after a call returns.  All arithmetic is f32 (`pub type Float = f32`, lib.rs:4).
 */
#ifndef TINYRT_H
#define TINYRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRT_ABI_VERSION 4   /* 2 (round 3): + trt_scene_trim, trt_streamed_launch_plan, trt_band_copy_plan
                             * 3 (round 4): tuning moved out of the process environment: trt_tuning (trt_render_params.tuning),
                             *              trt_scene_options + trt_scene_create_ex; trt_stats.gather_per_band
                             * 4 (round 5): + trt_world_add_spheres; trt_scene_options.top_nodes is ignored (the LDS cache of a large scene's upper
                             *              tree levels is gone: measured slower in every form); d_counters[12..15] = shades by material kind.
                             *              No struct changed size or moved a field. */

enum trt_status {
    TRT_OK = 0,
    TRT_ERR_INVALID_ARG = -1,     /* null pointer, bad index, empty world, bad range */
    TRT_ERR_DUPLICATE = -2,       /* material name already present (world.rs:29-31 panics) */
    TRT_ERR_NOT_FOUND = -3,       /* material name absent (world.rs:35-41 returns None) */
    TRT_ERR_HIP = -4,             /* HIP runtime error; message carries hipGetErrorString */
    TRT_ERR_NO_DEVICE = -5,       /* no gfx950 device visible: the product never falls back to a CPU path */
    TRT_ERR_OOM = -6
};

/* ---- POD layouts fixed by the reference (#[repr(C)], tightly packed, 4-byte aligned) ---- */
typedef struct { float x, y, z; } trt_vec3;                          /* math/vec3.rs:9-15      12 B */
typedef struct { trt_vec3 origin, direction; } trt_ray;              /* ray.rs:4-9             24 B */
typedef struct { uint32_t x, y; trt_ray ray; } trt_sample_point;     /* renderer/pointgen.rs:7-13  32 B */
typedef struct { uint32_t x, y; trt_vec3 color; } trt_sampled_color; /* renderer/imager.rs:9-15    20 B */

/* ---- materials (material/{lambertian,metal,dielectric,light}.rs) ---- */
enum trt_material_kind {
    TRT_LAMBERTIAN = 0,   /* Lambertian::new(albedo)                  lambertian.rs:10-12 */
    TRT_METAL = 1,        /* Metal::new(albedo, fuzz), fuzz clamped   metal.rs:12-14      */
    TRT_DIELECTRIC = 2,   /* Dielectric::new(albedo, refraction_index) dielectric.rs:12-14 */
    TRT_LIGHT = 3         /* Light::new(color)                        light.rs:11-13      */
};
typedef struct {
    uint32_t kind;        /* enum trt_material_kind */
    trt_vec3 albedo;      /* albedo, or emitted colour for TRT_LIGHT */
    float param;          /* fuzz (metal) | refraction index (dielectric) | unused */
} trt_material;

/* ---- World: the scene container (hittable/world.rs:10-45) ---- */
typedef struct trt_world trt_world;
int trt_world_create(trt_world **out);                                            /* World::new        world.rs:16-21 */
void trt_world_destroy(trt_world *w);
int trt_world_add_material(trt_world *w, const char *name, const trt_material *m); /* World::add_material world.rs:27-33 */
int trt_world_get_material(const trt_world *w, const char *name, uint32_t *index); /* World::get_material world.rs:35-41 */
/* World::add_geometry(Box::new(Sphere::new(center, radius, material)))  world.rs:23-25, sphere.rs:16-26 */
int trt_world_add_sphere(trt_world *w, trt_vec3 center, float radius, uint32_t material);
/* The same for n spheres in array order - exactly the loop `for i in 0..n { world.add_geometry(Sphere::new(..)) }` (world.rs:23-25), one call
 * instead of n for scenes of millions of primitives.  center_radius: 4n floats (x, y, z, radius); material: n indices.  All or nothing:
 * an index out of range adds no sphere. */
int trt_world_add_spheres(trt_world *w, uint32_t n, const float *center_radius, const uint32_t *material);
/* World::add_geometry(Box::new(Quad::new(corner, u, v, material)))      world.rs:23-25, quad.rs:20-29 */
int trt_world_add_quad(trt_world *w, trt_vec3 corner, trt_vec3 u, trt_vec3 v, uint32_t material);
int trt_world_num_geometries(const trt_world *w);
int trt_world_num_materials(const trt_world *w);

/* ---- Scene: World::get_bvh() (world.rs:43-45 -> bvh.rs:12-22,42-84), flattened for the GPU ----
 * Host-only work (BVH build in the reference's median-split order, threading into a
 * pre-order skip-link array, packing into 16-byte planes).  Device upload happens lazily
 * on first render, so a scene can be compiled and inspected on a machine without a GPU. */
typedef struct trt_scene trt_scene;
int trt_scene_create(const trt_world *w, trt_scene **out);                          /* = trt_scene_create_ex(w, NULL, out) */
/* How a scene is compiled and how much idle device scratch its handle keeps.  PLACEMENT ONLY: every value renders the same
 * frames, bit for bit (the reference has one BVH layout and no such choice: bvh.rs:42-84).  Fill with
 * trt_scene_options_default() and change fields; NULL = the defaults. */
typedef struct {
    float cull_prune;             /* culling tree: an inner node whose box is >= this share of its nearest kept ancestor's is dropped (0.5) */
    int32_t flat_walk;            /* lock-step leaf list instead of a tree walk: -1 = for at most 32 primitives (default), 0 never, 1 always */
    int32_t compact_nodes;        /* 16-byte f16 culling nodes: -1 = for scenes too large for LDS (default), 0 never, 1 always */
    uint32_t top_nodes;           /* ignored since ABI 4 (rounds 1-4: upper tree levels of a large scene cached in LDS; slower in every form measured) */
    uint64_t scratch_cap_bytes;   /* idle scratch (workspaces + context frames) kept per device between renders; default 32 GiB */
    uint32_t reserved[6];         /* zero */
} trt_scene_options;
void trt_scene_options_default(trt_scene_options *out);
int trt_scene_create_ex(const trt_world *w, const trt_scene_options *options, trt_scene **out);
/* Must not run while another host thread is inside a render call on this scene; renders enqueued with trt_render_device
 * that still run on the device are waited for. */
void trt_scene_destroy(trt_scene *s);
/* The scene handle caches device resources per device: the uploaded scene, render scratch ("workspaces": up to 8 per device,
 * each 12 bytes per pixel and sample of one launch: 3.2 GB for 64 spp at 2048x2048, at most 16 GB - trt_tuning.radiance_gb - for
 * renders of 256 spp and more; renders enqueued back to back on one stream share ONE) and, for the blocking entry points, contexts
 * (stream, events, counters, a device frame).
 * Idle scratch beyond 32 GiB per device (trt_scene_options.scratch_cap_bytes) is freed when a render ends; this call frees ALL idle
 * scratch now (whatever running renders own is skipped).  The uploaded scene stays. */
int trt_scene_trim(trt_scene *s);

typedef struct {
    uint32_t num_nodes, num_spheres, num_quads, num_materials;
    uint32_t max_depth;           /* BVH depth (root = 1) */
    uint32_t device_bytes;        /* size of the packed scene in HBM */
    uint32_t lds_bytes;           /* bytes the kernels stage into LDS (0 = traverses from global memory) */
    uint32_t num_cull_nodes;      /* nodes of the culling tree the kernels walk (same leaves, same order, fewer inner nodes) */
} trt_scene_info;
int trt_scene_get_info(const trt_scene *s, trt_scene_info *out);
/* Pre-order node dump of the REFERENCE tree (bvh.rs:42-84, node for node): bbox6[6*i..] = min.xyz,max.xyz;
 * prim[i] = geometry insertion index or -1 for an inner node; skip[i] = pre-order index of the next node
 * once subtree i is done. */
int trt_scene_get_nodes(const trt_scene *s, float *bbox6, int32_t *prim, int32_t *skip, uint32_t cap);
/* Same dump of the CULLING tree: another hierarchy over the reference tree's leaf sequence (identical leaf boxes
 * in identical order, inner boxes = exact unions), which gives bit-identical hits with fewer box tests. */
int trt_scene_get_cull_nodes(const trt_scene *s, float *bbox6, int32_t *prim, int32_t *skip, uint32_t cap);
/* Scenes too large for LDS also carry the culling tree as 16-byte nodes, boxes rounded OUTWARD to IEEE half precision
 * (one load per box step instead of two; postponed leaves are re-tested against their exact f32 boxes).  Copies
 * num_cull_nodes x 4 words: (lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16, link), link = skip index of an
 * inner node (the device copy keeps it as a byte offset, index x 16), or 0x80000000 | leaf sequence number.  Returns
 * TRT_ERR_NOT_FOUND if the scene has no such array. */
int trt_scene_get_compact_nodes(const trt_scene *s, uint32_t *words4, uint32_t cap);

/* ---- Camera (camera.rs:4-14, 17-56) ---- */
typedef struct {
    trt_vec3 position, viewport_upper_left, forward, horizontal, vertical;
    trt_vec3 defocus_disk_u, defocus_disk_v;
    uint32_t width, height;
} trt_camera;
/* Camera::new(focus_distance, defocus_angle[deg], position, look_at, up, vertical_fov[deg], width, height) */
int trt_camera_init(trt_camera *out, float focus_distance, float defocus_angle_deg, trt_vec3 position,
                    trt_vec3 look_at, trt_vec3 up, float vertical_fov_deg, uint32_t width, uint32_t height);

/* ---- Renderer (renderer/renderer.rs:12-35) ---- */
enum trt_backend {
    TRT_BACKEND_MEGAKERNEL = 0,   /* one persistent lane per pixel, whole bounce loop in one kernel */
    TRT_BACKEND_WAVEFRONT = 1,    /* workgroup-resident wavefront: path state SoA in HBM, ray queues in LDS,
                                     generate / extend / sort-by-material / shade as phases of one persistent kernel */
    TRT_BACKEND_AUTO = 2,         /* the fastest measured backend (currently TRT_BACKEND_STREAMED for every scene) */
    TRT_BACKEND_STREAMED = 3      /* samples as work items pulled by persistent waves; radiances folded per pixel in sample order */
};
/* Scheduling of a render.  EVERY field is scheduling or placement only: any value renders the same frame, bit for bit, with the
 * same ray count (each is covered by a bit-equality test).  The reference configures a render through constructor arguments
 * (renderer.rs:21-35), never through the environment; so does this library: fill with trt_tuning_default() - the built-in
 * defaults, overridden ONCE, when the library is loaded, by the TRT_* environment variables named below (for sweeps from a shell) -
 * change fields, and hand it over in trt_render_params.tuning (NULL = trt_tuning_default()).  Nothing on the launch path reads
 * the environment; two threads may render one scene with different tunings at the same time. */
typedef struct {
    uint32_t stream_waves_per_simd;   /* streamed backend: waves per SIMD grid and launch bound are sized for; 0 = by scene (6 for scenes
                                         in LDS, 7 for scenes in global memory that fit the 32 MiB of L2, 8 beyond); 5..8 (4 with dual_walk on a
                                         scene in global memory)                                            TRT_STREAM_MINW */
    uint32_t stream_big_threads;      /* lanes per workgroup for LDS scene copies above 20 KB: 0 = auto, 512, 768   TRT_BIG_THREADS */
    uint32_t stream_batch_spp;        /* samples per pixel in one work batch of a wave (8)                  TRT_STREAM_BATCH_SPP */
    uint32_t radiance_gb;             /* radiance records of one streamed launch, GiB (16; 1..64)           TRT_RADIANCE_GB */
    uint32_t leaf_slots;              /* postponed leaves per lane and walk; 0 = by launch plan             TRT_LEAF_SLOTS */
    uint32_t lds_leaf_stack;          /* where they live: 0 registers, 1 LDS where it costs no occupancy (default), 2 LDS always   TRT_LDS_LEAF_STACK */
    uint32_t ray_pool;                /* 1 (default): per-wave LDS pool of primary rays where it fits; 0: one ray in stock per lane   TRT_RAY_POOL */
    uint32_t stragglers;              /* 16-byte-node walk: lanes that may carry an unfinished walk into the next round (8; 0 = none)   TRT_STRAGGLERS */
    uint32_t lds_stragglers;          /* the same for the LDS tree walk (8)                                 TRT_LDS_STRAGGLERS */
    uint32_t dual_walk;               /* scenes in global memory: two paths per lane, two node loads in flight per wave: 0 = by scene (on where the
                                         scene's hot part exceeds the 32 MiB of L2), 1 = wherever the kernel exists, 2 = never   TRT_DUAL_WALK */
    uint32_t runtime_walk;            /* 1: the kernels that choose the walk at run time instead of the specialised ones (0)   TRT_RUNTIME_WALK */
    uint32_t xcd_remap;               /* 1: contiguous image regions per XCD (0: measured 2x slower)        TRT_XCD_REMAP */
    uint32_t mega_waves_per_simd;     /* megakernel backend: 0 = default (7)                                TRT_MINW */
    uint32_t mega_threads;            /* megakernel: lanes per workgroup, 0 = auto, 256, 512                TRT_MEGA_THREADS */
    uint32_t mega_global_waves8;      /* megakernel on scenes in global memory: 8 waves per SIMD (0)        TRT_MINW8 */
    uint32_t wf_waves_per_simd;       /* wavefront backend: 0 = default                                     TRT_WF_MINW */
    uint32_t wf_serve_min;            /* wavefront backend: lanes that must wait before a refill; 0 = default (12)   TRT_WF_SERVE_MIN */
    uint32_t reserved[7];             /* zero */
} trt_tuning;
void trt_tuning_default(trt_tuning *out);

typedef struct {
    uint32_t samples_per_pixel;   /* Renderer::samples_per_pixel: fixes the 1/spp scale (imager.rs:35) */
    uint32_t max_bounces;         /* Renderer::max_bounces */
    trt_vec3 background;          /* Renderer::background_color (None -> 0, renderer.rs:33) */
    uint32_t seed;                /* trt-rng v1 seed (the reference has no seed API: utils/random.rs:15-18) */
    uint32_t backend;             /* enum trt_backend */
    /* progressive / sharded rendering; zero-initialised = whole image, all samples */
    uint32_t sample_begin, sample_end;   /* render samples [begin,end) of 0..spp; end==0 means spp */
    uint32_t accumulate;                 /* 0: pixels start at 0; 1: continue the running sums in the buffer */
    /* image rows owned by this call: local row r (0..rows_local) is image row
     *   ((r / band_rows) * band_stride + band_offset) * band_rows + r % band_rows.
     * band_rows==0 means the identity map over all `height` rows. */
    uint32_t band_rows, band_stride, band_offset, rows_local;
    uint32_t collect_stats;       /* 0: count samples and rays only.  1: counting kernel variant walking the REFERENCE tree:
                                     node/primitive test counts equal the CPU path's (SURVEY §8d's algorithmic bytes).
                                     2: counting variant walking the culling tree: the box tests actually performed. */
    const trt_tuning *tuning;     /* scheduling knobs (see trt_tuning); NULL = trt_tuning_default().  Read during the call only. */
} trt_render_params;

typedef struct {
    uint64_t samples;             /* single_point_sampling calls */
    uint64_t rays;                /* closest-hit queries = world.hit calls (cpu.rs:48): the Mray/s unit */
    uint64_t node_tests;          /* AABB slab tests (bvh.rs:89); 0 unless collect_stats */
    uint64_t sphere_tests;        /* 0 unless collect_stats */
    uint64_t quad_plane_tests;    /* 0 unless collect_stats */
    uint64_t quad_inside_tests;   /* 0 unless collect_stats */
    uint64_t shades;              /* hits whose material was evaluated; 0 unless collect_stats */
    double kernel_ms;             /* device time of the launch(es), HIP events on the launch stream (host-buffer calls only) */
    uint64_t wave_trips[4];       /* diagnostics, collect_stats only: per-wave loop trips (bounce rounds, box-test steps,
                                     leaf phases, ray generations); lane-level counts / (64 x these) = SIMD utilisation */
    uint64_t gather_per_band;     /* trt_render_multi*: shards whose rows were gathered band by band instead of by one strided 2-D copy
                                     (no peer access between two devices, or the runtime refused the strided peer copy); else 0 */
} trt_stats;

/* Renderer::render(camera, world) (renderer.rs:37-79), synchronous.  `accum` is a HOST buffer
 * of rows*width*3 f32 linear running sums (Imager's `pixels`, imager.rs:43,50), rows =
 * rows_local or height.  Gamma/quantisation is not applied: see trt_tonemap_u8. */
int trt_render(trt_scene *s, const trt_camera *cam, const trt_render_params *p, float *accum, trt_stats *stats);

/* Renderer::render over several GPUs of one node: still ONE call that returns the whole frame (renderer.rs:37-79;
 * src/main.rs:19).  The scene is replicated, the image is cut into bands of 16 rows dealt round-robin over the shards
 * (band b -> shard b % ndev), one host thread per shard drives its device, and each shard's bands are copied straight to their
 * place in `accum` (HOST buffer, height*width*3 f32; read first when p->accumulate is set) with one strided 2-D copy
 * (trt_band_copy_plan).  The frame is bit-identical for
 * every ndev and equals trt_render's: the RNG is keyed by the image pixel and every pixel is folded in sample order.
 * `devices`: ndev device ordinals (a device may appear more than once: its shards then run concurrently on it), or NULL
 * for 0..ndev-1; ndev == 0 means every visible device.  p->band_rows must be 0.  stats (may be NULL): counters summed over
 * the shards, kernel_ms of the slowest one. */
int trt_render_multi(trt_scene *s, const trt_camera *cam, const trt_render_params *p, const int *devices, uint32_t ndev,
                     float *accum, trt_stats *stats);
/* The same with the frame gathered into HBM: `d_accum` is a buffer of height*width*3 f32 on devices[0] (device 0 when
 * `devices` is NULL); every shard sends its bands there with one strided 2-D device-to-device copy (xGMI between the GPUs of a node; staged
 * through the host per band where two devices have no peer access).
 * Synchronous: the frame is complete when the call returns. */
int trt_render_multi_device(trt_scene *s, const trt_camera *cam, const trt_render_params *p, const int *devices,
                            uint32_t ndev, float *d_accum, trt_stats *stats);
/* Rows of an image of `height` rows that shard `rank` of `ndev` owns under that band layout (host arithmetic only). */
int trt_band_rows_local(uint32_t height, uint32_t ndev, uint32_t rank, uint32_t *rows_local);
/* The gather of one shard as trt_render_multi[_device] performs it (host arithmetic only; bytes).  A shard keeps its rows
 * contiguous; in the frame its k-th band starts at row (k * ndev + rank) * 16, i.e. at a constant pitch: ONE strided 2-D copy
 * (`full_bands` rows of `band_bytes`, source pitch `local_pitch`, destination `frame_offset` + k * `frame_pitch`) moves all
 * full bands, one 1-D copy of `tail_bytes` the ragged last band if the shard owns it. */
typedef struct {
    uint32_t rows_local, full_bands, tail_rows, reserved;
    uint64_t band_bytes, local_pitch, frame_pitch, frame_offset;
    uint64_t tail_bytes, tail_local_offset, tail_frame_offset;
} trt_band_copy;
int trt_band_copy_plan(uint32_t width, uint32_t height, uint32_t ndev, uint32_t rank, trt_band_copy *out);

/* Same, on buffers already resident in HBM.  `d_accum`: device pointer, rows*width*3 f32.
 * `d_counters`: device pointer to 16 uint64 (zeroed by the caller; [0..6] = trt_stats' first seven
 * fields, [8..11] = wave_trips, [12..15] with collect_stats: `shades` by material kind in enum trt_material_kind order) or NULL.  `stream`: a hipStream_t (NULL = default stream).  Asynchronous: returns after
 * enqueueing; the caller synchronises the stream.
 *
 * Concurrency (all render entry points): a scene is immutable once created and may be rendered by several host threads
 * and on several streams at the same time; every render takes private device scratch from a pool on the scene handle
 * (at most 8 scratch buffers per device: further concurrent renders queue behind running ones on the device).
 * trt_scene_destroy must not run while a host thread is inside a render call of that scene. */
int trt_render_device(trt_scene *s, const trt_camera *cam, const trt_render_params *p, float *d_accum,
                      uint64_t *d_counters, void *stream);

/* The literal Sampler plug-in form (sampler/mod.rs:10-17): n SamplePoints in, n SampledColors
 * out, HOST buffers.  Point i uses RNG stream (seed, pixel=i, sample=0).  With `stats` non-NULL the counting kernel runs
 * (reference-order walk of the reference tree: its counters equal the CPU path's); with NULL the production walk. */
int trt_sample_batch(trt_scene *s, const trt_sample_point *in, uint32_t n, trt_sampled_color *out,
                     uint32_t max_bounces, trt_vec3 background, uint32_t seed, trt_stats *stats);

/* Imager finalisation + Image -> RgbImage (imager.rs:52-53; utils/image.rs:92-111): c^(1/gamma),
 * clamp to [0, 0.999], *255, truncate; NaN -> 0.  HOST buffers, npixels*3 each. */
int trt_tonemap_u8(const float *accum, uint32_t npixels, float gamma, uint8_t *rgb);

/* The same on buffers resident in HBM (device pointers), asynchronous on `stream` (a hipStream_t, NULL = default): the
 * frame never has to leave the GPU as f32.  Host form and device kernel evaluate c^(1/gamma) with the same function (trt-math v2
 * powf, csrc/trt_pow.h): their u8 frames are equal byte for byte.  Against the reference, which calls the platform's libm powf
 * (utils/image.rs:94-96), a channel may differ by one least-significant bit where that powf is not correctly rounded. */
int trt_tonemap_u8_device(const float *d_accum, uint32_t npixels, float gamma, uint8_t *d_rgb, void *stream);

/* How the streamed backend launches a render of this scene with these settings (host arithmetic only: works without a GPU).
 * The kernels' view of their dynamic LDS - scene copy | postponed-leaf stack (threads x leaf_slots x 8 B) | ray pool (36 B per
 * lane) - is decided in ONE place (streamed.hip streamed_launch_plan) and reported here, so that its invariants can be checked
 * for every scene size and every tuning knob without a device (tests/test_host_boundary.py). */
typedef struct {
    uint32_t scene_mode;              /* 0 scene read from global memory, 1 whole hot scene copied into LDS */
    uint32_t threads_per_workgroup, waves_per_simd, workgroups_per_cu;
    uint32_t lds_bytes, scene_lds_bytes;      /* dynamic LDS per workgroup; the scene copy's share */
    uint32_t leaf_slots, lds_leaf_stack, ray_pool;
    uint32_t walk;                    /* 1 tree walk with LDS leaf stack, 2 lock-step leaf list, 3 16-byte nodes, 5 tree walk with register slots */
    uint32_t specialised;             /* 1: a kernel with the walk fixed at compile time */
    uint32_t has_kernel;              /* 0 would be a bug: no instantiation for the plan (the launch then fails, it never falls back) */
    uint32_t kernel_waves_per_simd, kernel_threads, kernel_walk /* 0 = chosen at run time */, kernel_ray_pool, kernel_counting;
    uint32_t chunk_spp;               /* samples per pixel per launch under p->tuning (= trt_streamed_chunk_spp(width, rows) for the default tuning) */
    uint32_t dual_walk;               /* 1: two paths per lane (two leaf stacks per lane in LDS) */
    uint64_t workspace_bytes;         /* device scratch one render of this size takes from the scene's pool */
} trt_launch_plan;
int trt_streamed_launch_plan(const trt_scene *s, const trt_camera *cam, const trt_render_params *p, trt_launch_plan *out);

/* Samples per pixel the streamed backend traces per kernel launch for an image of this size under the default tuning (it splits
 * longer sample ranges into such chunks; one chunk = one tracing-kernel launch + one fold launch).  For another tuning:
 * trt_streamed_launch_plan's chunk_spp. */
uint32_t trt_streamed_chunk_spp(uint32_t width, uint32_t rows);

/* Measurement aid: between _begin and _end every launch of a render's dominant kernel (streamed backend: the sample
 * kernel, not the fold) is bracketed by HIP events on the stream it is launched on; _end waits for them and returns the
 * summed device time and the number of launches.  Process-wide switch; a launch's two events are paired on the host thread
 * that makes the launch, so renders on several threads, streams or devices (trt_render_multi) never mix their brackets. */
int trt_kernel_timing_begin(void);
int trt_kernel_timing_end(double *total_ms, uint32_t *launches);

/* Name of the GPU kernel that dominates a render of this scene with these settings ("trt::stream_pool_kernel", ...): what
 * a kernel trace of the call shows, for profiles and benchmark records.  "" on invalid arguments. */
const char *trt_dominant_kernel(const trt_scene *s, const trt_camera *cam, const trt_render_params *p);

/* ---- library ---- */
const char *trt_last_error(void);
int trt_device_count(void);               /* gfx950 devices visible; 0 without a GPU (never an error) */
int trt_set_device(int ordinal);          /* device used by this thread's later calls */
uint32_t trt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TINYRT_H */
