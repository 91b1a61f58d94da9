// capi.hip — the extern "C" boundary declared in include/tinyrt.h.
//
// Mirrors the reference's World / Camera / Renderer::render() surface (hittable/world.rs:16-45,
// camera.rs:17-56, renderer/renderer.rs:21-79) as opaque handles + POD structs.  The reference
// panics on misuse; here every failure is a negative status plus a thread-local message, and no
// C++ exception leaves this file.  There is no CPU fallback: without a gfx950 device every
// compute entry point fails with TRT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "scene.h"

using namespace trt;

struct trt_world {
    World w;
};

// trt_kernel_timing_begin / _end: HIP events around every dominant-kernel launch, process-wide
namespace trt {
namespace {
std::mutex g_timing_mu;
bool g_timing_on = false;
std::vector<hipEvent_t> g_timing_events;        // begin, end, begin, end, ...
}  // namespace
void timing_mark(hipStream_t stream, bool begin) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    if (!g_timing_on) return;
    if (begin != (g_timing_events.size() % 2 == 0)) return;       // concurrent renders interleaved their marks: keep pairs intact
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventRecord(ev, stream) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev); return; }
    g_timing_events.push_back(ev);
}
}  // namespace trt

// Device scratch of one render (wavefront: path-state planes; streamed: the radiance records of a chunk + the batch
// counter).  Renders of one scene may run concurrently - from several host threads, on several streams - so a render
// never shares a workspace with another one that may still be running: each takes one from a per-(scene, device) pool
// and gives it back with an event recorded behind its last kernel; a workspace is handed out again only to a stream that
// first waits for that event (stream-ordered reuse), and the pool prefers workspaces whose event has already fired.
struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;      // recorded behind the last kernel that used the workspace
    bool recorded = false;
    bool busy = false;              // a host thread is between acquire and release (its launches are not all enqueued yet)
};
constexpr size_t kMaxWorkspacesPerDevice = 8;      // beyond that, renders queue behind each other on the device

struct trt_scene {
    SceneHost host;
    std::mutex mu;
    std::condition_variable cv;
    std::unordered_map<int, float4*> device_blob;     // device ordinal -> packed scene in HBM
    std::unordered_map<int, std::vector<Workspace*>> ws_pool;   // device ordinal -> workspaces (grown on demand)
};

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
int fail_hip(hipError_t e, const char* what) {
    return fail(TRT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define TRT_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return fail_hip(e_, #call);    \
    } while (0)

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(TRT_ERR_NO_DEVICE, "no HIP device visible: libtinyrt has no CPU path, it needs a gfx950 GPU");
    }
    return TRT_OK;
}

// Uploads the packed scene to the current device on first use.
int scene_on_device(trt_scene* s, SceneDev& out) {
    int dev = 0;
    TRT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->device_blob.find(dev);
    float4* d = nullptr;
    if (it == s->device_blob.end()) {
        TRT_HIP(hipMalloc(reinterpret_cast<void**>(&d), s->host.layout.blob_bytes));
        hipError_t e = hipMemcpy(d, s->host.blob.data(), s->host.layout.blob_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail_hip(e, "hipMemcpy(scene)"); }
        s->device_blob.emplace(dev, d);
    } else {
        d = it->second;
    }
    out.blob = d;
    out.L = s->host.layout;
    return TRT_OK;
}

// Takes a workspace of at least `need` bytes for a render that will be enqueued on `stream` of device `dev`.
int workspace_acquire(trt_scene* s, int dev, size_t need, hipStream_t stream, Workspace** out) {
    std::unique_lock<std::mutex> lock(s->mu);
    std::vector<Workspace*>& pool = s->ws_pool[dev];
    for (;;) {
        Workspace* pick = nullptr;
        // 1. an idle one whose last render has finished (no dependency at all); the smallest that fits, else any to regrow
        Workspace* finished_small = nullptr;
        for (Workspace* w : pool) {
            if (w->busy) continue;
            const bool finished = !w->recorded || hipEventQuery(w->done) == hipSuccess;
            if (!finished) continue;
            if (w->bytes >= need) { if (!pick || w->bytes < pick->bytes) pick = w; }
            else if (!finished_small) finished_small = w;
        }
        (void)hipGetLastError();                              // hipEventQuery's hipErrorNotReady is not an error
        if (!pick && finished_small) {                        // grow a finished one in place
            Workspace* w = finished_small;
            if (w->ptr) { hipError_t e = hipFree(w->ptr); w->ptr = nullptr; w->bytes = 0; if (e != hipSuccess) return fail_hip(e, "hipFree(workspace)"); }
            hipError_t e = hipMalloc(&w->ptr, need);
            if (e != hipSuccess) { w->ptr = nullptr; return fail_hip(e, "hipMalloc(workspace)"); }
            w->bytes = need;
            w->recorded = false;
            pick = w;
        }
        // 2. a new one while the pool may grow
        if (!pick && pool.size() < kMaxWorkspacesPerDevice) {
            Workspace* w = new (std::nothrow) Workspace();
            if (!w) return fail(TRT_ERR_OOM, "out of memory");
            hipError_t e = hipEventCreateWithFlags(&w->done, hipEventDisableTiming);
            if (e == hipSuccess) e = hipMalloc(&w->ptr, need);
            if (e != hipSuccess) {
                if (w->done) (void)hipEventDestroy(w->done);
                delete w;
                if (pool.empty()) return fail_hip(e, "hipMalloc(workspace)");
                (void)hipGetLastError();                      // out of HBM for another copy: queue behind a running render instead
            } else {
                w->bytes = need;
                pool.push_back(w);
                pick = w;
            }
        }
        // 3. an idle one that is still in flight and large enough: this render queues behind it on the device
        if (!pick) {
            for (Workspace* w : pool) if (!w->busy && w->bytes >= need) { pick = w; break; }
        }
        if (pick) {
            if (pick->recorded) TRT_HIP(hipStreamWaitEvent(stream, pick->done, 0));
            pick->busy = true;
            *out = pick;
            return TRT_OK;
        }
        // 4. every workspace is being enqueued on by another host thread, or is in flight and too small: wait for a release
        bool any_busy = false;
        for (Workspace* w : pool) any_busy = any_busy || w->busy;
        if (any_busy) { s->cv.wait(lock); continue; }
        // all idle, all in flight, all too small: wait for the first to finish, then regrow it
        TRT_HIP(hipEventSynchronize(pool.front()->done));
    }
}

// Gives the workspace back; `stream` has all of the render's launches enqueued.
int workspace_release(trt_scene* s, Workspace* w, hipStream_t stream) {
    const hipError_t e = hipEventRecord(w->done, stream);
    {
        std::lock_guard<std::mutex> lock(s->mu);
        w->recorded = (e == hipSuccess);
        w->busy = false;
    }
    s->cv.notify_all();
    if (e != hipSuccess) return fail_hip(e, "hipEventRecord(workspace)");
    return TRT_OK;
}

void to_camera_dev(const trt_camera& c, CameraDev& d) {
    const trt_vec3* src[6] = {&c.position, &c.viewport_upper_left, &c.horizontal, &c.vertical, &c.defocus_disk_u, &c.defocus_disk_v};
    float* dst[6] = {d.pos, d.upper_left, d.horizontal, d.vertical, d.du, d.dv};
    for (int i = 0; i < 6; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    d.width = c.width;
    d.height = c.height;
}

// Validates params against the camera and fills the kernel arguments.  rows = rows the call owns.
int to_render_args(const trt_camera* cam, const trt_render_params* p, RenderArgs& ra, uint32_t& rows) {
    if (cam->width < 2 || cam->height < 2) return fail(TRT_ERR_INVALID_ARG, "camera width and height must be at least 2 (pointgen.rs:41-42 divides by width-1, height-1)");
    if (p->samples_per_pixel == 0) return fail(TRT_ERR_INVALID_ARG, "samples_per_pixel must be positive");
    uint32_t s1 = p->sample_end == 0 ? p->samples_per_pixel : p->sample_end;
    if (p->sample_begin > s1 || s1 > p->samples_per_pixel) return fail(TRT_ERR_INVALID_ARG, "sample range must satisfy begin <= end <= samples_per_pixel");
    if (p->backend > TRT_BACKEND_STREAMED) return fail(TRT_ERR_INVALID_ARG, "unknown backend");
    ra.background[0] = p->background.x; ra.background[1] = p->background.y; ra.background[2] = p->background.z;
    ra.inv_spp = 1.0f / (float)p->samples_per_pixel;
    ra.max_bounces = p->max_bounces;
    ra.seed_key = rng_seed_key(p->seed);
    ra.sample_begin = p->sample_begin;
    ra.sample_end = s1;
    ra.accumulate = p->accumulate ? 1u : 0u;
    if (p->band_rows == 0) {
        ra.band_rows = 0; ra.band_stride = 1; ra.band_offset = 0;
        rows = cam->height;
    } else {
        if (p->band_stride == 0 || p->band_offset >= p->band_stride) return fail(TRT_ERR_INVALID_ARG, "band_offset must be below band_stride");
        rows = p->rows_local;
        if (rows > 0) {
            uint32_t last = rows - 1;
            uint64_t y = ((uint64_t)(last / p->band_rows) * p->band_stride + p->band_offset) * p->band_rows + last % p->band_rows;
            if (y >= cam->height) return fail(TRT_ERR_INVALID_ARG, "rows_local maps past the last image row");
        }
        ra.band_rows = p->band_rows; ra.band_stride = p->band_stride; ra.band_offset = p->band_offset;
    }
    ra.rows_local = rows;
    ra.leaf_slots = 0u;                                    // rt_path.h walk_fast; 0 = the backend's default; scheduling only, any value renders the same frame
    if (const char* e = getenv("TRT_LEAF_SLOTS")) ra.leaf_slots = (uint32_t)atoi(e);
    ra.lds_leaf_stack = 1u;
    if (const char* e = getenv("TRT_LDS_LEAF_STACK")) ra.lds_leaf_stack = (uint32_t)atoi(e);   // 0 off, 1 where it costs no occupancy, 2 always
    ra.xcd_aware = getenv("TRT_XCD_REMAP") ? 1u : 0u;   // off: contiguous image regions per XCD measured 2x slower (load imbalance)
    ra.stragglers = 12u;                                   // scheduling only: any value renders the same frame
    if (const char* e = getenv("TRT_STRAGGLERS")) ra.stragglers = (uint32_t)atoi(e);
    ra.ref_tree = p->collect_stats == 1 ? 1u : 0u;      // 1: counters comparable with the CPU path; 2: count the culling tree's own tests
    return TRT_OK;
}

int enqueue_render(trt_scene* s, const trt_camera* cam, const trt_render_params* p, float* d_accum, uint64_t* d_counters,
                   hipStream_t stream, uint32_t* rows_out) {
    RenderArgs ra;
    uint32_t rows = 0;
    int rc = to_render_args(cam, p, ra, rows);
    if (rc != TRT_OK) return rc;
    if (rows_out) *rows_out = rows;
    SceneDev sc;
    rc = scene_on_device(s, sc);
    if (rc != TRT_OK) return rc;
    CameraDev cd;
    to_camera_dev(*cam, cd);
    const size_t bytes = (size_t)rows * cam->width * 3 * sizeof(float);
    if (rows == 0 || ra.sample_begin == ra.sample_end || ra.max_bounces == 0) {
        // nothing to trace: a path with no bounce budget returns colour 0 (cpu.rs:43-47,64)
        if (!ra.accumulate && bytes) TRT_HIP(hipMemsetAsync(d_accum, 0, bytes, stream));
        return TRT_OK;
    }
    const bool wavefront = p->backend == TRT_BACKEND_WAVEFRONT;
    const bool streamed = p->backend == TRT_BACKEND_STREAMED || p->backend == TRT_BACKEND_AUTO;     // fastest on every scene measured
    if (wavefront || streamed) {
        // Device workspace (wavefront: 72 B of path state per pixel; streamed: 12 B per pixel and sample of a chunk), private
        // to this render until its last kernel has run (workspace_acquire): concurrent renders of one scene are safe.
        int dev = 0;
        TRT_HIP(hipGetDevice(&dev));
        const size_t need = wavefront ? wavefront_workspace_bytes(cam->width, rows) : streamed_workspace_bytes(cam->width, rows);
        Workspace* ws = nullptr;
        rc = workspace_acquire(s, dev, need, stream, &ws);
        if (rc != TRT_OK) return rc;
        hipError_t le;
        if (streamed) {
            le = launch_streamed(sc, cd, ra, ws->ptr, d_accum, reinterpret_cast<unsigned long long*>(d_counters), p->collect_stats != 0, stream);
        } else {
            uint32_t serve_min = 0;
            if (const char* e = getenv("TRT_WF_SERVE_MIN")) serve_min = (uint32_t)atoi(e);
            le = launch_wavefront(sc, cd, ra, ws->ptr, d_accum, reinterpret_cast<unsigned long long*>(d_counters), p->collect_stats != 0,
                                  serve_min, stream);
        }
        rc = workspace_release(s, ws, stream);
        if (le != hipSuccess) return fail_hip(le, streamed ? "launch_streamed" : "launch_wavefront");
        return rc;
    }
    TRT_HIP(launch_megakernel(sc, cd, ra, d_accum, reinterpret_cast<unsigned long long*>(d_counters), p->collect_stats != 0, stream));
    return TRT_OK;
}

void counters_to_stats(const unsigned long long* c, trt_stats* st) {
    st->samples = c[CTR_SAMPLES]; st->rays = c[CTR_RAYS]; st->node_tests = c[CTR_NODE]; st->sphere_tests = c[CTR_SPHERE];
    st->quad_plane_tests = c[CTR_QUAD_PLANE]; st->quad_inside_tests = c[CTR_QUAD_INSIDE]; st->shades = c[CTR_SHADE];
    for (int i = 0; i < 4; i++) st->wave_trips[i] = c[CTR_W_ROUNDS + i];
}

}  // namespace

extern "C" {

uint32_t trt_abi_version(void) { return TRT_ABI_VERSION; }
const char* trt_last_error(void) { return g_last_error.c_str(); }

int trt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int trt_set_device(int ordinal) {
    int rc = require_device();
    if (rc != TRT_OK) return rc;
    TRT_HIP(hipSetDevice(ordinal));
    return TRT_OK;
}

// ---- World ----
int trt_world_create(trt_world** out) {
    if (!out) return fail(TRT_ERR_INVALID_ARG, "out is null");
    trt_world* w = new (std::nothrow) trt_world();
    if (!w) return fail(TRT_ERR_OOM, "out of memory");
    *out = w;
    return TRT_OK;
}
void trt_world_destroy(trt_world* w) { delete w; }

int trt_world_add_material(trt_world* w, const char* name, const trt_material* m) {
    if (!w || !name || !m) return fail(TRT_ERR_INVALID_ARG, "null argument");
    if (m->kind > TRT_LIGHT) return fail(TRT_ERR_INVALID_ARG, "unknown material kind");
    try {
        std::string key(name);
        if (w->w.material_index.count(key)) return fail(TRT_ERR_DUPLICATE, key + " key is already in the material table");   // world.rs:30
        trt_material mm = *m;
        if (mm.kind == TRT_METAL) mm.param = fminf(fmaxf(mm.param, 0.0f), 1.0f);      // Metal::new clamps fuzz (metal.rs:12-14)
        w->w.material_index.emplace(key, (uint32_t)w->w.materials.size());
        w->w.materials.push_back(mm);
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    }
    return TRT_OK;
}
int trt_world_get_material(const trt_world* w, const char* name, uint32_t* index) {
    if (!w || !name || !index) return fail(TRT_ERR_INVALID_ARG, "null argument");
    auto it = w->w.material_index.find(name);
    if (it == w->w.material_index.end()) return fail(TRT_ERR_NOT_FOUND, std::string("no material named ") + name);
    *index = it->second;
    return TRT_OK;
}
int trt_world_add_sphere(trt_world* w, trt_vec3 center, float radius, uint32_t material) {
    if (!w) return fail(TRT_ERR_INVALID_ARG, "world is null");
    if (material >= w->w.materials.size()) return fail(TRT_ERR_INVALID_ARG, "material index out of range");
    try {
        w->w.geometries.push_back(Geometry{0u, material, center, trt_vec3{radius, 0.0f, 0.0f}, trt_vec3{0.0f, 0.0f, 0.0f}});
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    }
    return TRT_OK;
}
int trt_world_add_quad(trt_world* w, trt_vec3 corner, trt_vec3 u, trt_vec3 v, uint32_t material) {
    if (!w) return fail(TRT_ERR_INVALID_ARG, "world is null");
    if (material >= w->w.materials.size()) return fail(TRT_ERR_INVALID_ARG, "material index out of range");
    try {
        w->w.geometries.push_back(Geometry{1u, material, corner, u, v});
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    }
    return TRT_OK;
}
int trt_world_num_geometries(const trt_world* w) { return w ? (int)w->w.geometries.size() : 0; }
int trt_world_num_materials(const trt_world* w) { return w ? (int)w->w.materials.size() : 0; }

// ---- Scene ----
int trt_scene_create(const trt_world* w, trt_scene** out) {
    if (!w || !out) return fail(TRT_ERR_INVALID_ARG, "null argument");
    try {
        trt_scene* s = new trt_scene();
        std::string msg;
        if (!compile_scene(w->w, s->host, msg)) { delete s; return fail(TRT_ERR_INVALID_ARG, msg); }
        *out = s;
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    }
    return TRT_OK;
}
void trt_scene_destroy(trt_scene* s) {
    if (!s) return;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (auto& kv : s->ws_pool) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        for (Workspace* w : kv.second) {
            if (w->recorded) (void)hipEventSynchronize(w->done);      // a render may still be using it
            if (w->ptr) (void)hipFree(w->ptr);
            if (w->done) (void)hipEventDestroy(w->done);
            delete w;
        }
    }
    for (auto& kv : s->device_blob) {
        if (hipSetDevice(kv.first) == hipSuccess) (void)hipFree(kv.second);
    }
    if (have_prev) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    delete s;
}
int trt_scene_get_info(const trt_scene* s, trt_scene_info* out) {
    if (!s || !out) return fail(TRT_ERR_INVALID_ARG, "null argument");
    const SceneLayout& L = s->host.layout;
    out->num_nodes = L.n_nodes; out->num_spheres = L.n_spheres; out->num_quads = L.n_quads; out->num_materials = L.n_materials;
    out->max_depth = s->host.max_depth;
    out->device_bytes = L.blob_bytes;
    out->lds_bytes = scene_lds_bytes(L);
    out->num_cull_nodes = L.n_cull_nodes;
    return TRT_OK;
}
static int copy_nodes(const NodeDump& d, float* bbox6, int32_t* prim, int32_t* skip, uint32_t cap) {
    const uint32_t n = (uint32_t)d.skip.size();
    if (cap < n) return fail(TRT_ERR_INVALID_ARG, "cap is smaller than the node count");
    for (size_t k = 0; k < 6 * (size_t)n; k++) bbox6[k] = d.bbox6[k];
    for (uint32_t i = 0; i < n; i++) { prim[i] = d.prim_geo[i]; skip[i] = d.skip[i]; }
    return TRT_OK;
}
int trt_scene_get_nodes(const trt_scene* s, float* bbox6, int32_t* prim, int32_t* skip, uint32_t cap) {
    if (!s || !bbox6 || !prim || !skip) return fail(TRT_ERR_INVALID_ARG, "null argument");
    return copy_nodes(s->host.reference, bbox6, prim, skip, cap);
}
int trt_scene_get_cull_nodes(const trt_scene* s, float* bbox6, int32_t* prim, int32_t* skip, uint32_t cap) {
    if (!s || !bbox6 || !prim || !skip) return fail(TRT_ERR_INVALID_ARG, "null argument");
    return copy_nodes(s->host.culling, bbox6, prim, skip, cap);
}

int trt_scene_get_compact_nodes(const trt_scene* s, uint32_t* words4, uint32_t cap) {
    if (!s || !words4) return fail(TRT_ERR_INVALID_ARG, "null argument");
    const SceneLayout& L = s->host.layout;
    if (L.off_compact == 0u) return fail(TRT_ERR_NOT_FOUND, "scene has no compact node array (it is walked from LDS)");
    if (cap < L.n_cull_nodes) return fail(TRT_ERR_INVALID_ARG, "buffer too small");
    memcpy(words4, s->host.blob.data() + 16u * (size_t)L.off_compact, 16u * (size_t)L.n_cull_nodes);
    return TRT_OK;
}

// ---- Camera ----
int trt_camera_init(trt_camera* out, float focus_distance, float defocus_angle_deg, trt_vec3 position, trt_vec3 look_at,
                    trt_vec3 up, float vertical_fov_deg, uint32_t width, uint32_t height) {
    if (!out) return fail(TRT_ERR_INVALID_ARG, "out is null");
    if (width == 0 || height == 0) return fail(TRT_ERR_INVALID_ARG, "width and height must be positive");
    camera_init(*out, focus_distance, defocus_angle_deg, position, look_at, up, vertical_fov_deg, width, height);
    return TRT_OK;
}

// ---- Renderer::render ----
int trt_render_device(trt_scene* s, const trt_camera* cam, const trt_render_params* p, float* d_accum, uint64_t* d_counters,
                      void* stream) {
    if (!s || !cam || !p || !d_accum) return fail(TRT_ERR_INVALID_ARG, "null argument");
    int rc = require_device();
    if (rc != TRT_OK) return rc;
    return enqueue_render(s, cam, p, d_accum, d_counters, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int trt_render(trt_scene* s, const trt_camera* cam, const trt_render_params* p, float* accum, trt_stats* stats) {
    if (!s || !cam || !p || !accum) return fail(TRT_ERR_INVALID_ARG, "null argument");
    int rc = require_device();
    if (rc != TRT_OK) return rc;
    RenderArgs probe;
    uint32_t rows = 0;
    rc = to_render_args(cam, p, probe, rows);
    if (rc != TRT_OK) return rc;
    const size_t bytes = (size_t)rows * cam->width * 3 * sizeof(float);
    float* d_accum = nullptr;
    unsigned long long* d_ctr = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long h_ctr[CTR_COUNT] = {0};
    float ms = 0.0f;
    auto cleanup = [&]() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
        if (d_accum) (void)hipFree(d_accum);
        if (d_ctr) (void)hipFree(d_ctr);
    };
#define TRT_HIP_C(call)                                                          \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) { cleanup(); return fail_hip(e_, #call); }         \
    } while (0)
    TRT_HIP_C(hipStreamCreate(&stream));
    TRT_HIP_C(hipEventCreate(&ev0));
    TRT_HIP_C(hipEventCreate(&ev1));
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_accum), bytes ? bytes : 16));
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_ctr), sizeof(h_ctr)));
    TRT_HIP_C(hipMemsetAsync(d_ctr, 0, sizeof(h_ctr), stream));
    if (p->accumulate && bytes) TRT_HIP_C(hipMemcpyAsync(d_accum, accum, bytes, hipMemcpyHostToDevice, stream));
    TRT_HIP_C(hipEventRecord(ev0, stream));
    rc = enqueue_render(s, cam, p, d_accum, reinterpret_cast<uint64_t*>(d_ctr), stream, nullptr);
    if (rc != TRT_OK) { cleanup(); return rc; }
    TRT_HIP_C(hipEventRecord(ev1, stream));
    if (bytes) TRT_HIP_C(hipMemcpyAsync(accum, d_accum, bytes, hipMemcpyDeviceToHost, stream));
    TRT_HIP_C(hipMemcpyAsync(h_ctr, d_ctr, sizeof(h_ctr), hipMemcpyDeviceToHost, stream));
    TRT_HIP_C(hipStreamSynchronize(stream));
    TRT_HIP_C(hipEventElapsedTime(&ms, ev0, ev1));
    cleanup();
    if (stats) { counters_to_stats(h_ctr, stats); stats->kernel_ms = ms; }
    return TRT_OK;
}

// ---- Renderer::render over several GPUs of one node ----
//
// renderer.rs:37-79 is ONE call that returns the whole Image; so is this.  The path shards by pixels (every sample reads
// only the immutable scene, cpu.rs:39-65): the scene is replicated, the image is cut into bands of kMultiBandRows rows dealt
// round-robin (band b -> device b % ndev, so expensive regions spread over all devices), the RNG is keyed by the image
// pixel, and every device folds its own pixels in sample order - the frame is bit-identical for every ndev.  One host
// thread per device enqueues that device's bands on a stream of its own; the only exchange is the gather of the
// finished f32 sums: each band goes straight to its place in the destination frame (a host buffer: one D2H copy per
// band; a buffer on devices[0]: one peer copy over xGMI per band), so no un-interleave pass exists.
namespace {

constexpr uint32_t kMultiBandRows = 16;

struct MultiShard {
    int device = 0;
    uint32_t rank = 0, rows_local = 0;
    int rc = TRT_OK;
    std::string error;
    unsigned long long ctr[CTR_COUNT] = {0};
    float ms = 0.0f;
};

uint32_t band_rows_local(uint32_t height, uint32_t ndev, uint32_t rank, uint32_t band_rows) {
    const uint32_t n_bands = (height + band_rows - 1u) / band_rows;
    uint32_t rows = 0;
    for (uint32_t b = rank; b < n_bands; b += ndev) rows += (b + 1u) * band_rows <= height ? band_rows : height - b * band_rows;
    return rows;
}

// One device's share.  dst: the whole frame, on the host (dst_device < 0) or on device dst_device.
void render_shard(trt_scene* s, const trt_camera* cam, const trt_render_params* p, uint32_t ndev, MultiShard* sh, float* dst,
                  int dst_device) {
    float* d_accum = nullptr;
    unsigned long long* d_ctr = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
        if (d_accum) (void)hipFree(d_accum);
        if (d_ctr) (void)hipFree(d_ctr);
    };
#define TRT_HIP_S(call)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) { sh->rc = fail_hip(e_, #call); sh->error = g_last_error; cleanup(); return; }    \
    } while (0)
    TRT_HIP_S(hipSetDevice(sh->device));
    if (sh->rows_local == 0) return;                                        // more devices than bands
    const size_t row_bytes = (size_t)cam->width * 3 * sizeof(float);
    trt_render_params q = *p;
    q.band_rows = kMultiBandRows; q.band_stride = ndev; q.band_offset = sh->rank; q.rows_local = sh->rows_local;
    TRT_HIP_S(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    TRT_HIP_S(hipEventCreate(&ev0));
    TRT_HIP_S(hipEventCreate(&ev1));
    TRT_HIP_S(hipMalloc(reinterpret_cast<void**>(&d_accum), row_bytes * sh->rows_local));
    TRT_HIP_S(hipMalloc(reinterpret_cast<void**>(&d_ctr), sizeof(sh->ctr)));
    TRT_HIP_S(hipMemsetAsync(d_ctr, 0, sizeof(sh->ctr), stream));
    // local rows [r0, r0 + n) = image rows [y0, y0 + n): one contiguous band
    auto for_each_band = [&](auto&& fn) -> hipError_t {
        uint32_t r0 = 0;
        for (uint32_t b = sh->rank; r0 < sh->rows_local; b += ndev) {
            const uint32_t y0 = b * kMultiBandRows;
            const uint32_t n = y0 + kMultiBandRows <= cam->height ? kMultiBandRows : cam->height - y0;
            hipError_t e = fn(r0, y0, n);
            if (e != hipSuccess) return e;
            r0 += n;
        }
        return hipSuccess;
    };
    if (p->accumulate) {                                                    // continue the running sums of the frame
        TRT_HIP_S(for_each_band([&](uint32_t r0, uint32_t y0, uint32_t n) {
            float* local = d_accum + (size_t)r0 * cam->width * 3;
            const float* frame = dst + (size_t)y0 * cam->width * 3;
            return dst_device < 0 ? hipMemcpyAsync(local, frame, row_bytes * n, hipMemcpyHostToDevice, stream)
                                  : hipMemcpyPeerAsync(local, sh->device, frame, dst_device, row_bytes * n, stream);
        }));
    }
    TRT_HIP_S(hipEventRecord(ev0, stream));
    sh->rc = enqueue_render(s, cam, &q, d_accum, reinterpret_cast<uint64_t*>(d_ctr), stream, nullptr);
    if (sh->rc != TRT_OK) { sh->error = g_last_error; (void)hipStreamSynchronize(stream); cleanup(); return; }
    TRT_HIP_S(hipEventRecord(ev1, stream));
    TRT_HIP_S(for_each_band([&](uint32_t r0, uint32_t y0, uint32_t n) {     // the gather: every band to its place in the frame
        const float* local = d_accum + (size_t)r0 * cam->width * 3;
        float* frame = dst + (size_t)y0 * cam->width * 3;
        return dst_device < 0 ? hipMemcpyAsync(frame, local, row_bytes * n, hipMemcpyDeviceToHost, stream)
                              : hipMemcpyPeerAsync(frame, dst_device, local, sh->device, row_bytes * n, stream);
    }));
    TRT_HIP_S(hipMemcpyAsync(sh->ctr, d_ctr, sizeof(sh->ctr), hipMemcpyDeviceToHost, stream));
    TRT_HIP_S(hipStreamSynchronize(stream));
    TRT_HIP_S(hipEventElapsedTime(&sh->ms, ev0, ev1));
    cleanup();
#undef TRT_HIP_S
}

int render_multi(trt_scene* s, const trt_camera* cam, const trt_render_params* p, const int* devices, uint32_t ndev, float* dst,
                 bool dst_on_device, trt_stats* stats) {
    if (!s || !cam || !p || !dst) return fail(TRT_ERR_INVALID_ARG, "null argument");
    int rc = require_device();
    if (rc != TRT_OK) return rc;
    if (p->band_rows != 0) return fail(TRT_ERR_INVALID_ARG, "trt_render_multi lays out the bands itself: band_rows must be 0");
    const int visible = trt_device_count();
    if (ndev == 0) ndev = (uint32_t)visible;
    if (ndev > 64) return fail(TRT_ERR_INVALID_ARG, "at most 64 shards");
    std::vector<MultiShard> shards(ndev);
    for (uint32_t r = 0; r < ndev; r++) {
        const int d = devices ? devices[r] : (int)r;
        if (d < 0 || d >= visible) return fail(TRT_ERR_INVALID_ARG, "device ordinal out of range");
        shards[r].device = d;
        shards[r].rank = r;
        shards[r].rows_local = band_rows_local(cam->height, ndev, r, kMultiBandRows);
    }
    {   // validate once on the calling thread, so argument errors do not depend on a device
        RenderArgs probe;
        uint32_t rows = 0;
        rc = to_render_args(cam, p, probe, rows);
        if (rc != TRT_OK) return rc;
    }
    int prev = 0;
    TRT_HIP(hipGetDevice(&prev));
    const int dst_device = dst_on_device ? shards[0].device : -1;
    if (dst_on_device) {
        for (uint32_t r = 1; r < ndev; r++) {                               // peer access for the gather (xGMI)
            if (shards[r].device == dst_device) continue;
            int can = 0;
            TRT_HIP(hipDeviceCanAccessPeer(&can, shards[r].device, dst_device));
            if (!can) continue;                                             // hipMemcpyPeerAsync then stages through the host
            TRT_HIP(hipSetDevice(shards[r].device));
            hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipSetDevice(prev); return fail_hip(e, "hipDeviceEnablePeerAccess"); }
            (void)hipGetLastError();
        }
        TRT_HIP(hipSetDevice(prev));
    }
    if (ndev == 1) {
        render_shard(s, cam, p, 1, &shards[0], dst, dst_device);
    } else {
        std::vector<std::thread> workers;
        try {
            for (uint32_t r = 0; r < ndev; r++) workers.emplace_back(render_shard, s, cam, p, ndev, &shards[r], dst, dst_device);
        } catch (...) {
            for (auto& t : workers) t.join();
            (void)hipSetDevice(prev);
            return fail(TRT_ERR_OOM, "could not start a host thread per device");
        }
        for (auto& t : workers) t.join();
    }
    (void)hipSetDevice(prev);
    unsigned long long total[CTR_COUNT] = {0};
    float ms = 0.0f;
    for (const MultiShard& sh : shards) {
        if (sh.rc != TRT_OK) return fail(sh.rc, "device " + std::to_string(sh.device) + ": " + sh.error);
        for (int k = 0; k < CTR_COUNT; k++) total[k] += sh.ctr[k];
        if (sh.ms > ms) ms = sh.ms;
    }
    if (stats) { counters_to_stats(total, stats); stats->kernel_ms = ms; }      // kernel_ms: the slowest device's
    return TRT_OK;
}

}  // namespace

int trt_render_multi(trt_scene* s, const trt_camera* cam, const trt_render_params* p, const int* devices, uint32_t ndev, float* accum,
                     trt_stats* stats) {
    try {
        return render_multi(s, cam, p, devices, ndev, accum, false, stats);
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    }
}

int trt_render_multi_device(trt_scene* s, const trt_camera* cam, const trt_render_params* p, const int* devices, uint32_t ndev,
                            float* d_accum, trt_stats* stats) {
    try {
        return render_multi(s, cam, p, devices, ndev, d_accum, true, stats);
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    }
}

int trt_band_rows_local(uint32_t height, uint32_t ndev, uint32_t rank, uint32_t* rows_local) {
    if (!rows_local || ndev == 0 || rank >= ndev) return fail(TRT_ERR_INVALID_ARG, "need 0 <= rank < ndev and a result pointer");
    *rows_local = band_rows_local(height, ndev, rank, kMultiBandRows);
    return TRT_OK;
}

// ---- Sampler::sampling, batch form ----
int trt_sample_batch(trt_scene* s, const trt_sample_point* in, uint32_t n, trt_sampled_color* out, uint32_t max_bounces,
                     trt_vec3 background, uint32_t seed, trt_stats* stats) {
    if (!s || (n && (!in || !out))) return fail(TRT_ERR_INVALID_ARG, "null argument");
    int rc = require_device();
    if (rc != TRT_OK) return rc;
    if (stats) { *stats = trt_stats{}; }
    if (n == 0) return TRT_OK;
    if (max_bounces == 0) {                       // cpu.rs:43-47: the loop body never runs, colour stays 0
        for (uint32_t i = 0; i < n; i++) { out[i].x = in[i].x; out[i].y = in[i].y; out[i].color = trt_vec3{0.0f, 0.0f, 0.0f}; }
        if (stats) stats->samples = n;
        return TRT_OK;
    }
    SceneDev sc;
    rc = scene_on_device(s, sc);
    if (rc != TRT_OK) return rc;
    RenderArgs ra{};
    ra.background[0] = background.x; ra.background[1] = background.y; ra.background[2] = background.z;
    ra.inv_spp = 1.0f;
    ra.max_bounces = max_bounces;
    ra.seed_key = rng_seed_key(seed);
    // with `stats` the counting kernel walks the reference tree (its counters then equal the CPU path's); without, the production
    // kernel walks the culling tree with postponed leaves: same colours, several times faster
    const bool counting = stats != nullptr;
    ra.ref_tree = counting ? 1u : 0u;
    ra.leaf_slots = counting ? 1u : 4u;
    trt_sample_point* d_in = nullptr;
    trt_sampled_color* d_out = nullptr;
    unsigned long long* d_ctr = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long h_ctr[CTR_COUNT] = {0};
    float ms = 0.0f;
    auto cleanup = [&]() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (d_ctr) (void)hipFree(d_ctr);
    };
    TRT_HIP_C(hipEventCreate(&ev0));
    TRT_HIP_C(hipEventCreate(&ev1));
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_in), (size_t)n * sizeof(trt_sample_point)));
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * sizeof(trt_sampled_color)));
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_ctr), sizeof(h_ctr)));
    TRT_HIP_C(hipMemset(d_ctr, 0, sizeof(h_ctr)));
    TRT_HIP_C(hipMemcpy(d_in, in, (size_t)n * sizeof(trt_sample_point), hipMemcpyHostToDevice));
    TRT_HIP_C(hipEventRecord(ev0, nullptr));
    TRT_HIP_C(launch_sample_batch(sc, d_in, n, d_out, ra, d_ctr, counting, nullptr));
    TRT_HIP_C(hipEventRecord(ev1, nullptr));
    TRT_HIP_C(hipMemcpy(out, d_out, (size_t)n * sizeof(trt_sampled_color), hipMemcpyDeviceToHost));
    TRT_HIP_C(hipMemcpy(h_ctr, d_ctr, sizeof(h_ctr), hipMemcpyDeviceToHost));
    TRT_HIP_C(hipEventElapsedTime(&ms, ev0, ev1));
    cleanup();
    if (stats) { counters_to_stats(h_ctr, stats); stats->kernel_ms = ms; }
    return TRT_OK;
#undef TRT_HIP_C
}

uint32_t trt_streamed_chunk_spp(uint32_t width, uint32_t rows) { return streamed_chunk_spp(width, rows); }

int trt_kernel_timing_begin(void) {
    std::lock_guard<std::mutex> lock(trt::g_timing_mu);
    for (hipEvent_t ev : trt::g_timing_events) (void)hipEventDestroy(ev);
    trt::g_timing_events.clear();
    trt::g_timing_on = true;
    return TRT_OK;
}
int trt_kernel_timing_end(double* total_ms, uint32_t* launches) {
    std::lock_guard<std::mutex> lock(trt::g_timing_mu);
    trt::g_timing_on = false;
    double sum = 0.0;
    uint32_t n = 0;
    int rc = TRT_OK;
    std::vector<hipEvent_t>& evs = trt::g_timing_events;
    for (size_t k = 0; k + 1 < evs.size(); k += 2) {
        float ms = 0.0f;
        hipError_t e = hipEventSynchronize(evs[k + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, evs[k], evs[k + 1]);
        if (e != hipSuccess) { rc = fail_hip(e, "kernel timing events"); break; }
        sum += ms;
        n++;
    }
    for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
    evs.clear();
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return rc;
}

// Name of the kernel that dominates a render with these settings (what a profile of it shows).
const char* trt_dominant_kernel(const trt_scene* s, const trt_camera* cam, const trt_render_params* p) {
    if (!s || !cam || !p) return "";
    RenderArgs ra;
    uint32_t rows = 0;
    if (to_render_args(cam, p, ra, rows) != TRT_OK) return "";
    if (p->backend == TRT_BACKEND_WAVEFRONT) return "trt::wavefront_kernel";
    if (p->backend == TRT_BACKEND_MEGAKERNEL) return "trt::megakernel";
    return streamed_kernel_name(s->host.layout, ra);
}

// ---- Imager finalisation ----
int trt_tonemap_u8(const float* accum, uint32_t npixels, float gamma, uint8_t* rgb) {
    if (npixels && (!accum || !rgb)) return fail(TRT_ERR_INVALID_ARG, "null argument");
    tonemap_u8(accum, npixels, gamma, rgb);
    return TRT_OK;
}

// The same on buffers resident in HBM; asynchronous on `stream`.
int trt_tonemap_u8_device(const float* d_accum, uint32_t npixels, float gamma, uint8_t* d_rgb, void* stream) {
    if (npixels && (!d_accum || !d_rgb)) return fail(TRT_ERR_INVALID_ARG, "null argument");
    if (!(gamma > 0.0f)) return fail(TRT_ERR_INVALID_ARG, "gamma must be positive");
    int rc = require_device();
    if (rc != TRT_OK) return rc;
    TRT_HIP(launch_tonemap_u8(d_accum, npixels, gamma, d_rgb, reinterpret_cast<hipStream_t>(stream)));
    return TRT_OK;
}

}  // extern "C"
