// capi.hip — the extern "C" boundary declared in include/tinyrt.h.
//
// Mirrors the reference's World / Camera / Renderer::render() surface (hittable/world.rs:16-45,
// camera.rs:17-56, renderer/renderer.rs:21-79) as opaque handles + POD structs.  The reference
// panics on misuse; here every failure is a negative status plus a thread-local message, and no
// C++ exception leaves this file.  There is no CPU fallback: without a gfx950 device every
// compute entry point fails with TRT_ERR_NO_DEVICE.
//
// Host-side resources (round 3).  Everything a render needs on a device besides the caller's buffers is cached on the scene
// handle, per device, and lives until trt_scene_destroy / trt_scene_trim:
//   * the packed scene (uploaded on first use);
//   * WORKSPACES: the device scratch of one render (streamed: radiance records of a chunk + batch counter; wavefront: path
//     state).  A render owns its workspace from acquire to release; release records an event behind the render's last kernel
//     and a workspace is handed out again only to a stream that first waits for that event;
//   * RENDER CONTEXTS for the blocking entry points (trt_render, trt_render_multi*): a stream, two timing events, the device
//     counters and a device frame buffer.  Streams are created once and never destroyed while the scene lives, so an event
//     recorded on one (a workspace's `done`) never outlives its stream.
// The scene mutex guards only the bookkeeping: an entry is marked busy under the lock and every HIP call that can block
// (hipMalloc, hipFree, hipEventSynchronize, hipStreamSynchronize) runs with the lock released.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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

// trt_kernel_timing_begin / _end: HIP events around every dominant-kernel launch.  A launcher calls timing_mark(stream, true)
// before and timing_mark(stream, false) after the launch ON THE SAME HOST THREAD, so the open `begin` is thread-local: pairs
// never mix threads, streams or devices (round 2 kept one process-wide list and paired marks by parity).
namespace trt {
namespace {
std::mutex g_timing_mu;
bool g_timing_on = false;
unsigned g_timing_epoch = 0;                              // bumped by every _begin: a stale thread-local `begin` is dropped
struct TimingPair { hipEvent_t begin, end; int device; };
std::vector<TimingPair> g_timing_pairs;                   // completed pairs, appended under g_timing_mu
struct TimingOpen { hipEvent_t ev = nullptr; hipStream_t stream = nullptr; int device = -1; unsigned epoch = 0; };
thread_local TimingOpen t_open;
}  // namespace
void timing_mark(hipStream_t stream, bool begin) {
    unsigned epoch;
    {
        std::lock_guard<std::mutex> lock(g_timing_mu);
        if (!g_timing_on) return;
        epoch = g_timing_epoch;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
    if (begin) {
        if (t_open.ev) { (void)hipEventDestroy(t_open.ev); t_open = TimingOpen{}; }       // a begin without its end: dropped
        hipEvent_t ev = nullptr;
        if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipEventRecord(ev, stream) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev); return; }
        t_open.ev = ev; t_open.stream = stream; t_open.device = dev; t_open.epoch = epoch;
        return;
    }
    if (!t_open.ev) return;
    TimingOpen o = t_open;
    t_open = TimingOpen{};
    hipEvent_t ev = nullptr;
    if (o.stream != stream || o.device != dev || o.epoch != epoch || hipEventCreate(&ev) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventDestroy(o.ev);
        return;
    }
    if (hipEventRecord(ev, stream) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev); (void)hipEventDestroy(o.ev); return; }
    std::lock_guard<std::mutex> lock(g_timing_mu);
    if (g_timing_on && g_timing_epoch == epoch) g_timing_pairs.push_back(TimingPair{o.ev, ev, dev});
    else { (void)hipEventDestroy(o.ev); (void)hipEventDestroy(ev); }
}
}  // namespace trt

// Device scratch of one render.  busy: a host thread owns the entry (between acquire and release, or while the pool itself
// regrows / frees it); recorded: `done` has been recorded behind the last kernel that used the buffer.
struct Workspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;            // the stream `done` was recorded on (valid while recorded)
    bool recorded = false;
    bool busy = false;
};
// Everything a blocking render call needs on one device besides its workspace.  One call owns a context from acquire to
// release and has synchronised `stream` before it releases, so an idle context has no work in flight.
struct RenderCtx {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* d_ctr = nullptr;          // CTR_COUNT counters
    float* d_accum = nullptr;                     // device frame (or band) buffer, grown on demand
    size_t accum_bytes = 0;
    bool busy = false;
};
constexpr size_t kMaxWorkspacesPerDevice = 8;      // beyond that, renders queue behind each other on the device

struct DeviceCache {
    float4* blob = nullptr;                       // packed scene in HBM
    bool blob_busy = false;                       // a thread is uploading it
    std::vector<Workspace*> ws;
    std::vector<RenderCtx*> ctx;
    uint32_t short_samples = 0;                   // != 0: the last streamed render of a full-size launch was granted scratch for this many samples only
};

struct trt_scene {
    SceneHost host;
    std::mutex mu;
    std::condition_variable cv;
    std::unordered_map<int, DeviceCache> dev;     // device ordinal -> cached device resources
    size_t scratch_cap_bytes = (size_t)32 << 30;  // idle scratch (workspaces + context frames) kept per device: trt_scene_options.scratch_cap_bytes
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

// The library defaults: the built-in values (kernels.h tuning_builtin, scene_host.cpp scene_options_builtin), overridden by the TRT_*
// environment variables ONCE (a function-local static: initialised exactly once, thread-safe; first touched when the library is loaded).  This is
// the only place the library reads its environment; nothing on the launch path does (round 3 read it in 21 places there).
struct Defaults {
    trt_tuning tuning;
    trt_scene_options scene;
};
const Defaults& defaults() {
    static const Defaults d = [] {
        Defaults x;
        x.tuning = tuning_builtin();
        x.scene = scene_options_builtin();
        auto env = [](const char* name) -> const char* { const char* e = getenv(name); return (e && *e) ? e : nullptr; };
        auto u32 = [&](const char* name, uint32_t& field) { if (const char* e = env(name)) field = (uint32_t)strtoul(e, nullptr, 10); };
        trt_tuning& t = x.tuning;
        u32("TRT_STREAM_MINW", t.stream_waves_per_simd); u32("TRT_BIG_THREADS", t.stream_big_threads); u32("TRT_STREAM_BATCH_SPP", t.stream_batch_spp);
        u32("TRT_RADIANCE_GB", t.radiance_gb); u32("TRT_LEAF_SLOTS", t.leaf_slots); u32("TRT_LDS_LEAF_STACK", t.lds_leaf_stack);
        u32("TRT_RAY_POOL", t.ray_pool); u32("TRT_STRAGGLERS", t.stragglers); u32("TRT_LDS_STRAGGLERS", t.lds_stragglers);
        u32("TRT_DUAL_WALK", t.dual_walk); u32("TRT_RUNTIME_WALK", t.runtime_walk); u32("TRT_XCD_REMAP", t.xcd_remap);
        u32("TRT_MINW", t.mega_waves_per_simd); u32("TRT_MEGA_THREADS", t.mega_threads); u32("TRT_MINW8", t.mega_global_waves8);
        u32("TRT_WF_MINW", t.wf_waves_per_simd); u32("TRT_WF_SERVE_MIN", t.wf_serve_min);
        trt_scene_options& o = x.scene;
        if (const char* e = env("TRT_CULL_PRUNE")) {                       // outside (0, 1] (or not a number): the built-in value stays (such a
            const float v = (float)atof(e);                                //   value made every trt_scene_create of the process fail with INVALID_ARG)
            if (v > 0.0f && v <= 1.0f) o.cull_prune = v;
        }
        if (const char* e = env("TRT_FLAT_WALK")) o.flat_walk = atoi(e) ? 1 : 0;
        if (const char* e = env("TRT_COMPACT_NODES")) o.compact_nodes = atoi(e) ? 1 : 0;
        if (const char* e = env("TRT_SCRATCH_CAP_MB")) o.scratch_cap_bytes = (uint64_t)strtoull(e, nullptr, 10) << 20;
        return x;
    }();
    return d;
}

// ... and that first need is the library's own load: a namespace-scope initialiser, so that "once, at load" holds whatever the host does later
[[maybe_unused]] const bool g_defaults_read_at_load = (defaults(), true);

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(TRT_ERR_NO_DEVICE, "no HIP device visible: libtinyrt has no CPU path, it needs a gfx950 GPU");
    }
    return TRT_OK;
}

// Uploads the packed scene to the current device on first use.  One thread uploads (lock released during the copy), the
// others wait for it.
int scene_on_device(trt_scene* s, SceneDev& out) {
    int dev = 0;
    TRT_HIP(hipGetDevice(&dev));
    std::unique_lock<std::mutex> lock(s->mu);
    DeviceCache& dc = s->dev[dev];                              // references into an unordered_map stay valid across inserts
    while (dc.blob == nullptr && dc.blob_busy) s->cv.wait(lock);
    if (dc.blob == nullptr) {
        dc.blob_busy = true;
        lock.unlock();
        float4* d = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), s->host.layout.blob_bytes);
        if (e == hipSuccess) {
            e = hipMemcpy(d, s->host.blob.data(), s->host.layout.blob_bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(d); d = nullptr; }
        }
        lock.lock();
        dc.blob_busy = false;
        dc.blob = d;
        s->cv.notify_all();
        if (e != hipSuccess) return fail_hip(e, "scene upload");
    }
    out.blob = dc.blob;
    out.L = s->host.layout;
    return TRT_OK;
}

bool ws_finished(Workspace* w) {                                  // call with the scene lock held, entry not busy
    if (!w->recorded) return true;
    const bool done = hipEventQuery(w->done) == hipSuccess;
    (void)hipGetLastError();                                      // hipErrorNotReady is not an error
    return done;
}

// Frees idle scratch on `dc` until at most `cap` bytes of IDLE scratch stay cached (largest first).  Called with the lock held; the
// hipFree calls run with the lock released (the entries are marked busy meanwhile).
void trim_locked(trt_scene* s, std::unique_lock<std::mutex>& lock, DeviceCache& dc, size_t cap) {
    for (;;) {
        size_t total = 0;                                         // idle entries only: a busy entry's fields belong to its owner
        for (Workspace* w : dc.ws) if (!w->busy) total += w->bytes;
        for (RenderCtx* c : dc.ctx) if (!c->busy) total += c->accum_bytes;
        if (total <= cap) return;
        Workspace* bw = nullptr;
        RenderCtx* bc = nullptr;
        for (Workspace* w : dc.ws) if (!w->busy && w->bytes && ws_finished(w) && (!bw || w->bytes > bw->bytes)) bw = w;
        for (RenderCtx* c : dc.ctx) if (!c->busy && c->accum_bytes && (!bc || c->accum_bytes > bc->accum_bytes)) bc = c;
        if (!bw && !bc) return;                                   // everything left is in use
        void* victim;
        if (bw && (!bc || bw->bytes >= bc->accum_bytes)) { bw->busy = true; victim = bw->ptr; bw->ptr = nullptr; bw->bytes = 0; bw->recorded = false; bc = nullptr; }
        else { bc->busy = true; victim = bc->d_accum; bc->d_accum = nullptr; bc->accum_bytes = 0; bw = nullptr; }
        lock.unlock();
        (void)hipFree(victim);
        (void)hipGetLastError();
        lock.lock();
        if (bw) bw->busy = false; else bc->busy = false;
        s->cv.notify_all();
    }
}

// Takes a workspace of at least `need` bytes for a render that will be enqueued on `stream` of device `dev`.
int workspace_acquire(trt_scene* s, int dev, size_t need, hipStream_t stream, Workspace** out) {
    std::unique_lock<std::mutex> lock(s->mu);
    DeviceCache& dc = s->dev[dev];
    bool may_grow = true;
    for (;;) {
        // 1. an idle one whose last render has finished (no dependency at all): the smallest that fits, else one to regrow
        Workspace* pick = nullptr;
        Workspace* regrow = nullptr;
        for (Workspace* w : dc.ws) {
            if (w->busy || !ws_finished(w)) continue;
            if (w->bytes >= need) { if (!pick || w->bytes < pick->bytes) pick = w; }
            else if (!regrow) regrow = w;
        }
        enum { USE, REGROW, CREATE, QUEUE, DRAIN } what = USE;
        // 1b. one whose last render is still in flight ON THIS STREAM: stream order already puts this render behind it - no wait, no second
        //     buffer.  (Back-to-back renders enqueued on one stream - bench.py's steps, a progressive viewer - otherwise grew the pool to its
        //     eight workspaces, 16 GB each at full size, and sent the idle ones through the byte cap: a 13 GB hipMalloc + hipFree per render.)
        //     hipStreamPerThread is one handle for many streams and never matches.
        //     The wait on `done` is still enqueued (SAME_STREAM below): behind its own stream's work it costs nothing, and it keeps the reuse
        //     safe should a caller destroy a stream with work pending and get the same handle back for a new one.
        bool same_stream = false;
        if (!pick && stream != hipStreamPerThread) {
            for (Workspace* w : dc.ws) if (!w->busy && w->recorded && w->last_stream == stream && w->bytes >= need && (!pick || w->bytes < pick->bytes)) pick = w;
            same_stream = pick != nullptr;
        }
        if (same_stream) what = QUEUE;
        if (!pick && regrow) { pick = regrow; what = REGROW; }
        // 2. a new one while the pool may grow
        if (!pick && may_grow && dc.ws.size() < kMaxWorkspacesPerDevice) {
            pick = new (std::nothrow) Workspace();
            if (!pick) return fail(TRT_ERR_OOM, "out of memory");
            dc.ws.push_back(pick);
            what = CREATE;
        }
        // 3. an idle one that is still in flight and large enough: this render queues behind it on the device
        if (!pick) {
            for (Workspace* w : dc.ws) if (!w->busy && w->bytes >= need) { pick = w; what = QUEUE; break; }
        }
        if (!pick) {
            // 4. every workspace is owned by another host thread: wait for a release
            bool any_busy = false;
            for (Workspace* w : dc.ws) any_busy = any_busy || w->busy;
            if (any_busy) { s->cv.wait(lock); continue; }
            if (dc.ws.empty()) return fail(TRT_ERR_OOM, "no device memory for a render workspace");
            // 5. all idle, all in flight, all too small: wait for the first to finish, then regrow it
            pick = dc.ws.front();
            what = DRAIN;
        }
        pick->busy = true;
        lock.unlock();
        // ---- HIP calls, lock released; `pick` is ours ----
        hipError_t e = hipSuccess;
        const char* where = "";
        if (what == CREATE) {
            e = hipEventCreateWithFlags(&pick->done, hipEventDisableTiming); where = "hipEventCreate(workspace)";
            if (e == hipSuccess) { e = hipMalloc(&pick->ptr, need); where = "hipMalloc(workspace)"; if (e == hipSuccess) pick->bytes = need; else pick->ptr = nullptr; }
        } else if (what == DRAIN || what == REGROW) {
            if (what == DRAIN) { e = hipEventSynchronize(pick->done); where = "hipEventSynchronize(workspace)"; }
            if (e == hipSuccess && pick->ptr) { e = hipFree(pick->ptr); where = "hipFree(workspace)"; pick->ptr = nullptr; pick->bytes = 0; }
            if (e == hipSuccess) { e = hipMalloc(&pick->ptr, need); where = "hipMalloc(workspace)"; if (e == hipSuccess) pick->bytes = need; else pick->ptr = nullptr; }
            if (e == hipSuccess) pick->recorded = false;
        } else if (what == QUEUE) {
            e = hipStreamWaitEvent(stream, pick->done, 0); where = "hipStreamWaitEvent(workspace)";
        }
        if (e == hipSuccess) { *out = pick; return TRT_OK; }
        (void)hipGetLastError();
        lock.lock();
        if (what == CREATE) {
            dc.ws.erase(std::find(dc.ws.begin(), dc.ws.end(), pick));
            if (pick->done) (void)hipEventDestroy(pick->done);
            delete pick;
            s->cv.notify_all();
            if (e == hipErrorOutOfMemory) {
                // out of HBM for another copy: queue behind a running render whose buffer is large enough, if there is one; else the caller
                // may ask for less (TRT_ERR_OOM: enqueue_render halves the samples per launch)
                bool fits = false;                                 // (a busy entry's fields belong to its owner: it may fit once released)
                for (Workspace* w : dc.ws) fits = fits || w->busy || w->bytes >= need;
                if (fits) { may_grow = false; continue; }
                return fail(TRT_ERR_OOM, std::string(where) + ": " + hipGetErrorString(e));
            }
            return fail_hip(e, where);
        }
        pick->busy = false;                                       // (a failed regrow leaves an empty, reusable entry)
        s->cv.notify_all();
        if (e == hipErrorOutOfMemory) return fail(TRT_ERR_OOM, std::string(where) + ": " + hipGetErrorString(e));
        return fail_hip(e, where);
    }
}

// Gives the workspace back; `stream` has all of the render's launches enqueued.
int workspace_release(trt_scene* s, int dev, Workspace* w, hipStream_t stream) {
    hipError_t e = hipEventRecord(w->done, stream);
    bool recorded = e == hipSuccess;
    if (!recorded) {
        // without the event nothing orders the next user behind this render: wait for the render itself
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream);
        (void)hipGetLastError();
    }
    {
        std::unique_lock<std::mutex> lock(s->mu);
        w->recorded = recorded;
        w->last_stream = stream;
        w->busy = false;
        trim_locked(s, lock, s->dev[dev], s->scratch_cap_bytes);
    }
    s->cv.notify_all();
    if (e != hipSuccess) return fail_hip(e, "hipEventRecord(workspace)");
    return TRT_OK;
}

// A render context on the current device with a device frame of at least `accum_bytes` (0: none needed).
int context_acquire(trt_scene* s, int dev, size_t accum_bytes, RenderCtx** out) {
    RenderCtx* c = nullptr;
    bool fresh = false;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        DeviceCache& dc = s->dev[dev];
        RenderCtx* fits = nullptr;                                 // the smallest idle frame that fits, else the largest idle one (regrown)
        RenderCtx* any = nullptr;
        for (RenderCtx* k : dc.ctx) {
            if (k->busy) continue;
            if (k->accum_bytes >= accum_bytes && (!fits || k->accum_bytes < fits->accum_bytes)) fits = k;
            if (!any || k->accum_bytes > any->accum_bytes) any = k;
        }
        c = fits ? fits : any;
        if (!c) {
            c = new (std::nothrow) RenderCtx();
            if (!c) return fail(TRT_ERR_OOM, "out of memory");
            dc.ctx.push_back(c);
            fresh = true;
        }
        c->busy = true;
    }
    hipError_t e = hipSuccess;
    const char* where = "";
    if (fresh) {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); where = "hipStreamCreate";
        if (e == hipSuccess) { e = hipEventCreate(&c->ev0); where = "hipEventCreate"; }
        if (e == hipSuccess) { e = hipEventCreate(&c->ev1); where = "hipEventCreate"; }
        if (e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&c->d_ctr), CTR_COUNT * sizeof(unsigned long long)); where = "hipMalloc(counters)"; }
    }
    if (e == hipSuccess && c->accum_bytes < accum_bytes) {
        if (c->d_accum) { e = hipFree(c->d_accum); where = "hipFree(frame)"; c->d_accum = nullptr; c->accum_bytes = 0; }
        if (e == hipSuccess) { e = hipMalloc(reinterpret_cast<void**>(&c->d_accum), accum_bytes); where = "hipMalloc(frame)"; if (e == hipSuccess) c->accum_bytes = accum_bytes; else c->d_accum = nullptr; }
    }
    if (e == hipSuccess) { *out = c; return TRT_OK; }
    (void)hipGetLastError();
    const int rc = fail_hip(e, where);
    std::lock_guard<std::mutex> lock(s->mu);
    if (fresh) {                                                  // half-built: take it out of the pool again
        DeviceCache& dc = s->dev[dev];
        dc.ctx.erase(std::find(dc.ctx.begin(), dc.ctx.end(), c));
        if (c->d_ctr) (void)hipFree(c->d_ctr);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
    } else {
        c->busy = false;
    }
    return rc;
}

void context_destroy(RenderCtx* c) {                              // idle context: nothing in flight on its stream
    if (c->d_accum) (void)hipFree(c->d_accum);
    if (c->d_ctr) (void)hipFree(c->d_ctr);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    (void)hipGetLastError();
    delete c;
}

// The caller has synchronised c->stream (or nothing was enqueued on it).  The context stays in the pool whatever the pool's size:
// its stream and events live until trt_scene_destroy (a workspace's `done` event may have been recorded on that stream and is
// queried by later acquires, so destroying the stream here - round 3 did, beyond 16 idle contexts - brought back the very pattern
// DESIGN.md's audit of round 2's abort removed).  What a burst of shards leaves behind is bounded by the byte cap: idle contexts'
// frames are freed largest first by trim_locked; a context without a frame is a stream, two events and 128 bytes of counters.
void context_release(trt_scene* s, int dev, RenderCtx* c) {
    std::unique_lock<std::mutex> lock(s->mu);
    DeviceCache& dc = s->dev[dev];
    c->busy = false;
    trim_locked(s, lock, dc, s->scratch_cap_bytes);
}

void to_camera_dev(const trt_camera& c, CameraDev& d) {
    const trt_vec3* src[6] = {&c.position, &c.viewport_upper_left, &c.horizontal, &c.vertical, &c.defocus_disk_u, &c.defocus_disk_v};
    float* dst[6] = {d.pos, d.upper_left, d.horizontal, d.vertical, d.du, d.dv};
    for (int i = 0; i < 6; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    d.width = c.width;
    d.height = c.height;
}

// Validates params against the camera and fills the kernel arguments.  rows = rows the call owns.
int to_render_args(const trt_camera* cam, const trt_render_params* p, RenderArgs& ra, uint32_t& rows, trt_tuning& tn) {
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
    // scheduling (tinyrt.h trt_tuning): the caller's, else the library defaults; scheduling only, any value renders the same frame
    tn = p->tuning ? *p->tuning : defaults().tuning;
    // whatever a caller writes into the knobs, the launch arithmetic stays in range (the launch plan clamps waves, lanes and slots itself)
    if (tn.stream_batch_spp > 256u) tn.stream_batch_spp = 256u;      // a batch is at most one launch's samples of a tile
    if (tn.stragglers > 63u) tn.stragglers = 63u;
    if (tn.lds_stragglers > 63u) tn.lds_stragglers = 63u;
    if (tn.radiance_gb > 64u) tn.radiance_gb = 64u;
    ra.leaf_slots = tn.leaf_slots;                         // rt_path.h walk_fast; 0 = the backend's default
    ra.lds_leaf_stack = tn.lds_leaf_stack;                 // 0 off, 1 where it costs no occupancy, 2 always
    ra.xcd_aware = tn.xcd_remap ? 1u : 0u;                 // off: contiguous image regions per XCD measured 2x slower (load imbalance)
    ra.stragglers = tn.stragglers;                         // profiles/r03_stragglers_sweep.txt
    ra.ref_tree = p->collect_stats == 1 ? 1u : 0u;      // 1: counters comparable with the CPU path; 2: count the culling tree's own tests
    return TRT_OK;
}

int enqueue_render(trt_scene* s, const trt_camera* cam, const trt_render_params* p, float* d_accum, uint64_t* d_counters,
                   hipStream_t stream, uint32_t* rows_out) {
    RenderArgs ra;
    trt_tuning tn;
    uint32_t rows = 0;
    int rc = to_render_args(cam, p, ra, rows, tn);
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
        // Device workspace (wavefront: 72 B of path state per pixel; streamed: 12 B per pixel and sample of a launch), private
        // to this render until its last kernel has run (workspace_acquire): concurrent renders of one scene are safe.
        int dev = 0;
        TRT_HIP(hipGetDevice(&dev));
        Workspace* ws = nullptr;
        if (wavefront) {
            rc = workspace_acquire(s, dev, wavefront_workspace_bytes(cam->width, rows), stream, &ws);
        } else {
            // The streamed launch is sized for 288 GB (up to 16 GB of records).  If the device cannot give that much right now - HBM held by
            // another scene's cached scratch, by the host application, by torch - ask for half the samples per launch, and half again: the
            // render then runs more, shorter launches (same frame: the fold adds the samples in order whatever the launch boundaries).
            uint32_t samples = ra.sample_end - ra.sample_begin;
            const uint32_t full = streamed_chunk_spp(cam->width, rows, tn.radiance_gb);
            if (samples > full) samples = full;
            const uint32_t wanted = samples;
            // Lasting pressure: while the device still cannot hold the full request, start where the last render ended up instead
            // of freeing the cached shorter workspace, failing the full-size hipMalloc and allocating the shorter one again on every render.
            // hipMemGetInfo says when the pressure is gone: free memory + this scene's own idle scratch (a regrow frees it first) covers the request.
            {
                uint32_t short_samples = 0;
                size_t idle = 0;
                {
                    std::lock_guard<std::mutex> lock(s->mu);
                    DeviceCache& dc = s->dev[dev];
                    short_samples = dc.short_samples;
                    for (Workspace* w : dc.ws) if (!w->busy) idle += w->bytes;
                }
                if (short_samples != 0u && short_samples < samples) {
                    size_t free_b = 0, total_b = 0;
                    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
                    (void)hipGetLastError();
                    if (!known || free_b + idle < streamed_workspace_bytes(cam->width, rows, samples, tn.radiance_gb)) samples = short_samples;
                }
            }
            bool trimmed = false;
            for (;;) {
                rc = workspace_acquire(s, dev, streamed_workspace_bytes(cam->width, rows, samples, tn.radiance_gb), stream, &ws);
                if (rc != TRT_ERR_OOM) break;
                if (!trimmed) {                      // first give back what this scene itself keeps idle on the device (frames of idle contexts, other workspaces), then ask again
                    trimmed = true;
                    std::unique_lock<std::mutex> lock(s->mu);
                    trim_locked(s, lock, s->dev[dev], 0);
                    continue;
                }
                if (samples <= 1u) break;
                samples = (samples + 1u) / 2u;
            }
            {
                std::lock_guard<std::mutex> lock(s->mu);
                s->dev[dev].short_samples = (rc == TRT_OK && samples < wanted) ? samples : 0u;
            }
        }
        if (rc != TRT_OK) return rc;
        hipError_t le;
        if (streamed) {
            le = launch_streamed(sc, cd, ra, tn, ws->ptr, ws->bytes, d_accum, reinterpret_cast<unsigned long long*>(d_counters), p->collect_stats != 0, stream);
        } else {
            le = launch_wavefront(sc, cd, ra, tn, ws->ptr, d_accum, reinterpret_cast<unsigned long long*>(d_counters), p->collect_stats != 0, stream);
        }
        rc = workspace_release(s, dev, ws, stream);
        if (le != hipSuccess) return fail_hip(le, streamed ? "launch_streamed" : "launch_wavefront");
        return rc;
    }
    TRT_HIP(launch_megakernel(sc, cd, ra, tn, d_accum, reinterpret_cast<unsigned long long*>(d_counters), p->collect_stats != 0, stream));
    return TRT_OK;
}

void counters_to_stats(const unsigned long long* c, trt_stats* st) {
    st->samples = c[CTR_SAMPLES]; st->rays = c[CTR_RAYS]; st->node_tests = c[CTR_NODE]; st->sphere_tests = c[CTR_SPHERE];
    st->quad_plane_tests = c[CTR_QUAD_PLANE]; st->quad_inside_tests = c[CTR_QUAD_INSIDE]; st->shades = c[CTR_SHADE];
    for (int i = 0; i < 4; i++) st->wave_trips[i] = c[CTR_W_ROUNDS + i];
    st->gather_per_band = 0;
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
int trt_world_add_spheres(trt_world* w, uint32_t n, const float* center_radius, const uint32_t* material) {
    if (!w) return fail(TRT_ERR_INVALID_ARG, "world is null");
    if (n == 0) return TRT_OK;
    if (!center_radius || !material) return fail(TRT_ERR_INVALID_ARG, "null array");
    for (uint32_t i = 0; i < n; i++)
        if (material[i] >= w->w.materials.size()) return fail(TRT_ERR_INVALID_ARG, "material index out of range");      // nothing is added then
    try {
        w->w.geometries.reserve(w->w.geometries.size() + n);
        for (uint32_t i = 0; i < n; i++) {
            const float* c = center_radius + 4u * (size_t)i;
            w->w.geometries.push_back(Geometry{0u, material[i], trt_vec3{c[0], c[1], c[2]}, trt_vec3{c[3], 0.0f, 0.0f}, trt_vec3{0.0f, 0.0f, 0.0f}});
        }
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
void trt_scene_options_default(trt_scene_options* out) { if (out) *out = defaults().scene; }
void trt_tuning_default(trt_tuning* out) { if (out) *out = defaults().tuning; }

int trt_scene_create_ex(const trt_world* w, const trt_scene_options* options, trt_scene** out) {
    if (!w || !out) return fail(TRT_ERR_INVALID_ARG, "null argument");
    try {
        const trt_scene_options opt = options ? *options : defaults().scene;
        if (!(opt.cull_prune > 0.0f && opt.cull_prune <= 1.0f)) return fail(TRT_ERR_INVALID_ARG, "cull_prune must be in (0, 1]");
        trt_scene* s = new trt_scene();
        std::string msg;
        if (!compile_scene(w->w, opt, s->host, msg)) { delete s; return fail(TRT_ERR_INVALID_ARG, msg); }
        s->scratch_cap_bytes = (size_t)opt.scratch_cap_bytes;
        *out = s;
    } catch (const std::bad_alloc&) {
        return fail(TRT_ERR_OOM, "out of memory");
    } catch (const std::exception& e) {                         // (the scene compiler builds its trees on several threads: nothing may leave the C ABI)
        return fail(TRT_ERR_INVALID_ARG, std::string("scene compilation failed: ") + e.what());
    }
    return TRT_OK;
}
int trt_scene_create(const trt_world* w, trt_scene** out) { return trt_scene_create_ex(w, nullptr, out); }
// Frees every idle cached device buffer of the scene (workspaces whose render has finished, idle contexts' frames); the
// packed scene stays.  Safe while renders of the scene run: what they own is skipped.
int trt_scene_trim(trt_scene* s) {
    if (!s) return fail(TRT_ERR_INVALID_ARG, "scene is null");
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    std::vector<int> devices;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        for (auto& kv : s->dev) devices.push_back(kv.first);
    }
    for (int d : devices) {
        if (hipSetDevice(d) != hipSuccess) continue;
        std::unique_lock<std::mutex> lock(s->mu);
        trim_locked(s, lock, s->dev[d], 0);
    }
    if (have_prev) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    return TRT_OK;
}
// Must not run while another host thread is inside a render call on this scene.  Renders enqueued through trt_render_device
// that are still running on the device are waited for (their workspace events) before anything is freed.
void trt_scene_destroy(trt_scene* s) {
    if (!s) return;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (auto& kv : s->dev) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        DeviceCache& dc = kv.second;
        for (Workspace* w : dc.ws) {
            if (w->recorded) (void)hipEventSynchronize(w->done);      // a render may still be using it
            if (w->ptr) (void)hipFree(w->ptr);
            if (w->done) (void)hipEventDestroy(w->done);
            delete w;
        }
        for (RenderCtx* c : dc.ctx) context_destroy(c);               // after the workspaces: their events were recorded on these streams
        if (dc.blob) (void)hipFree(dc.blob);
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
    for (uint32_t i = 0; i < L.n_cull_nodes; i++)                       // stored as byte offsets (the walk's cursor); reported as node indices
        if (!(words4[4u * (size_t)i + 3u] & 0x80000000u)) words4[4u * (size_t)i + 3u] >>= 4;
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
    trt_tuning probe_tn;
    uint32_t rows = 0;
    rc = to_render_args(cam, p, probe, rows, probe_tn);
    if (rc != TRT_OK) return rc;
    const size_t bytes = (size_t)rows * cam->width * 3 * sizeof(float);
    int dev = 0;
    TRT_HIP(hipGetDevice(&dev));
    RenderCtx* c = nullptr;
    rc = context_acquire(s, dev, bytes ? bytes : 16, &c);
    if (rc != TRT_OK) return rc;
    unsigned long long h_ctr[CTR_COUNT] = {0};
    float ms = 0.0f;
    // from here on every exit drains the context's stream before the context goes back to the pool
    auto finish = [&](int code) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        context_release(s, dev, c);
        return code;
    };
#define TRT_HIP_C(call)                                                          \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) return finish(fail_hip(e_, #call));                \
    } while (0)
    TRT_HIP_C(hipMemsetAsync(c->d_ctr, 0, sizeof(h_ctr), c->stream));
    if (p->accumulate && bytes) TRT_HIP_C(hipMemcpyAsync(c->d_accum, accum, bytes, hipMemcpyHostToDevice, c->stream));
    TRT_HIP_C(hipEventRecord(c->ev0, c->stream));
    rc = enqueue_render(s, cam, p, c->d_accum, reinterpret_cast<uint64_t*>(c->d_ctr), c->stream, nullptr);
    if (rc != TRT_OK) return finish(rc);
    TRT_HIP_C(hipEventRecord(c->ev1, c->stream));
    if (bytes) TRT_HIP_C(hipMemcpyAsync(accum, c->d_accum, bytes, hipMemcpyDeviceToHost, c->stream));
    TRT_HIP_C(hipMemcpyAsync(h_ctr, c->d_ctr, sizeof(h_ctr), hipMemcpyDeviceToHost, c->stream));
    TRT_HIP_C(hipStreamSynchronize(c->stream));
    TRT_HIP_C(hipEventElapsedTime(&ms, c->ev0, c->ev1));
#undef TRT_HIP_C
    context_release(s, dev, c);
    if (stats) { counters_to_stats(h_ctr, stats); stats->kernel_ms = ms; }
    return TRT_OK;
}

// ---- Renderer::render over several GPUs of one node ----
//
// renderer.rs:37-79 is ONE call that returns the whole Image; so is this.  The path shards by pixels (every sample reads
// only the immutable scene, cpu.rs:39-65): the scene is replicated, the image is cut into bands of kMultiBandRows rows dealt
// round-robin (band b -> shard b % ndev, so expensive regions spread over all devices), the RNG is keyed by the image
// pixel, and every device folds its own pixels in sample order - the frame is bit-identical for every ndev.  One host
// thread per shard drives its device on a cached context (stream, events, counters, band buffer: nothing is created per
// call once the pool is warm).  The only exchange is the gather of the finished f32 sums: a shard's bands sit at a
// constant pitch of ndev x 16 rows in the frame, so ONE strided 2-D copy moves all its full bands straight to their place
// (a host buffer: D2H; a buffer on devices[0]: device to device over xGMI), plus one 1-D copy if the shard owns the ragged
// last band.  No padded gather buffer and no un-interleave pass exist.
namespace {

constexpr uint32_t kMultiBandRows = 16;

struct MultiShard {
    int device = 0;
    uint32_t rank = 0, rows_local = 0;
    bool peer_ok = true;                      // the gather may address the destination device directly
    bool per_band = false;                    // the gather went band by band (no peer access, or the strided peer copy was refused)
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

// Where shard `rank` of `ndev` keeps its rows and where they belong in the frame (bytes).  Local rows are contiguous; in the
// frame the shard's k-th band starts at row (k * ndev + rank) * 16.
void band_copy_plan(uint32_t width, uint32_t height, uint32_t ndev, uint32_t rank, trt_band_copy* o) {
    const uint64_t row_bytes = (uint64_t)width * 3u * sizeof(float);
    const uint32_t rows = band_rows_local(height, ndev, rank, kMultiBandRows);
    o->rows_local = rows;
    o->full_bands = rows / kMultiBandRows;
    o->tail_rows = rows % kMultiBandRows;                                   // only the image's last band can be short
    o->band_bytes = row_bytes * kMultiBandRows;
    o->local_pitch = o->band_bytes;
    o->frame_pitch = o->band_bytes * ndev;
    o->frame_offset = (uint64_t)rank * o->band_bytes;
    o->tail_bytes = row_bytes * o->tail_rows;
    o->tail_local_offset = (uint64_t)o->full_bands * o->band_bytes;
    o->tail_frame_offset = ((uint64_t)o->full_bands * ndev + rank) * o->band_bytes;
}

// Moves one shard's rows between its local buffer and the frame (to_frame: the gather; else: reading the running sums).
// *per_band (may be null) is set when the rows went band by band instead of as one strided 2-D copy: two devices without peer
// access, or a runtime that refused the strided copy between two devices (never seen, but that path has only ever run against
// the simulated runtime of tests/native: on refusal the per-band peer copies - round 2's gather - take over).
hipError_t copy_bands(const trt_band_copy& pl, char* local, char* frame, bool to_frame, int local_device, int frame_device,
                      bool peer_ok, hipStream_t stream, bool* per_band = nullptr) {
    const bool host_frame = frame_device < 0;
    const hipMemcpyKind kind = host_frame ? (to_frame ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice) : hipMemcpyDeviceToDevice;
    const bool cross = !host_frame && frame_device != local_device;
    bool direct = host_frame || !cross || peer_ok;
    if (pl.full_bands) {
        if (direct) {
            hipError_t e = to_frame
                ? hipMemcpy2DAsync(frame + pl.frame_offset, pl.frame_pitch, local, pl.local_pitch, pl.band_bytes, pl.full_bands, kind, stream)
                : hipMemcpy2DAsync(local, pl.local_pitch, frame + pl.frame_offset, pl.frame_pitch, pl.band_bytes, pl.full_bands, kind, stream);
            if (e != hipSuccess) {
                if (!cross) return e;
                (void)hipGetLastError();
                direct = false;                                   // the strided cross-device copy was refused: band by band
            }
        }
        if (!direct) {                                            // no peer access (the runtime stages each band through the host), or the fallback
            if (per_band) *per_band = true;
            for (uint32_t k = 0; k < pl.full_bands; k++) {
                char* l = local + (uint64_t)k * pl.local_pitch;
                char* f = frame + pl.frame_offset + (uint64_t)k * pl.frame_pitch;
                hipError_t e = to_frame ? hipMemcpyPeerAsync(f, frame_device, l, local_device, pl.band_bytes, stream)
                                        : hipMemcpyPeerAsync(l, local_device, f, frame_device, pl.band_bytes, stream);
                if (e != hipSuccess) return e;
            }
        }
    }
    if (pl.tail_rows) {
        char* l = local + pl.tail_local_offset;
        char* f = frame + pl.tail_frame_offset;
        if (direct) return to_frame ? hipMemcpyAsync(f, l, pl.tail_bytes, kind, stream) : hipMemcpyAsync(l, f, pl.tail_bytes, kind, stream);
        if (per_band) *per_band = true;
        return to_frame ? hipMemcpyPeerAsync(f, frame_device, l, local_device, pl.tail_bytes, stream)
                        : hipMemcpyPeerAsync(l, local_device, f, frame_device, pl.tail_bytes, stream);
    }
    return hipSuccess;
}

// One device's share.  dst: the whole frame, on the host (dst_device < 0) or on device dst_device.
void render_shard(trt_scene* s, const trt_camera* cam, const trt_render_params* p, uint32_t ndev, MultiShard* sh, float* dst,
                  int dst_device) {
    hipError_t e = hipSetDevice(sh->device);
    if (e != hipSuccess) { sh->rc = fail_hip(e, "hipSetDevice"); sh->error = g_last_error; return; }
    if (sh->rows_local == 0) return;                                        // more shards than bands
    trt_band_copy pl;
    band_copy_plan(cam->width, cam->height, ndev, sh->rank, &pl);
    trt_render_params q = *p;
    q.band_rows = kMultiBandRows; q.band_stride = ndev; q.band_offset = sh->rank; q.rows_local = sh->rows_local;
    RenderCtx* c = nullptr;
    sh->rc = context_acquire(s, sh->device, (size_t)pl.rows_local * cam->width * 3 * sizeof(float), &c);
    if (sh->rc != TRT_OK) { sh->error = g_last_error; return; }
    auto finish = [&]() {
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        context_release(s, sh->device, c);
    };
#define TRT_HIP_S(call)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess) { sh->rc = fail_hip(e_, #call); sh->error = g_last_error; finish(); return; }     \
    } while (0)
    char* local = reinterpret_cast<char*>(c->d_accum);
    char* frame = reinterpret_cast<char*>(dst);
    TRT_HIP_S(hipMemsetAsync(c->d_ctr, 0, sizeof(sh->ctr), c->stream));
    if (p->accumulate) TRT_HIP_S(copy_bands(pl, local, frame, false, sh->device, dst_device, sh->peer_ok, c->stream));   // continue the frame's running sums
    TRT_HIP_S(hipEventRecord(c->ev0, c->stream));
    sh->rc = enqueue_render(s, cam, &q, c->d_accum, reinterpret_cast<uint64_t*>(c->d_ctr), c->stream, nullptr);
    if (sh->rc != TRT_OK) { sh->error = g_last_error; finish(); return; }
    TRT_HIP_S(hipEventRecord(c->ev1, c->stream));
    TRT_HIP_S(copy_bands(pl, local, frame, true, sh->device, dst_device, sh->peer_ok, c->stream, &sh->per_band));        // the gather
    TRT_HIP_S(hipMemcpyAsync(sh->ctr, c->d_ctr, sizeof(sh->ctr), hipMemcpyDeviceToHost, c->stream));
    TRT_HIP_S(hipStreamSynchronize(c->stream));
    TRT_HIP_S(hipEventElapsedTime(&sh->ms, c->ev0, c->ev1));
#undef TRT_HIP_S
    context_release(s, sh->device, c);
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
        trt_tuning probe_tn;
        uint32_t rows = 0;
        rc = to_render_args(cam, p, probe, rows, probe_tn);
        if (rc != TRT_OK) return rc;
    }
    int prev = 0;
    TRT_HIP(hipGetDevice(&prev));
    const int dst_device = dst_on_device ? shards[0].device : -1;
    if (dst_on_device) {
        for (uint32_t r = 0; r < ndev; r++) {                               // peer access for the gather (xGMI)
            if (shards[r].device == dst_device) continue;
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, shards[r].device, dst_device);
            if (e != hipSuccess) { (void)hipSetDevice(prev); return fail_hip(e, "hipDeviceCanAccessPeer"); }
            if (!can) { shards[r].peer_ok = false; continue; }             // hipMemcpyPeerAsync then stages through the host
            e = hipSetDevice(shards[r].device);
            if (e == hipSuccess) e = hipDeviceEnablePeerAccess(dst_device, 0);
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
    uint64_t per_band = 0;
    for (const MultiShard& sh : shards) {
        if (sh.rc != TRT_OK) return fail(sh.rc, "device " + std::to_string(sh.device) + ": " + sh.error);
        for (int k = 0; k < CTR_COUNT; k++) total[k] += sh.ctr[k];
        if (sh.ms > ms) ms = sh.ms;
        per_band += sh.per_band ? 1u : 0u;
    }
    if (stats) { counters_to_stats(total, stats); stats->kernel_ms = ms; stats->gather_per_band = per_band; }      // kernel_ms: the slowest device's
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

int trt_band_copy_plan(uint32_t width, uint32_t height, uint32_t ndev, uint32_t rank, trt_band_copy* out) {
    if (!out || ndev == 0 || rank >= ndev || width == 0) return fail(TRT_ERR_INVALID_ARG, "need 0 <= rank < ndev, a positive width and a result pointer");
    band_copy_plan(width, height, ndev, rank, out);
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
    int dev = 0;
    TRT_HIP(hipGetDevice(&dev));
    RenderCtx* c = nullptr;
    rc = context_acquire(s, dev, 0, &c);
    if (rc != TRT_OK) return rc;
    trt_sample_point* d_in = nullptr;
    trt_sampled_color* d_out = nullptr;
    unsigned long long h_ctr[CTR_COUNT] = {0};
    float ms = 0.0f;
    auto finish = [&](int code) {
        (void)hipStreamSynchronize(c->stream);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        (void)hipGetLastError();
        context_release(s, dev, c);
        return code;
    };
#define TRT_HIP_C(call)                                                          \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) return finish(fail_hip(e_, #call));                \
    } while (0)
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_in), (size_t)n * sizeof(trt_sample_point)));
    TRT_HIP_C(hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * sizeof(trt_sampled_color)));
    TRT_HIP_C(hipMemsetAsync(c->d_ctr, 0, sizeof(h_ctr), c->stream));
    TRT_HIP_C(hipMemcpyAsync(d_in, in, (size_t)n * sizeof(trt_sample_point), hipMemcpyHostToDevice, c->stream));
    TRT_HIP_C(hipEventRecord(c->ev0, c->stream));
    TRT_HIP_C(launch_sample_batch(sc, d_in, n, d_out, ra, c->d_ctr, counting, c->stream));
    TRT_HIP_C(hipEventRecord(c->ev1, c->stream));
    TRT_HIP_C(hipMemcpyAsync(out, d_out, (size_t)n * sizeof(trt_sampled_color), hipMemcpyDeviceToHost, c->stream));
    TRT_HIP_C(hipMemcpyAsync(h_ctr, c->d_ctr, sizeof(h_ctr), hipMemcpyDeviceToHost, c->stream));
    TRT_HIP_C(hipStreamSynchronize(c->stream));
    TRT_HIP_C(hipEventElapsedTime(&ms, c->ev0, c->ev1));
#undef TRT_HIP_C
    (void)finish(TRT_OK);
    if (stats) { counters_to_stats(h_ctr, stats); stats->kernel_ms = ms; }
    return TRT_OK;
}

uint32_t trt_streamed_chunk_spp(uint32_t width, uint32_t rows) { return streamed_chunk_spp(width, rows, defaults().tuning.radiance_gb); }

// How the streamed backend would launch this render (host arithmetic only: works without a GPU).
int trt_streamed_launch_plan(const trt_scene* s, const trt_camera* cam, const trt_render_params* p, trt_launch_plan* out) {
    if (!s || !cam || !p || !out) return fail(TRT_ERR_INVALID_ARG, "null argument");
    RenderArgs ra;
    trt_tuning tn;
    uint32_t rows = 0;
    int rc = to_render_args(cam, p, ra, rows, tn);
    if (rc != TRT_OK) return rc;
    const StreamLaunchPlan pl = streamed_launch_plan(s->host.layout, ra, tn, p->collect_stats != 0);
    memset(out, 0, sizeof(*out));
    out->scene_mode = (uint32_t)pl.mode;
    out->threads_per_workgroup = (uint32_t)pl.threads;
    out->waves_per_simd = (uint32_t)pl.waves_per_simd;
    out->workgroups_per_cu = pl.wg_per_cu;
    out->lds_bytes = (uint32_t)pl.lds_bytes;
    out->scene_lds_bytes = (uint32_t)pl.scene_lds_bytes;
    out->leaf_slots = pl.slots;
    out->lds_leaf_stack = pl.lds_stack ? 1u : 0u;
    out->ray_pool = pl.pool ? 1u : 0u;
    out->walk = (uint32_t)pl.walk;
    out->specialised = pl.specialised ? 1u : 0u;
    out->has_kernel = pl.kernel != nullptr ? 1u : 0u;
    out->kernel_waves_per_simd = (uint32_t)pl.kernel_minw;
    out->kernel_threads = (uint32_t)pl.kernel_threads;
    out->kernel_walk = (uint32_t)pl.kernel_walk;
    out->kernel_ray_pool = pl.kernel_pool ? 1u : 0u;
    out->kernel_counting = pl.kernel_stats ? 1u : 0u;
    out->chunk_spp = streamed_chunk_spp(cam->width, rows, tn.radiance_gb);
    out->dual_walk = pl.dual ? 1u : 0u;
    out->workspace_bytes = streamed_workspace_bytes(cam->width, rows, ra.sample_end - ra.sample_begin, tn.radiance_gb);
    return TRT_OK;
}

int trt_kernel_timing_begin(void) {
    std::lock_guard<std::mutex> lock(trt::g_timing_mu);
    for (const trt::TimingPair& pr : trt::g_timing_pairs) { (void)hipEventDestroy(pr.begin); (void)hipEventDestroy(pr.end); }
    trt::g_timing_pairs.clear();
    trt::g_timing_epoch++;
    trt::g_timing_on = true;
    return TRT_OK;
}
int trt_kernel_timing_end(double* total_ms, uint32_t* launches) {
    std::vector<trt::TimingPair> pairs;
    {
        std::lock_guard<std::mutex> lock(trt::g_timing_mu);
        trt::g_timing_on = false;
        pairs.swap(trt::g_timing_pairs);
    }
    double sum = 0.0;
    uint32_t n = 0;
    int rc = TRT_OK;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (const trt::TimingPair& pr : pairs) {
        float ms = 0.0f;
        hipError_t e = hipSetDevice(pr.device);
        if (e == hipSuccess) e = hipEventSynchronize(pr.end);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.begin, pr.end);
        if (e != hipSuccess) { if (rc == TRT_OK) rc = fail_hip(e, "kernel timing events"); continue; }
        sum += ms;
        n++;
    }
    for (const trt::TimingPair& pr : pairs) { (void)hipEventDestroy(pr.begin); (void)hipEventDestroy(pr.end); }
    if (have_prev) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return rc;
}

// Name of the kernel that dominates a render with these settings (what a profile of it shows).
const char* trt_dominant_kernel(const trt_scene* s, const trt_camera* cam, const trt_render_params* p) {
    if (!s || !cam || !p) return "";
    RenderArgs ra;
    trt_tuning tn;
    uint32_t rows = 0;
    if (to_render_args(cam, p, ra, rows, tn) != TRT_OK) return "";
    if (p->backend == TRT_BACKEND_WAVEFRONT) return "trt::wavefront_kernel";
    if (p->backend == TRT_BACKEND_MEGAKERNEL) return "trt::megakernel";
    return streamed_kernel_name(s->host.layout, ra, tn);
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
