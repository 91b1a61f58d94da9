// capi_host_check.cpp — drives the HOST logic of tiny-raytracer_amd/csrc/capi.hip (workspace / context pools, multi-GPU gather,
// timing brackets, error paths) on a simulated HIP runtime (tests/native/hipstub) under ThreadSanitizer or ASan+UBSan.
// TEST INFRASTRUCTURE: built and run by tests/test_host_sanitizers.py; prints "ok ..." lines and returns 0, or says what failed.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tinyrt.h"

extern "C" long launch_stub_corruptions(void);
extern "C" long launch_stub_launches(void);
extern "C" long launch_stub_short_launches(void);
namespace trt { float stub_pattern(uint32_t y, uint32_t x, int c, uint32_t seed_key, uint32_t s0, uint32_t s1); }

static int g_failures = 0;
#define CHECK(cond, ...)                                                                  \
    do {                                                                                  \
        if (!(cond)) { printf("FAIL %s:%d: %s | ", __FILE__, __LINE__, #cond); printf(__VA_ARGS__); printf("\n"); g_failures++; } \
    } while (0)

static uint32_t seed_key(uint32_t seed) {
    uint32_t x = seed + 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static std::vector<float> expected(uint32_t w, uint32_t h, uint32_t seed, uint32_t s0, uint32_t s1, int passes = 1) {
    std::vector<float> f((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                float v = 0.0f;
                for (int k = 0; k < passes; k++) v = k == 0 ? trt::stub_pattern(y, x, c, seed_key(seed), s0, s1) : v + trt::stub_pattern(y, x, c, seed_key(seed), s0, s1);
                f[((size_t)y * w + x) * 3 + c] = v;
            }
    return f;
}

static trt_scene* make_scene(const trt_scene_options* opt = nullptr) {
    trt_world* w = nullptr;
    trt_world_create(&w);
    trt_material m{TRT_LAMBERTIAN, {0.5f, 0.5f, 0.5f}, 0.0f};
    trt_world_add_material(w, "m", &m);
    for (int i = 0; i < 5; i++) trt_world_add_sphere(w, trt_vec3{(float)i, 0.0f, -3.0f}, 0.4f, 0);
    trt_scene* s = nullptr;
    if (trt_scene_create_ex(w, opt, &s) != TRT_OK) { printf("scene: %s\n", trt_last_error()); exit(2); }
    trt_world_destroy(w);
    return s;
}
static trt_camera make_camera(uint32_t w, uint32_t h) {
    trt_camera c;
    trt_camera_init(&c, 1.0f, 0.0f, trt_vec3{0, 0, 0}, trt_vec3{0, 0, -1}, trt_vec3{0, 1, 0}, 60.0f, w, h);
    return c;
}
static trt_render_params params(uint32_t seed, uint32_t backend, uint32_t s0 = 0, uint32_t s1 = 0, uint32_t acc = 0) {
    trt_render_params p;
    memset(&p, 0, sizeof(p));                                              // tuning = NULL: the library defaults
    p.samples_per_pixel = 8; p.max_bounces = 4; p.seed = seed; p.backend = backend; p.sample_begin = s0; p.sample_end = s1; p.accumulate = acc;
    return p;
}

// six host threads x three repetitions on ONE scene handle, both scratch-using backends, mixed image sizes (so workspaces are
// regrown while others are in flight)
static void concurrent_renders(trt_scene* s) {
    const uint32_t sizes[6][2] = {{48, 40}, {64, 33}, {48, 40}, {96, 70}, {31, 17}, {64, 33}};
    for (uint32_t backend : {TRT_BACKEND_STREAMED, TRT_BACKEND_WAVEFRONT, TRT_BACKEND_MEGAKERNEL}) {
        for (int rep = 0; rep < 3; rep++) {
            std::vector<std::thread> th;
            std::vector<std::vector<float>> got(6);
            std::atomic<int> errors{0};
            for (int t = 0; t < 6; t++) {
                th.emplace_back([&, t] {
                    trt_camera cam = make_camera(sizes[t][0], sizes[t][1]);
                    trt_render_params p = params(100 + t, backend);
                    got[t].assign((size_t)sizes[t][0] * sizes[t][1] * 3, -1.0f);
                    trt_stats st;
                    if (trt_render(s, &cam, &p, got[t].data(), &st) != TRT_OK) { errors++; return; }
                    if (st.samples != (uint64_t)sizes[t][0] * sizes[t][1] * 8) errors++;
                });
            }
            for (auto& t : th) t.join();
            CHECK(errors == 0, "backend %u rep %d: %s", backend, rep, trt_last_error());
            for (int t = 0; t < 6; t++) CHECK(got[t] == expected(sizes[t][0], sizes[t][1], 100 + t, 0, 8), "backend %u thread %d frame differs", backend, t);
        }
    }
    CHECK(launch_stub_corruptions() == 0, "%ld renders shared a workspace", launch_stub_corruptions());
    printf("ok concurrent renders: %ld launches, %ld live allocations, %ld streams\n", launch_stub_launches(), hipstub_live_allocations(), hipstub_live_streams());
}

// trt_render_device on three un-synchronised streams, one of them needing a larger workspace, back to back on one stream
static void device_streams(trt_scene* s) {
    hipStream_t st[3];
    for (auto& x : st) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    trt_camera small = make_camera(40, 30), big = make_camera(120, 90);
    float *a, *b, *c;
    hipMalloc((void**)&a, 40 * 30 * 12); hipMalloc((void**)&b, 40 * 30 * 12); hipMalloc((void**)&c, 120 * 90 * 12);
    for (int rep = 0; rep < 4; rep++) {
        trt_render_params pa = params(1, TRT_BACKEND_AUTO), pb = params(2, TRT_BACKEND_WAVEFRONT), pc = params(3, TRT_BACKEND_STREAMED);
        CHECK(trt_render_device(s, &small, &pa, a, nullptr, st[0]) == TRT_OK, "%s", trt_last_error());
        CHECK(trt_render_device(s, &small, &pb, b, nullptr, st[1]) == TRT_OK, "%s", trt_last_error());
        CHECK(trt_render_device(s, &big, &pc, c, nullptr, st[2]) == TRT_OK, "%s", trt_last_error());
        CHECK(trt_render_device(s, &small, &pa, a, nullptr, st[0]) == TRT_OK, "%s", trt_last_error());
    }
    for (auto& x : st) hipStreamSynchronize(x);
    std::vector<float> ha(40 * 30 * 3), hb(40 * 30 * 3), hc(120 * 90 * 3);
    hipMemcpy(ha.data(), a, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, hb.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
    CHECK(ha == expected(40, 30, 1, 0, 8) && hb == expected(40, 30, 2, 0, 8) && hc == expected(120, 90, 3, 0, 8), "async frames differ");
    CHECK(launch_stub_corruptions() == 0, "%ld renders shared a workspace", launch_stub_corruptions());
    // renders enqueued back to back on ONE stream need ONE workspace: stream order serialises them (workspace_acquire, step 1b)
    CHECK(trt_scene_trim(s) == TRT_OK, "%s", trt_last_error());
    const long before = hipstub_live_allocations();
    for (int k = 0; k < 12; k++) {
        trt_render_params pc = params(3, TRT_BACKEND_STREAMED);
        CHECK(trt_render_device(s, &big, &pc, c, nullptr, st[2]) == TRT_OK, "%s", trt_last_error());
    }
    CHECK(hipstub_live_allocations() - before <= 1, "%ld workspaces for twelve renders on one stream", hipstub_live_allocations() - before);
    hipStreamSynchronize(st[2]);
    hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
    CHECK(hc == expected(120, 90, 3, 0, 8), "back-to-back frames differ");
    CHECK(launch_stub_corruptions() == 0, "%ld renders shared a workspace", launch_stub_corruptions());
    hipFree(a); hipFree(b); hipFree(c);
    for (auto& x : st) hipStreamDestroy(x);
    printf("ok device streams\n");
}

// one call, N shards: host frame and device frame, peer access on and off, ragged heights, more shards than bands, progressive
static void multi(trt_scene* s) {
    const int dev_lists[][13] = {{0}, {0, 1}, {0, 1, 2, 3}, {1, 0, 1}, {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1}};
    const uint32_t dev_n[] = {1, 2, 4, 3, 13};
    for (uint32_t h : {15u, 16u, 17u, 70u, 250u}) {
        const uint32_t w = 24;
        trt_camera cam = make_camera(w, h);
        for (int k = 0; k < 5; k++) {
            trt_render_params p = params(7, TRT_BACKEND_AUTO);
            std::vector<float> frame((size_t)w * h * 3, -1.0f);
            trt_stats st;
            CHECK(trt_render_multi(s, &cam, &p, dev_lists[k], dev_n[k], frame.data(), &st) == TRT_OK, "%s", trt_last_error());
            CHECK(frame == expected(w, h, 7, 0, 8), "multi host frame h=%u list %d", h, k);
            CHECK(st.samples == (uint64_t)w * h * 8, "multi samples %llu", (unsigned long long)st.samples);
            // progressive: two passes of the same range accumulate
            trt_render_params p2 = params(7, TRT_BACKEND_AUTO, 0, 8, 1);
            CHECK(trt_render_multi(s, &cam, &p2, dev_lists[k], dev_n[k], frame.data(), nullptr) == TRT_OK, "%s", trt_last_error());
            CHECK(frame == expected(w, h, 7, 0, 8, 2), "multi accumulate h=%u list %d", h, k);
            // frame gathered into HBM on devices[0]
            for (const char* peer : {"1", "0"}) {
                setenv("HIPSTUB_PEER", peer, 1);
                hipSetDevice(dev_lists[k][0]);
                float* d = nullptr;
                hipMalloc((void**)&d, frame.size() * 4);
                CHECK(trt_render_multi_device(s, &cam, &p, dev_lists[k], dev_n[k], d, nullptr) == TRT_OK, "%s", trt_last_error());
                std::vector<float> back(frame.size());
                hipMemcpy(back.data(), d, back.size() * 4, hipMemcpyDeviceToHost);
                CHECK(back == expected(w, h, 7, 0, 8), "multi device frame h=%u list %d peer %s", h, k, peer);
                hipFree(d);
                hipSetDevice(0);
            }
        }
    }
    CHECK(launch_stub_corruptions() == 0, "%ld renders shared a workspace", launch_stub_corruptions());
    printf("ok multi: %ld live streams, %ld live allocations\n", hipstub_live_streams(), hipstub_live_allocations());
}

// more shards on ONE device than round 3's idle-context limit (16): no stream or event is destroyed while the scene lives (ADVICE r3:
// context_release used to destroy surplus contexts, whose streams workspace events had been recorded on), later calls reuse the pool
static void many_shards_on_one_device(trt_scene* s) {
    std::vector<int> devs(40, 1);
    const uint32_t w = 24, h = 40 * 16 + 5;
    trt_camera cam = make_camera(w, h);
    std::vector<float> frame((size_t)w * h * 3);
    const long errors_before = hipstub_errors();
    long streams_after_first = 0;
    for (int rep = 0; rep < 3; rep++) {
        trt_render_params p = params(21 + rep, TRT_BACKEND_AUTO);
        trt_stats st;
        CHECK(trt_render_multi(s, &cam, &p, devs.data(), 40, frame.data(), &st) == TRT_OK, "%s", trt_last_error());
        CHECK(frame == expected(w, h, 21 + rep, 0, 8), "40 shards on one device, rep %d", rep);
        CHECK(st.gather_per_band == 0, "host frame gathered band by band");
        if (rep == 0) streams_after_first = hipstub_live_streams();
        else CHECK(hipstub_live_streams() <= streams_after_first + 8, "the context pool keeps growing: %ld -> %ld streams", streams_after_first, hipstub_live_streams());
    }
    CHECK(hipstub_errors() == errors_before, "%ld uses of a destroyed stream / event", hipstub_errors() - errors_before);
    CHECK(launch_stub_corruptions() == 0, "%ld renders shared a workspace", launch_stub_corruptions());
    printf("ok 40 shards on one device: %ld live streams\n", hipstub_live_streams());
}

// What the gather relies on, read off the order in which the simulated runtime EXECUTED things: on every shard's stream the running sums
// come in (progressive pass) before the kernel, the kernel runs before the shard's rows leave, the rows of different shards never overlap
// in the frame, together they cover it exactly once, and nothing touches the frame after the call has returned - with peer access (one
// strided 2-D copy per shard) and without (band by band, reported in trt_stats.gather_per_band).
static void gather_ordering(trt_scene* s) {
    const int devs[4] = {0, 1, 2, 3};
    const uint32_t w = 20, h = 16 * 9 + 7;                                   // ten bands, the last one ragged
    trt_camera cam = make_camera(w, h);
    const size_t frame_bytes = (size_t)w * h * 12;
    for (const char* peer : {"1", "0"}) {
        setenv("HIPSTUB_PEER", peer, 1);
        hipSetDevice(0);
        float* d = nullptr;
        hipMalloc((void**)&d, frame_bytes);
        for (uint32_t acc = 0; acc < 2; acc++) {
            trt_render_params p = params(31, TRT_BACKEND_AUTO, 0, 8, acc);
            trt_stats st;
            hipstub_oplog_start();
            CHECK(trt_render_multi_device(s, &cam, &p, devs, 4, d, &st) == TRT_OK, "%s", trt_last_error());
            std::vector<hipstub_op> log = hipstub_oplog_stop();
            const char* lo = reinterpret_cast<const char*>(d);
            auto in_frame = [&](const void* q) { return q >= (const void*)lo && q < (const void*)(lo + frame_bytes); };
            std::vector<int> written(frame_bytes / 4, 0);
            std::vector<hipStream_t> streams;
            for (const hipstub_op& op : log) if (std::find(streams.begin(), streams.end(), op.stream) == streams.end()) streams.push_back(op.stream);
            int kernels = 0;
            for (hipStream_t sm : streams) {
                int phase = 0;                                               // 0 before the kernel, 1 after it
                bool saw_read = false, saw_write = false;
                for (const hipstub_op& op : log) {
                    if (op.stream != sm) continue;
                    if (!strcmp(op.kind, "kernel")) { CHECK(phase == 0, "two kernels on one shard stream"); phase = 1; kernels++; continue; }
                    if (in_frame(op.src)) { CHECK(phase == 0, "peer %s acc %u: running sums read AFTER the kernel", peer, acc); saw_read = true; }
                    if (in_frame(op.dst)) {
                        CHECK(phase == 1, "peer %s acc %u: rows written to the frame BEFORE the kernel", peer, acc);
                        saw_write = true;
                        for (size_t r = 0; r < op.height; r++)
                            for (size_t b = 0; b < op.width; b += 4) written[((const char*)op.dst - lo + r * op.dpitch + b) / 4]++;
                    }
                }
                if (phase == 1) CHECK(saw_write && saw_read == (acc == 1), "peer %s acc %u: a shard that rendered did not gather (or read sums it should not)", peer, acc);
            }
            CHECK(kernels == 4, "%d shard kernels", kernels);
            bool once = true;
            for (int c : written) once = once && c == 1;
            CHECK(once, "peer %s acc %u: the shards' rows do not tile the frame exactly once", peer, acc);
            CHECK(st.gather_per_band == (strcmp(peer, "0") ? 0u : 3u), "peer %s: gather_per_band %llu", peer, (unsigned long long)st.gather_per_band);
            // the call has returned: nothing is still queued that touches the frame
            hipstub_oplog_start();
            std::vector<float> back(frame_bytes / 4);
            hipMemcpy(back.data(), d, frame_bytes, hipMemcpyDeviceToHost);
            log = hipstub_oplog_stop();
            for (const hipstub_op& op : log) CHECK(!in_frame(op.dst), "a write to the frame ran after trt_render_multi_device returned");
            CHECK(back == expected(w, h, 31, 0, 8, acc ? 2 : 1), "peer %s acc %u frame", peer, acc);
        }
        hipFree(d);
    }
    // the runtime refuses the strided copy between two devices: the per-band peer copies take over, and the stats say so
    setenv("HIPSTUB_PEER", "1", 1);
    {
        hipSetDevice(0);
        float* d = nullptr;
        hipMalloc((void**)&d, frame_bytes);
        trt_render_params p = params(32, TRT_BACKEND_AUTO);
        trt_stats st;
        setenv("HIPSTUB_REFUSE_CROSS_2D", "1", 1);
        CHECK(trt_render_multi_device(s, &cam, &p, devs, 4, d, &st) == TRT_OK, "%s", trt_last_error());
        unsetenv("HIPSTUB_REFUSE_CROSS_2D");
        std::vector<float> back(frame_bytes / 4);
        hipMemcpy(back.data(), d, frame_bytes, hipMemcpyDeviceToHost);
        CHECK(back == expected(w, h, 32, 0, 8), "frame after the refused strided copy");
        CHECK(st.gather_per_band == 3, "fallback not reported: %llu", (unsigned long long)st.gather_per_band);
        hipFree(d);
    }
    printf("ok gather ordering\n");
}

// timing brackets with several rendering threads and a multi-device render in flight: pairs never mix
static void timing(trt_scene* s) {
    CHECK(trt_kernel_timing_begin() == TRT_OK, "begin");
    trt_camera cam = make_camera(32, 32);
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.emplace_back([&, t] {
        std::vector<float> f(32 * 32 * 3);
        trt_render_params p = params(t, t & 1 ? TRT_BACKEND_WAVEFRONT : TRT_BACKEND_STREAMED);
        for (int k = 0; k < 3; k++) trt_render(s, &cam, &p, f.data(), nullptr);
    });
    const int devs[4] = {0, 1, 0, 1};
    std::vector<float> f(32 * 32 * 3);
    trt_render_params p = params(9, TRT_BACKEND_AUTO);
    trt_render_multi(s, &cam, &p, devs, 4, f.data(), nullptr);
    for (auto& t : th) t.join();
    double ms = -1.0;
    uint32_t n = 0;
    CHECK(trt_kernel_timing_end(&ms, &n) == TRT_OK, "%s", trt_last_error());
    CHECK(n == 4 * 3 + 2 && ms >= 0.0, "timing pairs %u (want 14: two of the four shards own no band of a 32-row image)", n);
    trt_render(s, &cam, &p, f.data(), nullptr);
    CHECK(trt_kernel_timing_end(&ms, &n) == TRT_OK && n == 0, "timing off");
    printf("ok timing\n");
}

// the byte cap: idle scratch is freed when renders end; trim frees the rest
static void cap_and_trim() {
    trt_scene_options opt;
    trt_scene_options_default(&opt);
    opt.scratch_cap_bytes = 1u << 20;
    trt_scene* s = make_scene(&opt);
    const size_t base = hipstub_live_bytes();
    trt_camera cam = make_camera(256, 200);                                  // streamed workspace 2.4 MB, frame 0.6 MB
    std::vector<std::thread> th;
    for (int t = 0; t < 5; t++) th.emplace_back([&, t] {
        std::vector<float> f(256 * 200 * 3);
        trt_render_params p = params(t, TRT_BACKEND_STREAMED);
        for (int k = 0; k < 2; k++) if (trt_render(s, &cam, &p, f.data(), nullptr) != TRT_OK) { printf("FAIL cap render: %s\n", trt_last_error()); g_failures++; }
    });
    for (auto& t : th) t.join();
    const size_t after = hipstub_live_bytes() - base;
    CHECK(after <= (1u << 20) + 4096u * 64u, "idle scratch %zu bytes exceeds the 1 MiB cap", after);
    CHECK(trt_scene_trim(s) == TRT_OK, "trim");
    const size_t trimmed = hipstub_live_bytes() - base;
    CHECK(trimmed < 64u * 1024u, "after trim %zu bytes of scratch remain (scene blob and counters only expected)", trimmed);
    std::vector<float> f(256 * 200 * 3);
    trt_render_params p = params(3, TRT_BACKEND_STREAMED);
    CHECK(trt_render(s, &cam, &p, f.data(), nullptr) == TRT_OK && f == expected(256, 200, 3, 0, 8), "render after trim");
    trt_scene_destroy(s);
    printf("ok cap and trim\n");
}

// every HIP call failing once, at every position: an error comes back, nothing leaks, nothing hangs, the next render works
static void failure_injection() {
    const char* apis[] = {"hipMalloc", "hipEventCreate", "hipStreamCreate", "hipEventRecord", "hipMemcpyAsync", "hipMemsetAsync",
                          "hipStreamSynchronize", "hipMemcpy2DAsync", "hipStreamWaitEvent", "hipMemcpy"};
    int failed_calls = 0;
    for (const char* api : apis) {
        for (long at = 0; at < 8; at++) {
            trt_scene* s = make_scene();
            trt_camera cam = make_camera(40, 36);
            std::vector<float> f(40 * 36 * 3);
            trt_render_params p = params(5, TRT_BACKEND_STREAMED);
            hipstub_fail_after(api, at);
            const int devs[3] = {0, 1, 0};
            int rc1 = trt_render(s, &cam, &p, f.data(), nullptr);
            int rc2 = trt_render_multi(s, &cam, &p, devs, 3, f.data(), nullptr);
            hipstub_fail_after(api, -1);
            if (rc1 != TRT_OK || rc2 != TRT_OK) failed_calls++;
            CHECK((rc1 == TRT_OK || rc1 == TRT_ERR_HIP) && (rc2 == TRT_OK || rc2 == TRT_ERR_HIP), "%s@%ld: rc %d %d", api, at, rc1, rc2);
            int rc3 = trt_render(s, &cam, &p, f.data(), nullptr);                       // the pools recovered
            CHECK(rc3 == TRT_OK && f == expected(40, 36, 5, 0, 8), "%s@%ld: render after the failure: %d %s", api, at, rc3, trt_last_error());
            int rc4 = trt_render_multi(s, &cam, &p, devs, 3, f.data(), nullptr);
            CHECK(rc4 == TRT_OK && f == expected(40, 36, 5, 0, 8), "%s@%ld: multi after the failure: %d %s", api, at, rc4, trt_last_error());
            trt_scene_destroy(s);
        }
    }
    CHECK(failed_calls > 20, "only %d injected failures surfaced", failed_calls);
    // device memory exhausted by concurrent renders: later ones queue behind running ones instead of failing
    {
        trt_scene* s = make_scene();
        trt_camera cam = make_camera(128, 100);                              // workspace 0.6 MB
        std::vector<float> warm(128 * 100 * 3);
        trt_render_params p0 = params(0, TRT_BACKEND_STREAMED);
        trt_render(s, &cam, &p0, warm.data(), nullptr);
        char cap[32];
        snprintf(cap, sizeof cap, "%zu", hipstub_live_bytes() + 1600u * 1024u);       // room for two more frames / workspaces at most
        setenv("HIPSTUB_HBM_BYTES", cap, 1);
        std::atomic<int> ok{0}, oom{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 6; t++) th.emplace_back([&, t] {
            std::vector<float> f(128 * 100 * 3);
            trt_render_params p = params(t, TRT_BACKEND_STREAMED);
            int rc = trt_render(s, &cam, &p, f.data(), nullptr);
            if (rc == TRT_OK && f == expected(128, 100, t, 0, 8)) ok++; else if (rc == TRT_ERR_HIP || rc == TRT_ERR_OOM) oom++; else { printf("FAIL oom render rc %d\n", rc); g_failures++; }
        });
        for (auto& t : th) t.join();
        unsetenv("HIPSTUB_HBM_BYTES");
        CHECK(ok >= 1 && ok + oom == 6, "ok %d oom %d", ok.load(), oom.load());
        trt_scene_destroy(s);
    }
    // a FIRST render that cannot have its full-size scratch (torch, or another scene's cached scratch, holds the HBM): it asks for
    // half the samples per launch, and half again, instead of failing; the frame is the same
    {
        trt_scene* s = make_scene();
        trt_camera cam = make_camera(128, 100);                              // full-size workspace: 128 x 100 x 4 spp x 12 B = 614 KB; frame 154 KB
        char cap[32];
        snprintf(cap, sizeof cap, "%zu", hipstub_live_bytes() + 154u * 1024u + 400u * 1024u);      // room for the frame and two thirds of the workspace
        setenv("HIPSTUB_HBM_BYTES", cap, 1);
        const long before = launch_stub_short_launches();
        std::vector<float> f(128 * 100 * 3);
        trt_render_params p = params(11, TRT_BACKEND_STREAMED);
        const int rc = trt_render(s, &cam, &p, f.data(), nullptr);
        unsetenv("HIPSTUB_HBM_BYTES");
        CHECK(rc == TRT_OK && f == expected(128, 100, 11, 0, 8), "render with HBM short: rc %d %s", rc, trt_last_error());
        CHECK(launch_stub_short_launches() > before, "the render was not granted a smaller workspace");
        // the pressure LASTS: the next renders start from the size that was granted - no hipFree + failing full-size hipMalloc +
        // hipMalloc of the shorter workspace per render - and still render the same frame
        setenv("HIPSTUB_HBM_BYTES", cap, 1);
        const long mallocs_before = hipstub_malloc_calls();
        for (int k = 0; k < 3; k++) {
            const int rck = trt_render(s, &cam, &p, f.data(), nullptr);
            CHECK(rck == TRT_OK && f == expected(128, 100, 11, 0, 8), "render %d under lasting pressure: rc %d %s", k, rck, trt_last_error());
        }
        CHECK(hipstub_malloc_calls() == mallocs_before, "%ld device allocations in three renders under lasting pressure (expected none: the granted workspace is reused)",
              hipstub_malloc_calls() - mallocs_before);
        // ... and once the memory is back (hipMemGetInfo), the next render takes its full-size launch again
        unsetenv("HIPSTUB_HBM_BYTES");
        const long short_before = launch_stub_short_launches();
        const int rcf = trt_render(s, &cam, &p, f.data(), nullptr);
        CHECK(rcf == TRT_OK && f == expected(128, 100, 11, 0, 8) && launch_stub_short_launches() == short_before, "render after the pressure: rc %d, %ld short launches",
              rcf, launch_stub_short_launches() - short_before);
        trt_scene_destroy(s);
    }
    printf("ok failure injection (%d failed calls recovered from)\n", failed_calls);
}

int main(int argc, char** argv) {
    setenv("HIPSTUB_DEVICES", "4", 1);
    const size_t base_allocs = hipstub_live_allocations();
    trt_scene* s = make_scene();
    concurrent_renders(s);
    device_streams(s);
    multi(s);
    many_shards_on_one_device(s);
    gather_ordering(s);
    timing(s);
    trt_scene_destroy(s);
    CHECK(hipstub_live_allocations() == (long)base_allocs && hipstub_live_streams() == 0 && hipstub_live_events() == 0,
          "leak after destroy: %ld allocations, %ld streams, %ld events", hipstub_live_allocations(), hipstub_live_streams(), hipstub_live_events());
    cap_and_trim();
    if (argc < 2 || strcmp(argv[1], "quick") != 0) failure_injection();
    CHECK(hipstub_live_allocations() == (long)base_allocs && hipstub_live_streams() == 0 && hipstub_live_events() == 0,
          "leak at exit: %ld allocations, %ld streams, %ld events", hipstub_live_allocations(), hipstub_live_streams(), hipstub_live_events());
    CHECK(hipstub_errors() == 0, "%ld uses of a destroyed stream / event / allocation", hipstub_errors());
    CHECK(launch_stub_corruptions() == 0, "%ld renders shared a workspace", launch_stub_corruptions());
    if (g_failures) { printf("%d check(s) failed\n", g_failures); return 1; }
    printf("ok all\n");
    return 0;
}
