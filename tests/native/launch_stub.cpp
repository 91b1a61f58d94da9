// launch_stub.cpp — stand-ins for the kernel launchers of kernels.hip / streamed.hip / wavefront.hip (TEST INFRASTRUCTURE, host
// only; see tests/native/hipstub/hip/hip_runtime.h).  A "render" is a task on the stream that
//   * stamps its workspace with a token of its own, works for a moment, and checks the stamp again: two renders that were
//     handed one workspace at the same time (round 1's bug) corrupt each other's stamp, and the sanitizers see the race;
//   * fills its rows of the frame with a pattern keyed by the IMAGE pixel and the sample range (like the kernels' RNG), adding
//     to the running sums when asked to: a frame assembled from any number of shards must equal the one-shard frame;
//   * adds to the counters like flush_counters does.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <thread>

#include "../../tiny-raytracer_amd/csrc/kernels.h"

namespace trt {

std::atomic<long> g_stub_corruptions{0}, g_stub_launches{0};
static std::atomic<unsigned long long> g_token{1};

static uint32_t stub_image_row(const RenderArgs& ra, uint32_t row) {
    if (ra.band_rows == 0u) return row;
    return ((row / ra.band_rows) * ra.band_stride + ra.band_offset) * ra.band_rows + row % ra.band_rows;
}
float stub_pattern(uint32_t y, uint32_t x, int c, uint32_t seed_key, uint32_t s0, uint32_t s1) {
    return (float)((y * 131u + x * 17u + (uint32_t)c * 5u + seed_key % 13u) % 1000u) * (float)(s1 - s0);
}

static hipError_t stub_render(const CameraDev& cam, const RenderArgs& ra, void* workspace, size_t ws_bytes, float* d_accum,
                              unsigned long long* d_counters, hipStream_t stream) {
    if (ra.rows_local == 0 || cam.width == 0) return hipSuccess;
    const CameraDev c = cam;
    const RenderArgs a = ra;
    timing_mark(stream, true);
    hipstub_enqueue(stream, [=] {
        g_stub_launches++;
        hipstub_log("kernel", d_accum, nullptr, (size_t)c.width * 12u, a.rows_local, (size_t)c.width * 12u);
        const unsigned long long token = g_token++;
        volatile unsigned long long* w = static_cast<unsigned long long*>(workspace);
        const size_t n = ws_bytes / sizeof(unsigned long long);
        if (w && n) { w[0] = token; w[n / 2] = token; w[n - 1] = token; }
        std::this_thread::sleep_for(std::chrono::microseconds(150));
        for (uint32_t row = 0; row < a.rows_local; row++) {
            const uint32_t y = stub_image_row(a, row);
            for (uint32_t x = 0; x < c.width; x++)
                for (int ch = 0; ch < 3; ch++) {
                    float* o = d_accum + 3ull * ((unsigned long long)row * c.width + x) + ch;
                    const float v = stub_pattern(y, x, ch, a.seed_key, a.sample_begin, a.sample_end);
                    *o = a.accumulate ? *o + v : v;
                }
        }
        if (w && n && (w[0] != token || w[n / 2] != token || w[n - 1] != token)) g_stub_corruptions++;
        if (d_counters) {
            d_counters[CTR_SAMPLES] += (unsigned long long)a.rows_local * c.width * (a.sample_end - a.sample_begin);
            d_counters[CTR_RAYS] += 2ull * a.rows_local * c.width * (a.sample_end - a.sample_begin);
        }
    });
    timing_mark(stream, false);
    return hipSuccess;
}

hipError_t launch_megakernel(const SceneDev&, const CameraDev& cam, const RenderArgs& ra, const trt_tuning&, float* d_accum, unsigned long long* d_counters, bool,
                             hipStream_t stream) {
    return stub_render(cam, ra, nullptr, 0, d_accum, d_counters, stream);
}
size_t wavefront_workspace_bytes(uint32_t width, uint32_t rows) { return (size_t)width * rows * 72u + 64u; }
hipError_t launch_wavefront(const SceneDev&, const CameraDev& cam, const RenderArgs& ra, const trt_tuning&, void* workspace, float* d_accum,
                            unsigned long long* d_counters, bool, hipStream_t stream) {
    return stub_render(cam, ra, workspace, wavefront_workspace_bytes(cam.width, ra.rows_local), d_accum, d_counters, stream);
}
uint32_t streamed_chunk_spp(uint32_t, uint32_t, uint32_t) { return 4; }
size_t streamed_workspace_bytes(uint32_t width, uint32_t rows, uint32_t samples, uint32_t gb) {
    const uint32_t chunk = streamed_chunk_spp(width, rows, gb);
    return (size_t)width * rows * (samples < chunk ? (samples ? samples : 1u) : chunk) * 12u + 256u;
}
uint32_t streamed_chunk_that_fits(uint32_t width, uint32_t rows, size_t bytes) { return bytes > 256u ? (uint32_t)((bytes - 256u) / ((size_t)width * rows * 12u)) : 0u; }
const char* streamed_kernel_name(const SceneLayout&, const RenderArgs&, const trt_tuning&) { return "stub"; }
StreamLaunchPlan streamed_launch_plan(const SceneLayout&, const RenderArgs&, const trt_tuning&, bool) { return StreamLaunchPlan{}; }
std::atomic<long> g_stub_short_launches{0};
hipError_t launch_streamed(const SceneDev&, const CameraDev& cam, const RenderArgs& ra, const trt_tuning& tn, void* workspace, size_t workspace_bytes,
                           float* d_accum, unsigned long long* d_counters, bool, hipStream_t stream) {
    if (streamed_chunk_that_fits(cam.width, ra.rows_local, workspace_bytes) == 0u) return hipErrorInvalidValue;
    if (workspace_bytes < streamed_workspace_bytes(cam.width, ra.rows_local, ra.sample_end - ra.sample_begin, tn.radiance_gb)) g_stub_short_launches++;   // granted less than a full launch
    return stub_render(cam, ra, workspace, workspace_bytes, d_accum, d_counters, stream);
}
hipError_t launch_tonemap_u8(const float*, unsigned long long, float, uint8_t*, hipStream_t) { return hipSuccess; }
hipError_t launch_sample_batch(const SceneDev&, const trt_sample_point* d_in, uint32_t n, trt_sampled_color* d_out, const RenderArgs&,
                               unsigned long long* d_counters, bool, hipStream_t stream) {
    hipstub_enqueue(stream, [=] {
        for (uint32_t i = 0; i < n; i++) { d_out[i].x = d_in[i].x; d_out[i].y = d_in[i].y; d_out[i].color = trt_vec3{1.0f, 2.0f, 3.0f}; }
        if (d_counters) d_counters[CTR_SAMPLES] += n;
    });
    return hipSuccess;
}

}  // namespace trt

extern "C" long launch_stub_corruptions(void) { return trt::g_stub_corruptions.load(); }
extern "C" long launch_stub_launches(void) { return trt::g_stub_launches.load(); }
extern "C" long launch_stub_short_launches(void) { return trt::g_stub_short_launches.load(); }
