// hip_runtime.h — a HOST-ONLY stand-in for the slice of the HIP runtime API that tiny-raytracer_amd/csrc/capi.hip uses.
//
// TEST INFRASTRUCTURE (tests/test_host_sanitizers.py): capi.hip's host logic - the per-scene pools of workspaces and render
// contexts, the multi-GPU gather, the timing brackets - is compiled against this header with g++ -fsanitize=thread (and
// address,undefined) and driven by tests/native/capi_host_check.cpp.  Streams are real worker threads that run their queue in
// order, events complete when the stream reaches them, hipFree waits for the device like the real one, "device memory" is the
// C heap (so the sanitizers see every access), and the kernel launchers are replaced by tests/native/launch_stub.cpp.  Nothing
// in the product includes this file; the product's own build uses ROCm's <hip/hip_runtime.h>.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };

typedef enum {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorInvalidDeviceFunction = 98,
    hipErrorInvalidConfiguration = 9, hipErrorNotReady = 600, hipErrorPeerAccessAlreadyEnabled = 704, hipErrorUnknown = 999
} hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef struct hipstub_stream* hipStream_t;
typedef struct hipstub_event* hipEvent_t;
#define hipStreamPerThread ((hipStream_t)2)       /* the real runtime's per-thread default stream handle: one value, many streams */
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpyPeerAsync(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipDeviceCanAccessPeer(int* can, int dev, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);

// ---- for the test driver and the launcher stubs ----
#include <functional>
void hipstub_enqueue(hipStream_t s, std::function<void()> task);      // run `task` in stream order (s == nullptr: the current device's default stream)
void hipstub_fail_after(const char* api, long calls);                 // the `calls`-th next call of `api` ("hipMalloc", "hipEventRecord", ...) fails once; < 0: off
long hipstub_live_allocations(void);
long hipstub_live_streams(void);
long hipstub_live_events(void);
size_t hipstub_live_bytes(void);
long hipstub_malloc_calls(void);                                      // hipMalloc calls so far (failed ones included)
size_t hipstub_peak_bytes(void);
void hipstub_reset_peak(void);
// Operation log (ordering tests): while on, every copy / memset the stub EXECUTES - and whatever a launcher stub reports through
// hipstub_log - is appended in execution order with the stream it ran on.
#include <vector>
struct hipstub_op {
    unsigned long long seq;
    hipStream_t stream;
    int device;                    // the device the stream belongs to
    const char* kind;              // "copy", "copy2d", "peer", "memset", or what hipstub_log was given ("kernel")
    const void* dst;
    const void* src;
    size_t width, height, dpitch;  // bytes per row, rows, destination pitch (1-D operations: height 1)
};
void hipstub_oplog_start(void);
std::vector<hipstub_op> hipstub_oplog_stop(void);
void hipstub_log(const char* kind, const void* dst, const void* src, size_t width, size_t height, size_t dpitch);    // from inside a stream task
long hipstub_errors(void);                                            // protocol violations the stub itself saw (use of a destroyed stream / event, ...)
