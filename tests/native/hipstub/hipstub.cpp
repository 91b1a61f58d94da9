// hipstub.cpp — implementation of tests/native/hipstub/hip/hip_runtime.h (test infrastructure, host only).
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

struct hipstub_event {
    std::mutex mu;
    std::condition_variable cv;
    unsigned long long recorded = 0, completed = 0;          // generations
    std::chrono::steady_clock::time_point stamp;
    int device = 0;
};

static void t_running_on_set(struct hipstub_stream* s);
struct hipstub_stream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, running = false;
    int device = 0;
    std::thread worker;
    void loop() {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv.wait(lock, [&] { return stop || !q.empty(); });
            if (q.empty()) { if (stop) return; continue; }
            std::function<void()> t = std::move(q.front());
            q.pop_front();
            running = true;
            lock.unlock();
            t_running_on_set(this);
            t();
            lock.lock();
            running = false;
            cv.notify_all();
        }
    }
    void drain() {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return q.empty() && !running; });
    }
};

namespace {
std::mutex g_mu;
std::shared_mutex g_life;                                 // shared: somebody walks the streams of a device; exclusive: a stream object is deleted
std::set<hipstub_stream*> g_streams;
std::set<hipstub_event*> g_events;
std::map<void*, size_t> g_allocs;
std::map<void*, int> g_alloc_dev;                          // allocation -> device it was made on
std::map<int, hipstub_stream*> g_default_stream;
size_t g_live_bytes = 0, g_peak_bytes = 0;
std::map<std::string, long> g_fail;
std::atomic<long> g_errors{0};
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;
thread_local hipstub_stream* t_running_on = nullptr;      // the stream whose worker thread this is
std::mutex g_log_mu;
bool g_log_on = false;
std::vector<hipstub_op> g_log;

int device_count() { const char* e = getenv("HIPSTUB_DEVICES"); int n = e ? atoi(e) : 2; return n; }
bool inject(const char* api) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_fail.find(api);
    if (it == g_fail.end() || it->second < 0) return false;
    if (it->second == 0) { it->second = -1; return true; }
    it->second--;
    return false;
}
hipError_t ret(hipError_t e) { if (e != hipSuccess) t_last = e; return e; }
hipstub_stream* make_stream(int dev) {
    hipstub_stream* s = new hipstub_stream();
    s->device = dev;
    s->worker = std::thread([s] { s->loop(); });
    return s;
}
hipstub_stream* resolve(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (s == nullptr) {
        hipstub_stream*& d = g_default_stream[t_device];
        if (!d) d = make_stream(t_device);
        return d;
    }
    if (!g_streams.count(s)) { g_errors++; return nullptr; }         // use of a destroyed / unknown stream
    return s;
}
bool live_event(hipEvent_t e) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_events.count(e)) { g_errors++; return false; }
    return true;
}
std::vector<hipstub_stream*> streams_of(int dev) {
    std::lock_guard<std::mutex> lock(g_mu);
    std::vector<hipstub_stream*> v;
    for (hipstub_stream* s : g_streams) if (s->device == dev) v.push_back(s);
    auto it = g_default_stream.find(dev);
    if (it != g_default_stream.end() && it->second) v.push_back(it->second);
    return v;
}
}  // namespace

static void t_running_on_set(hipstub_stream* s) { t_running_on = s; }
void hipstub_oplog_start(void) { std::lock_guard<std::mutex> lock(g_log_mu); g_log.clear(); g_log_on = true; }
std::vector<hipstub_op> hipstub_oplog_stop(void) { std::lock_guard<std::mutex> lock(g_log_mu); g_log_on = false; return std::move(g_log); }
void hipstub_log(const char* kind, const void* dst, const void* src, size_t width, size_t height, size_t dpitch) {
    std::lock_guard<std::mutex> lock(g_log_mu);
    if (!g_log_on) return;
    g_log.push_back(hipstub_op{(unsigned long long)g_log.size(), t_running_on, t_running_on ? t_running_on->device : -1, kind, dst, src, width, height, dpitch});
}
void hipstub_enqueue(hipStream_t s, std::function<void()> task) {
    hipstub_stream* st = resolve(s);
    if (!st) return;
    std::lock_guard<std::mutex> lock(st->mu);
    st->q.push_back(std::move(task));
    st->cv.notify_all();
}
void hipstub_fail_after(const char* api, long calls) { std::lock_guard<std::mutex> lock(g_mu); g_fail[api] = calls; }
long hipstub_live_allocations(void) { std::lock_guard<std::mutex> lock(g_mu); return (long)g_allocs.size(); }
long hipstub_live_streams(void) { std::lock_guard<std::mutex> lock(g_mu); return (long)g_streams.size(); }
long hipstub_live_events(void) { std::lock_guard<std::mutex> lock(g_mu); return (long)g_events.size(); }
size_t hipstub_live_bytes(void) { std::lock_guard<std::mutex> lock(g_mu); return g_live_bytes; }
size_t hipstub_peak_bytes(void) { std::lock_guard<std::mutex> lock(g_mu); return g_peak_bytes; }
void hipstub_reset_peak(void) { std::lock_guard<std::mutex> lock(g_mu); g_peak_bytes = g_live_bytes; }
long hipstub_errors(void) { return g_errors.load(); }

const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorNotReady: return "device not ready";
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorInvalidDevice: return "invalid device ordinal";
        default: return "stub error";
    }
}
hipError_t hipGetLastError(void) { hipError_t e = t_last; t_last = hipSuccess; return e; }
hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return ret(hipErrorInvalidDevice); t_device = d; return hipSuccess; }

static std::atomic<long> g_malloc_calls{0};
long hipstub_malloc_calls(void) { return g_malloc_calls.load(); }
hipError_t hipMalloc(void** p, size_t bytes) {
    g_malloc_calls++;
    if (inject("hipMalloc")) { *p = nullptr; return ret(hipErrorOutOfMemory); }
    const char* cap = getenv("HIPSTUB_HBM_BYTES");
    std::lock_guard<std::mutex> lock(g_mu);
    if (cap && g_live_bytes + bytes > (size_t)strtoull(cap, nullptr, 10)) { *p = nullptr; return ret(hipErrorOutOfMemory); }
    void* q = malloc(bytes ? bytes : 1);
    if (!q) { *p = nullptr; return ret(hipErrorOutOfMemory); }
    memset(q, 0xA5, bytes);                                    // device memory is not zeroed
    g_allocs[q] = bytes;
    g_alloc_dev[q] = t_device;
    g_live_bytes += bytes;
    if (g_live_bytes > g_peak_bytes) g_peak_bytes = g_live_bytes;
    *p = q;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
    if (inject("hipMemGetInfo")) return ret(hipErrorUnknown);
    const char* cap = getenv("HIPSTUB_HBM_BYTES");
    std::lock_guard<std::mutex> lock(g_mu);
    const size_t total = cap ? (size_t)strtoull(cap, nullptr, 10) : ((size_t)288 << 30);
    *total_b = total;
    *free_b = total > g_live_bytes ? total - g_live_bytes : 0;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    {   // the real hipFree synchronises the device (and holds references to its streams while it does: so does the stub)
        std::shared_lock<std::shared_mutex> life(g_life);
        for (hipstub_stream* s : streams_of(t_device)) s->drain();
    }
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_allocs.find(p);
    if (it == g_allocs.end()) { g_errors++; return ret(hipErrorInvalidValue); }
    g_live_bytes -= it->second;
    g_allocs.erase(it);
    g_alloc_dev.erase(p);
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    if (inject("hipMemcpyAsync")) return ret(hipErrorUnknown);
    hipstub_enqueue(s, [=] { hipstub_log("copy", dst, src, bytes, 1, bytes); memcpy(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind k) {
    if (inject("hipMemcpy")) return ret(hipErrorUnknown);
    hipstub_stream* st = resolve(nullptr);
    hipstub_enqueue(nullptr, [=] { memcpy(dst, src, bytes); });
    st->drain();
    (void)k;
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
    if (inject("hipMemcpy2DAsync")) return ret(hipErrorUnknown);
    if (width > dpitch || width > spitch) return ret(hipErrorInvalidValue);
    if (getenv("HIPSTUB_REFUSE_CROSS_2D")) {                  // a runtime that has no strided copy between two devices
        std::lock_guard<std::mutex> lock(g_mu);
        auto owner = [&](const void* q) { auto it = g_allocs.upper_bound(const_cast<void*>(q)); if (it == g_allocs.begin()) return -1; --it;
                                          return (const char*)q < (const char*)it->first + it->second ? g_alloc_dev[it->first] : -1; };
        const int a = owner(dst), b = owner(src);
        if (a >= 0 && b >= 0 && a != b) return ret(hipErrorInvalidValue);
    }
    hipstub_enqueue(s, [=] { hipstub_log("copy2d", dst, src, width, height, dpitch); for (size_t r = 0; r < height; r++) memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, width); });
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void* dst, int, const void* src, int, size_t bytes, hipStream_t s) {
    if (inject("hipMemcpyPeerAsync")) return ret(hipErrorUnknown);
    hipstub_enqueue(s, [=] { hipstub_log("peer", dst, src, bytes, 1, bytes); memcpy(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s) {
    if (inject("hipMemsetAsync")) return ret(hipErrorUnknown);
    hipstub_enqueue(s, [=] { hipstub_log("memset", dst, nullptr, bytes, 1, bytes); memset(dst, value, bytes); });
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    if (inject("hipStreamCreate")) { *s = nullptr; return ret(hipErrorOutOfMemory); }
    hipstub_stream* st = make_stream(t_device);
    std::lock_guard<std::mutex> lock(g_mu);
    g_streams.insert(st);
    *s = st;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_streams.erase(s)) { g_errors++; return ret(hipErrorInvalidValue); }
    }
    { std::lock_guard<std::mutex> lock(s->mu); s->stop = true; s->cv.notify_all(); }      // pending work still runs to its end
    s->worker.join();
    std::unique_lock<std::shared_mutex> life(g_life);                                      // nobody is draining it any more
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    if (inject("hipStreamSynchronize")) return ret(hipErrorUnknown);
    hipstub_stream* st = resolve(s);
    if (!st) return ret(hipErrorInvalidValue);
    st->drain();
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
    if (inject("hipEventCreate")) { *e = nullptr; return ret(hipErrorOutOfMemory); }
    hipstub_event* ev = new hipstub_event();
    ev->device = t_device;
    std::lock_guard<std::mutex> lock(g_mu);
    g_events.insert(ev);
    *e = ev;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_events.erase(e)) { g_errors++; return ret(hipErrorInvalidValue); }
    }
    // a pending record / wait may still refer to it: real HIP keeps the object alive until then; so does the stub
    {
        std::shared_lock<std::shared_mutex> life(g_life);
        for (int d = 0; d < device_count(); d++) for (hipstub_stream* s : streams_of(d)) s->drain();
    }
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (inject("hipEventRecord")) return ret(hipErrorUnknown);
    if (!live_event(e)) return ret(hipErrorInvalidValue);
    unsigned long long gen;
    { std::lock_guard<std::mutex> lock(e->mu); gen = ++e->recorded; }
    hipstub_enqueue(s, [e, gen] {
        std::lock_guard<std::mutex> lock(e->mu);
        if (gen > e->completed) e->completed = gen;
        e->stamp = std::chrono::steady_clock::now();
        e->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    if (!live_event(e)) return ret(hipErrorInvalidValue);
    std::lock_guard<std::mutex> lock(e->mu);
    return e->completed >= e->recorded ? hipSuccess : ret(hipErrorNotReady);
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    if (inject("hipEventSynchronize")) return ret(hipErrorUnknown);
    if (!live_event(e)) return ret(hipErrorInvalidValue);
    std::unique_lock<std::mutex> lock(e->mu);
    const unsigned long long gen = e->recorded;
    e->cv.wait(lock, [&] { return e->completed >= gen; });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if (inject("hipStreamWaitEvent")) return ret(hipErrorUnknown);
    if (!live_event(e)) return ret(hipErrorInvalidValue);
    unsigned long long gen;
    { std::lock_guard<std::mutex> lock(e->mu); gen = e->recorded; }
    hipstub_enqueue(s, [e, gen] {
        std::unique_lock<std::mutex> lock(e->mu);
        e->cv.wait(lock, [&] { return e->completed >= gen; });
    });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    if (!live_event(a) || !live_event(b)) return ret(hipErrorInvalidValue);
    if (a->device != b->device) return ret(hipErrorInvalidValue);          // the real runtime refuses events of two devices
    std::scoped_lock lock(a->mu, b->mu);
    if (a->completed < a->recorded || b->completed < b->recorded || a->recorded == 0 || b->recorded == 0) return ret(hipErrorNotReady);
    *ms = std::chrono::duration<float, std::milli>(b->stamp - a->stamp).count();
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { const char* e = getenv("HIPSTUB_PEER"); *can = e ? atoi(e) : 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) {
    static std::mutex mu;
    static std::set<std::pair<int, int>> on;
    std::lock_guard<std::mutex> lock(mu);
    return on.insert({t_device, 0}).second ? hipSuccess : ret(hipErrorPeerAccessAlreadyEnabled);
}
