// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): the HIP runtime the ThreadSanitizer build of the host layers links against.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

struct StubStream { std::atomic<unsigned long long> seq{0}; };
struct StubEvent { std::atomic<unsigned long long> seq{0}; };

namespace {
StubStream g_null_stream;                       // the legacy default stream: blocking streams are ordered with it
std::atomic<unsigned long long> g_device{0};    // hipDeviceSynchronize
thread_local int t_device = 0;
std::mutex g_mem_mutex;
std::map<const char*, std::pair<size_t, hipMemoryType>> g_mem;   // allocation base -> (bytes, kind)

StubStream* S(hipStream_t s) { return s ? s : &g_null_stream; }
void queued(hipStream_t s) {       // an operation enters stream s: ordered after what the stream held (acquire) and visible to its later waiters (release)
    S(s)->seq.fetch_add(1, std::memory_order_acq_rel);
    g_null_stream.seq.fetch_add(1, std::memory_order_acq_rel);       // (every stream of the product is a blocking stream or waited for explicitly: the null stream sees all)
    g_device.fetch_add(1, std::memory_order_acq_rel);
}
void waited(hipStream_t s) { (void)S(s)->seq.load(std::memory_order_acquire); }

hipError_t alloc(void** p, size_t n, hipMemoryType kind) {
    if (!p) return hipErrorInvalidValue;
    void* q = nullptr;
    if (posix_memalign(&q, 4096, n ? n : 1) != 0) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    g_mem[static_cast<const char*>(q)] = {n ? n : 1, kind};
    *p = q;
    return hipSuccess;
}
hipError_t release(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mem_mutex);
        auto it = g_mem.find(static_cast<const char*>(p));
        if (it == g_mem.end()) return hipErrorInvalidValue;
        g_mem.erase(it);
    }
    std::free(p);
    return hipSuccess;
}
}  // namespace

void stub_stream_work(hipStream_t stream) { queued(stream); }

struct StubHandle { size_t bytes; };
hipError_t hipMemGetAllocationGranularity(size_t* g, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) { *g = 4096; return hipSuccess; }
hipError_t hipMemAddressReserve(void** p, size_t n, size_t, void*, unsigned long long) { return alloc(p, n, hipMemoryTypeDevice); }
hipError_t hipMemAddressFree(void* p, size_t) { return release(p); }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t n, const hipMemAllocationProp*, unsigned long long) { *h = new StubHandle{n}; return hipSuccess; }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { delete h; return hipSuccess; }
hipError_t hipMemMap(void*, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return hipSuccess; }
hipError_t hipMemUnmap(void*, size_t) { return hipSuccess; }
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t) { return hipSuccess; }

hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub HIP error"; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidValue; t_device = d; return hipSuccess; }
hipError_t hipDeviceSynchronize() { (void)g_device.load(std::memory_order_acquire); waited(nullptr); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { return alloc(p, n, hipMemoryTypeDevice); }
hipError_t hipFree(void* p) { return release(p); }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return alloc(p, n, hipMemoryTypeHost); }
hipError_t hipHostFree(void* p) { return release(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { waited(nullptr); if (n) std::memmove(d, s, n); queued(nullptr); waited(nullptr); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { waited(st); if (n) std::memmove(d, s, n); queued(st); return hipSuccess; }
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
    waited(nullptr);
    for (size_t y = 0; y < h; ++y) std::memmove(static_cast<char*>(d) + y * dp, static_cast<const char*>(s) + y * sp, w);
    queued(nullptr);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { waited(nullptr); if (n) std::memset(d, v, n); queued(nullptr); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { waited(st); if (n) std::memset(d, v, n); queued(st); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new StubStream; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { waited(s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { (void)e->seq.load(std::memory_order_acquire); queued(s); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new StubEvent; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { waited(s); e->seq.fetch_add(1, std::memory_order_acq_rel); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e->seq.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { (void)e->seq.load(std::memory_order_acquire); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    auto it = g_mem.upper_bound(static_cast<const char*>(p));
    if (it != g_mem.begin()) {
        --it;
        if (static_cast<const char*>(p) < it->first + it->second.first) {
            a->type = it->second.second;
            a->device = 0;
            a->devicePointer = const_cast<void*>(p);
            a->hostPointer = it->second.second == hipMemoryTypeHost ? const_cast<void*>(p) : nullptr;
            a->isManaged = 0;
            a->allocationFlags = 0;
            return hipSuccess;
        }
    }
    return hipErrorInvalidValue;                 // pageable host memory: what the real runtime says too
}
