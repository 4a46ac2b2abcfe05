// TEST INFRASTRUCTURE, not product: a stand-in for <hip/hip_runtime.h> so that the product's threaded HOST layers (jsp_api.cpp, msv1_codec.cpp,
// sp_codec.cpp, jsp_shard.cpp and the host stages) can be built with g++ -fsanitize=thread and driven without a GPU (GPU sanitizers are not
// available on this pool; tools/tsan_cpu.sh).  "Device memory" is host memory, every stream operation runs at the call, kernels are no-ops
// (tests/tsan/kernel_stubs.cpp) — what is under test is the host threads' synchronisation with EACH OTHER, not pixels.  What a real runtime
// orders through streams and events is given to ThreadSanitizer as release / acquire pairs on one atomic per stream / event (hip_stub.cpp):
// work queued on a stream happens-before the return of a wait on that stream or on an event recorded behind it.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
struct StubStream;
struct StubEvent;
typedef StubStream* hipStream_t;
typedef StubEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t {
    hipMemoryType type;
    int device;
    void* devicePointer;
    void* hostPointer;
    int isManaged;
    unsigned allocationFlags;
};
constexpr unsigned hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;

hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t);
hipError_t hipGetDeviceCount(int*);
hipError_t hipGetDevice(int*);
hipError_t hipSetDevice(int);
hipError_t hipDeviceSynchronize();
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
hipError_t hipMalloc(void**, size_t);
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned flags = 0);
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, flags); }
hipError_t hipHostFree(void*);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t stream = nullptr);
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t stream = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned flags = 0);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t stream = nullptr);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventQuery(hipEvent_t);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*);

// virtual memory management (the frame pool's first form): the reserved range IS the memory, handles are tokens
struct StubHandle;
typedef StubHandle* hipMemGenericAllocationHandle_t;
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAllocationGranularity_flags { hipMemAllocationGranularityMinimum = 0, hipMemAllocationGranularityRecommended = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; int requestedHandleType; hipMemLocation location; void* win32HandleMetaData; struct { unsigned char compressionType, gpuDirectRDMACapable; unsigned short usage; } allocFlags; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };
hipError_t hipMemGetAllocationGranularity(size_t*, const hipMemAllocationProp*, hipMemAllocationGranularity_flags);
hipError_t hipMemAddressReserve(void**, size_t, size_t alignment, void* addr, unsigned long long flags);
hipError_t hipMemAddressFree(void*, size_t);
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t*, size_t, const hipMemAllocationProp*, unsigned long long flags);
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t);
hipError_t hipMemMap(void*, size_t, size_t offset, hipMemGenericAllocationHandle_t, unsigned long long flags);
hipError_t hipMemUnmap(void*, size_t);
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t count);

// what the kernel stubs call: "a kernel was queued on this stream"
void stub_stream_work(hipStream_t stream);
