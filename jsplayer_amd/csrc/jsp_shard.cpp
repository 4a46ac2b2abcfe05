// Stream-per-GPU sharding for a native caller (the caller side of Manager.hx:97-142: one Manager + one decoder per
// stream).  Independent AVI streams go one per device — stream i -> devices[i mod G], no frame ever crosses xGMI — and the only
// collective is the sum of the per-device (frames, pixels) counters: an all-reduce over RCCL when librccl can be loaded and the
// devices are distinct, the same sum on the host otherwise.  RCCL is looked up at run time (dlopen): the library does not depend
// on it.  The Python side of the same thing is jsplayer_amd/sharding.py (one process per GPU under torch.distributed).
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "codec.h"
#include "../../include/jsplayer_amd_lab.h"

// The few RCCL types and constants the counter reduce needs, spelled out here (values as in rccl/rccl.h of ROCm 7: part of the NCCL
// ABI): every entry point is taken from dlsym, and a ROCm install without the RCCL headers still builds the library.
typedef struct ncclComm* ncclComm_t;
enum ncclResult_t : int { ncclSuccess = 0 };
enum ncclDataType_t : int { ncclUint64 = 5 };
enum ncclRedOp_t : int { ncclSum = 0 };

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string tried;
    Rccl() {
        // JSP_RCCL_LIB: the library to load instead of the usual names (an install that keeps it elsewhere; tests point it at nothing)
        const char* named = std::getenv("JSP_RCCL_LIB");
        const std::vector<const char*> names = named && *named ? std::vector<const char*>{named}
                                                                : std::vector<const char*>{"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (so) break;
            tried += (tried.empty() ? "" : ", ") + std::string(name);
        }
        if (!so) return;
        auto sym = [&](const char* n) { return dlsym(so, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce;
    }
};
Rccl& rccl() {
    static Rccl r;     // (never unloaded: communicators may outlive any one call)
    return r;
}

std::string& shard_error() {
    thread_local std::string e;
    return e;
}

// The all-reduce proper: one communicator per device of this process, every device contributes its two counters and every device
// receives the sums; device 0's copy is handed back, all copies must agree.
bool reduce_over_rccl(const int* devices, int ndev, const uint64_t* per_device, uint64_t* total) {
    Rccl& r = rccl();
    if (!r.ok) { shard_error() = r.so ? "librccl lacks an entry point the counter reduce needs" : "librccl not loadable (tried " + r.tried + ")"; return false; }
    std::vector<ncclComm_t> comms(ndev, nullptr);
    std::vector<hipStream_t> streams(ndev, nullptr);
    std::vector<uint64_t*> bufs(ndev, nullptr);
    bool good = true;
    auto check = [&](ncclResult_t rc, const char* what) {
        if (rc == ncclSuccess) return true;
        shard_error() = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "rccl error");
        return false;
    };
    int before = 0;
    (void)hipGetDevice(&before);
    good = check(r.CommInitAll(comms.data(), ndev, devices), "ncclCommInitAll");
    for (int i = 0; i < ndev && good; ++i) {
        good = hipSetDevice(devices[i]) == hipSuccess && hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) == hipSuccess &&
               hipMalloc(reinterpret_cast<void**>(&bufs[i]), 4 * sizeof(uint64_t)) == hipSuccess &&
               hipMemcpyAsync(bufs[i], per_device + 2 * i, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, streams[i]) == hipSuccess;
        if (!good) shard_error() = "HIP error while setting up the counter reduce";
    }
    if (good) {
        good = check(r.GroupStart(), "ncclGroupStart");
        for (int i = 0; i < ndev && good; ++i)
            good = check(r.AllReduce(bufs[i], bufs[i] + 2, 2, ncclUint64, ncclSum, comms[i], streams[i]), "ncclAllReduce");
        good = check(r.GroupEnd(), "ncclGroupEnd") && good;
    }
    std::vector<uint64_t> got((size_t)ndev * 2, 0);
    for (int i = 0; i < ndev && good; ++i) {
        good = hipSetDevice(devices[i]) == hipSuccess &&
               hipMemcpyAsync(&got[2 * i], bufs[i] + 2, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, streams[i]) == hipSuccess &&
               hipStreamSynchronize(streams[i]) == hipSuccess;
        if (!good) shard_error() = "HIP error while reading the reduced counters back";
    }
    for (int i = 0; i < ndev && good; ++i)
        if (got[2 * i] != got[0] || got[2 * i + 1] != got[1]) { good = false; shard_error() = "the devices disagree about the reduced counters"; }
    if (good) { total[0] = got[0]; total[1] = got[1]; }
    for (int i = 0; i < ndev; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) continue;
        if (bufs[i]) (void)hipFree(bufs[i]);
        if (streams[i]) (void)hipStreamDestroy(streams[i]);
        if (comms[i]) (void)r.CommDestroy(comms[i]);
    }
    (void)hipSetDevice(before);
    (void)hipGetLastError();
    return good;
}

}  // namespace

extern "C" {

int jsp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int jsp_assign_stream(int stream_index, const int* devices, int ndev) {
    if (!devices || ndev <= 0 || stream_index < 0) return -1;
    return devices[stream_index % ndev];
}

int jsp_reduce_counters(const int* devices, int ndev, const uint64_t* per_device, uint64_t* total, int* via_rccl) {
    if (via_rccl) *via_rccl = 0;
    if (!devices || ndev <= 0 || !per_device || !total) return JSP_ERROR_OCCURED;
    uint64_t sum[2] = {0, 0};
    for (int i = 0; i < ndev; ++i) { sum[0] += per_device[2 * i]; sum[1] += per_device[2 * i + 1]; }
    // RCCL wants one rank per device: a device list with repeats (several streams per GPU) is summed per device first
    std::vector<int> distinct;
    std::vector<uint64_t> folded;
    for (int i = 0; i < ndev; ++i) {
        size_t k = 0;
        while (k < distinct.size() && distinct[k] != devices[i]) ++k;
        if (k == distinct.size()) { distinct.push_back(devices[i]); folded.push_back(0); folded.push_back(0); }
        folded[2 * k] += per_device[2 * i];
        folded[2 * k + 1] += per_device[2 * i + 1];
    }
    // (JSP_SHARD_ASSUME_DEVICES: tests of the fall-back on a box without that many devices — the RCCL attempt then fails where it fails)
    const char* assume = std::getenv("JSP_SHARD_ASSUME_DEVICES");
    const int have = assume && *assume ? std::atoi(assume) : jsp_device_count();
    bool usable = have > 0;
    shard_error().clear();
    if (!usable) shard_error() = "no HIP device visible: counters summed on the host";
    for (int d : distinct)
        if (usable && (d < 0 || d >= have)) { usable = false; shard_error() = "device ordinal " + std::to_string(d) + " is not one of the " + std::to_string(have) + " visible: counters summed on the host"; }
    uint64_t reduced[2] = {0, 0};
    if (usable && reduce_over_rccl(distinct.data(), (int)distinct.size(), folded.data(), reduced)) {
        if (reduced[0] != sum[0] || reduced[1] != sum[1]) return JSP_ERROR_OCCURED;   // (a collective that loses counts is an error, not a fallback)
        if (via_rccl) *via_rccl = 1;
    }
    total[0] = sum[0];
    total[1] = sum[1];
    return JSP_ZERO_STATE;
}

const char* jsp_shard_last_error(void) { return shard_error().c_str(); }

// What the bus of this box delivers when asked for nothing else: `copies` host-to-device copies of `bytes_per_copy` from pinned
// memory on each of `nstreams` HIP streams of `device_id` side by side, wall clock from the first submission to the last stream's
// idle; best of three passes.  What the end-to-end decode rates are to be held against (compressed bytes are all that crosses).
int jsp_measure_h2d(int device_id, size_t bytes_per_copy, int nstreams, int copies, double* gbytes_per_s) {
    if (!gbytes_per_s || bytes_per_copy == 0 || nstreams < 1 || nstreams > 16 || copies < 1) return JSP_ERROR_OCCURED;
    int before = 0;
    (void)hipGetDevice(&before);
    if (hipSetDevice(device_id) != hipSuccess) { (void)hipGetLastError(); return JSP_ERROR_OCCURED; }
    std::vector<hipStream_t> streams(nstreams, nullptr);
    std::vector<void*> host(nstreams, nullptr), dev(nstreams, nullptr);
    bool good = true;
    for (int i = 0; i < nstreams && good; ++i)
        good = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) == hipSuccess && hipHostMalloc(&host[i], bytes_per_copy, hipHostMallocDefault) == hipSuccess &&
               hipMalloc(&dev[i], bytes_per_copy) == hipSuccess;
    double best = 0;
    if (good) {
        for (int i = 0; i < nstreams; ++i) std::memset(host[i], 0x5A, bytes_per_copy);      // (touched: no first-use page faults inside the timed copies)
        for (int pass = 0; pass < 4 && good; ++pass) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int c = 0; c < copies && good; ++c)
                for (int i = 0; i < nstreams && good; ++i)
                    good = hipMemcpyAsync(dev[i], host[i], bytes_per_copy, hipMemcpyHostToDevice, streams[i]) == hipSuccess;
            for (int i = 0; i < nstreams && good; ++i) good = hipStreamSynchronize(streams[i]) == hipSuccess;
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double rate = (double)bytes_per_copy * copies * nstreams / s / 1e9;
            if (pass > 0 && rate > best) best = rate;                                        // (the first pass warms the path up)
        }
    }
    for (int i = 0; i < nstreams; ++i) {
        if (dev[i]) (void)hipFree(dev[i]);
        if (host[i]) (void)hipHostFree(host[i]);
        if (streams[i]) (void)hipStreamDestroy(streams[i]);
    }
    (void)hipSetDevice(before);
    (void)hipGetLastError();
    if (!good) return JSP_ERROR_OCCURED;
    *gbytes_per_s = best;
    return JSP_ZERO_STATE;
}

}  // extern "C"
