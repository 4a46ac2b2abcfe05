// Shared helpers for the jsplayer_amd native library (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <sched.h>

#include <chrono>
#include <cstdlib>
#include <thread>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace jsp {

// Thread-local last-error text surfaced by jsp_last_error().
std::string& last_error_slot();
void set_error(const char* fmt, ...);

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define JSP_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            char _buf[512];                                                                    \
            std::snprintf(_buf, sizeof _buf, "%s failed: %s (%s:%d)", #expr,                   \
                          hipGetErrorString(_e), __FILE__, __LINE__);                          \
            throw ::jsp::HipError(_buf);                                                       \
        }                                                                                      \
    } while (0)

// Host threads this process may keep busy: the machine's, narrowed by the affinity mask and by the cgroup's CPU quota (a GPU
// box shows all 256 threads of its host to a container that may use 16 of them: worker pools sized by
// hardware_concurrency() there spend their time being throttled).  Read once.
inline int usable_cpus() {
    static const int n = [] {
        int c = (int)std::thread::hardware_concurrency();
        if (c < 1) c = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1 && a < c) c = a; }
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            long long period = 0;
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
                const long long quota = std::atoll(q);
                if (quota > 0) { const int k = (int)((quota + period - 1) / period); if (k >= 1 && k < c) c = k; }
            }
            std::fclose(f);
        } else {                                                                   // cgroup v1
            long long quota = -1, period = 0;
            if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lld", &quota) != 1) quota = -1; std::fclose(fq); }
            if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%lld", &period) != 1) period = 0; std::fclose(fp); }
            if (quota > 0 && period > 0) { const int k = (int)((quota + period - 1) / period); if (k >= 1 && k < c) c = k; }
        }
        return c;
    }();
    return n;
}

inline double now_ms() {
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

// Growable device buffer (never shrinks).
struct DeviceBuffer {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) JSP_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 256;
        JSP_HIP(hipMalloc(&p, want));
        cap = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
};

// Growable pinned host buffer.
struct PinnedBuffer {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) JSP_HIP(hipHostFree(p));
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 256;
        JSP_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    ~PinnedBuffer() { release(); }
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
};

}  // namespace jsp
