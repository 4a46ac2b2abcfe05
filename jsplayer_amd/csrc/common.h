// Shared helpers for the jsplayer_amd native library (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
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
