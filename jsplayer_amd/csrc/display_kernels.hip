// The two per-pixel passes of the reference's Manager that sit directly after the codec
// (SURVEY.md §8f-1, §8f-3), as HBM-bound HIP kernels so a decoded frame never has to visit the CPU:
//   display_convert : Manager.fill_bitmap_data (Manager.hx:325-390) — RGB32 0x00RRGGBB -> canvas pixels
//   frames_differ   : the pixel loop of frames_differ_significantly (Manager.hx:413-419)
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/jsplayer_amd.h"
#include "../../include/jsplayer_amd_lab.h"
#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t convert(uint32_t c, int mode) {
    switch (mode) {
        case JSP_DISPLAY_CANVAS: return 0xFF000000u | ((c & 0xFFu) << 16) | (c & 0xFF00u) | ((c >> 16) & 0xFFu);  // :379
        case JSP_DISPLAY_CANVAS_RGB15: return 0xFF000000u | (c << 3);                                           // :370
        case JSP_DISPLAY_SETPIXELS: return 0xFF000000u | c;                                                      // :351
        default: return c << 11;                                                                                  // :340
    }
}

__global__ __launch_bounds__(256) void display_convert_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                              int X, int Y, int mode, int flip, int vec) {
    const int y = blockIdx.y;
    const int ys = flip ? Y - 1 - y : y;
    const uint32_t* s = src + (size_t)ys * X;
    uint32_t* d = dst + (size_t)y * X;
    if (vec) {
        for (int x = (blockIdx.x * 256 + threadIdx.x) * 4; x < X; x += gridDim.x * 256 * 4) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s + x));
            __builtin_nontemporal_store(u32x4{convert(v.x, mode), convert(v.y, mode), convert(v.z, mode), convert(v.w, mode)},
                                        reinterpret_cast<u32x4*>(d + x));
        }
    } else {
        for (int x = blockIdx.x * 256 + threadIdx.x; x < X; x += gridDim.x * 256) d[x] = convert(s[x], mode);
    }
}

__global__ __launch_bounds__(256) void frames_differ_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                            size_t first, size_t n, uint32_t* __restrict__ flag) {
    bool diff = false;
    // 16-byte body when both pointers allow it, scalar head/tail otherwise
    const size_t stride = (size_t)gridDim.x * 256;
    const bool vec_ok = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    size_t lo = first, hi = n;
    if (vec_ok) {
        const size_t lo4 = (first + 3) & ~size_t(3), hi4 = n & ~size_t(3);
        if (lo4 < hi4) {
            for (size_t i = lo4 + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < hi4; i += stride * 4) {
                const u32x4 p = *reinterpret_cast<const u32x4*>(a + i), q = *reinterpret_cast<const u32x4*>(b + i);
                diff |= (p.x != q.x) | (p.y != q.y) | (p.z != q.z) | (p.w != q.w);
            }
            for (size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x; i < lo4; i += stride) diff |= a[i] != b[i];
            lo = hi4;
        }
    }
    for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) diff |= a[i] != b[i];
    if (__ballot(diff) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// What the memory system takes when it is asked for nothing but stores in the friendliest shape this chip has been found to
// have (profiles/archive/r03_sp_store_lab.txt): 16 bytes per lane, ONE store per lane, workgroups of 256 lanes handed out in address order.
typedef uint32_t fill_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ceiling_fill_kernel(fill_u32x4* __restrict__ dst, size_t n, uint32_t v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = fill_u32x4{v, v + 1u, v + 2u, v + 3u};
}

// What a frame pool's memory takes from the decode kernels' store shape: a workgroup per (frame, run of `T` 4x4 blocks), handed out
// tile-major — run j of every frame, then run j + 1: as many write fronts as the pool has frames —, lane = block, four 16-byte row
// stores per block.  Physical memory differs in what it takes from this shape by a quarter (jsp_pool_create probes with it).
__global__ __launch_bounds__(256) void pool_probe_kernel(uint32_t* const* __restrict__ frames, int nframes, int X, int nbx, int nblocks, int T, uint32_t v) {
    const int j = blockIdx.x / nframes, f = blockIdx.x - j * nframes;
    uint32_t* dst = frames[f];
    for (int r = 0; r < T; r += 256) {
        const int blk = j * T + r + (int)threadIdx.x;
        if (blk < nblocks) {
            const int by = blk / nbx, bx = blk - by * nbx;
            uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
#pragma unroll
            for (int y = 0; y < 4; ++y) *reinterpret_cast<fill_u32x4*>(p + (size_t)y * X) = fill_u32x4{v, v, v, v};
        }
    }
}

}  // namespace

namespace jsp {
// The pixel loop of frames_differ_significantly (Manager.hx:413-419), queued: *d_flag (zeroed here, on the stream) is OR-ed with 1 when
// a[i] != b[i] for some first_pixel <= i < npixels.  Asynchronous on `stream`.
void launch_frames_differ(const int32_t* a, const int32_t* b, size_t first_pixel, size_t npixels, uint32_t* d_flag, hipStream_t stream) {
    JSP_HIP(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), stream));
    if (first_pixel >= npixels) return;
    const size_t count = npixels - first_pixel;
    const int grid = (int)std::min<size_t>((count / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(frames_differ_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(a), reinterpret_cast<const uint32_t*>(b),
                       first_pixel, npixels, d_flag);
}
// GB/s of pool_probe_kernel over the `nframes` frames of X x Y pixels whose addresses are in the device table `d_frames` (X, Y
// multiples of 4, 16-byte aligned frames); leaves the frames filled with `fill`.  Synchronous.
double pool_store_rate(uint32_t* const* d_frames, int nframes, int X, int Y, uint32_t fill) {
    const int nbx = X / 4, nblocks = nbx * (Y / 4), T = 8192, runs = (nblocks + T - 1) / T;
    hipEvent_t e0, e1;
    JSP_HIP(hipEventCreate(&e0));
    JSP_HIP(hipEventCreate(&e1));
    float best = 0;
    for (int pass = 0; pass < 3; ++pass) {
        JSP_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(pool_probe_kernel, dim3((unsigned)(runs * nframes)), dim3(256), 0, nullptr, d_frames, nframes, X, nbx, nblocks, T, fill);
        JSP_HIP(hipEventRecord(e1, nullptr));
        JSP_HIP(hipEventSynchronize(e1));
        float ms = 0;
        JSP_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 1 || (pass == 2 && ms < best)) best = ms;        // (the first pass faults the pages in)
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    JSP_HIP(hipGetLastError());
    return (double)nblocks * 64.0 * nframes / ((double)best * 1e6);
}
// GB/s of the plain fill (one 16-byte store per lane, workgroups in address order) over the same memory: the yardstick.
double pool_fill_rate(uint32_t* slab, size_t nbytes) {
    if (nbytes > ((size_t)8 << 30)) nbytes = (size_t)8 << 30;          // (a launch has fewer than 2^32 lanes; 8 GiB say as much as 80)
    const size_t n = nbytes / 16;
    hipEvent_t e0, e1;
    JSP_HIP(hipEventCreate(&e0));
    JSP_HIP(hipEventCreate(&e1));
    float best = 0;
    for (int pass = 0; pass < 3; ++pass) {
        JSP_HIP(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(ceiling_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, reinterpret_cast<fill_u32x4*>(slab), n, 0u);
        JSP_HIP(hipEventRecord(e1, nullptr));
        JSP_HIP(hipEventSynchronize(e1));
        float ms = 0;
        JSP_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 1 || (pass == 2 && ms < best)) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return (double)n * 16.0 / ((double)best * 1e6);
}
}  // namespace jsp

extern "C" {

int jsp_measure_fill(int32_t* device, size_t nbytes, int reps, double* gbytes_per_s, void* hip_stream) {
    try {
        if (!device || !gbytes_per_s || nbytes < 4096 || reps < 1 || (reinterpret_cast<uintptr_t>(device) & 15)) throw std::runtime_error("bad argument");
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        const size_t n = nbytes / 16;
        if (n >= (1ull << 32)) throw std::runtime_error("buffer too large for one launch (64 GiB or more)");
        const dim3 grid((unsigned)((n + 255) / 256));
        hipEvent_t e0, e1;
        JSP_HIP(hipEventCreate(&e0));
        JSP_HIP(hipEventCreate(&e1));
        float best = 0;
        for (int pass = 0; pass < 3; ++pass) {                   // (the first pass also warms the launch path up)
            JSP_HIP(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                hipLaunchKernelGGL(ceiling_fill_kernel, grid, dim3(256), 0, s, reinterpret_cast<fill_u32x4*>(device), n, (uint32_t)(pass * 16 + r));
            JSP_HIP(hipEventRecord(e1, s));
            JSP_HIP(hipEventSynchronize(e1));
            float ms = 0;
            JSP_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 0 || ms < best) best = ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        JSP_HIP(hipGetLastError());
        *gbytes_per_s = (double)n * 16.0 * reps / ((double)best * 1e6);
        return 0;
    } catch (const std::exception& e) {
        jsp::set_error("%s", e.what());
        return JSP_ERROR_OCCURED;
    }
}


int jsp_display_convert(const int32_t* frame, int32_t* out, int width, int height, int mode, int flip_rows,
                        void* hip_stream) {
    try {
        if (!frame || !out || width <= 0 || height <= 0 || mode < 0 || mode > 3) throw std::runtime_error("bad argument");
        const int vec = ((width & 3) == 0 && (((uintptr_t)frame | (uintptr_t)out) & 15) == 0) ? 1 : 0;
        int gx = (width / (vec ? 4 : 1) + 255) / 256;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(display_convert_kernel, dim3(gx, height), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                           reinterpret_cast<const uint32_t*>(frame), reinterpret_cast<uint32_t*>(out), width, height, mode,
                           flip_rows ? 1 : 0, vec);
        JSP_HIP(hipGetLastError());
        return 0;
    } catch (const std::exception& e) {
        jsp::set_error("%s", e.what());
        return JSP_ERROR_OCCURED;
    }
}

int jsp_frames_differ(const int32_t* a, const int32_t* b, size_t first_pixel, size_t npixels, int* differ,
                      void* hip_stream) {
    try {
        if (!a || !b || !differ) throw std::runtime_error("null argument");
        *differ = 0;
        if (first_pixel >= npixels) return 0;
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        // one result word per host thread and device, allocated once (the call is synchronous)
        struct Scratch { int device = -1; uint32_t* word = nullptr; };
        thread_local Scratch scratch;
        int dev = 0;
        JSP_HIP(hipGetDevice(&dev));
        if (scratch.device != dev) {
            scratch.word = nullptr;
            JSP_HIP(hipMalloc(reinterpret_cast<void**>(&scratch.word), sizeof(uint32_t)));
            scratch.device = dev;
        }
        uint32_t* d_flag = scratch.word;
        JSP_HIP(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), s));
        const size_t count = npixels - first_pixel;
        int grid = (int)std::min<size_t>((count / 4 + 255) / 256 + 1, 2048);
        hipLaunchKernelGGL(frames_differ_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(a),
                           reinterpret_cast<const uint32_t*>(b), first_pixel, npixels, d_flag);
        uint32_t h = 0;
        JSP_HIP(hipMemcpyAsync(&h, d_flag, sizeof h, hipMemcpyDeviceToHost, s));
        JSP_HIP(hipStreamSynchronize(s));
        *differ = h ? 1 : 0;
        return 0;
    } catch (const std::exception& e) {
        jsp::set_error("%s", e.what());
        return JSP_ERROR_OCCURED;
    }
}
}
