// lab: what store rate does a kernel SHAPED like msv1_fused_kernel reach, with the parse replaced by a delay?
// One workgroup (256 lanes) per "tile": (A) load the tile's input (8 bytes per block, coalesced 16-byte loads) and wait for
// it, (B) spin for C cycles (stands for the parse), (C) rounds of 256 blocks: D dependent VALU instructions, then the block's
// four 16-byte row stores (lane = block, raster order over 1920x1080 RGB32 frames), optionally throttled with s_waitcnt vmcnt.
// Tiles are taken frame-major or tile-major; occupancy is set with dynamic LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/store_lab.hip -o /tmp/store_lab && /tmp/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int X = 1920, Y = 1080, NBX = X / 4, NBLK = (X / 4) * (Y / 4);

template <int VM>
__device__ __forceinline__ void throttle() { if (VM < 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory"); }

// T = blocks per tile (multiple of 256)
template <int VM, bool NT>
__global__ __launch_bounds__(256) void tile_kernel(uint32_t* __restrict__ out, const uint8_t* __restrict__ in, int T, int tiles_per_frame,
                                                   int nframes, int tile_major, int C, int D, uint32_t* __restrict__ sink) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x;
    int f, j;
    if (tile_major) { j = blockIdx.x / nframes; f = blockIdx.x - j * nframes; }
    else { f = blockIdx.x / tiles_per_frame; j = blockIdx.x - f * tiles_per_frame; }
    // (A) the tile's input: T * 8 bytes
    const uint8_t* src = in + ((size_t)f * tiles_per_frame + j) * (size_t)T * 8;
    uint32_t acc = 0;
    for (int o = tid * 16; o < T * 8; o += 256 * 16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + o);
        lds[(o / 4) & 4095] = v.x ^ v.y ^ v.z ^ v.w;
        acc ^= v.x;
    }
    __syncthreads();
    // (B) the parse: C cycles of waiting (s_sleep: nothing issued) or, C < 0, -C dependent VALU instructions
    if (C > 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)C) __builtin_amdgcn_s_sleep(4);
    } else {
        for (int i = 0; i < -C; ++i) acc = acc * 1664525u + 1013904223u;
    }
    // (C) rounds of 256 blocks
    uint32_t* dst = out + (size_t)f * X * Y;
    const int b0 = j * T;
    for (int r = 0; r < T; r += 256) {
        const int blk = b0 + r + tid;
        uint32_t a = acc + lds[(r + tid) & 4095];
        for (int i = 0; i < D; ++i) a = a * 1664525u + 1013904223u;
        if (blk < NBLK) {
            const int by = blk / NBX, bx = blk - by * NBX;
            uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
            throttle<VM>();
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const u32x4 x = u32x4{a, a + 1, a + 2, a + (uint32_t)y};
                if (NT) __builtin_nontemporal_store(x, (gu32x4*)(p + (size_t)y * X));
                else *(gu32x4*)(p + (size_t)y * X) = x;
            }
        }
        acc = a;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// Persistent form: grid = resident workgroups, tiles taken with stride gridDim; the NEXT tile's input is requested before this
// tile's stores go out (PF = 1) or after them (PF = 0); NOLOAD: no input at all.
template <int VM, int PF>
__global__ __launch_bounds__(256) void persist_kernel(uint32_t* __restrict__ out, const uint8_t* __restrict__ in, int T, int tiles_per_frame,
                                                      int nframes, int ntiles, int C, int D, uint32_t* __restrict__ sink) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    u32x4 pre[4];                                        // T * 8 <= 16 KB: up to 4 loads per lane
    auto request = [&](int tile) {
        const int j = tile / nframes, f = tile - j * nframes;   // tile-major
        const uint8_t* src = in + ((size_t)f * tiles_per_frame + j) * (size_t)T * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (tid * 16 + q * 4096 < T * 8) pre[q] = *reinterpret_cast<const u32x4*>(src + tid * 16 + q * 4096);
    };
    if (PF != 2) request(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int j = tile / nframes, f = tile - j * nframes;
        if (PF != 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (tid * 16 + q * 4096 < T * 8) { lds[(tid * 4 + q * 1024) & 4095] = pre[q].x ^ pre[q].y ^ pre[q].z ^ pre[q].w; acc ^= pre[q].x; }
        }
        __syncthreads();
        if (C > 0) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < (unsigned long long)C) __builtin_amdgcn_s_sleep(4);
        } else {
            for (int i = 0; i < -C; ++i) acc = acc * 1664525u + 1013904223u;
        }
        if (PF == 1 && tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
        uint32_t* dst = out + (size_t)f * X * Y;
        const int b0 = j * T;
        for (int r = 0; r < T; r += 256) {
            const int blk = b0 + r + tid;
            uint32_t a = acc + lds[(r + tid) & 4095];
            for (int i = 0; i < D; ++i) a = a * 1664525u + 1013904223u;
            if (blk < NBLK) {
                const int by = blk / NBX, bx = blk - by * NBX;
                uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
                throttle<VM>();
#pragma unroll
                for (int y = 0; y < 4; ++y) __builtin_nontemporal_store(u32x4{a, a + 1, a + 2, a + (uint32_t)y}, (gu32x4*)(p + (size_t)y * X));
            }
            acc = a;
        }
        if (PF == 0 && tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static double time_us(F&& launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 512;
    uint32_t *out, *sink;
    uint8_t* in;
    const size_t out_bytes = (size_t)F * X * Y * 4, in_bytes = (size_t)F * (NBLK + 4096) * 8;
    CK(hipMalloc(&out, out_bytes));
    CK(hipMalloc(&in, in_bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, in_bytes));
    CK(hipMemset(out, 0, out_bytes));
    CK(hipFuncSetAttribute((const void*)tile_kernel<63, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)tile_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)tile_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)tile_kernel<63, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    printf("%d frames 1920x1080: %.0f MB written + %.0f MB read per launch\n", F, out_bytes / 1e6, (double)F * NBLK * 8 / 1e6);
    printf("%6s %4s %3s %3s %7s %5s %3s | %9s %9s\n", "T", "wg", "tm", "vm", "C", "D", "nt", "us", "GB/s");
    auto run = [&](int T, int wgs_per_cu, int tile_major, int vm, int C, int D, int nt) {
        const int tpf = (NBLK + T - 1) / T;
        const size_t lds = wgs_per_cu >= 8 ? 16384 : (size_t)(160 * 1024 / wgs_per_cu - 1024) & ~(size_t)255;
        const int grid = tpf * F;
        auto launch = [&] {
            if (!nt) hipLaunchKernelGGL((tile_kernel<63, false>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, tile_major, C, D, sink);
            else if (vm == 0) hipLaunchKernelGGL((tile_kernel<0, true>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, tile_major, C, D, sink);
            else if (vm == 4) hipLaunchKernelGGL((tile_kernel<4, true>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, tile_major, C, D, sink);
            else hipLaunchKernelGGL((tile_kernel<63, true>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, tile_major, C, D, sink);
        };
        const double us = time_us(launch, 5);
        CK(hipGetLastError());
        printf("%6d %4d %3d %3d %7d %5d %3d | %9.1f %9.0f\n", T, wgs_per_cu, tile_major, vm, C, D, nt, us, (out_bytes + (double)F * NBLK * 8) / us / 1e3);
        fflush(stdout);
    };
    auto runp = [&](int T, int wgs_per_cu, int pf, int C, int D) {
        const int tpf = (NBLK + T - 1) / T, ntiles = tpf * F;
        const size_t lds = wgs_per_cu >= 8 ? 16384 : (size_t)(160 * 1024 / wgs_per_cu - 1024) & ~(size_t)255;
        const int grid = 256 * wgs_per_cu;
        auto launch = [&] {
            if (pf == 1) hipLaunchKernelGGL((persist_kernel<63, 1>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, ntiles, C, D, sink);
            else if (pf == 0) hipLaunchKernelGGL((persist_kernel<63, 0>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, ntiles, C, D, sink);
            else hipLaunchKernelGGL((persist_kernel<63, 2>), dim3(grid), dim3(256), lds, 0, out, in, T, tpf, F, ntiles, C, D, sink);
        };
        const double us = time_us(launch, 5);
        CK(hipGetLastError());
        printf("persist T %d wg %d pf %d C %d D %d | %9.1f us %9.0f GB/s\n", T, wgs_per_cu, pf, C, D, us, (out_bytes + (double)F * NBLK * 8) / us / 1e3);
        fflush(stdout);
    };
    CK(hipFuncSetAttribute((const void*)persist_kernel<63, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)persist_kernel<63, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)persist_kernel<63, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    if (argc > 2) {
        for (int C : {10000, 20000, 40000})
            for (int wg : {4, 5, 8})
                for (int pf : {0, 1, 2}) runp(2048, wg, pf, C, 100);
        for (int wg : {4, 8}) for (int pf : {0, 1, 2}) runp(2048, wg, pf, -1000, 100);
        for (int wg : {4, 8}) for (int pf : {0, 1, 2}) runp(1024, wg, pf, 10000, 100);
        return 0;
    }
    // 1. pure shape: no parse, no decode work; tile size and occupancy
    for (int T : {256, 512, 1024, 2048, 4096})
        for (int wg : {4, 8}) run(T, wg, 0, 63, 0, 0, 1);
    run(2048, 4, 0, 63, 0, 0, 0);
    run(2048, 4, 1, 63, 0, 0, 1);
    // 2. the decode's arithmetic between the stores
    for (int D : {50, 100, 200}) { run(2048, 4, 0, 63, 0, D, 1); run(256, 8, 0, 63, 0, D, 1); }
    // 3. a parse-like wait in front of the stores (cycles of sleep), by occupancy and throttle
    for (int C : {10000, 20000, 40000})
        for (int wg : {4, 5, 6, 8}) run(2048, wg, 0, 63, C, 100, 1);
    for (int C : {10000, 20000, 40000}) { run(2048, 4, 0, 4, C, 100, 1); run(2048, 4, 0, 0, C, 100, 1); run(2048, 4, 1, 63, C, 100, 1); }
    // 4. the parse as VALU work instead of sleep (dependent instructions per lane)
    for (int V : {1000, 2000, 4000}) for (int wg : {4, 6, 8}) run(2048, wg, 0, 63, -V, 100, 1);
    // 5. smaller tiles with the same total work per block
    for (int T : {512, 1024}) for (int wg : {4, 8}) { run(T, wg, 0, 63, 20000 * T / 2048, 100, 1); run(T, wg, 0, 63, -2000 * T / 2048, 100, 1); }
    return 0;
}
