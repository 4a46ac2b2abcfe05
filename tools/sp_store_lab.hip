// lab: what store rate does a kernel SHAPED like sp_iframe_tile_kernel reach — every wave walks down a band of rows and writes,
// per row, its own piece of the row — as a function of how the pieces are laid out and which waves run together?
//   frames F of 1920x1080 RGB32; a tile = B rows x (S KB of a row); a workgroup = W waves = W tiles side by side in one band;
//   per row a wave issues D dependent VALU instructions (stands for the decode), then S stores of 1 KB (lane = 16 bytes);
//   order 0: blockIdx.x = frame fastest (what launch_iframe_tiles does: grid (frames, tiles)), 1: tile fastest within a frame;
//   L = 1: the W waves of a workgroup meet at a barrier every row (lockstep); occupancy capped with dynamic LDS (bytes per wave).
//   hipcc -O3 --offload-arch=gfx950 tools/sp_store_lab.hip -o /tmp/sp_store_lab && /tmp/sp_store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int Y = 1080;
__constant__ int X = 1920;   // (lab: the row pitch is a parameter)
static int hX = 1920;

template <int S, bool NT>
__global__ void band_kernel(uint32_t* __restrict__ out, int nframes, int B, int tiles_x, int bands, int order, int D, int L,
                            uint32_t* __restrict__ sink) {
    const int W = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const int groups_x = (tiles_x + W - 1) / W, per_frame = groups_x * bands;
    int f, g;
    if (order == 0) { g = blockIdx.x / nframes; f = blockIdx.x - g * nframes; }
    else { f = blockIdx.x / per_frame; g = blockIdx.x - f * per_frame; }
    const int band = g / groups_x, tx = (g - band * groups_x) * W + wave;
    const int y0 = band * B, y1 = y0 + B < Y ? y0 + B : Y;
    const int x0 = tx * (S * 256) + lane * 4;      // first of the lane's 4 pixels in segment 0
    uint32_t* dst = out + (size_t)f * X * Y;
    uint32_t a = (uint32_t)(f * 131 + g * 7 + lane);
    for (int y = y0; y < y1; ++y) {
        for (int i = 0; i < D; ++i) a = a * 1664525u + 1013904223u;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int x = x0 + s * 256;
            if (tx < tiles_x && x < X) {
                const u32x4 v = u32x4{a, a + 1, a + 2, a + (uint32_t)s};
                if (NT) __builtin_nontemporal_store(v, (gu32x4*)(dst + (size_t)y * X + x));
                else *(gu32x4*)(dst + (size_t)y * X + x) = v;
            }
        }
        if (L == 1) __syncthreads();
        else if (L == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (L == 3) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (L == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    if (a == 0x12345u) sink[0] = a;
}

// one wave per 1 KB piece, and that is all it does; the order the pieces are handed out in is
//   order 0: address order;  order 1: the TIME order of the band kernel (all first rows of all bands of all frames, then all second rows ...)
__global__ __launch_bounds__(64) void piece_kernel(uint32_t* __restrict__ out, int nframes, int B, int bands, int order) {
    const size_t p = blockIdx.x;
    int f, y, sx;
    if (order == 0) { sx = (int)(p & 7); y = (int)((p >> 3) % Y); f = (int)((p >> 3) / Y); }
    else {
        sx = (int)(p & 7);
        size_t q = p >> 3;
        const int band = (int)(q % bands); q /= bands;
        f = (int)(q % nframes); q /= nframes;
        y = band * B + (int)q;                       // q = row step
    }
    const int x = sx * 256 + (int)threadIdx.x * 4;
    if (y < Y && x < X && f < nframes) *(gu32x4*)(out + (size_t)f * X * Y + (size_t)y * X + x) = u32x4{(uint32_t)p, 1u, 2u, 3u};
}

// long-lived workgroups sweeping linearly (piece k * grid + blockIdx, 4 KB each), optionally in GLOBAL lockstep: every `sync_every`
// iterations all workgroups meet at a counter (all resident: grid <= 8 per CU).  Does a compact write front matter?
__global__ __launch_bounds__(256) void sweep_kernel(u32x4* __restrict__ dst, size_t n_pieces, int sync_every, unsigned int* __restrict__ counter,
                                                    unsigned int base) {
    unsigned int arrived = base;
    int since = 0;
    for (size_t p = blockIdx.x; p < n_pieces; p += gridDim.x) {
        *(gu32x4*)(dst + p * 256 + threadIdx.x) = u32x4{(uint32_t)p, 1u, 2u, 3u};
        if (sync_every < 0) {                                  // no lockstep, but at most -sync_every - 1 stores of the wave in flight
            if (sync_every == -1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (sync_every == -2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (sync_every == -3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (sync_every == -5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        if (sync_every > 0 && ++since == sync_every) {
            since = 0;
            arrived += gridDim.x;
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < (1 << 16); ++spin) {
                    if ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - arrived) >= 0) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
        }
    }
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 256;
    uint32_t *out, *sink;
    CK(hipMalloc(&out, (size_t)F * 2048 * Y * 4));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double bytes = (double)F * hX * Y * 4;
    printf("%d frames 1920x1080: %.0f MB written per launch\n", F, bytes / 1e6);
    printf("   B  S  W  L order     D  lds/wave nt |        us      GB/s\n");
    auto run = [&](int B, int S, int W, int L, int order, int D, int lds_per_wave, int nt) {
        const int tiles_x = (hX + S * 256 - 1) / (S * 256), bands = (Y + B - 1) / B, groups_x = (tiles_x + W - 1) / W;
        const dim3 grid((unsigned)((size_t)F * groups_x * bands)), block(64 * W);
        const size_t lds = (size_t)lds_per_wave * W;
        auto launch = [&] {
#define GO(SS, NN) hipLaunchKernelGGL((band_kernel<SS, NN>), grid, block, lds, 0, out, F, B, tiles_x, bands, order, D, L, sink)
            if (S == 1) { if (nt) GO(1, true); else GO(1, false); }
            else if (S == 2) { if (nt) GO(2, true); else GO(2, false); }
            else if (S == 4) { if (nt) GO(4, true); else GO(4, false); }
            else { if (nt) GO(8, true); else GO(8, false); }
#undef GO
        };
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 5 < best) best = ms / 5;
        }
        printf("%4d %2d %2d %2d %5d %5d %9d %2d | %9.1f %9.0f\n", B, S, W, L, order, D, lds_per_wave, nt, best * 1000, bytes / best / 1e6);
        fflush(stdout);
    };
    auto run_pieces = [&](int B, int order) {
        const int bands = (Y + B - 1) / B;
        const size_t n = order == 0 ? (size_t)F * Y * 8 : (size_t)B * F * bands * 8;
        auto launch = [&] { hipLaunchKernelGGL(piece_kernel, dim3((unsigned)n), dim3(64), 0, 0, out, F, B, bands, order); };
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 5 < best) best = ms / 5;
        }
        printf("pieces: one wave per 1 KB, B %4d, %s order | %9.1f us %9.0f GB/s\n", B, order ? "band-time" : "address", best * 1000, bytes / best / 1e6);
        fflush(stdout);
    };
    for (int per : {1, 2, 4, 8, 16, 64}) {   // short-lived workgroups in address order: `per` pieces of 4 KB each (sweep_kernel with a grid of n / per)
        const size_t n_pieces = (size_t)F * 1920 * Y * 4 / 4096;
        const unsigned grid = (unsigned)(n_pieces / per);
        auto launch = [&] { hipLaunchKernelGGL(sweep_kernel, dim3(grid), dim3(256), 0, 0, (u32x4*)out, n_pieces, 0, (unsigned int*)sink, 0u); };
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 3 < best) best = ms / 3;
        }
        printf("sweep: %u workgroups handed out in order, %d pieces of 4 KB each (grid-stride) | %9.1f us %9.0f GB/s\n", grid, per, best * 1000,
               (double)n_pieces * 4096 / best / 1e6);
        fflush(stdout);
    }
    {
        unsigned int* counter;
        CK(hipMalloc(&counter, 64));
        CK(hipMemset(counter, 0, 64));
        const size_t n_pieces = (size_t)F * 1920 * Y * 4 / 4096;
        unsigned int base = 0;
        for (int grid : {2025, 8100}) for (int sync_every : {0, -1}) {
            const size_t iters = (n_pieces + grid - 1) / grid;
            auto launch = [&] {
                hipLaunchKernelGGL(sweep_kernel, dim3(grid), dim3(256), 0, 0, (u32x4*)out, n_pieces, sync_every, counter, base);
                if (sync_every > 0) base += (unsigned int)(iters / sync_every) * grid;   // (every workgroup makes the same number of visits: n_pieces is a multiple of the grid)
            };
            if (n_pieces % grid) { printf("pieces not a multiple of the grid\n"); break; }
            launch();
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 3 < best) best = ms / 3;
            }
            printf("sweep: %d long-lived workgroups, 4 KB pieces, stores in flight per wave at most %2d (0: no limit) | %9.1f us %9.0f GB/s\n", grid, -sync_every, best * 1000,
                   (double)n_pieces * 4096 / best / 1e6);
            fflush(stdout);
        }
    }
    // row pitch: do rows a multiple of 64 KB apart written at the same time help?
    for (int xx : {1920, 2048}) {
        hX = xx;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(X), &hX, sizeof(int)));
        bytes = (double)F * hX * Y * 4;
        printf("row pitch %d bytes\n", hX * 4);
        for (int B : {90}) for (int order : {0}) for (int L : {0}) run(B, 1, 1, L, order, 20, 4608, 0);
    }
    hX = 1920;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(X), &hX, sizeof(int)));
    bytes = (double)F * hX * Y * 4;
    return 0;
    run_pieces(90, 0);
    run_pieces(90, 1);
    run_pieces(24, 1);
    run_pieces(1080, 1);
    // short bands: waves that live for a few rows only
    for (int B : {1, 2, 4, 8, 12, 16}) for (int order : {0, 1}) run(B, 1, 1, 0, order, 20, 4608, 0);
    for (int B : {1, 2, 4, 8, 16}) run(B, 1, 8, 0, 1, 20, 4608, 0);
    // the shape the kernel has today: 90-row bands, 1 KB pieces, one wave per workgroup, frame fastest, 32 waves per CU
    for (int D : {0, 20}) run(90, 1, 1, 0, 0, D, 4608, 0);
    // order
    for (int D : {0, 20}) run(90, 1, 1, 0, 1, D, 4608, 0);
    // waves of a band together (and in lockstep)
    for (int W : {2, 4, 8}) for (int L : {0, 1}) for (int order : {0, 1}) run(90, 1, W, L, order, 20, 4608, 0);
    // wider pieces
    for (int S : {2, 4, 8}) for (int order : {0, 1}) run(90, S, 1, 0, order, 20 * S, 4608 * (S > 2 ? 2 : 1), 0);
    // band height
    for (int B : {24, 45, 180, 270, 1080}) for (int order : {0, 1}) run(B, 1, 1, 0, order, 20, 4608, 0);
    for (int B : {24, 45, 180, 270, 1080}) run(B, 1, 8, 1, 1, 20, 4608, 0);
    // non-temporal
    run(90, 1, 1, 0, 0, 20, 4608, 1);
    run(90, 1, 8, 1, 1, 20, 4608, 1);
    // occupancy
    for (int ldsw : {4608, 9216, 18432}) run(90, 1, 1, 0, 0, 20, ldsw, 0);
    return 0;
}
