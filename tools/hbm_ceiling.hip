// Measured HBM ceilings on the box for the access shapes the decode kernels use
// ("also report against a measured streaming-store ceiling").
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o /tmp/hbm_ceiling && /tmp/hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill16(u32x4* __restrict__ dst, size_t n, uint32_t v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        u32x4 x = u32x4{v, v + 1, v + 2, v + 3};
        if (NT) __builtin_nontemporal_store(x, dst + i); else dst[i] = x;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void copy16(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        u32x4 x = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(x, dst + i); else dst[i] = x;
    }
}
// the MSVideo1 store shape: lane = 4x4 block, 4 row stores of 16 B, rows X ints apart
template <bool NT>
__global__ __launch_bounds__(256) void fill_blocks(uint32_t* __restrict__ base, int nblocks, int nbx, int X, size_t frame_ints) {
    uint32_t* dst = base + (size_t)blockIdx.y * frame_ints;
    int blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
    u32x4 x = u32x4{(uint32_t)blk, (uint32_t)blk + 1, (uint32_t)blk + 2, (uint32_t)blk + 3};
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        if (NT) __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(p + (size_t)y * X));
        else *reinterpret_cast<u32x4*>(p + (size_t)y * X) = x;
    }
}

// A: linear, 4 stores per lane, workgroup covers 16 KB contiguous
__global__ __launch_bounds__(256) void fill_lin4(u32x4* __restrict__ dst, size_t n) {
    size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k * 256 < n) dst[base + k * 256] = u32x4{1u, 2u, 3u, (uint32_t)k};
}
// D: one workgroup (512 lanes, 480 active) per block row: 30 KB contiguous, written linearly
__global__ __launch_bounds__(512) void fill_rowlinear(uint32_t* __restrict__ base, int nbx, int X, size_t frame_ints) {
    u32x4* dst = reinterpret_cast<u32x4*>(base + (size_t)blockIdx.y * frame_ints + (size_t)blockIdx.x * 4 * X);
    const int per_row = X / 4;  // 16-byte units per pixel row == nbx
    for (int i = threadIdx.x; i < per_row * 4; i += 512) dst[i] = u32x4{1u, 2u, 3u, (uint32_t)i};
}
// E: block-shaped, XCD-aware: workgroup id remapped so the 8 workgroups an XCD gets in a round are
// memory-adjacent chunks (blockIdx%8 picks the XCD under round-robin dispatch)
__global__ __launch_bounds__(256) void fill_blocks_xcd(uint32_t* __restrict__ base, int nblocks, int nbx, int X, size_t frame_ints, int wgs_per_frame) {
    uint32_t* dst = base + (size_t)blockIdx.y * frame_ints;
    const int per_xcd = (wgs_per_frame + 7) / 8;
    const int wg = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (wg >= wgs_per_frame) return;
    int blk = wg * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
    u32x4 x = u32x4{(uint32_t)blk, 1u, 2u, 3u};
#pragma unroll
    for (int y = 0; y < 4; ++y) *reinterpret_cast<u32x4*>(p + (size_t)y * X) = x;
}

// F: block-shaped, ONE store per lane: workgroup = 64 blocks x 4 rows, wave w writes row w (1 KiB contiguous)
__global__ __launch_bounds__(256) void fill_blocks_rowsplit(uint32_t* __restrict__ base, int nblocks, int nbx, int X, size_t frame_ints) {
    uint32_t* dst = base + (size_t)blockIdx.y * frame_ints;
    const int blk = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = threadIdx.x >> 6;
    if (blk >= nblocks) return;
    int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* p = dst + ((size_t)by * 4 + y) * X + bx * 4;
    *reinterpret_cast<u32x4*>(p) = u32x4{(uint32_t)blk, 1u, 2u, 3u};
}
// G: two stores per lane: workgroup = 128 blocks x 2 row pairs
__global__ __launch_bounds__(256) void fill_blocks_rowsplit2(uint32_t* __restrict__ base, int nblocks, int nbx, int X, size_t frame_ints) {
    uint32_t* dst = base + (size_t)blockIdx.y * frame_ints;
    const int blk = blockIdx.x * 128 + (threadIdx.x & 127);
    const int y = (threadIdx.x >> 7) * 2;
    if (blk >= nblocks) return;
    int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* p = dst + ((size_t)by * 4 + y) * X + bx * 4;
    *reinterpret_cast<u32x4*>(p) = u32x4{(uint32_t)blk, 1u, 2u, 3u};
    *reinterpret_cast<u32x4*>(p + X) = u32x4{(uint32_t)blk, 1u, 2u, 4u};
}

template <class F>
static double time_us(F&& launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}

// Temporal shapes: a workgroup owns a tile of the picture and writes it once per frame, frame after frame
// (what the inter-frame group kernels do).  TR rows x TC pixels per workgroup, 256 lanes, each lane TR*TC/1024
// stores of 16 B per frame; lanes run along the row first.
template <int TR, int TC>
__global__ __launch_bounds__(256) void temporal_fill(uint32_t* __restrict__ base, int X, int Y, size_t frame_ints, int F) {
    constexpr int LANES_PER_ROW = TC / 4, ROWS_PER_PASS = 256 / LANES_PER_ROW, PASSES = TR / ROWS_PER_PASS;
    static_assert(PASSES >= 1 && TR % ROWS_PER_PASS == 0, "shape");
    const int tiles_x = (X + TC - 1) / TC;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * TC + (threadIdx.x % LANES_PER_ROW) * 4, r0 = ty * TR + threadIdx.x / LANES_PER_ROW;
    if (x >= X) return;
    u32x4 v = u32x4{(uint32_t)x, (uint32_t)r0, 3u, 4u};
    for (int f = 0; f < F; ++f) {
        uint32_t* dst = base + (size_t)f * frame_ints;
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
            const int y = r0 + k * ROWS_PER_PASS;
            if (y < Y) *reinterpret_cast<u32x4*>(dst + (size_t)y * X + x) = v;
        }
        v.x += 1;
    }
}
// same as temporal_fill<16, 64>, but every frame's base pointer comes from a table staged in LDS (what the
// ScreenPressor group kernel does with its per-frame destinations)
__global__ __launch_bounds__(256) void temporal_fill_lds_table(uint32_t* const* __restrict__ table, int X, int Y, int F) {
    __shared__ uint32_t* s_tab[128];
    const int tiles_x = (X + 63) / 64;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x = tx * 64 + (threadIdx.x % 16) * 4, y = ty * 16 + threadIdx.x / 16;
    u32x4 v = u32x4{(uint32_t)x, (uint32_t)y, 3u, 4u};
    for (int f0 = 0; f0 < F; f0 += 128) {
        const int nf = F - f0 < 128 ? F - f0 : 128;
        __syncthreads();
        if ((int)threadIdx.x < nf) s_tab[threadIdx.x] = table[f0 + threadIdx.x];
        __syncthreads();
        if (x < X && y < Y)
            for (int f = 0; f < nf; ++f) *reinterpret_cast<u32x4*>(s_tab[f] + (size_t)y * X + x) = v;
    }
}

template <int TR, int TC>
static double run_temporal(uint32_t* d, int X, int Y, size_t frame_ints, int F) {
    const int tiles = ((X + TC - 1) / TC) * ((Y + TR - 1) / TR);
    return time_us([&] { temporal_fill<TR, TC><<<tiles, 256>>>(d, X, Y, frame_ints, F); }, 30);
}

int main(int argc, char** argv) {
    // optional argument: number of frames (default 64 = 531 MB; 299 = 2.5 GB, far beyond the 256 MB Infinity Cache)
    const int X = 1920, Y = 1080, F = argc > 1 ? atoi(argv[1]) : 64;
    const size_t frame_ints = (size_t)X * Y, bytes = frame_ints * 4 * F, n16 = bytes / 16;
    u32x4 *d, *s;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&s, bytes));
    CK(hipMemset(s, 1, bytes));
    const int nblocks = (X / 4) * (Y / 4);
    printf("buffer %.1f MB (%d frames 1920x1080 RGB32)\n", bytes / 1e6, F);
    for (int grid : {2048, 8192, 32768, (int)((n16 + 255) / 256)}) {
        double t1 = time_us([&] { fill16<false><<<grid, 256>>>(d, n16, 7); }, 50);
        double t2 = time_us([&] { fill16<true><<<grid, 256>>>(d, n16, 7); }, 50);
        double t3 = time_us([&] { copy16<false><<<grid, 256>>>(d, s, n16); }, 50);
        double t4 = time_us([&] { copy16<true><<<grid, 256>>>(d, s, n16); }, 50);
        printf("grid %8d  fill %7.1f us %6.0f GB/s | fill nt %7.1f us %6.0f GB/s | copy %7.1f us %6.0f GB/s (r+w) | copy nt %7.1f us %6.0f GB/s\n",
               grid, t1, bytes / t1 / 1e3, t2, bytes / t2 / 1e3, t3, 2.0 * bytes / t3 / 1e3, t4, 2.0 * bytes / t4 / 1e3);
    }
    dim3 g((nblocks + 255) / 256, F);
    double t5 = time_us([&] { fill_blocks<false><<<g, 256>>>((uint32_t*)d, nblocks, X / 4, X, frame_ints); }, 50);
    double t6 = time_us([&] { fill_blocks<true><<<g, 256>>>((uint32_t*)d, nblocks, X / 4, X, frame_ints); }, 50);
    printf("block-shaped fill (4 x 16 B rows per lane): %7.1f us %6.0f GB/s | nt %7.1f us %6.0f GB/s\n",
           t5, bytes / t5 / 1e3, t6, bytes / t6 / 1e3);
    {
        double a = time_us([&] { fill_lin4<<<(int)((n16 + 1023) / 1024), 256>>>(d, n16); }, 50);
        double dd = time_us([&] { fill_rowlinear<<<dim3(Y / 4, F), 512>>>((uint32_t*)d, X / 4, X, frame_ints); }, 50);
        int wpf = (nblocks + 255) / 256;
        double e = time_us([&] { fill_blocks_xcd<<<dim3(((wpf + 7) / 8) * 8, F), 256>>>((uint32_t*)d, nblocks, X / 4, X, frame_ints, wpf); }, 50);
        printf("A linear 4 stores/lane %7.1f us %6.0f GB/s | D row-linear 512-lane WG %7.1f us %6.0f GB/s | E block-shaped xcd-remap %7.1f us %6.0f GB/s\n",
               a, bytes / a / 1e3, dd, bytes / dd / 1e3, e, bytes / e / 1e3);
    }
    {
        double f = time_us([&] { fill_blocks_rowsplit<<<dim3((nblocks + 63) / 64, F), 256>>>((uint32_t*)d, nblocks, X / 4, X, frame_ints); }, 50);
        double g2 = time_us([&] { fill_blocks_rowsplit2<<<dim3((nblocks + 127) / 128, F), 256>>>((uint32_t*)d, nblocks, X / 4, X, frame_ints); }, 50);
        printf("F block-shaped 1 store/lane (wave = row) %7.1f us %6.0f GB/s | G 2 stores/lane %7.1f us %6.0f GB/s\n", f, bytes / f / 1e3, g2, bytes / g2 / 1e3);
    }
    {
        struct { const char* name; double us; int wgs; } r[] = {
            {"16 rows x 64 px (1 store/lane)", run_temporal<16, 64>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"4 rows x 256 px (1 store/lane, wave = 1 KiB row)", run_temporal<4, 256>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"1 row x 1024 px (1 store/lane)", run_temporal<1, 1024>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"8 rows x 256 px (2 stores/lane)", run_temporal<8, 256>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"16 rows x 256 px (4 stores/lane)", run_temporal<16, 256>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"32 rows x 64 px (2 stores/lane)", run_temporal<32, 64>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"64 rows x 64 px (4 stores/lane)", run_temporal<64, 64>((uint32_t*)d, X, Y, frame_ints, F), 0},
            {"4 rows x 1024 px (4 stores/lane)", run_temporal<4, 1024>((uint32_t*)d, X, Y, frame_ints, F), 0},
        };
        {
            std::vector<uint32_t*> h(F);
            for (int f = 0; f < F; ++f) h[f] = (uint32_t*)d + (size_t)f * frame_ints;
            uint32_t** dt;
            CK(hipMalloc(&dt, sizeof(uint32_t*) * F));
            CK(hipMemcpy(dt, h.data(), sizeof(uint32_t*) * F, hipMemcpyHostToDevice));
            const int tiles = ((X + 63) / 64) * ((Y + 15) / 16);
            double us = time_us([&] { temporal_fill_lds_table<<<tiles, 256>>>(dt, X, Y, F); }, 30);
            printf("temporal 16 rows x 64 px, frame pointers from an LDS table: %7.1f us %6.0f GB/s\n", us, bytes / us / 1e3);
        }
        printf("temporal tile fill, all frames written by long-lived workgroups (one tile each, frame after frame):\n");
        for (auto& e : r) printf("  %-52s %7.1f us %6.0f GB/s\n", e.name, e.us, bytes / e.us / 1e3);
    }
    double t7 = time_us([&] { (void)hipMemsetAsync(d, 0, bytes, 0); }, 20);
    double t8 = time_us([&] { (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, 20);
    printf("hipMemsetAsync %7.1f us %6.0f GB/s | hipMemcpyDtoD %7.1f us %6.0f GB/s (r+w)\n", t7, bytes / t7 / 1e3, t8, 2.0 * bytes / t8 / 1e3);
    return 0;
}
