// lab: does it matter WHERE the destination frames lie?  The store shape of msv1_fused_kernel on all-solid frames — a workgroup per
// (frame, tile of T blocks), tiles handed out tile-major (tile j of every frame, then tile j + 1 ...: up to F write fronts at once),
// lane = 4x4 block, four 16-byte row stores per block — with the frames
//   A  back to back in one allocation,          B  one hipMalloc per frame,
//   C  two frames per 20 MB hipMalloc (what torch's caching allocator does with 8.3 MB tensors),
//   D  one VA range backed by separately created 2 MB-granular physical chunks (hipMemCreate / hipMemMap), frames back to back.
//   hipcc -O3 --offload-arch=gfx950 tools/front_lab.hip -o tools/front_lab.bin && tools/front_lab.bin [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int X = 1920, Y = 1080, NBX = X / 4, NBLK = (X / 4) * (Y / 4);
constexpr size_t FRAME_BYTES = (size_t)X * Y * 4;

__global__ __launch_bounds__(256) void front_kernel(uint32_t* const* __restrict__ frames, int nframes, int T, int tiles_per_frame, int tile_major) {
    int f, j;
    if (tile_major == 1) { j = blockIdx.x / nframes; f = blockIdx.x - j * nframes; }
    else if (tile_major == 0) { f = blockIdx.x / tiles_per_frame; j = blockIdx.x - f * tiles_per_frame; }
    else if (tile_major == 2) {                    // staggered: frame f runs (f mod S) rounds late; rounds = tiles_per_frame + S, empty slots leave at once
        const int S = 64;
        const int r = blockIdx.x / nframes;
        f = blockIdx.x - r * nframes;
        j = r - f % S;
        if (j < 0 || j >= tiles_per_frame) return;
    } else {                                       // a pseudo-random permutation of all (frame, tile) pairs
        const unsigned long long n = (unsigned long long)nframes * tiles_per_frame;
        const unsigned long long p = ((unsigned long long)blockIdx.x * 2654435761ull + 12345ull) % n;   // (n is not a multiple of the odd multiplier's factors: a bijection when gcd = 1)
        f = (int)(p % nframes);
        j = (int)(p / nframes);
    }
    uint32_t* dst = frames[f];
    const int b0 = j * T;
    for (int r = 0; r < T; r += 256) {
        const int blk = b0 + r + (int)threadIdx.x;
        if (blk < NBLK) {
            const int by = blk / NBX, bx = blk - by * NBX;
            uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
#pragma unroll
            for (int y = 0; y < 4; ++y) *(gu32x4*)(p + (size_t)y * X) = u32x4{(uint32_t)blk, 1u, 2u, (uint32_t)y};
        }
    }
}

// plain fill: one 16-byte store per lane, workgroups in address order
__global__ __launch_bounds__(256) void fill_kernel(u32x4* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) *(gu32x4*)(dst + i) = u32x4{(uint32_t)i, 1u, 2u, 3u};
}

// the ScreenPressor key-frame kernel's store shape: one wave per (frame, band of B rows, 1 KB of a row), D dependent instructions per row
__global__ __launch_bounds__(64) void band_kernel(uint32_t* __restrict__ pool, int nframes, int B, int bands, int D) {
    const int f = blockIdx.x % nframes, g = blockIdx.x / nframes;   // frames fastest, as launch_iframe_tiles
    const int band = g / 8, sx = g - band * 8;
    const int x = sx * 256 + (int)threadIdx.x * 4;
    if (x >= X) return;
    uint32_t* dst = pool + (size_t)f * X * Y;
    uint32_t a = (uint32_t)(f + g);
    const int y1 = (band + 1) * B < Y ? (band + 1) * B : Y;
    for (int y = band * B; y < y1; ++y) {
        for (int i = 0; i < D; ++i) a = a * 1664525u + 1013904223u;
        *(gu32x4*)(dst + (size_t)y * X + x) = u32x4{a, a + 1, a + 2, a + 3};
    }
}

// the ScreenPressor inter-frame group kernel's store shape: workgroups that own a piece of the picture and write it into `nframes` consecutive frames
//   wide = 0: 256 lanes = 8 blocks x 16 rows, a wave's store = two 512-byte row segments (rows r and r + 8), 2 stores per lane and frame (sp_pframe_group_kernel)
//   wide = 1: 256 lanes = 16 blocks x 16 rows, a wave's store = ONE 1 KB row segment, 4 stores per lane and frame (rows 4w .. 4w + 3 of wave w)
__global__ __launch_bounds__(256) void group_kernel(uint32_t* __restrict__ pool, int nframes, int wide) {
    const int tid = threadIdx.x;
    uint32_t a = (uint32_t)(blockIdx.x * 131 + blockIdx.y * 7 + tid);
    if (!wide) {
        const int r = tid >> 5, ch = tid & 31;
        const int x0 = (int)blockIdx.x * 128 + ch * 4, ya = (int)blockIdx.y * 16 + r, yb = ya + 8;
        if (x0 >= X) return;
        for (int f = 0; f < nframes; ++f) {
            uint32_t* dst = pool + (size_t)f * X * Y;
            a = a * 1664525u + 1013904223u;
            if (ya < Y) *(gu32x4*)(dst + (size_t)ya * X + x0) = u32x4{a, a + 1, a + 2, a + 3};
            if (yb < Y) *(gu32x4*)(dst + (size_t)yb * X + x0) = u32x4{a, a + 1, a + 2, a + 4};
        }
    } else {
        const int w = tid >> 6, lane = tid & 63;
        const int x0 = (int)blockIdx.x * 256 + lane * 4, y0 = (int)blockIdx.y * 16 + w * 4;
        if (x0 >= X) return;
        for (int f = 0; f < nframes; ++f) {
            uint32_t* dst = pool + (size_t)f * X * Y;
            a = a * 1664525u + 1013904223u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (y0 + k < Y) *(gu32x4*)(dst + (size_t)(y0 + k) * X + x0) = u32x4{a, a + 1, a + 2, a + (uint32_t)k};
        }
    }
}


// round 5: FULL-WIDTH STRIPS.  A workgroup owns R whole rows of the picture (R x 7 680 contiguous bytes; with P > 1 the strip is cut
// into P pieces of 7 680 / P bytes per row) and writes them into `nframes` consecutive frames: lane = 16-byte column chunk, R row stores per
// lane and frame (a wave's store instruction = 1 KB of one row, the workgroup's waves side by side in the row).  col_major = 0: all lanes
// store row 0, then row 1 ...; 1: linear — the strip is one run of R x 480 chunks, chunk = t + k x WG (the same bytes, every store
// instruction of a wave 1 KB further along the run).
__global__ void strip_kernel(uint32_t* __restrict__ pool, int nframes, int R, int P, int linear) {
    const int tid = threadIdx.x, WG = blockDim.x;
    const int s = (int)blockIdx.x / P, part = (int)blockIdx.x - s * P;
    const int y0 = s * R;
    const int rows = y0 + R <= Y ? R : Y - y0;
    if (rows <= 0) return;
    const int cpr = (X / 4) / P;                     // chunks per row piece
    uint32_t a = (uint32_t)(blockIdx.x * 131 + tid);
    if (linear && P == 1) {
        const int n = rows * (X / 4);
        for (int f = 0; f < nframes; ++f) {
            u32x4* dst = (u32x4*)(pool + (size_t)f * X * Y + (size_t)y0 * X);
            a = a * 1664525u + 1013904223u;
            for (int c = tid; c < n; c += WG) *(gu32x4*)(dst + c) = u32x4{a, a + 1, a + 2, (uint32_t)c};
        }
        return;
    }
    for (int f = 0; f < nframes; ++f) {
        uint32_t* dst = pool + (size_t)f * X * Y + (size_t)y0 * X + part * cpr * 4;
        a = a * 1664525u + 1013904223u;
        for (int c = tid; c < cpr; c += WG)
            for (int r = 0; r < rows; ++r) *(gu32x4*)(dst + (size_t)r * X + c * 4) = u32x4{a, a + 1, a + 2, (uint32_t)r};
    }
}


// round 5: is it the UNEQUAL PROGRESS of looping workgroups?  The group shape (8 blocks x 16 rows per workgroup, 299 frames) with all workgroups
// kept in step — every `sync_every` frames each workgroup arrives on a counter and waits until all have (all are resident: 1 020 x 4 waves) —
// and the same shape with FRESH workgroups: grid.z = chunks of C frames, a workgroup writes its piece into C frames and leaves.
__global__ __launch_bounds__(256) void group_sync_kernel(uint32_t* __restrict__ pool, int nframes, int sync_every, unsigned* counter, unsigned base) {
    const int tid = threadIdx.x;
    const unsigned nwg = gridDim.x * gridDim.y;
    uint32_t a = (uint32_t)(blockIdx.x * 131 + blockIdx.y * 7 + tid);
    const int r = tid >> 5, ch = tid & 31;
    const int x0 = (int)blockIdx.x * 128 + ch * 4, ya = (int)blockIdx.y * 16 + r, yb = ya + 8;
    unsigned round = 0;
    for (int f = 0; f < nframes; ++f) {
        if (sync_every > 0 && f > 0 && f % sync_every == 0) {
            ++round;
            if (tid == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < (1 << 20); ++spin) {
                    if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base >= round * nwg) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (x0 >= X) continue;
        uint32_t* dst = pool + (size_t)f * X * Y;
        a = a * 1664525u + 1013904223u;
        if (ya < Y) *(gu32x4*)(dst + (size_t)ya * X + x0) = u32x4{a, a + 1, a + 2, a + 3};
        if (yb < Y) *(gu32x4*)(dst + (size_t)yb * X + x0) = u32x4{a, a + 1, a + 2, a + 4};
    }
}
__global__ __launch_bounds__(256) void group_fresh_kernel(uint32_t* __restrict__ pool, int nframes, int C) {
    const int tid = threadIdx.x;
    uint32_t a = (uint32_t)(blockIdx.x * 131 + blockIdx.y * 7 + tid);
    const int r = tid >> 5, ch = tid & 31;
    const int x0 = (int)blockIdx.x * 128 + ch * 4, ya = (int)blockIdx.y * 16 + r, yb = ya + 8;
    if (x0 >= X) return;
    const int f0 = (int)blockIdx.z * C, f1 = f0 + C < nframes ? f0 + C : nframes;
    for (int f = f0; f < f1; ++f) {
        uint32_t* dst = pool + (size_t)f * X * Y;
        a = a * 1664525u + 1013904223u;
        if (ya < Y) *(gu32x4*)(dst + (size_t)ya * X + x0) = u32x4{a, a + 1, a + 2, a + 3};
        if (yb < Y) *(gu32x4*)(dst + (size_t)yb * X + x0) = u32x4{a, a + 1, a + 2, a + 4};
    }
}


// round 5: any piece geometry for a frame-walking workgroup: a workgroup owns PR rows x PB bytes (PB / 16 lanes per row) of the picture and
// writes it into `nframes` consecutive frames; its 256 lanes take the piece's 16-byte chunks row-major, chunk = tid + k x 256.
__global__ __launch_bounds__(1024) void piece_kernel(uint32_t* __restrict__ pool, int nframes, int PR, int PB, int pieces_per_row) {
    const int tid = threadIdx.x, WG = blockDim.x;
    const int L = PB / 16;
    const int prow = (int)blockIdx.x / pieces_per_row, pcol = (int)blockIdx.x - prow * pieces_per_row;
    const int y0 = prow * PR, xb = pcol * PB;
    uint32_t a = (uint32_t)(blockIdx.x * 131 + tid);
    const int n = PR * L;
    for (int f = 0; f < nframes; ++f) {
        uint8_t* dst = (uint8_t*)(pool + (size_t)f * X * Y);
        a = a * 1664525u + 1013904223u;
        for (int c = tid; c < n; c += WG) {
            const int r = c / L, l = c - r * L;
            const int y = y0 + r, x = xb + l * 16;
            if (y < Y && x < X * 4) *(gu32x4*)(dst + (size_t)y * X * 4 + x) = u32x4{a, a + 1, a + 2, (uint32_t)c};
        }
    }
}


// table-based forms of the band-walking and frame-walking shapes (frames anywhere): LAB_SPREAD
__global__ __launch_bounds__(64) void band_table_kernel(uint32_t* const* __restrict__ frames, int nframes, int B, int bands, int D) {
    const int f = blockIdx.x % nframes, g = blockIdx.x / nframes;
    const int band = g / 8, sx = g - band * 8;
    const int x = sx * 256 + (int)threadIdx.x * 4;
    if (x >= X) return;
    uint32_t* dst = frames[f];
    uint32_t a = (uint32_t)(f + g);
    const int y1 = (band + 1) * B < Y ? (band + 1) * B : Y;
    for (int y = band * B; y < y1; ++y) {
        for (int i = 0; i < D; ++i) a = a * 1664525u + 1013904223u;
        *(gu32x4*)(dst + (size_t)y * X + x) = u32x4{a, a + 1, a + 2, a + 3};
    }
}
__global__ __launch_bounds__(256) void group_table_kernel(uint32_t* const* __restrict__ frames, int nframes) {
    const int tid = threadIdx.x;
    uint32_t a = (uint32_t)(blockIdx.x * 131 + blockIdx.y * 7 + tid);
    const int r = tid >> 5, ch = tid & 31;
    const int x0 = (int)blockIdx.x * 128 + ch * 4, ya = (int)blockIdx.y * 16 + r, yb = ya + 8;
    if (x0 >= X) return;
    for (int f = 0; f < nframes; ++f) {
        uint32_t* dst = frames[f];
        a = a * 1664525u + 1013904223u;
        if (ya < Y) *(gu32x4*)(dst + (size_t)ya * X + x0) = u32x4{a, a + 1, a + 2, a + 3};
        if (yb < Y) *(gu32x4*)(dst + (size_t)yb * X + x0) = u32x4{a, a + 1, a + 2, a + 4};
    }
}


// the band-walking shape with the 8 spans of a band in ONE workgroup (8 waves side by side in the row: a row's 7 680 bytes go out from one CU at about
// the same time); sync = 1: an LDS-only barrier after every row keeps the waves in step
__global__ __launch_bounds__(512) void band_wg_table_kernel(uint32_t* const* __restrict__ frames, int nframes, int B, int bands, int D, int sync) {
    const int f = blockIdx.x % nframes, band = blockIdx.x / nframes;
    const int x = (int)threadIdx.x * 4;
    uint32_t* dst = frames[f];
    uint32_t a = (uint32_t)(f + band + threadIdx.x);
    const int y1 = (band + 1) * B < Y ? (band + 1) * B : Y;
    for (int y = band * B; y < y1; ++y) {
        for (int i = 0; i < D; ++i) a = a * 1664525u + 1013904223u;
        if (x < X) *(gu32x4*)(dst + (size_t)y * X + x) = u32x4{a, a + 1, a + 2, a + 3};
        if (sync) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}


// the fronts shape with NONTEMPORAL row stores (what msv1_fused_kernel issues), and with plain stores but 8 KB per lane visit order reversed (rows 3..0)
__global__ __launch_bounds__(256) void front_nt_kernel(uint32_t* const* __restrict__ frames, int nframes, int T, int tiles_per_frame, int nt) {
    const int j = blockIdx.x / nframes, f = blockIdx.x - j * nframes;
    uint32_t* dst = frames[f];
    const int b0 = j * T;
    for (int r = 0; r < T; r += 256) {
        const int blk = b0 + r + (int)threadIdx.x;
        if (blk < NBLK) {
            const int by = blk / NBX, bx = blk - by * NBX;
            uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const u32x4 v = u32x4{(uint32_t)blk, 1u, 2u, (uint32_t)y};
                if (nt) __builtin_nontemporal_store(v, (gu32x4*)(p + (size_t)y * X));
                else *(gu32x4*)(p + (size_t)y * X) = v;
            }
        }
    }
}


// fresh workgroups, one BLOCK ROW each (480 blocks = 4 rows x 7 680 contiguous bytes), frame-major or row-major over the frames (what a descriptor-fed
// block kernel with one workgroup per block row would store)
__global__ __launch_bounds__(512) void blockrow_kernel(uint32_t* const* __restrict__ frames, int nframes, int row_major) {
    int f, by;
    if (row_major) { by = blockIdx.x / nframes; f = blockIdx.x - by * nframes; } else { f = blockIdx.x / (Y / 4); by = blockIdx.x - f * (Y / 4); }
    const int bx = threadIdx.x;
    if (bx >= NBX) return;
    uint32_t* p = frames[f] + (size_t)by * 4 * X + bx * 4;
#pragma unroll
    for (int y = 0; y < 4; ++y) *(gu32x4*)(p + (size_t)y * X) = u32x4{(uint32_t)bx, 1u, 2u, (uint32_t)y};
}
// fronts with workgroups of WG lanes (64 ... 1024): T blocks per workgroup, tile-major
__global__ __launch_bounds__(1024) void front_wg_kernel(uint32_t* const* __restrict__ frames, int nframes, int T) {
    const int WG = blockDim.x;
    const int j = blockIdx.x / nframes, f = blockIdx.x - j * nframes;
    uint32_t* dst = frames[f];
    const int b0 = j * T;
    for (int r = 0; r < T; r += WG) {
        const int blk = b0 + r + (int)threadIdx.x;
        if (blk < NBLK) {
            const int by = blk / NBX, bx = blk - by * NBX;
            uint32_t* p = dst + (size_t)by * 4 * X + bx * 4;
#pragma unroll
            for (int y = 0; y < 4; ++y) *(gu32x4*)(p + (size_t)y * X) = u32x4{(uint32_t)blk, 1u, 2u, (uint32_t)y};
        }
    }
}


// the same bytes as blockrow_kernel (one block row = 30 720 contiguous bytes per fresh workgroup, frame-major) written LINEARLY: lane t takes the 16-byte chunks t, t + 512, ...
__global__ __launch_bounds__(512) void blockrow_linear_kernel(uint32_t* const* __restrict__ frames, int nframes) {
    const int f = blockIdx.x / (Y / 4), by = blockIdx.x - f * (Y / 4);
    u32x4* p = (u32x4*)(frames[f] + (size_t)by * 4 * X);
    for (int c = threadIdx.x; c < X; c += 512) *(gu32x4*)(p + c) = u32x4{(uint32_t)c, 1u, 2u, 3u};   // X / 4 chunks per row x 4 rows = X chunks
}


// plain linear fills of one allocation by workgroup extent: a workgroup of L lanes writes L x SPL consecutive 16-byte chunks (chunk = first + k x L + lane), workgroups in address order
__global__ __launch_bounds__(1024) void linear_fill_kernel(u32x4* __restrict__ dst, size_t n, int spl) {
    const size_t first = (size_t)blockIdx.x * blockDim.x * spl;
    for (int k = 0; k < spl; ++k) {
        const size_t i = first + (size_t)k * blockDim.x + threadIdx.x;
        if (i < n) *(gu32x4*)(dst + i) = u32x4{(uint32_t)i, 1u, 2u, 3u};
    }
}

// translation probe: every lane reads 4 bytes from a page of its own, pages picked by a multiplicative hash over the whole buffer;
// `page` = distance between candidate addresses.  Bound by address translation when the mapping's fragments are small.
__global__ __launch_bounds__(256) void page_probe_kernel(const uint32_t* __restrict__ buf, size_t npages, size_t page_words, int rounds, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        i = (i * 2654435761ull + 40503ull * (r + 1)) % npages;
        acc += buf[i * page_words + (threadIdx.x & 15)];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 512;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    uint32_t** d_table;
    CK(hipMalloc(&d_table, sizeof(uint32_t*) * F));
    auto measure = [&](const char* what, const std::vector<uint32_t*>& frames) {
        CK(hipMemcpy(d_table, frames.data(), sizeof(uint32_t*) * F, hipMemcpyHostToDevice));
        for (int T : {8192, 2048}) for (int tm : {1, 0}) {
            const int tpf = (NBLK + T - 1) / T;
            auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * F), dim3(256), 0, 0, d_table, F, T, tpf, tm); };
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
            printf("%-58s T %5d %s | %8.1f us %7.0f GB/s\n", what, T, tm ? "tile-major " : "frame-major", best * 1000, (double)F * FRAME_BYTES / best / 1e6);
            fflush(stdout);
        }
    };
    if (getenv("LAB_TIME")) {   // round 6, last question: is "fast" a property of the MEMORY a pool got, or of the MOMENT it is measured?  Pools made of per-frame 8 MB handles, probed (the 512-front
        // shape), then churn (allocations made and freed, host work with the GPU idle), then the SAME pools probed again next to fresh ones, several rounds.
        const int NF = F / 2;                                   // frames per pool (256: 2.1 GB)
        const int T = 8192, tpf = (NBLK + T - 1) / T;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        struct Pool { void* va; size_t bytes; std::vector<hipMemGenericAllocationHandle_t> h; std::vector<uint32_t*> fr; };
        const size_t HB = (size_t)8 << 20;
        auto make = [&](Pool& p) {
            p.bytes = (size_t)NF * HB;
            CK(hipMemAddressReserve(&p.va, p.bytes, HB, nullptr, 0));
            p.h.resize(NF);
            for (int i = 0; i < NF; ++i) { CK(hipMemCreate(&p.h[i], HB, &prop, 0)); CK(hipMemMap((char*)p.va + (size_t)i * HB, HB, 0, p.h[i], 0)); }
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(p.va, p.bytes, &acc, 1));
            p.fr.resize(NF);
            for (int i = 0; i < NF; ++i) p.fr[i] = (uint32_t*)((char*)p.va + (size_t)i * HB);
        };
        auto drop = [&](Pool& p) { (void)hipMemUnmap(p.va, p.bytes); for (auto h : p.h) (void)hipMemRelease(h); (void)hipMemAddressFree(p.va, p.bytes); };
        auto probe = [&](const Pool& p) {
            CK(hipMemcpy(d_table, p.fr.data(), sizeof(uint32_t*) * NF, hipMemcpyHostToDevice));
            auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * NF), dim3(256), 0, 0, d_table, NF, T, tpf, 1); };
            launch();
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms / 3 < best ? ms / 3 : best;
            }
            return (double)NF * FRAME_BYTES / best / 1e6;
        };
        std::vector<Pool> kept(4);
        printf("four pools of %d frames made one after the other and kept:", NF);
        for (auto& p : kept) { make(p); printf(" %5.0f", probe(p)); }
        printf(" GB/s\n");
        for (int round = 0; round < 6; ++round) {
            // churn: what a bench leg does between pools — big allocations made and freed, pinned host memory, then the host busy and the GPU idle
            {
                std::vector<void*> junk;
                for (int i = 0; i < 24; ++i) { void* d = nullptr; if (hipMalloc(&d, (size_t)(64 + 37 * i) << 20) == hipSuccess) junk.push_back(d); }
                void* hp = nullptr;
                (void)hipHostMalloc(&hp, (size_t)512 << 20, hipHostMallocDefault);
                for (size_t i = 0; i < junk.size(); i += 2) (void)hipFree(junk[i]);
                timespec ts{(round % 3 == 2) ? 4 : 1, 0};        // (every third round: four seconds of idle)
                nanosleep(&ts, nullptr);
                for (size_t i = 1; i < junk.size(); i += 2) (void)hipFree(junk[i]);
                if (hp) (void)hipHostFree(hp);
            }
            printf("round %d (after churn%s): the kept pools again:", round, round % 3 == 2 ? " + 4 s idle" : "");
            for (auto& p : kept) printf(" %5.0f", probe(p));
            Pool fresh[3];
            printf(" | three fresh pools (held together):");
            for (auto& p : fresh) { make(p); printf(" %5.0f", probe(p)); }
            printf(" | the kept pools once more:");
            for (auto& p : kept) printf(" %5.0f", probe(p));
            printf(" GB/s\n");
            fflush(stdout);
            for (auto& p : fresh) drop(p);
        }
        for (auto& p : kept) drop(p);
        return 0;
    }
    if (getenv("LAB_VMM")) {   // round 6: the pool built from hipMemCreate handles mapped into one VA range — holds exactly the pool; does an arrangement of the handles match the dealt hipMalloc pool?
        // three store shapes per arrangement: the 512 write fronts (what jsp_pool_create probes with), band-walking waves (the ScreenPressor key-frame kernel's), frame-walking workgroups (the group kernels')
        auto rates = [&](const char* what, const std::vector<uint32_t*>& fr, double setup_ms, double held_x) {
            CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * F, hipMemcpyHostToDevice));
            const int T = 8192, tpf = (NBLK + T - 1) / T;
            auto time3 = [&](auto&& launch) {
                launch();
                CK(hipDeviceSynchronize());
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    float ms = 0;
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 3; ++i) launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms / 3 < best ? ms / 3 : best;
                }
                return (double)F * FRAME_BYTES / best / 1e6;
            };
            const double fronts = time3([&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * F), dim3(256), 0, 0, d_table, F, T, tpf, 1); });
            const int B = 90, bands = (Y + B - 1) / B;
            const double band = time3([&] { hipLaunchKernelGGL(band_wg_table_kernel, dim3(F * bands), dim3(512), 0, 0, d_table, F, B, bands, 8, 0); });
            const double walk = time3([&] { hipLaunchKernelGGL(group_table_kernel, dim3((X + 127) / 128, (Y + 15) / 16), dim3(256), 0, 0, d_table, F); });
            printf("%-86s fronts %5.0f | bands %5.0f | walkers %5.0f GB/s | set up in %7.1f ms, holds %.2f x the pool\n", what, fronts, band, walk, setup_ms, held_x);
            fflush(stdout);
        };
        auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; };
        const int CHF = 16, NCH = F / CHF;
        {   // what jsp_pool_create does: 4 x NCH hipMalloc chunks of 16 frames, every fourth kept, frames dealt
            const double t0 = now_ms();
            std::vector<uint32_t*> chunk(NCH * 4);
            for (auto& c : chunk) CK(hipMalloc(&c, FRAME_BYTES * CHF));
            std::vector<uint32_t*> fr(F);
            for (int i = 0; i < F; ++i) fr[i] = chunk[(i % NCH) * 4] + (size_t)(i / NCH) * X * Y;
            const double t1 = now_ms();
            rates("hipMalloc chunks of 16 frames, 4 x as many as needed, every fourth, frames dealt", fr, t1 - t0, 4.0);
            for (int i = 0; i < F; ++i) fr[i] = chunk[i / CHF] + (size_t)(i % CHF) * X * Y;
            rates("  the same chunks, the first quarter of them, frames in order (neighbours)", fr, 0, 4.0);
            for (auto c : chunk) CK(hipFree(c));
        }
        {   // one hipMalloc
            const double t0 = now_ms();
            uint32_t* one;
            CK(hipMalloc(&one, FRAME_BYTES * F));
            std::vector<uint32_t*> fr(F);
            for (int i = 0; i < F; ++i) fr[i] = one + (size_t)i * X * Y;
            rates("one hipMalloc, frames back to back", fr, now_ms() - t0, 1.0);
            for (int i = 0; i < F; ++i) fr[i] = one + (size_t)((i * 17) % F) * X * Y;
            rates("  the same, frame i in slot 17 i mod F", fr, 0, 1.0);
            CK(hipFree(one));
        }
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) { printf("no virtual memory management here\n"); return 0; }
        size_t gran_rec = 0;
        (void)hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
        printf("allocation granularity: minimum %zu KB, recommended %zu KB\n", gran >> 10, gran_rec >> 10);
        // a frame takes PIECES handles of HB bytes (its stride in the VA range is rounded up to that: + 1.1 % for 2 MB handles)
        for (size_t HB : {(size_t)2 << 20, (size_t)8 << 20, (size_t)128 << 20}) {
            if (HB % gran) continue;
            const size_t per_frame = HB >= FRAME_BYTES ? 1 : (FRAME_BYTES + HB - 1) / HB;     // handles per frame (HB < frame) ...
            const size_t frames_per = HB >= FRAME_BYTES ? HB / ((FRAME_BYTES + (2u << 20) - 1) / (2u << 20) * (2u << 20)) : 1;   // ... or frames per handle, each on a 2 MB boundary
            const size_t stride = HB >= FRAME_BYTES ? HB / frames_per : per_frame * HB;       // bytes from one frame to the next in VA
            const size_t nh = HB >= FRAME_BYTES ? ((size_t)F + frames_per - 1) / frames_per : (size_t)F * per_frame;
            const size_t total = nh * HB;
            // arrangement: which handle (in creation order) backs VA slot s
            for (int arr = 0; arr < 4; ++arr) {
                if (HB >= FRAME_BYTES && arr == 3) continue;
                if (HB > ((size_t)8 << 20) && arr >= 2) continue;
                const double t0 = now_ms();
                void* va = nullptr;
                if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); break; }
                std::vector<hipMemGenericAllocationHandle_t> handles(nh);
                bool ok = true;
                for (size_t h = 0; h < nh && ok; ++h) ok = hipMemCreate(&handles[h], HB, &prop, 0) == hipSuccess;
                const double t_create = now_ms();
                // VA slot s (HB bytes) <- handle order[s]
                std::vector<size_t> order(nh);
                const size_t units = HB >= FRAME_BYTES ? nh : (size_t)F;                       // what is dealt: handles, or frames (groups of per_frame handles)
                const size_t grp = HB >= FRAME_BYTES ? 1 : per_frame;
                const size_t nchunks = units / 16 ? units / 16 : 1;
                for (size_t u = 0; u < units; ++u) {
                    size_t src = u;                                                             // arr 0: in creation order
                    if (arr == 1 || arr == 3) src = (u % nchunks) * (units / nchunks) + u / nchunks;   // arr 1: dealt — unit u lies in "chunk" u mod nchunks of the creation order
                    if (arr == 2) src = (u * 17) % units;                                       // arr 2: strided
                    for (size_t k = 0; k < grp; ++k)
                        order[u * grp + k] = arr == 3 ? k * units + src                         // arr 3: dealt, and a frame's own pieces in different quarters of the creation order
                                                      : src * grp + k;
                }
                for (size_t sidx = 0; sidx < nh && ok; ++sidx) ok = hipMemMap((char*)va + sidx * HB, HB, 0, handles[order[sidx]], 0) == hipSuccess;
                hipMemAccessDesc acc{};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                ok = ok && hipMemSetAccess(va, total, &acc, 1) == hipSuccess;
                const double t1 = now_ms();
                if (ok) {
                    std::vector<uint32_t*> fr(F);
                    for (int i = 0; i < F; ++i) fr[i] = (uint32_t*)((char*)va + (size_t)i * stride);
                    char what[160];
                    static const char* names[] = {"in creation order", "frames dealt over 16-frame stretches of the creation order", "strided (x 17)", "dealt, a frame's own pieces a quarter of the pool apart"};
                    std::snprintf(what, sizeof what, "handles of %3zu MB (%zu created in %.1f ms), %s", HB >> 20, nh, t_create - t0, names[arr]);
                    rates(what, fr, t1 - t0, (double)total / ((double)F * FRAME_BYTES));
                } else printf("mapping handles of %zu MB failed (%s)\n", HB >> 20, hipGetErrorString(hipGetLastError()));
                const double t2 = now_ms();
                (void)hipMemUnmap(va, total);
                for (auto h : handles) (void)hipMemRelease(h);
                (void)hipMemAddressFree(va, total);
                printf("      (torn down in %.1f ms)\n", now_ms() - t2);
            }
        }
        return 0;
    }
    if (getenv("LAB_TSWEEP")) {   // round 5: how the fronts shape depends on the extent a workgroup writes (T blocks = T x 64 bytes), tile-major and frame-major, on a chunked pool and on one allocation
        const int NC = F / 16 * 4;
        std::vector<uint32_t*> chunk(NC);
        for (int c = 0; c < NC; ++c) { CK(hipMalloc(&chunk[c], FRAME_BYTES * 16)); CK(hipMemset(chunk[c], 0, FRAME_BYTES * 16)); }
        uint32_t* one;
        CK(hipMalloc(&one, FRAME_BYTES * F));
        CK(hipMemset(one, 0, FRAME_BYTES * F));
        std::vector<uint32_t*> dealt, contiguous(F);
        for (int slot = 0; slot < 16; ++slot) for (int c = 0; c < F / 16; ++c) dealt.push_back(chunk[c * 4] + (size_t)slot * X * Y);
        for (int i = 0; i < F; ++i) contiguous[i] = one + (size_t)i * X * Y;
        for (int which = 0; which < 2; ++which) {
            const std::vector<uint32_t*>& fr = which ? contiguous : dealt;
            CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * F, hipMemcpyHostToDevice));
            printf("%s:", which ? "one allocation, frames back to back" : "16-frame chunks, every fourth, frames dealt");
            for (int T : {256, 512, 1024, 2048, 4096, 8192, 32768}) for (int tm : {1, 0}) {
                const int tpf = (NBLK + T - 1) / T;
                auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * F), dim3(256), 0, 0, d_table, F, T, tpf, tm); };
                launch();
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" T%d%s %4.0f |", T, tm ? "t" : "f", (double)F * FRAME_BYTES * 3 / ms / 1e6);
            }
            printf(" GB/s (t = tile-major, f = frame-major)\n");
            for (int nt : {0, 1}) {
                const int T = 8192, tpf = (NBLK + T - 1) / T;
                auto launch = [&] { hipLaunchKernelGGL(front_nt_kernel, dim3(tpf * F), dim3(256), 0, 0, d_table, F, T, tpf, nt); };
                launch();
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("   T8192 tile-major, %s stores: %4.0f GB/s\n", nt ? "nontemporal" : "plain", (double)F * FRAME_BYTES * 3 / ms / 1e6);
            }
            printf("   workgroups of 64 / 128 / 512 / 1024 lanes (T8192 tile-major):");
            for (int wg : {64, 128, 512, 1024}) {
                const int T = 8192, tpf = (NBLK + T - 1) / T;
                auto launch = [&] { hipLaunchKernelGGL(front_wg_kernel, dim3(tpf * F), dim3(wg), 0, 0, d_table, F, T); };
                launch();
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %4.0f", (double)F * FRAME_BYTES * 3 / ms / 1e6);
            }
            printf(" | one workgroup per block row, frame-major / row-major:");
            for (int rm : {0, 1}) {
                auto launch = [&] { hipLaunchKernelGGL(blockrow_kernel, dim3(F * (Y / 4)), dim3(512), 0, 0, d_table, F, rm); };
                launch();
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %4.0f", (double)F * FRAME_BYTES * 3 / ms / 1e6);
            }
            {
                auto launch = [&] { hipLaunchKernelGGL(blockrow_linear_kernel, dim3(F * (Y / 4)), dim3(512), 0, 0, d_table, F); };
                launch();
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" | the block row's 30 KB written linearly: %4.0f", (double)F * FRAME_BYTES * 3 / ms / 1e6);
            }
            printf(" GB/s\n");
        }
        {
            const size_t n16 = FRAME_BYTES * F / 16;
            auto fill = [&] { hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (u32x4*)one, n16); };
            fill();
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventRecord(e0));
            for (int i = 0; i < 3; ++i) fill();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("plain fill %4.0f GB/s\n", (double)F * FRAME_BYTES * 3 / ms / 1e6);
            printf("linear fills of the one allocation, lanes x stores per lane (KB per workgroup):");
            for (int L : {64, 256, 512, 1024}) for (int spl : {1, 2, 4, 8, 16}) {
                const size_t per = (size_t)L * spl;
                const unsigned grid = (unsigned)((n16 + per - 1) / per);
                auto go = [&] { hipLaunchKernelGGL(linear_fill_kernel, dim3(grid), dim3(L), 0, 0, (u32x4*)one, n16, spl); };
                go();
                CK(hipDeviceSynchronize());
                float ms2 = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) go();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms2, e0, e1));
                printf(" %dx%d(%zuK) %4.0f |", L, spl, per * 16 / 1024, (double)F * FRAME_BYTES * 3 / ms2 / 1e6);
            }
            printf(" GB/s\n");
        }
        return 0;
    }
    if (argc > 2 && getenv("LAB_SUBSETS")) {   // round 5: is "fast" a property of WHICH frames are written together?  2F single-frame allocations (or chunks of LAB_CH frames), random F-subsets, per-frame scores
        const int CH = getenv("LAB_CH") ? atoi(getenv("LAB_CH")) : 1, OVER = atoi(argv[2]);      // OVER x F frames allocated
        const int NA = F * OVER / CH;                                                              // allocations
        std::vector<uint32_t*> alloc(NA);
        for (int a = 0; a < NA; ++a) { CK(hipMalloc(&alloc[a], FRAME_BYTES * CH)); }
        for (int a = 0; a < NA; ++a) CK(hipMemset(alloc[a], 0, FRAME_BYTES * CH));
        const int T = 8192, tpf = (NBLK + T - 1) / T;
        auto rate = [&](const std::vector<uint32_t*>& fr) {
            const int n = (int)fr.size();
            CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * n, hipMemcpyHostToDevice));
            auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * n), dim3(256), 0, 0, d_table, n, T, tpf, 1); };
            launch();
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventRecord(e0));
            for (int i = 0; i < 2; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            return (double)n * FRAME_BYTES * 2 / ms / 1e6;
        };
        const int need = F / CH, NS = 96;
        std::vector<double> score(NA, 0.0), cnt(NA, 0.0), rates;
        double total = 0;
        unsigned long long seed = 777;
        auto frames_of = [&](const std::vector<int>& ids) {
            std::vector<uint32_t*> fr;
            for (int slot = 0; slot < CH; ++slot) for (int id : ids) fr.push_back(alloc[id] + (size_t)slot * X * Y);   // dealt over the allocations
            return fr;
        };
        for (int t = 0; t < NS; ++t) {
            std::vector<int> all(NA), ids;
            for (int i = 0; i < NA; ++i) all[i] = i;
            for (int q = 0; q < need; ++q) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; const int j = q + (int)((seed >> 33) % (unsigned)(NA - q)); std::swap(all[q], all[j]); ids.push_back(all[q]); }
            const double r = rate(frames_of(ids));
            rates.push_back(r);
            total += r;
            for (int id : ids) { score[id] += r; cnt[id] += 1; }
        }
        std::vector<double> sorted = rates;
        std::sort(sorted.begin(), sorted.end());
        printf("%d allocations of %d frame(s), %d random subsets of %d: min %.0f | 10%% %.0f | median %.0f | 90%% %.0f | max %.0f GB/s\n", NA, CH, NS, need, sorted.front(), sorted[NS / 10], sorted[NS / 2], sorted[NS * 9 / 10], sorted.back());
        // per-allocation score: mean rate of the subsets it was in, against the overall mean
        std::vector<int> order(NA);
        for (int i = 0; i < NA; ++i) { order[i] = i; score[i] = cnt[i] > 0 ? score[i] / cnt[i] : total / NS; }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return score[a] > score[b]; });
        std::vector<int> top(order.begin(), order.begin() + need), bottom(order.end() - need, order.end());
        printf("scores: best allocation %.0f, worst %.0f (mean %.0f)\n", score[order.front()], score[order.back()], total / NS);
        printf("the %d best-scored allocations as one pool: %.0f GB/s | the %d worst: %.0f GB/s | allocation order [0, %d): %.0f | every %d-th: %.0f\n", need, rate(frames_of(top)), need, rate(frames_of(bottom)), need,
               rate(frames_of([&] { std::vector<int> v; for (int i = 0; i < need; ++i) v.push_back(i); return v; }())), OVER, rate(frames_of([&] { std::vector<int> v; for (int i = 0; i < need; ++i) v.push_back(i * OVER); return v; }())));
        // a second round: subsets drawn from the better-scored two thirds only
        {
            const int pool2 = NA * 2 / 3 > need ? NA * 2 / 3 : need;
            std::vector<double> r2;
            for (int t = 0; t < 24; ++t) {
                std::vector<int> all(order.begin(), order.begin() + pool2), ids;
                for (int q = 0; q < need; ++q) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; const int j = q + (int)((seed >> 33) % (unsigned)(pool2 - q)); std::swap(all[q], all[j]); ids.push_back(all[q]); }
                r2.push_back(rate(frames_of(ids)));
            }
            std::sort(r2.begin(), r2.end());
            printf("24 random subsets of the better-scored two thirds: min %.0f | median %.0f | max %.0f GB/s\n", r2.front(), r2[12], r2.back());
        }
        return 0;
    }
    if (argc > 2 && getenv("LAB_SPREAD")) {   // round 5: pools put together from separately allocated chunks of CH frames: neighbours, every g-th, random
        const int NC = atoi(argv[2]), CH = getenv("LAB_CH") ? atoi(getenv("LAB_CH")) : 64, NCH = F / CH;
        std::vector<uint32_t*> chunk(NC);
        for (int c = 0; c < NC; ++c) { CK(hipMalloc(&chunk[c], FRAME_BYTES * CH)); CK(hipMemset(chunk[c], 0, FRAME_BYTES * CH)); }
        printf("%d chunks of %d frames, first at %p, second at %p, last at %p\n", NC, CH, (void*)chunk[0], (void*)chunk[1], (void*)chunk[NC - 1]);
        const int T = 8192, tpf = (NBLK + T - 1) / T;
        auto rate = [&](const std::vector<uint32_t*>& fr) {
            const int n = (int)fr.size();
            CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * n, hipMemcpyHostToDevice));
            auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * n), dim3(256), 0, 0, d_table, n, T, tpf, 1); };
            launch();
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 2; ++rep) {
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 3 < best) best = ms / 3;
            }
            return (double)n * FRAME_BYTES / best / 1e6;
        };
        auto timed = [&](auto&& launch, double bytes) {
            launch();
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 2; ++rep) {
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 3 < best) best = ms / 3;
            }
            return bytes / best / 1e6;
        };
        auto shapes = [&](const std::vector<uint32_t*>& fr) {   // "fronts / band-walkers (256 frames) / frame-walkers (299 frames)"
            static char buf[160];
            const double a = rate(fr);
            CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * fr.size(), hipMemcpyHostToDevice));
            const int nb = 256 < (int)fr.size() ? 256 : (int)fr.size(), ng = 299 < (int)fr.size() ? 299 : (int)fr.size();
            const double b = timed([&] { hipLaunchKernelGGL(band_table_kernel, dim3((unsigned)(nb * 12 * 8)), dim3(64), 4608, 0, d_table, nb, 90, 12, 20); }, (double)nb * FRAME_BYTES);
            const double c = timed([&] { hipLaunchKernelGGL(group_table_kernel, dim3(15, 68), dim3(256), 0, 0, d_table, ng); }, (double)ng * FRAME_BYTES);
            if (getenv("LAB_BANDWG")) {
                const double b8 = timed([&] { hipLaunchKernelGGL(band_wg_table_kernel, dim3((unsigned)(nb * 12)), dim3(512), 0, 0, d_table, nb, 90, 12, 20, 0); }, (double)nb * FRAME_BYTES);
                const double b8s = timed([&] { hipLaunchKernelGGL(band_wg_table_kernel, dim3((unsigned)(nb * 12)), dim3(512), 0, 0, d_table, nb, 90, 12, 20, 1); }, (double)nb * FRAME_BYTES);
                const double b8t = timed([&] { hipLaunchKernelGGL(band_wg_table_kernel, dim3((unsigned)(nb * 36)), dim3(512), 0, 0, d_table, nb, 30, 36, 20, 0); }, (double)nb * FRAME_BYTES);
                std::snprintf(buf, sizeof buf, " %4.0f/%4.0f[wg8 %4.0f sync %4.0f 30rows %4.0f]/%4.0f", a, b, b8, b8s, b8t, c);
                return buf;
            }
            std::snprintf(buf, sizeof buf, " %4.0f/%4.0f/%4.0f", a, b, c);
            return buf;
        };
        auto pool_of = [&](const std::vector<int>& ids) {
            std::vector<uint32_t*> fr;
            for (int id : ids) for (int i = 0; i < CH; ++i) fr.push_back(chunk[id] + (size_t)i * X * Y);
            return fr;
        };
        printf("(fronts / band-walkers / frame-walkers, GB/s)\n");
        if (getenv("LAB_PERM")) {   // ONE allocation per pool, frames back to back, but taken in another ORDER: frame i lies in slot (i x K) mod F
            const int P = 3;
            for (int k = 0; k < P; ++k) {
                uint32_t* pool;
                CK(hipMalloc(&pool, FRAME_BYTES * F));
                CK(hipMemset(pool, 0, FRAME_BYTES * F));
                printf("one allocation %d, frame i in slot (i x K) mod %d:", k, F);
                for (int K : {1, 3, 7, 17, 37, 65, 101, 171, 255}) {
                    std::vector<uint32_t*> fr(F);
                    for (int i = 0; i < F; ++i) fr[i] = pool + (size_t)((i * K) % F) * X * Y;
                    printf(" K=%d%s |", K, shapes(fr));
                }
                printf(" GB/s\n");
                fflush(stdout);
                // (kept allocated: the next pool lies elsewhere)
            }
            return 0;
        }
        printf("neighbouring chunks [a, a + %d):", NCH);
        for (int a = 0; a + NCH <= NC; a += NCH) { std::vector<int> ids; for (int q = 0; q < NCH; ++q) ids.push_back(a + q); printf("%s", shapes(pool_of(ids))); }
        printf(" GB/s\n");
        const int g = NC / NCH;
        printf("every %d-th chunk, from a:", g);
        for (int a = 0; a < g; ++a) { std::vector<int> ids; for (int q = 0; q < NCH; ++q) ids.push_back(a + q * g); printf("%s", shapes(pool_of(ids))); }
        printf(" GB/s\n");
        printf("random chunks:");
        unsigned long long seed = 12345;
        for (int t = 0; t < g; ++t) {
            std::vector<int> all(NC), ids;
            for (int c = 0; c < NC; ++c) all[c] = c;
            for (int q = 0; q < NCH; ++q) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; const int j = q + (int)((seed >> 33) % (unsigned)(NC - q)); std::swap(all[q], all[j]); ids.push_back(all[q]); }
            printf("%s", shapes(pool_of(ids)));
        }
        printf(" GB/s\n");
        // frames dealt round-robin over ALL chunks: frame i in chunk i %% NC (each chunk holds only F / NC of the pool's frames)
        {
            std::vector<uint32_t*> fr;
            for (int i = 0; i < F; ++i) fr.push_back(chunk[i % NC] + (size_t)(i / NC) * X * Y);
            printf("frames dealt round-robin over all %d chunks: %s GB/s\n", NC, shapes(fr));
        }
        return 0;
    }
    if (argc > 2 && getenv("LAB_CHUNKS")) {   // round 5: is a slow pool slow EVERYWHERE?  Chunks of 64 frames probed one by one, then a pool put together from the fastest chunks of all pools
        const int K = atoi(argv[2]), CH = 64, NCH = F / CH;
        std::vector<uint32_t*> pools(K);
        for (int k = 0; k < K; ++k) { CK(hipMalloc(&pools[k], FRAME_BYTES * F)); CK(hipMemset(pools[k], 0, FRAME_BYTES * F)); }
        const int T = 8192, tpf = (NBLK + T - 1) / T;
        auto rate = [&](const std::vector<uint32_t*>& fr) {
            const int n = (int)fr.size();
            CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * n, hipMemcpyHostToDevice));
            auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(tpf * n), dim3(256), 0, 0, d_table, n, T, tpf, 1); };
            launch();
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int rep = 0; rep < 2; ++rep) {
                float ms = 0;
                CK(hipEventRecord(e0));
                for (int i = 0; i < 3; ++i) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 3 < best) best = ms / 3;
            }
            return (double)n * FRAME_BYTES / best / 1e6;
        };
        struct Chunk { double r; int k, c; };
        std::vector<Chunk> chunks;
        for (int k = 0; k < K; ++k) {
            std::vector<uint32_t*> all(F);
            for (int i = 0; i < F; ++i) all[i] = pools[k] + (size_t)i * X * Y;
            printf("pool %d: whole (512 fronts) %5.0f | chunks of %d frames:", k, rate(all), CH);
            for (int c = 0; c < NCH; ++c) {
                std::vector<uint32_t*> fr(all.begin() + c * CH, all.begin() + (c + 1) * CH);
                const double r = rate(fr);
                chunks.push_back({r, k, c});
                printf(" %5.0f", r);
            }
            // interleaved: every 8th frame (64 frames spread over the whole pool)
            printf(" | every 8th frame:");
            for (int o = 0; o < 2; ++o) {
                std::vector<uint32_t*> fr;
                for (int i = o; i < F; i += 8) fr.push_back(all[i]);
                printf(" %5.0f", rate(fr));
            }
            printf(" GB/s\n");
            fflush(stdout);
        }
        std::sort(chunks.begin(), chunks.end(), [](const Chunk& a, const Chunk& b) { return a.r > b.r; });
        for (int pick = 0; pick < 2; ++pick) {    // the fastest NCH chunks, then the slowest
            std::vector<uint32_t*> fr;
            printf("%s %d chunks:", pick ? "slowest" : "fastest", NCH);
            for (int q = 0; q < NCH; ++q) {
                const Chunk& ch = pick ? chunks[chunks.size() - 1 - q] : chunks[q];
                printf(" p%dc%d(%.0f)", ch.k, ch.c, ch.r);
                for (int i = 0; i < CH; ++i) fr.push_back(pools[ch.k] + (size_t)(ch.c * CH + i) * X * Y);
            }
            printf(" -> as one pool of %d fronts: %5.0f GB/s\n", F, rate(fr));
        }
        return 0;
    }
    if (argc > 2) {   // several pools in ONE process, none freed before the last is measured: do they differ?
        const int K = atoi(argv[2]);
        uint32_t* sink;
        CK(hipMalloc(&sink, 64));
        std::vector<uint32_t*> pools(K);
        const int clear = argc > 3 ? atoi(argv[3]) : 1;
        for (int k = 0; k < K; ++k) { CK(hipMalloc(&pools[k], FRAME_BYTES * F)); if (clear) CK(hipMemset(pools[k], 0, FRAME_BYTES * F)); }
        printf("%d pools of %d frames, %s\n", K, F, clear ? "cleared with hipMemset first" : "not touched before the first launch");
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < K; ++k) {
                std::vector<uint32_t*> fr(F);
                for (int i = 0; i < F; ++i) fr[i] = pools[k] + (size_t)i * X * Y;
                CK(hipMemcpy(d_table, fr.data(), sizeof(uint32_t*) * F, hipMemcpyHostToDevice));
                const int T = 8192, tpf = (NBLK + T - 1) / T;
                float ms = 0, by_order[4] = {0, 0, 0, 0};
                for (int order : {1, 0, 2, 3}) {
                    const int grid = order == 2 ? (tpf + 64) * F : tpf * F;
                    auto launch = [&] { hipLaunchKernelGGL(front_kernel, dim3(grid), dim3(256), 0, 0, d_table, F, T, tpf, order); };
                    launch();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 3; ++i) launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&by_order[order], e0, e1));
                }
                ms = by_order[1];
                float fill_ms = 0;
                {
                    const size_t n16 = FRAME_BYTES * F / 16;
                    auto fill = [&] { hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (u32x4*)pools[k], n16); };
                    fill();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 3; ++i) fill();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&fill_ms, e0, e1));
                }
                float band_ms = 0;
                {
                    const int B = 90, bands = (Y + B - 1) / B;
                    auto go = [&] { hipLaunchKernelGGL(band_kernel, dim3((unsigned)(F * bands * 8)), dim3(64), 4608, 0, pools[k], F, B, bands, 20); };
                    go();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 3; ++i) go();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&band_ms, e0, e1));
                }
                float group_ms[2] = {0, 0};
                for (int wide = 0; wide < 2; ++wide) {
                    const int GF = F < 299 ? F : 299;
                    auto go = [&] { hipLaunchKernelGGL(group_kernel, dim3(wide ? 8 : 15, 68), dim3(256), 0, 0, pools[k], GF, wide); };
                    go();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 3; ++i) go();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&group_ms[wide], e0, e1));
                    group_ms[wide] = (float)((double)GF * FRAME_BYTES * 3 / group_ms[wide] / 1e6);
                }
                printf("pool %d: group shape (299 frames per workgroup): 8 blocks x 16 rows, 2 x 512 B per wave %5.0f | 16 blocks x 16 rows, 1 KB per wave %5.0f GB/s\n", k, group_ms[0], group_ms[1]);
                {   // round 5: full-width strips (R rows x 7 680 B contiguous per workgroup and frame), next to the group shapes above
                    const int GF = F < 299 ? F : 299;
                    struct Shape { int R, P, WG, linear; };
                    const Shape shapes[] = {{1, 1, 512, 0}, {2, 1, 512, 0}, {4, 1, 512, 0}, {8, 1, 512, 0}, {16, 1, 512, 0}, {4, 1, 256, 0}, {4, 1, 1024, 1}, {2, 1, 256, 1},
                                            {4, 2, 256, 0}, {8, 2, 256, 0}, {16, 2, 256, 0}, {16, 4, 128, 0}, {16, 15, 64, 0}};
                    printf("pool %d: strips (R rows x 7680/P bytes per workgroup, %d frames each; R.P.WG[l = linear run]):", k, GF);
                    for (const Shape& sh : shapes) {
                        const int strips = (Y + sh.R - 1) / sh.R;
                        auto go = [&] { hipLaunchKernelGGL(strip_kernel, dim3(strips * sh.P), dim3(sh.WG), 0, 0, pools[k], GF, sh.R, sh.P, sh.linear); };
                        go();
                        CK(hipDeviceSynchronize());
                        float ms = 0;
                        CK(hipEventRecord(e0));
                        for (int i = 0; i < 3; ++i) go();
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        printf(" %d.%d.%d%s %4.0f |", sh.R, sh.P, sh.WG, sh.linear ? "l" : "", (double)GF * FRAME_BYTES * 3 / ms / 1e6);
                    }
                    printf(" GB/s\n");
                }
                {   // round 5: the group shape in step / with fresh workgroups
                    const int GF = F < 299 ? F : 299;
                    static unsigned* counter = nullptr;
                    static unsigned base = 0;
                    if (!counter) { CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64)); }
                    printf("pool %d: group shape kept in step (arrive + wait every n frames): ", k);
                    for (int se : {0, 1, 4, 16, 64}) {
                        const int rounds = se ? (GF - 1) / se : 0;
                        auto go = [&] { hipLaunchKernelGGL(group_sync_kernel, dim3(15, 68), dim3(256), 0, 0, pools[k], GF, se, counter, base); base += (unsigned)rounds * 15u * 68u; };
                        go();
                        CK(hipDeviceSynchronize());
                        float ms = 0;
                        CK(hipEventRecord(e0));
                        for (int i = 0; i < 3; ++i) go();
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        printf(" n=%d %4.0f |", se, (double)GF * FRAME_BYTES * 3 / ms / 1e6);
                    }
                    printf(" fresh workgroups per chunk of C frames: ");
                    for (int C : {1, 2, 4, 16, 64}) {
                        auto go = [&] { hipLaunchKernelGGL(group_fresh_kernel, dim3(15, 68, (GF + C - 1) / C), dim3(256), 0, 0, pools[k], GF, C); };
                        go();
                        CK(hipDeviceSynchronize());
                        float ms = 0;
                        CK(hipEventRecord(e0));
                        for (int i = 0; i < 3; ++i) go();
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        printf(" C=%d %4.0f |", C, (double)GF * FRAME_BYTES * 3 / ms / 1e6);
                    }
                    printf(" GB/s\n");
                }
                if (getenv("LAB_PIECES2")) {   // round 5: FEW, FAT frame-walking workgroups (rows x bytes . lanes per workgroup)
                    const int GF = F < 299 ? F : 299;
                    struct Sh { int PR, PB, WG; };
                    const Sh shapes[] = {{16, 512, 256}, {16, 3840, 256}, {16, 3840, 512}, {16, 3840, 1024}, {8, 7680, 256}, {8, 7680, 512}, {8, 7680, 1024}, {16, 7680, 512},
                                         {16, 7680, 1024}, {32, 3840, 512}, {32, 3840, 1024}, {32, 7680, 1024}, {24, 7680, 1024}, {48, 7680, 1024}, {12, 7680, 512}, {4, 7680, 256}, {4, 7680, 128}, {2, 7680, 64}};
                    printf("pool %d: few fat walkers (rows x bytes . lanes, workgroups):", k);
                    for (const Sh& sh : shapes) {
                        const int ppr = (X * 4 + sh.PB - 1) / sh.PB, prow = (Y + sh.PR - 1) / sh.PR;
                        auto go = [&] { hipLaunchKernelGGL(piece_kernel, dim3(ppr * prow), dim3(sh.WG), 0, 0, pools[k], GF, sh.PR, sh.PB, ppr); };
                        go();
                        CK(hipDeviceSynchronize());
                        float ms = 0;
                        CK(hipEventRecord(e0));
                        for (int i = 0; i < 3; ++i) go();
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        printf(" %dx%d.%d(%dwg) %4.0f |", sh.PR, sh.PB, sh.WG, ppr * prow, (double)GF * FRAME_BYTES * 3 / ms / 1e6);
                    }
                    printf(" GB/s\n");
                    fflush(stdout);
                }
                if (getenv("LAB_PIECES")) {   // round 5: piece geometries for frame-walking workgroups
                    const int GF = F < 299 ? F : 299;
                    printf("pool %d: walkers by piece (rows x bytes per workgroup and frame, %d frames):", k, GF);
                    for (int PB : {512, 1536, 2560, 3840, 7680})
                        for (int PR : {1, 2, 4, 8, 16, 32}) {
                            const int L = PB / 16, n = PR * L;
                            if (n < 128 || n > 256 * 16) continue;
                            const int ppr = (X * 4 + PB - 1) / PB, prow = (Y + PR - 1) / PR;
                            auto go = [&] { hipLaunchKernelGGL(piece_kernel, dim3(ppr * prow), dim3(256), 0, 0, pools[k], GF, PR, PB, ppr); };
                            go();
                            CK(hipDeviceSynchronize());
                            float ms = 0;
                            CK(hipEventRecord(e0));
                            for (int i = 0; i < 3; ++i) go();
                            CK(hipEventRecord(e1));
                            CK(hipEventSynchronize(e1));
                            CK(hipEventElapsedTime(&ms, e0, e1));
                            printf(" %dx%d(%dwg) %4.0f |", PR, PB, ppr * prow, (double)GF * FRAME_BYTES * 3 / ms / 1e6);
                        }
                    printf(" GB/s\n");
                    fflush(stdout);
                }
                printf("pool %d: plain fill %5.0f | band-walking waves %5.0f | ", k, (double)F * FRAME_BYTES * 3 / fill_ms / 1e6, (double)F * FRAME_BYTES * 3 / band_ms / 1e6);
                printf("tile-major %5.0f | frame-major %5.0f | staggered %5.0f | scattered %5.0f GB/s\n", (double)F * FRAME_BYTES * 3 / by_order[1] / 1e6,
                       (double)F * FRAME_BYTES * 3 / by_order[0] / 1e6, (double)F * FRAME_BYTES * 3 / by_order[2] / 1e6, (double)F * FRAME_BYTES * 3 / by_order[3] / 1e6);
                float pr[2];
                int q = 0;
                for (size_t page : {(size_t)4096, (size_t)2 << 20}) {
                    const size_t npages = FRAME_BYTES * F / page;
                    auto probe = [&] { hipLaunchKernelGGL(page_probe_kernel, dim3(4096), dim3(256), 0, 0, pools[k], npages, page / 4, 64, sink); };
                    probe();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0));
                    probe();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&pr[q++], e0, e1));
                }
                printf("pool %d at %p: stores tile-major %7.0f GB/s | 67 M scattered 4-byte reads: one per 4 KB page %7.1f us, one per 2 MB %7.1f us\n", k, (void*)pools[k],
                       (double)F * FRAME_BYTES * 3 / ms / 1e6, pr[0] * 1000, pr[1] * 1000);
                fflush(stdout);
            }
        return 0;
    }
    {   // A
        uint32_t* pool;
        CK(hipMalloc(&pool, FRAME_BYTES * F));
        std::vector<uint32_t*> fr(F);
        for (int i = 0; i < F; ++i) fr[i] = pool + (size_t)i * X * Y;
        measure("A one allocation, frames back to back", fr);
        CK(hipFree(pool));
    }
    {   // B
        std::vector<uint32_t*> fr(F);
        for (int i = 0; i < F; ++i) CK(hipMalloc(&fr[i], FRAME_BYTES));
        measure("B one hipMalloc per frame", fr);
        printf("   (frame 0 at %p, 1 at %p, 2 at %p)\n", (void*)fr[0], (void*)fr[1], (void*)fr[2]);
        for (auto* p : fr) CK(hipFree(p));
    }
    {   // C
        std::vector<uint32_t*> fr(F), segs;
        for (int i = 0; i < F; i += 2) {
            uint32_t* s;
            CK(hipMalloc(&s, 20u << 20));
            segs.push_back(s);
            fr[i] = s;
            if (i + 1 < F) fr[i + 1] = s + (size_t)X * Y;
        }
        measure("C two frames per 20 MB hipMalloc", fr);
        printf("   (segment 0 at %p, 1 at %p, 2 at %p)\n", (void*)segs[0], (void*)segs[1], (void*)segs[2]);
        for (auto* p : segs) CK(hipFree(p));
    }
    {   // D
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { printf("D: no virtual memory management here\n"); return 0; }
        for (size_t chunk : {(size_t)256 << 20, (size_t)1024 << 20}) {
            chunk = (chunk + gran - 1) / gran * gran;
            const size_t total = (FRAME_BYTES * F + chunk - 1) / chunk * chunk;
            void* va = nullptr;
            if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("D: reserve failed\n"); break; }
            std::vector<hipMemGenericAllocationHandle_t> handles;
            bool ok = true;
            for (size_t off = 0; off < total && ok; off += chunk) {
                hipMemGenericAllocationHandle_t h;
                ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess && hipMemMap((char*)va + off, chunk, 0, h, 0) == hipSuccess;
                if (ok) handles.push_back(h);
            }
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            ok = ok && hipMemSetAccess(va, total, &acc, 1) == hipSuccess;
            if (ok) {
                std::vector<uint32_t*> fr(F);
                for (int i = 0; i < F; ++i) fr[i] = (uint32_t*)va + (size_t)i * X * Y;
                char what[96];
                std::snprintf(what, sizeof what, "D one VA range over physical chunks of %zu MB (granularity %zu KB)", chunk >> 20, gran >> 10);
                measure(what, fr);
            } else printf("D: mapping chunks of %zu MB failed\n", chunk >> 20);
            (void)hipMemUnmap(va, total);
            for (auto h : handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(va, total);
        }
    }
    return 0;
}
