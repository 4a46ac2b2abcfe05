// lab: does it matter WHERE the destination frames lie?  The store shape of msv1_fused_kernel on all-solid frames — a workgroup per
// (frame, tile of T blocks), tiles handed out tile-major (tile j of every frame, then tile j + 1 ...: up to F write fronts at once),
// lane = 4x4 block, four 16-byte row stores per block — with the frames
//   A  back to back in one allocation,          B  one hipMalloc per frame,
//   C  two frames per 20 MB hipMalloc (what torch's caching allocator does with 8.3 MB tensors),
//   D  one VA range backed by separately created 2 MB-granular physical chunks (hipMemCreate / hipMemMap), frames back to back.
//   hipcc -O3 --offload-arch=gfx950 tools/front_lab.hip -o tools/front_lab.bin && tools/front_lab.bin [frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
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
