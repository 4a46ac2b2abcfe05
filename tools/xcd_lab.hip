// lab: why do workgroups that KEEP storing get less from the memory system than workgroups that store once and leave?
// (profiles/archive/r03_sp_store_lab.txt: the same 2.1 GB written in 4 KB pieces — 6.9 TB/s with one piece per workgroup, 5.3 with 2 025
// workgroups that loop.)  One explanation is a static split: a looping workgroup owns 1/N of the bytes whatever its CU / XCD gets
// from memory, so the slowest finishes last, while fresh workgroups go wherever a slot frees up.  This tool writes the same buffer
//   mode 0  one 4 KB piece per workgroup, address order                       (the plain fill)
//   mode 1  N looping workgroups, piece = wg + k * N                           (static split, grid-stride)
//   mode 2  N looping workgroups, pieces handed out by an atomic counter       (dynamic split, same workgroups)
//   mode 3  N looping workgroups, each owns a CONTIGUOUS run of pieces         (static split, blocked)
//   mode 4  N looping workgroups, runs of 16 pieces (64 KB) handed out by an atomic counter   (dynamic split, coarse: one atomic per 64 KB)
//   mode 5  one run of 16 pieces per workgroup, address order                   (fresh workgroups, 64 KB each)
//   mode 6  as 4, the next ticket asked for before the current run is written   (the ticket's round trip hidden)
//   mode 7  one piece per workgroup, but the eight dispatch classes (workgroup id mod 8 = the XCD the hardware deals it to) own
//           UNEQUAL contiguous shares of the buffer, sized from the classes' end times in the pass before (three rounds): the XCDs
//           that get less from memory get fewer pieces, surplus workgroups of a class leave at once
// and has every workgroup note its XCC id, CU id and the clock when it began and ended; the summary is per XCD.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_lab.hip -o tools/xcd_lab.bin && tools/xcd_lab.bin [MB] [N]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Note { unsigned long long t0, t1; uint32_t xcc, cu, pieces, pad; };

__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }
__device__ __forceinline__ uint32_t hw_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }

struct Shares { uint32_t first[8], count[8]; };
__global__ __launch_bounds__(256) void lab_kernel(u32x4* __restrict__ dst, size_t npieces, int mode, unsigned long long* __restrict__ counter, Note* __restrict__ notes, Shares sh) {
    __shared__ unsigned long long s_next;
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    uint32_t done = 0;
    const size_t N = gridDim.x;
    auto piece = [&](size_t p) {
        *(gu32x4*)(dst + p * 256 + threadIdx.x) = u32x4{(uint32_t)p, 1u, 2u, 3u};
        ++done;
    };
    if (mode == 0) {
        if (blockIdx.x < npieces) piece(blockIdx.x);
    } else if (mode == 1) {
        for (size_t p = blockIdx.x; p < npieces; p += N) piece(p);
    } else if (mode == 3) {
        const size_t per = (npieces + N - 1) / N, lo = blockIdx.x * per, hi = lo + per < npieces ? lo + per : npieces;
        for (size_t p = lo; p < hi; ++p) piece(p);
    } else if (mode == 7) {
        const uint32_t c = blockIdx.x & 7u, j = blockIdx.x >> 3;
        if (j < sh.count[c]) piece((size_t)sh.first[c] + j);
    } else if (mode == 5) {
        for (size_t p = (size_t)blockIdx.x * 16; p < (size_t)blockIdx.x * 16 + 16 && p < npieces; ++p) piece(p);
    } else if (mode == 4 || mode == 6) {
        const size_t nruns = (npieces + 15) / 16;
        if (threadIdx.x == 0) s_next = atomicAdd(counter, 1ull);
        __syncthreads();
        size_t run = s_next;
        while (run < nruns) {
            __syncthreads();
            if (mode == 6 && threadIdx.x == 0) s_next = atomicAdd(counter, 1ull);      // (returns while the run below is being written)
            for (size_t p = run * 16; p < run * 16 + 16 && p < npieces; ++p) piece(p);
            if (mode == 4 && threadIdx.x == 0) s_next = atomicAdd(counter, 1ull);
            __syncthreads();
            run = s_next;
        }
    } else {
        for (;;) {
            if (threadIdx.x == 0) s_next = atomicAdd(counter, 1ull);
            __syncthreads();
            const size_t p = s_next;
            __syncthreads();
            if (p >= npieces) break;
            piece(p);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        const uint32_t h = hw_id();
        notes[blockIdx.x] = Note{t0, (unsigned long long)wall_clock64(), xcc_id(), ((h >> 8) & 0xFu) | (((h >> 13) & 0x7u) << 4), done, 0u};   // cu_id | se_id << 4
    }
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 2048;
    const int N = argc > 2 ? atoi(argv[2]) : 2048;
    const size_t bytes = mb << 20, npieces = bytes / 4096;
    u32x4* buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    unsigned long long* counter;
    CK(hipMalloc(&counter, 8));
    Note* d_notes;
    const size_t maxwg = 2 * std::max<size_t>(npieces, (size_t)N);
    CK(hipMalloc(&d_notes, sizeof(Note) * maxwg));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("%zu MB in 4 KB pieces, %d looping workgroups, wall clock %d kHz\n", mb, N, clk_khz);
    const char* names[8] = {"one piece per workgroup", "looping, grid-stride", "looping, atomic queue", "looping, contiguous runs", "looping, queue of 64 KB runs",
                            "one 64 KB run per workgroup", "looping, 64 KB runs, ticket ahead", "one piece per workgroup, XCD shares"};
    double share[8], class_end[8];
    Shares sh{};
    for (int c = 0; c < 8; ++c) { share[c] = 0.125; class_end[c] = 1; }
    for (int pass = 0; pass < 2; ++pass)
        for (int mode : {0, 1, 3, 7, 7, 7, 7, 0}) {
            if (mode == 0) for (int c = 0; c < 8; ++c) share[c] = 1.0 / 8;
            if (mode == 7) {   // new shares from the end times of the launch before: a class that ended late gets less
                double sum = 0;
                for (int c = 0; c < 8; ++c) { share[c] = share[c] / class_end[c]; sum += share[c]; }
                uint32_t at = 0;
                for (int c = 0; c < 8; ++c) {
                    sh.first[c] = at;
                    sh.count[c] = c == 7 ? (uint32_t)npieces - at : (uint32_t)(share[c] / sum * npieces);
                    share[c] = (double)sh.count[c] / npieces;
                    at += sh.count[c];
                }
            }
            uint32_t most = 0;
            for (int c = 0; c < 8; ++c) most = std::max(most, sh.count[c]);
            const unsigned grid = mode == 0 ? (unsigned)npieces : mode == 7 ? most * 8u : mode == 5 ? (unsigned)((npieces + 15) / 16) : (unsigned)N;
            if (grid > maxwg) { printf("shares too far apart\n"); return 1; }
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(counter, 0, 8));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(lab_kernel, dim3(grid), dim3(256), 0, 0, buf, npieces, mode, counter, d_notes, sh);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("mode %d %-26s | %8.1f us %7.0f GB/s", mode, names[mode], best * 1000, (double)bytes / best / 1e6);
            std::vector<Note> notes(grid);
            CK(hipMemcpy(notes.data(), d_notes, sizeof(Note) * grid, hipMemcpyDeviceToHost));
            unsigned long long start = ~0ull, end = 0;
            for (const Note& n : notes) { start = std::min(start, n.t0); end = std::max(end, n.t1); }
            // per XCD: workgroups, pieces, when its last workgroup ended (us after the first began), the span of its workgroups' end times
            printf(" | per XCD (workgroups, pieces, last end us):");
            for (int x = 0; x < 8; ++x) {
                unsigned long long last = 0, first_end = ~0ull, pieces = 0;
                int wgs = 0;
                for (const Note& n : notes)
                    if ((int)n.xcc == x) { ++wgs; pieces += n.pieces; last = std::max(last, n.t1); first_end = std::min(first_end, n.t1); }
                if (wgs) printf(" [%d %llu %.0f]", wgs, pieces, (double)(last - start) * 1e3 / clk_khz);
            }
            if (mode == 0 || mode == 7) {   // per dispatch class (workgroup id mod 8): is it one XCD, when did it end, what share did it have
                printf(" | per class (xcc, share %%, last end us):");
                for (int c = 0; c < 8; ++c) {
                    unsigned long long last = 0;
                    int xcc = -1;
                    bool one = true;
                    for (size_t g = c; g < notes.size(); g += 8) {
                        if (notes[g].pieces == 0) continue;
                        if (xcc < 0) xcc = (int)notes[g].xcc;
                        one = one && xcc == (int)notes[g].xcc;
                        last = std::max(last, notes[g].t1);
                    }
                    class_end[c] = (double)(last - start) * 1e3 / clk_khz;
                    printf(" [%d%s %.2f %.0f]", xcc, one ? "" : "!", 100.0 * (mode == 7 ? share[c] : 0.125), class_end[c]);
                }
            }
            if (mode != 0 && mode != 5 && mode != 7) {   // how far apart do the looping workgroups finish?
                std::vector<double> ends;
                for (const Note& n : notes) ends.push_back((double)(n.t1 - start) * 1e3 / clk_khz);
                std::sort(ends.begin(), ends.end());
                printf(" | workgroup end times us: min %.0f median %.0f p90 %.0f max %.0f", ends.front(), ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back());
            }
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
