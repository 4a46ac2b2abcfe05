// Kernel lab for the MSVideo1 block kernel: variants and ablations timed side by side on
// synthetic M1 streams (25 % solid / 50 % 2-colour / 25 % 8-colour), 64 frames of 1920x1080.
// Not part of the product; winners are ported into jsplayer_amd/csrc/msv1_kernels.hip.
//   hipcc -O3 --offload-arch=gfx950 tools/msv1_lab.hip -o tools/msv1_lab.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct FrameArgs { uint32_t* dst; uint32_t stream_end; uint32_t desc_base; };

__device__ __forceinline__ uint32_t rgb(uint32_t c) { return ((c & 0x1Fu) << 3) | ((c & 0x3E0u) << 6) | ((c & 0x7C00u) << 9); }
__device__ __forceinline__ uint32_t ld16(const uint8_t* s, uint32_t o, uint32_t end) {
    return (o + 1u < end) ? (uint32_t) * reinterpret_cast<const uint16_t*>(s + o) : 0u;
}

// decode the colours + flags of one block (16-bit)
__device__ __forceinline__ void decode(const uint8_t* __restrict__ stream, uint32_t o, uint32_t end, uint32_t (&c)[8], uint32_t& flags) {
    const bool b_ok = o + 1u < end;
    const uint32_t w = b_ok ? (uint32_t) * reinterpret_cast<const uint16_t*>(stream + o) : (o < end ? (uint32_t)stream[o] : 0u);
    const uint32_t b = w >> 8;
    if (b_ok && b < 0x80u) {
        flags = w ^ 0xFFFFu;
        const uint32_t q0 = ld16(stream, o + 2u, end), q1 = ld16(stream, o + 4u, end);
        c[0] = rgb(q0); c[1] = rgb(q1);
        if (q0 & 0x8000u) {
#pragma unroll
            for (int k = 2; k < 8; ++k) c[k] = rgb(ld16(stream, o + 2u + 2u * k, end));
        } else { c[2] = c[4] = c[6] = c[0]; c[3] = c[5] = c[7] = c[1]; }
    } else {
        flags = 0;
        const uint32_t v = rgb(w);
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = v;
    }
}

// MODE 0: product kernel. 1: no stores (reads + compute). 2: no stream reads (descriptor + stores).
template <int MODE>
__global__ __launch_bounds__(256) void k_base(const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
                                              const FrameArgs* __restrict__ frames, int nblocks, int nbx, int X, uint32_t* sink) {
    const FrameArgs fa = frames[blockIdx.y];
    const int blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    const uint32_t o = desc[fa.desc_base + blk];
    const int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* dst = fa.dst + (size_t)by * 4u * X + (size_t)bx * 4u;
    uint32_t c[8], flags;
    if (MODE == 2) { flags = o * 2654435761u; for (int k = 0; k < 8; ++k) c[k] = o + k; }
    else decode(stream, o, fa.stream_end, c, flags);
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = (((i >> 2) & 2) << 1) + (i & 2); px[i] = ((flags >> i) & 1u) ? c[q + 1] : c[q]; }
    if (MODE == 1) {
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) a ^= px[i];
        if (a == 0x12345678u) sink[0] = a;
        return;
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) *reinterpret_cast<uint4*>(dst + (size_t)y * X) = make_uint4(px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
}

// V3: one store per lane; a workgroup = 256 consecutive blocks of ONE pixel row (4 KiB contiguous);
// the four rows of the same blocks are handled by workgroups b, b+8, b+16, b+24 (same XCD under
// round-robin dispatch, so the stream/descriptor lines are shared in one L2)
__global__ __launch_bounds__(256) void k_rowsplit(const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
                                                  const FrameArgs* __restrict__ frames, int nblocks, int nbx, int X, int ngroups) {
    const FrameArgs fa = frames[blockIdx.y];
    const int b = blockIdx.x;
    const int group = (b >> 5) * 8 + (b & 7);
    const int y = (b >> 3) & 3;
    if (group >= ngroups) return;
    const int blk = group * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    const uint32_t o = desc[fa.desc_base + blk];
    const int by = blk / nbx, bx = blk - by * nbx;
    uint32_t c[8], flags;
    decode(stream, o, fa.stream_end, c, flags);
    const int qy = (y & 2) << 1;
    const uint32_t f = flags >> (y * 4);
    uint4 v;
    v.x = (f & 1u) ? c[qy + 1] : c[qy];
    v.y = (f & 2u) ? c[qy + 1] : c[qy];
    v.z = (f & 4u) ? c[qy + 3] : c[qy + 2];
    v.w = (f & 8u) ? c[qy + 3] : c[qy + 2];
    *reinterpret_cast<uint4*>(fa.dst + ((size_t)by * 4u + y) * X + (size_t)bx * 4u) = v;
}

// V4: stream bytes staged through LDS: the workgroup's span [first code, last code + 18) is loaded
// with coalesced 16-byte reads, lanes then pick their code out of LDS
__global__ __launch_bounds__(256) void k_lds(const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
                                             const FrameArgs* __restrict__ frames, int nblocks, int nbx, int X) {
    __shared__ __align__(16) uint8_t sbuf[256 * 18 + 64];
    __shared__ uint32_t s_lo;
    const FrameArgs fa = frames[blockIdx.y];
    const int blk0 = blockIdx.x * 256;
    const int blk = blk0 + threadIdx.x;
    const bool live = blk < nblocks;
    const uint32_t o = live ? desc[fa.desc_base + blk] : 0xFFFFFFFFu;
    if (threadIdx.x == 0) s_lo = o & ~15u;          // codes are in raster order: lane 0 has the lowest offset
    __syncthreads();
    const uint32_t lo = s_lo;
    const int last = (nblocks - blk0 < 256 ? nblocks - blk0 : 256) - 1;
    // span end: offset of the last live lane + 18, found by that lane
    __shared__ uint32_t s_hi;
    if ((int)threadIdx.x == last) s_hi = o + 18u;
    __syncthreads();
    const uint32_t hi = s_hi < fa.stream_end ? s_hi : fa.stream_end;
    for (uint32_t p = lo + threadIdx.x * 16u; p < hi; p += 256u * 16u)
        *reinterpret_cast<uint4*>(sbuf + (p - lo)) = *reinterpret_cast<const uint4*>(stream + p);
    __syncthreads();
    if (!live) return;
    const int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* dst = fa.dst + (size_t)by * 4u * X + (size_t)bx * 4u;
    uint32_t c[8], flags;
    decode(sbuf, o - lo, hi - lo, c, flags);
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = (((i >> 2) & 2) << 1) + (i & 2); px[i] = ((flags >> i) & 1u) ? c[q + 1] : c[q]; }
#pragma unroll
    for (int y = 0; y < 4; ++y) *reinterpret_cast<uint4*>(dst + (size_t)y * X) = make_uint4(px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
}

// V5: V4 made general: span bounds from the first/last CODED lane of each wave (sentinels allowed),
// one barrier.  NT: nontemporal stores.  WG: workgroup size.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int WG, bool NT>
__global__ __launch_bounds__(WG) void k_lds2(const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
                                              const FrameArgs* __restrict__ frames, int nblocks, int nbx, int X) {
    constexpr int NW = WG / 64;
    __shared__ __align__(16) uint8_t sbuf[WG * 18 + 64];
    __shared__ uint32_t s_wlo[NW], s_whi[NW];
    const FrameArgs fa = frames[blockIdx.y];
    const int blk = blockIdx.x * WG + threadIdx.x;
    const bool live = blk < nblocks;
    const uint32_t o = live ? desc[fa.desc_base + blk] : 0xFFFFFFFFu;
    const bool coded = o < 0xFFFFFFFEu;
    const unsigned long long m = __ballot(coded);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (m == 0ull) { if (lane == 0) { s_wlo[wv] = 0xFFFFFFFFu; s_whi[wv] = 0u; } }
    else {
        if (lane == __ffsll((long long)m) - 1) s_wlo[wv] = o;
        if (lane == 63 - __clzll((long long)m)) s_whi[wv] = o + 18u;
    }
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int k = 0; k < NW; ++k) { lo = min(lo, s_wlo[k]); hi = max(hi, s_whi[k]); }
    lo &= ~15u;
    hi = hi < fa.stream_end ? hi : fa.stream_end;
    for (uint32_t p = lo + threadIdx.x * 16u; p < hi; p += WG * 16u)
        *reinterpret_cast<uint4*>(sbuf + (p - lo)) = *reinterpret_cast<const uint4*>(stream + p);
    __syncthreads();
    if (!coded) return;
    const int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* dst = fa.dst + (size_t)by * 4u * X + (size_t)bx * 4u;
    uint32_t c[8], flags;
    decode(sbuf, o - lo, hi - lo, c, flags);
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = (((i >> 2) & 2) << 1) + (i & 2); px[i] = ((flags >> i) & 1u) ? c[q + 1] : c[q]; }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        if (NT) __builtin_nontemporal_store(u32x4{px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]}, reinterpret_cast<u32x4*>(dst + (size_t)y * X));
        else *reinterpret_cast<uint4*>(dst + (size_t)y * X) = make_uint4(px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
    }
}

// V6: descriptors as 16-bit offsets relative to a per-workgroup base (halves the descriptor bytes)
template <int WG>
__global__ __launch_bounds__(WG) void k_lds16(const uint8_t* __restrict__ stream, const uint16_t* __restrict__ rel,
                                               const uint32_t* __restrict__ wgbase, const FrameArgs* __restrict__ frames,
                                               int nblocks, int nbx, int X, int wgs_per_frame) {
    __shared__ __align__(16) uint8_t sbuf[WG * 18 + 64];
    const FrameArgs fa = frames[blockIdx.y];
    const int blk = blockIdx.x * WG + threadIdx.x;
    const bool live = blk < nblocks;
    const uint32_t base = wgbase[blockIdx.y * wgs_per_frame + blockIdx.x];      // 16-byte aligned start of the span
    const uint32_t span = wgbase[blockIdx.y * wgs_per_frame + blockIdx.x + 1];  // next workgroup's base
    const uint32_t r = live ? rel[(size_t)fa.desc_base + blk] : 0xFFFFu;
    uint32_t hi = span + 32u;
    hi = hi < fa.stream_end ? hi : fa.stream_end;
    for (uint32_t p = base + threadIdx.x * 16u; p < hi; p += WG * 16u)
        *reinterpret_cast<uint4*>(sbuf + (p - base)) = *reinterpret_cast<const uint4*>(stream + p);
    __syncthreads();
    if (r >= 0xFFFEu) return;
    const int by = blk / nbx, bx = blk - by * nbx;
    uint32_t* dst = fa.dst + (size_t)by * 4u * X + (size_t)bx * 4u;
    uint32_t c[8], flags;
    decode(sbuf, r, hi - base, c, flags);
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = (((i >> 2) & 2) << 1) + (i & 2); px[i] = ((flags >> i) & 1u) ? c[q + 1] : c[q]; }
#pragma unroll
    for (int y = 0; y < 4; ++y) *reinterpret_cast<uint4*>(dst + (size_t)y * X) = make_uint4(px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
}

template <class F>
static double time_us(F&& launch, int reps = 30) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3 / reps;
}

int main() {
    const int X = 1920, Y = 1080, F = 64, nbx = X / 4, nblocks = nbx * (Y / 4);
    std::vector<uint8_t> stream;
    std::vector<uint32_t> desc((size_t)nblocks * F);
    std::vector<uint32_t> ends(F);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int f = 0; f < F; ++f) {
        for (int b = 0; b < nblocks; ++b) {
            desc[(size_t)f * nblocks + b] = (uint32_t)stream.size();
            const uint32_t r = rnd() & 3;
            auto put16 = [&](uint32_t v) { stream.push_back(v & 0xFF); stream.push_back((v >> 8) & 0xFF); };
            if (r == 0) { uint32_t c = (rnd() & 0x7FFF) | 0x8000; if ((c & 0xFC00) == 0x8400) c ^= 0x1000; put16(c); }
            else if (r == 3) { put16(rnd() & 0x7FFF); put16(rnd() | 0x8000); for (int k = 0; k < 7; ++k) put16(rnd()); }
            else { put16(rnd() & 0x7FFF); put16(rnd() & 0x7FFF); put16(rnd()); }
        }
        ends[f] = (uint32_t)stream.size();
    }
    printf("stream %.1f MB, descriptors %.1f MB, frames %.1f MB\n", stream.size() / 1e6, desc.size() * 4 / 1e6, (double)X * Y * 4 * F / 1e6);
    uint8_t* d_stream; uint32_t *d_desc, *d_out, *d_sink; FrameArgs* d_fa;
    CK(hipMalloc(&d_stream, stream.size() + 4096)); CK(hipMalloc(&d_desc, desc.size() * 4));
    CK(hipMalloc(&d_out, (size_t)X * Y * 4 * F)); CK(hipMalloc(&d_fa, sizeof(FrameArgs) * F)); CK(hipMalloc(&d_sink, 64));
    CK(hipMemcpy(d_stream, stream.data(), stream.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice));
    std::vector<FrameArgs> fa(F);
    for (int f = 0; f < F; ++f) fa[f] = {d_out + (size_t)f * X * Y, ends[f], (uint32_t)((size_t)f * nblocks)};
    CK(hipMemcpy(d_fa, fa.data(), sizeof(FrameArgs) * F, hipMemcpyHostToDevice));
    const double A = stream.size() + (double)X * Y * 4 * F;
    dim3 g((nblocks + 255) / 256, F);
    auto report = [&](const char* name, double us) { printf("%-46s %7.1f us  %6.0f GB/s algorithmic  (%.3f of 8 TB/s)\n", name, us, A / us / 1e3, A / us / 1e3 / 8000); };
    report("V0 product kernel", time_us([&] { k_base<0><<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X, d_sink); }));
    // reference result for the variants
    std::vector<uint32_t> ref((size_t)X * Y), got((size_t)X * Y);
    CK(hipMemcpy(ref.data(), d_out + (size_t)3 * X * Y, ref.size() * 4, hipMemcpyDeviceToHost));
    auto check = [&](const char* name) {
        CK(hipMemcpy(got.data(), d_out + (size_t)3 * X * Y, got.size() * 4, hipMemcpyDeviceToHost));
        printf("   %s %s\n", name, got == ref ? "matches V0" : "DIFFERS from V0");
    };
    report("V1 no stores (reads + compute)", time_us([&] { k_base<1><<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X, d_sink); }));
    report("V2 no stream reads (descriptor + stores)", time_us([&] { k_base<2><<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X, d_sink); }));
    CK(hipMemset(d_out, 0, (size_t)X * Y * 4 * F));
    const int ngroups = (nblocks + 255) / 256;
    dim3 g3(((ngroups + 7) / 8) * 32, F);
    report("V3 row-split, 1 store/lane, XCD-shared reads", time_us([&] { k_rowsplit<<<g3, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X, ngroups); }));
    check("V3");
    CK(hipMemset(d_out, 0, (size_t)X * Y * 4 * F));
    report("V4 stream staged through LDS", time_us([&] { k_lds<<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X); }));
    check("V4");
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(d_out, 0, (size_t)X * Y * 4 * F));
        report("V5 LDS stage, ballot span bounds, WG 256", time_us([&] { k_lds2<256, false><<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X); }));
        check("V5");
        report("V5 + nontemporal stores", time_us([&] { k_lds2<256, true><<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X); }));
        report("V5 WG 512", time_us([&] { k_lds2<512, false><<<dim3((nblocks + 511) / 512, F), 512>>>(d_stream, d_desc, d_fa, nblocks, nbx, X); }));
        report("V5 WG 128", time_us([&] { k_lds2<128, false><<<dim3((nblocks + 127) / 128, F), 128>>>(d_stream, d_desc, d_fa, nblocks, nbx, X); }));
        report("V0 product kernel (again)", time_us([&] { k_base<0><<<g, 256>>>(d_stream, d_desc, d_fa, nblocks, nbx, X, d_sink); }));
    }
    {   // V6 tables
        const int WGS = 256, wpf = (nblocks + WGS - 1) / WGS;
        std::vector<uint16_t> rel((size_t)nblocks * F);
        std::vector<uint32_t> wgb((size_t)wpf * F + 1);
        for (int f = 0; f < F; ++f)
            for (int w = 0; w < wpf; ++w) {
                const uint32_t b0 = desc[(size_t)f * nblocks + (size_t)w * WGS] & ~15u;
                wgb[(size_t)f * wpf + w] = b0;
                for (int k = 0; k < WGS && w * WGS + k < nblocks; ++k)
                    rel[(size_t)f * nblocks + (size_t)w * WGS + k] = (uint16_t)(desc[(size_t)f * nblocks + (size_t)w * WGS + k] - b0);
            }
        wgb[(size_t)wpf * F] = (uint32_t)stream.size();
        uint16_t* d_rel; uint32_t* d_wgb;
        CK(hipMalloc(&d_rel, rel.size() * 2)); CK(hipMalloc(&d_wgb, wgb.size() * 4));
        CK(hipMemcpy(d_rel, rel.data(), rel.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_wgb, wgb.data(), wgb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_out, 0, (size_t)X * Y * 4 * F));
        report("V6 16-bit relative descriptors + LDS stage", time_us([&] { k_lds16<256><<<g, 256>>>(d_stream, d_rel, d_wgb, d_fa, nblocks, nbx, X, wpf); }));
        check("V6");
    }
    return 0;
}
