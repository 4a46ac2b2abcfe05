// MSVideo1 4x4 block reconstruction for gfx950 (MI355X).  Integer work, HBM-bound, no MFMA.
//
// One work-item per 4x4 block, blocks in raster order, so a wave covers 64 consecutive blocks:
//   * descriptor reads are one coalesced dword per lane;
//   * the wave's code words are neighbours in the stream (2..18 bytes per lane);
//   * each of the four row stores is 16 B per lane = up to 1 KiB contiguous per wave.
// grid.y = frame: a key-frame-only batch is one launch.
//
// Reference semantics reproduced per block: MSVideo1.hx:124-181 (16-bit), :307-364 (8-bit),
// copy_block :74-84, fromRGB15 :211-214, stage-2 significance compare :195-204.
#include <cstdlib>
#include <mutex>

#include "msv1.h"
#include "msv1_decode.h"

namespace jsp {
namespace {

constexpr int WG = 256;

__device__ __forceinline__ uint32_t rgb555_to_rgb32(uint32_t c) {
    return ((c & 0x1Fu) << 3) | ((c & 0x3E0u) << 6) | ((c & 0x7C00u) << 9);
}

// Little-endian 16-bit read at an even offset; 0 when either byte lies beyond `end`
// (the reference reads NaN there, which every later expression turns into 0).
__device__ __forceinline__ uint32_t ld16(const uint8_t* __restrict__ s, uint32_t o, uint32_t end) {
    return (o + 1u < end) ? (uint32_t) * reinterpret_cast<const uint16_t*>(s + o) : 0u;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Frame rows are written once and never read back by this launch: nontemporal stores keep them from
// evicting the stream / descriptor lines out of L2 (measured: 168 -> 91 us per 64-frame batch together
// with the LDS staging below, tools/msv1_lab.hip).
// (dst/prev reach the kernel inside a struct read from memory: without the explicit global address
// space the accesses would be FLAT instructions)
typedef __attribute__((address_space(1))) u32x4 gu32x4;
typedef const __attribute__((address_space(1))) u32x4 cgu32x4;
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef const __attribute__((address_space(1))) uint32_t cgu32;
__device__ __forceinline__ void store_row(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    __builtin_nontemporal_store(u32x4{a, b, c, d}, (gu32x4*)p);
}
__device__ __forceinline__ uint4 load_row(const uint32_t* p) {
    const u32x4 v = *(cgu32x4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int BITS, bool VEC>
__global__ __launch_bounds__(WG) void msv1_blocks_kernel(
    const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
    const Msv1FrameArgs* __restrict__ frames, const int32_t* __restrict__ palette, int nblocks,
    int nbx, int X) {
    constexpr int NW = WG / 64;
    constexpr int MAXCODE = 18;  // longest code: 16-bit 8-colour block
    __shared__ __align__(16) uint8_t sbuf[WG * MAXCODE + 64];
    __shared__ uint32_t s_wlo[NW], s_whi[NW];
    __shared__ uint32_t s_pal[BITS == 8 ? 256 : 1];
    if (BITS == 8) s_pal[threadIdx.x] = (uint32_t)palette[threadIdx.x];
    const Msv1FrameArgs fa = frames[blockIdx.y];
    const int blk = blockIdx.x * WG + threadIdx.x;
    const bool live = blk < nblocks;
    const uint32_t o = live ? desc[fa.desc_base + blk] : MSV1_DESC_UNTOUCHED;
    const bool coded = o < MSV1_DESC_UNTOUCHED;
    const int by = blk / nbx;
    const int bx = blk - by * nbx;
    const size_t di = (size_t)by * 4u * (size_t)X + (size_t)bx * 4u;
    // Inter frames: the block's pixels in the previous frame are needed either way (copied by skipped
    // blocks, compared by coded ones), so fetch them now, independent of the descriptor -> slice ->
    // decode chain; a short one-frame launch is latency bound and this removes a round trip.
    const bool preload = VEC && live && (fa.pad & MSV1_FRAME_USES_PREV);
    uint4 pr0 = make_uint4(0, 0, 0, 0), pr1 = pr0, pr2 = pr0, pr3 = pr0;
    if (preload) {
        const uint32_t* __restrict__ prev = reinterpret_cast<const uint32_t*>(fa.prev) + di;
        pr0 = load_row(prev);
        pr1 = load_row(prev + X);
        pr2 = load_row(prev + 2 * (size_t)X);
        pr3 = load_row(prev + 3 * (size_t)X);
    }

    // Stage the workgroup's slice of the code stream in LDS with coalesced 16-byte reads.  Codes are
    // in raster order, so the slice runs from the first coded lane's offset to the last coded lane's
    // offset + 18 (at most 256 codes of at most 18 bytes).
    {
        const unsigned long long m = __ballot(coded);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (m == 0ull) {
            if (lane == 0) { s_wlo[wv] = 0xFFFFFFFFu; s_whi[wv] = 0u; }
        } else {
            if (lane == __ffsll((long long)m) - 1) s_wlo[wv] = o;
            if (lane == 63 - __clzll((long long)m)) s_whi[wv] = o + MAXCODE;
        }
    }
    __syncthreads();
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int k = 0; k < NW; ++k) { lo = min(lo, s_wlo[k]); hi = max(hi, s_whi[k]); }
    lo &= ~15u;
    hi = hi < fa.stream_end ? hi : fa.stream_end;      // never read past the frame's bytes
    for (uint32_t p = lo + threadIdx.x * 16u; p < hi; p += WG * 16u)
        *reinterpret_cast<uint4*>(sbuf + (p - lo)) = *reinterpret_cast<const uint4*>(stream + p);
    __syncthreads();
    if (o == MSV1_DESC_UNTOUCHED) return;

    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(fa.dst) + di;

    if (o == MSV1_DESC_SKIP) {
        const uint32_t* __restrict__ prev = reinterpret_cast<const uint32_t*>(fa.prev) + di;
        if (VEC) {
            if (!preload) {
                pr0 = load_row(prev);
                pr1 = load_row(prev + X);
                pr2 = load_row(prev + 2 * (size_t)X);
                pr3 = load_row(prev + 3 * (size_t)X);
            }
            store_row(dst, pr0.x, pr0.y, pr0.z, pr0.w);
            store_row(dst + X, pr1.x, pr1.y, pr1.z, pr1.w);
            store_row(dst + 2 * (size_t)X, pr2.x, pr2.y, pr2.z, pr2.w);
            store_row(dst + 3 * (size_t)X, pr3.x, pr3.y, pr3.z, pr3.w);
        } else {
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) *(gu32*)(dst + (size_t)y * X + x) = *(cgu32*)(prev + (size_t)y * X + x);
        }
        return;
    }

    // From here on the code is read from LDS: offsets relative to `lo`, end of data relative too.
    const uint8_t* __restrict__ code = sbuf;
    const uint32_t end = hi > lo ? hi - lo : 0u;
    const uint32_t r = o - lo;
    // code word: a = low byte, b = high byte; a missing high byte makes the block "solid"
    const bool b_ok = r + 1u < end;
    const uint32_t w = b_ok ? (uint32_t) * reinterpret_cast<const uint16_t*>(code + r)
                            : (r < end ? (uint32_t)code[r] : 0u);
    const uint32_t b = w >> 8;
    uint32_t c[8];
    uint32_t flags;
    if (BITS == 16) {
        if (b_ok && b < 0x80u) {
            flags = w ^ 0xFFFFu;
            const uint32_t q0 = ld16(code, r + 2u, end);
            const uint32_t q1 = ld16(code, r + 4u, end);
            c[0] = rgb555_to_rgb32(q0);
            c[1] = rgb555_to_rgb32(q1);
            if (q0 & 0x8000u) {
#pragma unroll
                for (int k = 2; k < 8; ++k) c[k] = rgb555_to_rgb32(ld16(code, r + 2u + 2u * k, end));
            } else {
                c[2] = c[4] = c[6] = c[0];
                c[3] = c[5] = c[7] = c[1];
            }
        } else {
            flags = 0;
            const uint32_t v = rgb555_to_rgb32(w);
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = v;
        }
    } else {
        if (b_ok && b < 0x80u) {
            flags = w;
            // first index byte is the colour of SET bits (p2[1]), second of clear bits (p2[0])
            const uint32_t i0 = (r + 2u < end) ? s_pal[code[r + 2u]] : 0u;
            const uint32_t i1 = (r + 3u < end) ? s_pal[code[r + 3u]] : 0u;
            c[0] = c[2] = c[4] = c[6] = i1;
            c[1] = c[3] = c[5] = c[7] = i0;
        } else if (b_ok && b >= 0x90u) {
            flags = w ^ 0xFFFFu;
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = (r + 2u + k < end) ? s_pal[code[r + 2u + k]] : 0u;
        } else {
            flags = 0;
            const uint32_t v = (r < end) ? s_pal[w & 0xFFu] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = v;
        }
    }

    // pixel (x,y): quadrant q = ((y&2)<<1) + (x&2) is static, only the flag bit is dynamic
    uint32_t px[16];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int q = ((y & 2) << 1) + (x & 2);
            px[y * 4 + x] = ((flags >> (y * 4 + x)) & 1u) ? c[q + 1] : c[q];
        }

    if (VEC) {
#pragma unroll
        for (int y = 0; y < 4; ++y) store_row(dst + (size_t)y * X, px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
    } else {
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) *(gu32*)(dst + (size_t)y * X + x) = px[y * 4 + x];
    }

    // stage-2 significance: does any pixel at or above row cmp_row_lo differ from prev?
    // (skipped blocks are equal by construction; only coded blocks can differ)
    if (fa.cmp_row_lo != 0xFFFFFFFFu) {
        const uint32_t* __restrict__ prev = reinterpret_cast<const uint32_t*>(fa.prev) + di;
        bool diff = false;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            if ((uint32_t)(by * 4 + y) >= fa.cmp_row_lo) {
                if (VEC) {
                    const uint4 p = preload ? (y == 0 ? pr0 : y == 1 ? pr1 : y == 2 ? pr2 : pr3) : load_row(prev + (size_t)y * X);
                    diff |= (p.x != px[y * 4]) | (p.y != px[y * 4 + 1]) | (p.z != px[y * 4 + 2]) |
                            (p.w != px[y * 4 + 3]);
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) diff |= *(cgu32*)(prev + (size_t)y * X + x) != px[y * 4 + x];
                }
            }
        }
        // one word per frame: thousands of waves OR-ing the same address serialise (measured: 24 us per
        // inter frame, almost all of it here), so look before setting — an agent-scope load is enough,
        // a stale 0 only costs one more atomic
        if (__ballot(diff) != 0ull && (threadIdx.x & 63) == __ffsll((long long)__ballot(diff)) - 1 &&
            __hip_atomic_load(fa.signif, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            atomicOr(fa.signif, 1u);
    }
}

// Inter-frame batches in ONE launch.  A block's pixels depend only on the same block of earlier
// frames, so a workgroup takes a spatial tile of 256 blocks and walks the frames in order, keeping
// the tile's current pixels in registers: a skipped block is a plain re-store of what the lane already
// holds (no previous-frame read, no launch boundary between frames), the stage-2 compare is a register
// compare.  Frames of the group must write all their blocks or none (no UNTOUCHED sentinels apart
// from no-op frames) — msv1_codec.cpp forms the groups.
// Measured alternatives (64 x 1080p inter frames, this kernel: 163 us): a wave per pixel row with
// wave-private code slices and no barriers, loads issued a frame ahead: 410 us (four times the waves, each
// paying the store round trip that a vmcnt wait after a store implies — loads and stores share the
// counter); two row waves fed through LDS by a loader wave that never stores: 254 us (one load latency
// under full write pressure per frame on the critical path).  Both bit-exact, both dropped.
template <int BITS>
__global__ __launch_bounds__(WG) void msv1_blocks_temporal1_kernel(
    const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
    const Msv1FrameArgs* __restrict__ frames, int nframes, const int32_t* __restrict__ palette, int nblocks,
    int nbx, int X) {
    constexpr int NW = WG / 64;
    constexpr int MAXCODE = 18;
    __shared__ __align__(16) uint8_t sbuf[WG * MAXCODE + 64];
    __shared__ uint32_t s_wlo[NW], s_whi[NW];
    __shared__ uint32_t s_pal[BITS == 8 ? 256 : 1];
    if (BITS == 8) s_pal[threadIdx.x] = (uint32_t)palette[threadIdx.x];
    const int blk = blockIdx.x * WG + threadIdx.x;
    const bool live = blk < nblocks;
    const int by = blk / nbx;
    const int bx = blk - by * nbx;
    const size_t di = (size_t)by * 4u * (size_t)X + (size_t)bx * 4u;
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) px[i] = 0;
    bool have = false;  // px holds the pixels of the latest written frame
    uint32_t o_next = live ? desc[frames[0].desc_base + blk] : MSV1_DESC_UNTOUCHED;
    for (int f = 0; f < nframes; ++f) {
        const Msv1FrameArgs fa = frames[f];
        const uint32_t o = o_next;
        if (f + 1 < nframes) o_next = live ? desc[frames[f + 1].desc_base + blk] : MSV1_DESC_UNTOUCHED;
        if (fa.pad & MSV1_FRAME_NOOP) continue;       // early-out frame: nothing is written
        const bool coded = o < MSV1_DESC_UNTOUCHED;
        // the first written frame of the group may need the frame before the group
        if (!have && live && (fa.pad & MSV1_FRAME_USES_PREV)) {
            const uint32_t* __restrict__ prev = reinterpret_cast<const uint32_t*>(fa.prev) + di;
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const uint4 r = load_row(prev + (size_t)y * X);
                px[y * 4] = r.x; px[y * 4 + 1] = r.y; px[y * 4 + 2] = r.z; px[y * 4 + 3] = r.w;
            }
        }
        have = true;
        {
            const unsigned long long m = __ballot(coded);
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            if (m == 0ull) {
                if (lane == 0) { s_wlo[wv] = 0xFFFFFFFFu; s_whi[wv] = 0u; }
            } else {
                if (lane == __ffsll((long long)m) - 1) s_wlo[wv] = o;
                if (lane == 63 - __clzll((long long)m)) s_whi[wv] = o + MAXCODE;
            }
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
        bool diff = false;
        if (coded) {
            const uint8_t* __restrict__ code = sbuf;
            const uint32_t end = hi > lo ? hi - lo : 0u;
            const uint32_t r = o - lo;
            const bool b_ok = r + 1u < end;
            const uint32_t w = b_ok ? (uint32_t) * reinterpret_cast<const uint16_t*>(code + r)
                                    : (r < end ? (uint32_t)code[r] : 0u);
            const uint32_t b = w >> 8;
            uint32_t c[8];
            uint32_t flags;
            if (BITS == 16) {
                if (b_ok && b < 0x80u) {
                    flags = w ^ 0xFFFFu;
                    const uint32_t q0 = ld16(code, r + 2u, end);
                    const uint32_t q1 = ld16(code, r + 4u, end);
                    c[0] = rgb555_to_rgb32(q0);
                    c[1] = rgb555_to_rgb32(q1);
                    if (q0 & 0x8000u) {
#pragma unroll
                        for (int k = 2; k < 8; ++k) c[k] = rgb555_to_rgb32(ld16(code, r + 2u + 2u * k, end));
                    } else {
                        c[2] = c[4] = c[6] = c[0];
                        c[3] = c[5] = c[7] = c[1];
                    }
                } else {
                    flags = 0;
                    const uint32_t v = rgb555_to_rgb32(w);
#pragma unroll
                    for (int k = 0; k < 8; ++k) c[k] = v;
                }
            } else {
                if (b_ok && b < 0x80u) {
                    flags = w;
                    const uint32_t i0 = (r + 2u < end) ? s_pal[code[r + 2u]] : 0u;
                    const uint32_t i1 = (r + 3u < end) ? s_pal[code[r + 3u]] : 0u;
                    c[0] = c[2] = c[4] = c[6] = i1;
                    c[1] = c[3] = c[5] = c[7] = i0;
                } else if (b_ok && b >= 0x90u) {
                    flags = w ^ 0xFFFFu;
#pragma unroll
                    for (int k = 0; k < 8; ++k) c[k] = (r + 2u + k < end) ? s_pal[code[r + 2u + k]] : 0u;
                } else {
                    flags = 0;
                    const uint32_t v = (r < end) ? s_pal[w & 0xFFu] : 0u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) c[k] = v;
                }
            }
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int q = ((y & 2) << 1) + (x & 2);
                    const uint32_t v = ((flags >> (y * 4 + x)) & 1u) ? c[q + 1] : c[q];
                    if (fa.cmp_row_lo != 0xFFFFFFFFu && (uint32_t)(by * 4 + y) >= fa.cmp_row_lo) diff |= v != px[y * 4 + x];
                    px[y * 4 + x] = v;
                }
        }
        if (o != MSV1_DESC_UNTOUCHED) {  // coded or skipped: the block is (re)written in this frame's buffer
            uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(fa.dst) + di;
#pragma unroll
            for (int y = 0; y < 4; ++y) store_row(dst + (size_t)y * X, px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
        }
        if (fa.cmp_row_lo != 0xFFFFFFFFu) {
            const unsigned long long dm = __ballot(diff);
            if (dm != 0ull && (threadIdx.x & 63) == __ffsll((long long)dm) - 1 &&
                __hip_atomic_load(fa.signif, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                atomicOr(fa.signif, 1u);
        }
        // (the next iteration's first barrier comes after its ballot; sbuf is only rewritten after it)
    }
}

// Pixels outside the block grid (x >= 4*nbx or y >= 4*nby) are never written by the decoder, but
// the reference's compare loop still covers them (MSVideo1.hx:197-203).
__global__ __launch_bounds__(WG) void msv1_edge_compare_kernel(const Msv1FrameArgs* __restrict__ frames,
                                                               int X, int Y, int covered_x, int covered_y) {
    const Msv1FrameArgs fa = frames[blockIdx.y];
    if (fa.cmp_row_lo == 0xFFFFFFFFu) return;
    const int strip = X - covered_x;             // right-hand columns, all rows
    const long right = (long)strip * Y;
    const long top = (long)covered_x * (Y - covered_y);  // rows above the grid, covered columns
    const long total = right + top;
    bool diff = false;
    for (long i = (long)blockIdx.x * WG + threadIdx.x; i < total; i += (long)gridDim.x * WG) {
        int x, y;
        if (i < right) {
            y = (int)(i / strip);
            x = covered_x + (int)(i - (long)y * strip);
        } else {
            const long j = i - right;
            y = covered_y + (int)(j / covered_x);
            x = (int)(j - (long)(y - covered_y) * covered_x);
        }
        if ((uint32_t)y >= fa.cmp_row_lo) {
            const size_t k = (size_t)y * X + x;
            diff |= fa.dst[k] != fa.prev[k];
        }
    }
    if (__ballot(diff) != 0ull && (threadIdx.x & 63) == __ffsll((long long)__ballot(diff)) - 1 &&
        __hip_atomic_load(fa.signif, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        atomicOr(fa.signif, 1u);
}


// ---------------------------------------------------------------------------------------------------------
// Inter-frame groups, second form: a LOADER wave next to the four worker waves.
//
// A workgroup owns 256 blocks and walks the group's frames with their pixels in registers, as above.  What bounds
// that walk is not bandwidth but latency: a CU's loads and stores share one in-order queue (and a wave's vmcnt counts
// both), so the per-frame fetches — block table entry, then the code bytes it points at — each waited behind the row
// stores of the frame before: ~3 us per frame, 1.5 ms per 512 frames at 0.37 of peak.  Here the worker waves never
// issue a load inside the frame loop: a fifth wave fetches, CHUNK frames at a time, the tile's table entries and the
// slice of the code stream they point into straight into LDS (global_load_lds: no registers), a whole chunk ahead of
// the workers (two chunk buffers), and its own vmcnt only ever holds loads.  Chunks are handed over through two LDS
// counters (`ready`, `consumed`); nobody waits at a workgroup barrier inside the loop.
constexpr int TW = 4;                        // worker waves (lane = block)
constexpr int TWG = (TW + 1) * 64;
constexpr int T_NF = 16;                     // frames per chunk, at most
constexpr int T_CODE = 14 * 1024;            // bytes of code stream per chunk (a frame needs at most 256 * 18 + 32)
constexpr int T_SPIN = 1 << 24;              // polls before a wait gives up
constexpr int T_SLACK = 64;                  // readable bytes behind a chunk's code bytes (a block reads 20 bytes from its code)
struct TFrame {                              // what the workers need to know about one frame of a chunk
    uint32_t* dst;
    uint32_t code_at;                        // LDS byte offset (within the chunk's code area) of stream byte `lo`
    uint32_t lo;                             // first stream byte held
    uint32_t end;                            // end of the frame's readable stream bytes (16-bit: whole words only)
    uint32_t cmp_row_lo;
    uint32_t flags;                          // bit 0: the block's pixels come from `prev` first; bits 8..15: the odd last byte of a 16-bit stream
    uint32_t index;                          // frame number within the group (significance bit)
    uint32_t span;                           // stream bytes held (a multiple of 16)
};
struct TChunk {
    uint32_t desc[T_NF][WG];
    __attribute__((aligned(16))) uint8_t code[T_CODE + T_SLACK];
    TFrame fr[T_NF];
    int fidx[T_NF];                          // (loader) the frames whose table rows were fetched
    const uint32_t* prev;                    // (chunk 0) the frame before the group
    int nf;                                  // frames in this chunk
    int next;                                // first frame of the group not yet in a chunk (== nframes: this is the last chunk)
};

typedef __attribute__((address_space(1))) const void t_gvoid;
typedef __attribute__((address_space(3))) void t_lvoid;

// COMPACT: the frames' tables are the 2-byte form (msv1.h: an entry per block in `tab16`, a base per group of 256 blocks — this workgroup's tile IS one group —
// in `bases`): the loader fetches half the bytes per frame into the back half of the frame's LDS row and expands them in place into the 4-byte offsets the
// workers read; nothing else differs.
template <int BITS, bool COMPACT>
__global__ __launch_bounds__(TWG) void msv1_blocks_temporal_kernel(
    const uint8_t* __restrict__ stream, const uint32_t* __restrict__ desc,
    const Msv1FrameArgs* __restrict__ frames, int nframes, const int32_t* __restrict__ palette, int nblocks,
    int nbx, int X, const uint16_t* __restrict__ tab16, const uint32_t* __restrict__ bases, int pitch16, int ngroups) {
    extern __shared__ __align__(16) uint8_t t_lds[];
    TChunk* chunks = reinterpret_cast<TChunk*>(t_lds);                          // [2]
    uint32_t* s_pal = reinterpret_cast<uint32_t*>(t_lds + 2 * sizeof(TChunk));  // [256]
    uint32_t* s_sig = s_pal + 256;                                              // [(nframes + 31) / 32] stage-2 significance bits
    // (LDS address space spelled out: through generic pointers the polls would be FLAT loads, which count on vmcnt too — a poll
    // would then wait for the wave's row stores)
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
    lds_vu32* s_ready = (lds_vu32*)(s_sig + ((nframes + 31) >> 5));             // chunks handed over by the loader
    lds_vu32* s_done = s_ready + 1;                                             // [TW] chunks each worker wave is through with
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (provably uniform: the loader's loops then run on the scalar unit, its frame records are scalar loads)
    if (BITS == 8 && tid < 256) s_pal[tid] = (uint32_t)palette[tid];
    for (int i = tid; i < ((nframes + 31) >> 5) + 1 + TW; i += TWG) s_sig[i] = 0u;   // (+ ready, done[TW])
    __syncthreads();
    const int blk0 = blockIdx.x * WG;

    if (wave == TW) {
        // ---------------------------------------- loader ----------------------------------------
        const int live_blocks = nblocks - blk0 < WG ? nblocks - blk0 : WG;      // (the last tile is partial)
        int f = 0, c = 0;
        bool first = true;
        while (f < nframes) {
            TChunk& ck = chunks[c & 1];
            // the buffer was chunk c - 2's: all its readers must be through
            // (EVERY worker wave: a count summed over the waves would let a wave that is a chunk ahead stand in for one that is
            // still reading the buffer)
            if (c >= 2)
                for (int spin = 0; spin < T_SPIN; ++spin) {
                    const uint32_t slowest = min(min(s_done[0], s_done[1]), min(s_done[2], s_done[3]));
                    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)slowest) >= (uint32_t)(c - 1)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            // 1. table entries of up to T_NF frames (frames that write nothing are left out), one 1 KiB row each
            int nf = 0, scan = f;
            uint32_t v_base = 0, v_fr = 0;                                          // (COMPACT) lane i: the group base / the number in the batch of the chunk's frame i
#pragma unroll 1
            for (; scan < nframes && nf < T_NF; ++scan) {
                const Msv1FrameArgs fa = frames[scan];
                if (fa.pad & MSV1_FRAME_NOOP) continue;
                if (COMPACT) {
                    const uint32_t fr = fa.desc_base / (uint32_t)nblocks;           // the frame's number in the batch (desc_base = frame x nblocks)
                    if (lane * 8 < live_blocks)                                     // 512 bytes: eight entries per lane, 32 lanes, into the row's back half
                        __builtin_amdgcn_global_load_lds((t_gvoid*)(tab16 + (size_t)fr * (size_t)pitch16 + blk0 + lane * 8), (t_lvoid*)&ck.desc[nf][WG / 2], 16, 0, 0);
                    v_fr = lane == nf ? fr : v_fr;                                   // (its base is fetched below, all frames' at once: a scalar load per frame in this loop would put a memory round trip between the requests)
                } else if (lane * 4 < live_blocks)
                    __builtin_amdgcn_global_load_lds((t_gvoid*)(desc + fa.desc_base + blk0 + lane * 4), (t_lvoid*)&ck.desc[nf][0], 16, 0, 0);
                if (lane == 0) ck.fidx[nf] = scan;
                ++nf;
            }
            if (COMPACT && lane < nf) v_base = *(cgu32*)(bases + (size_t)v_fr * (size_t)ngroups + blockIdx.x);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // 2. per frame: which stream bytes its coded blocks point into, as long as the chunk's code area has room.  (The
            //    extents are parked in the lanes of three registers: an LDS read issued after an LDS-DMA is made to wait for it,
            //    and the requests of step 3 must all go out before any is waited for.)
            uint32_t used = 0, v_lo = 0, v_need = 0, v_at = 0;
            int kept = 0;
#pragma unroll 1
            for (int i = 0; i < nf; ++i) {
                const int fi = __builtin_amdgcn_readfirstlane(ck.fidx[i]);
                const Msv1FrameArgs fa = frames[fi];
                uint32_t lo = 0xFFFFFFFFu, hi = 0u;
                if (lane * 4 < live_blocks) {
                    uint32_t dd[4];
                    if (COMPACT) {
                        // four entries of the row's back half -> four 4-byte offsets at the row's front.  (Every lane's read is issued before any lane's write:
                        // one wave, one instruction stream, and LDS takes a wave's operations in order.)
                        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)v_base, i);
                        const uint2 e2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(&ck.desc[i][WG / 2]) + lane * 4);
                        const uint32_t e[4] = {e2.x & 0xFFFFu, e2.x >> 16, e2.y & 0xFFFFu, e2.y >> 16};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint32_t c = (base & ~0x7FFFu) | e[k];                  // the group's codes lie within 4 608 bytes of its base: the nearest value with these low 15 bits
                            c = c + 0x4000u < base ? c + 0x8000u : (c > base + 0x4000u ? c - 0x8000u : c);
                            dd[k] = (e[k] & 0x8000u) ? (e[k] == MSV1_TAB16_SKIP ? MSV1_DESC_SKIP : MSV1_DESC_UNTOUCHED) : c;
                        }
                        *reinterpret_cast<uint4*>(&ck.desc[i][lane * 4]) = make_uint4(dd[0], dd[1], dd[2], dd[3]);
                    } else {
                        const uint4 d4 = *reinterpret_cast<const uint4*>(&ck.desc[i][lane * 4]);
                        dd[0] = d4.x; dd[1] = d4.y; dd[2] = d4.z; dd[3] = d4.w;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (lane * 4 + k < live_blocks && dd[k] < MSV1_DESC_UNTOUCHED) { lo = min(lo, dd[k]); hi = max(hi, dd[k] + 18u); }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
                    hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
                }
                lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo) & ~15u;       // (uniform by construction; said so: the loop stays scalar)
                hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
                hi = hi < fa.stream_end ? hi : fa.stream_end;
                const uint32_t len = hi > lo ? hi - lo : 0u, need = (len + 15u) & ~15u;
                if (used + need > (uint32_t)T_CODE && kept > 0) break;          // this frame opens the next chunk
                if (lane == 0) {
                    TFrame tf;
                    tf.dst = reinterpret_cast<uint32_t*>(fa.dst);
                    tf.code_at = used;
                    tf.lo = lo;
                    tf.end = BITS == 16 ? (fa.stream_end & ~1u) : fa.stream_end;
                    tf.cmp_row_lo = fa.cmp_row_lo;
                    tf.flags = (first && (fa.pad & MSV1_FRAME_USES_PREV)) ? 1u : 0u;
                    tf.index = (uint32_t)fi;
                    tf.span = need;
                    ck.fr[kept] = tf;
                    if (first) ck.prev = reinterpret_cast<const uint32_t*>(fa.prev);
                }
                v_lo = lane == kept ? lo : v_lo;
                v_need = lane == kept ? need : v_need;
                v_at = lane == kept ? used : v_at;
                first = false;
                used = (used + need + T_SLACK + 15u) & ~15u;
                ++kept;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 3. ask for them
#pragma unroll 1
            for (int i = 0; i < kept; ++i) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v_lo, i), need = (uint32_t)__builtin_amdgcn_readlane((int)v_need, i),
                               at = (uint32_t)__builtin_amdgcn_readlane((int)v_at, i);
                for (uint32_t k = lane * 16u; k < need; k += 1024u)
                    __builtin_amdgcn_global_load_lds((t_gvoid*)(stream + lo + k), (t_lvoid*)(ck.code + at + (k & ~1023u)), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // 4. bytes past the end of a frame's data read as zero (MSVideo1.hx: a missing byte is NaN, and every later expression
            //    turns that into 0): zero [end, lo + span + T_SLACK) where it lies inside what a block may read; the odd last byte
            //    of a 16-bit stream — half a word — is kept aside for the one case that reads it (a code word cut in two)
#pragma unroll 1
            for (int i = 0; i < kept; ++i) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck.fr[i].lo), end = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck.fr[i].end),
                               span = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck.fr[i].span), at = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck.fr[i].code_at);
                if (span && lo + span + T_SLACK > end) {
                    const uint32_t z0 = end > lo ? end - lo : 0u;
                    if (BITS == 16 && lane == 0) {
                        const uint32_t true_end = frames[ck.fr[i].index].stream_end;
                        if ((true_end & 1u) && true_end > lo && true_end - 1u - lo < span) ck.fr[i].flags |= (uint32_t)ck.code[at + (true_end - 1u - lo)] << 8;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (uint32_t k = z0 + lane; k < span + T_SLACK; k += 64u) ck.code[at + k] = 0;
                }
            }
            f = kept < nf ? __builtin_amdgcn_readfirstlane(ck.fidx[kept]) : scan;
            if (lane == 0) { ck.nf = kept; ck.next = f; }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) *s_ready = (uint32_t)(c + 1);
            ++c;
        }
        return;
    }

    // ---------------------------------------- workers ----------------------------------------
    const int blk = blk0 + tid;
    const bool live = blk < nblocks;
    const int by = blk / nbx;
    const int bx = blk - by * nbx;
    const size_t di = (size_t)by * 4u * (size_t)X + (size_t)bx * 4u;
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) px[i] = 0;
    // the first written frame of the group may need the frame before the group: fetched here, once, so that the frame loop
    // below holds no load at all (a load inside it would make every round wait for the row stores in flight)
    for (int spin = 0; *s_ready < 1u && spin < T_SPIN; ++spin) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (chunks[0].nf > 0 && (chunks[0].fr[0].flags & 1u) && live) {
        const uint32_t* __restrict__ prev = chunks[0].prev + di;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const uint4 r = load_row(prev + (size_t)y * X);
            px[y * 4] = r.x; px[y * 4 + 1] = r.y; px[y * 4 + 2] = r.z; px[y * 4 + 3] = r.w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int c = 0;; ++c) {
        int spin = 0;
        for (; *s_ready < (uint32_t)(c + 1) && spin < T_SPIN; ++spin) __builtin_amdgcn_s_sleep(1);
        if (spin >= T_SPIN) return;               // (cannot happen: every wait in this kernel is bounded so that a mistake ends the launch)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const TChunk& ck = chunks[c & 1];
        const int nf = ck.nf, next = ck.next;
        for (int i = 0; i < nf; ++i) {
            const TFrame tf = ck.fr[i];
            const uint32_t o = live ? ck.desc[i][tid] : MSV1_DESC_UNTOUCHED;
            bool diff = false;
            if (o < MSV1_DESC_UNTOUCHED) {
                uint32_t nx[16];
                const uint32_t avail = tf.end > o ? tf.end - o : 0u;
                const uint32_t tail = (tf.flags >> 8) & 0xFFu;
                // (16-bit: `end` counts whole words, the odd last byte is in flags; 8-bit: `end` is exact)
                if (BITS == 16 ? (tail != 0u && o == tf.end) : avail == 1u) {
                    // only the first byte of the code word exists: neither a skip nor a pattern test holds, the reference paints it solid
                    const uint32_t a = BITS == 16 ? tail : (uint32_t)ck.code[tf.code_at + (o - tf.lo)];
                    const uint32_t v = BITS == 16 ? rgb555_to_rgb32(a) : s_pal[a];
#pragma unroll
                    for (int k = 0; k < 16; ++k) nx[k] = v;
                } else if (avail == 0u) {
                    // the whole code lies past the end of the data (the table of a truncated frame): every byte reads as
                    // missing, which the reference paints as colour 0 — and nothing of it is in LDS
#pragma unroll
                    for (int k = 0; k < 16; ++k) nx[k] = 0u;
                } else {
                    decode_block<BITS>(ck.code + tf.code_at + (o - tf.lo), avail, s_pal, nx);
                }
                if (tf.cmp_row_lo != 0xFFFFFFFFu) {
#pragma unroll
                    for (int y = 0; y < 4; ++y)
                        if ((uint32_t)(by * 4 + y) >= tf.cmp_row_lo)
                            diff |= (nx[y * 4] != px[y * 4]) | (nx[y * 4 + 1] != px[y * 4 + 1]) | (nx[y * 4 + 2] != px[y * 4 + 2]) | (nx[y * 4 + 3] != px[y * 4 + 3]);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) px[k] = nx[k];
            }
            if (o != MSV1_DESC_UNTOUCHED) {       // coded or skipped: the block is (re)written in this frame's buffer
                uint32_t* __restrict__ dst = tf.dst + di;
#pragma unroll
                for (int y = 0; y < 4; ++y) store_row(dst + (size_t)y * X, px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]);
            }
            if (tf.cmp_row_lo != 0xFFFFFFFFu && __ballot(diff) != 0ull && lane == 0)
                __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)&s_sig[tf.index >> 5], 1u << (tf.index & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) s_done[wave] = (uint32_t)(c + 1);
        if (next >= nframes) break;
    }
    // significance bits -> the frames' words (MSVideo1.hx:195-204), once the four worker waves are through
    for (int spin = 0; min(min(s_done[0], s_done[1]), min(s_done[2], s_done[3])) < *s_ready && spin < T_SPIN; ++spin) __builtin_amdgcn_s_sleep(1);
    for (int fi = tid; fi < nframes; fi += TW * 64)
        if ((s_sig[fi >> 5] >> (fi & 31)) & 1u) {
            gu32* sg = (gu32*)frames[fi].signif;
            if (__hip_atomic_load(sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __hip_atomic_fetch_or(sg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}

}  // namespace

void msv1_launch_blocks(const Msv1Geometry& geo, const uint8_t* d_stream, const uint32_t* d_desc,
                        const Msv1FrameArgs* d_frames, int nframes, const int32_t* d_palette,
                        bool vec_ok, hipStream_t stream) {
    if (geo.nblocks <= 0 || nframes <= 0) return;
    dim3 grid((geo.nblocks + WG - 1) / WG, nframes), block(WG);
    if (geo.bits == 16) {
        if (vec_ok)
            hipLaunchKernelGGL((msv1_blocks_kernel<16, true>), grid, block, 0, stream, d_stream, d_desc,
                               d_frames, d_palette, geo.nblocks, geo.nbx, geo.X);
        else
            hipLaunchKernelGGL((msv1_blocks_kernel<16, false>), grid, block, 0, stream, d_stream, d_desc,
                               d_frames, d_palette, geo.nblocks, geo.nbx, geo.X);
    } else {
        if (vec_ok)
            hipLaunchKernelGGL((msv1_blocks_kernel<8, true>), grid, block, 0, stream, d_stream, d_desc,
                               d_frames, d_palette, geo.nblocks, geo.nbx, geo.X);
        else
            hipLaunchKernelGGL((msv1_blocks_kernel<8, false>), grid, block, 0, stream, d_stream, d_desc,
                               d_frames, d_palette, geo.nblocks, geo.nbx, geo.X);
    }
}

void msv1_launch_blocks_temporal(const Msv1Geometry& geo, const uint8_t* d_stream, const uint32_t* d_desc,
                                 const Msv1FrameArgs* d_frames, int nframes, const int32_t* d_palette,
                                 hipStream_t stream, const uint16_t* d_tab16, const uint32_t* d_bases) {
    if (geo.nblocks <= 0 || nframes <= 0) return;
    static const bool old_form = std::getenv("JSP_MSV1_TEMPORAL_OLD") != nullptr;   // lab: the frame-at-a-time kernel
    if (old_form) {
        dim3 grid((geo.nblocks + WG - 1) / WG), block(WG);
        if (geo.bits == 16)
            hipLaunchKernelGGL((msv1_blocks_temporal1_kernel<16>), grid, block, 0, stream, d_stream, d_desc, d_frames, nframes,
                               d_palette, geo.nblocks, geo.nbx, geo.X);
        else
            hipLaunchKernelGGL((msv1_blocks_temporal1_kernel<8>), grid, block, 0, stream, d_stream, d_desc, d_frames, nframes,
                               d_palette, geo.nblocks, geo.nbx, geo.X);
        return;
    }
    const size_t lds = 2 * sizeof(TChunk) + 256 * 4 + (((size_t)nframes + 31) / 32 + 1 + TW) * 4;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msv1_blocks_temporal_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msv1_blocks_temporal_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msv1_blocks_temporal_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msv1_blocks_temporal_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    dim3 grid((geo.nblocks + WG - 1) / WG), block(TWG);
    const int pitch16 = msv1_tab16_pitch(geo.nblocks), ngroups = msv1_tab16_groups(geo.nblocks);
#define JSP_TEMPORAL(BITS, C) hipLaunchKernelGGL((msv1_blocks_temporal_kernel<BITS, C>), grid, block, lds, stream, d_stream, d_desc, d_frames, nframes, \
                                                 d_palette, geo.nblocks, geo.nbx, geo.X, d_tab16, d_bases, pitch16, ngroups)
    if (d_tab16) { if (geo.bits == 16) JSP_TEMPORAL(16, true); else JSP_TEMPORAL(8, true); }
    else { if (geo.bits == 16) JSP_TEMPORAL(16, false); else JSP_TEMPORAL(8, false); }
#undef JSP_TEMPORAL
}

// The compact form of 4-byte tables (msv1.h): one workgroup per (group of 256 blocks, listed frame).
__global__ __launch_bounds__(WG) void msv1_tables_compact_kernel(const uint32_t* __restrict__ desc, uint16_t* __restrict__ tab16, uint32_t* __restrict__ bases,
                                                                 const uint32_t* __restrict__ list, int nblocks, int pitch16, int ngroups) {
    __shared__ uint32_t s_min[WG / 64];
    const uint32_t fr = list[blockIdx.y];
    const int blk = (int)blockIdx.x * WG + (int)threadIdx.x;
    const uint32_t o = blk < nblocks ? desc[(size_t)fr * (size_t)nblocks + blk] : MSV1_DESC_UNTOUCHED;
    uint32_t m = o < MSV1_DESC_UNTOUCHED ? o : 0xFFFFFFFFu;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t all = s_min[0];
        for (int w = 1; w < WG / 64; ++w) all = min(all, s_min[w]);
        if (all != 0xFFFFFFFFu) bases[(size_t)fr * (size_t)ngroups + blockIdx.x] = all;
    }
    if (blk < nblocks)
        tab16[(size_t)fr * (size_t)pitch16 + blk] = (uint16_t)(o == MSV1_DESC_SKIP ? MSV1_TAB16_SKIP : (o == MSV1_DESC_UNTOUCHED ? MSV1_TAB16_UNTOUCHED : (o & 0x7FFFu)));
}

void msv1_launch_tables_compact(const Msv1Geometry& geo, const uint32_t* d_desc, uint16_t* d_tab16, uint32_t* d_bases, const uint32_t* d_list, int count,
                                hipStream_t stream) {
    if (geo.nblocks <= 0 || count <= 0) return;
    hipLaunchKernelGGL(msv1_tables_compact_kernel, dim3(msv1_tab16_groups(geo.nblocks), count), dim3(WG), 0, stream, d_desc, d_tab16, d_bases, d_list,
                       geo.nblocks, msv1_tab16_pitch(geo.nblocks), msv1_tab16_groups(geo.nblocks));
}

void msv1_launch_edge_compare(const Msv1Geometry& geo, const Msv1FrameArgs* d_frames, int nframes,
                              hipStream_t stream) {
    const int cx = geo.nbx * 4, cy = geo.nby * 4;
    if (cx == geo.X && cy == geo.Y) return;
    const long total = (long)(geo.X - cx) * geo.Y + (long)cx * (geo.Y - cy);
    if (total <= 0) return;
    int gx = (int)((total + WG - 1) / WG);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(msv1_edge_compare_kernel, dim3(gx, nframes), dim3(WG), 0, stream, d_frames, geo.X,
                       geo.Y, cx, cy);
}

}  // namespace jsp
