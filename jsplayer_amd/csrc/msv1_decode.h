// Block reconstruction shared by the MSVideo1 kernels that decode a 4x4 block from its code in LDS
// (msv1_fused_kernel, msv1_blocks_temporal_kernel): MSVideo1.hx:135-181 (16-bit), :319-364 (8-bit).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace jsp {
namespace {

typedef uint32_t fu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) fu32x4 fgu32x4;
typedef const __attribute__((address_space(1))) fu32x4 fcgu32x4;

__device__ __forceinline__ uint32_t rgb555(uint32_t c) {
    return ((c & 0x1Fu) << 3) | ((c & 0x3E0u) << 6) | ((c & 0x7C00u) << 9);
}

// The 16 pixels of one coded block from its code in LDS.  `avail` = bytes between the code's first byte and the
// end of the frame's data (8-bit only: the 16-bit stream is zeroed past its end in LDS, and a zero word decodes
// as the reference decodes a missing one).
typedef unsigned short fus2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fus2 as_us2(uint32_t v) { fus2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t as_u32(fus2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }

// Two RGB555 colours (one per 16-bit half of `p`) -> two 0x00RRGGBB words (MSVideo1.hx:211-214), with packed
// 16-bit arithmetic: the green/blue bytes and the red byte of both colours are built side by side.
__device__ __forceinline__ void rgb555_pair(uint32_t p, uint32_t& lo, uint32_t& hi) {
    const fus2 v = as_us2(p);
    const fus2 m_g = {0x03E0, 0x03E0}, m_b = {0x001F, 0x001F}, m_r = {0x7C00, 0x7C00};
    const uint32_t gb = as_u32((fus2)(((v & m_g) << (fus2){6, 6}) | ((v & m_b) << (fus2){3, 3})));
    const uint32_t r = as_u32((fus2)((v & m_r) >> (fus2){7, 7}));
    lo = __builtin_amdgcn_perm(r, gb, 0x05040100u);   // {gb.b0, gb.b1, r.b0, r.b1}
    hi = __builtin_amdgcn_perm(r, gb, 0x07060302u);   // {gb.b2, gb.b3, r.b2, r.b3}
}

// The 16 pixels of one coded block from its code in LDS.  `avail` = bytes between the code's first byte and the
// end of the frame's data (8-bit only: the 16-bit stream is zeroed past its end in LDS, and a zero word decodes
// as the reference decodes a missing one).
template <int BITS>
__device__ __forceinline__ void decode_block(const uint8_t* code, uint32_t avail, const uint32_t* s_pal, uint32_t (&px)[16]) {
    if (BITS == 16) {
        // Branch-free.  The code word's own bits select: bit set -> first colour of the quadrant's pair, clear ->
        // second (MSVideo1.hx:140-168: flags ^= 0xFFFF, then pal[q + (flags & 1)]).  A 2-colour code uses its pair in
        // every quadrant; a solid code (high byte >= 0x80) is the pair {word, word}.
        uint32_t w0;
        fu32x4 q;
        __builtin_memcpy(&w0, __builtin_assume_aligned(code, 2), 4);        // code word | first colour << 16
        __builtin_memcpy(&q, __builtin_assume_aligned(code + 4, 2), 16);    // colours 1..7 (+ 2 bytes of whatever follows)
        const bool pattern = (w0 & 0x8000u) == 0u;                          // high byte < 0x80
        const bool eight = (w0 & 0x80008000u) == 0x80000000u;               // ... and bit 15 of the first colour set
        uint32_t p0 = __builtin_amdgcn_alignbit(q.x, w0, 16);               // {c0, c1}
        uint32_t p1 = __builtin_amdgcn_alignbit(q.y, q.x, 16);              // {c2, c3}
        uint32_t p2 = __builtin_amdgcn_alignbit(q.z, q.y, 16);
        uint32_t p3 = __builtin_amdgcn_alignbit(q.w, q.z, 16);
        p0 = pattern ? p0 : __builtin_amdgcn_perm(w0, w0, 0x01000100u);     // solid: {word, word}
        p1 = eight ? p1 : p0;
        p2 = eight ? p2 : p0;
        p3 = eight ? p3 : p0;
        uint32_t c[8];
        rgb555_pair(p0, c[0], c[1]);
        rgb555_pair(p1, c[2], c[3]);
        rgb555_pair(p2, c[4], c[5]);
        rgb555_pair(p3, c[6], c[7]);
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int qd = ((y & 2) << 1) + (x & 2);
                // 0 or ~0 from the pixel's bit, then a bit-field insert: two instructions per pixel (written as
                // asm so that the compiler does not turn the pair back into and + compare + select)
                uint32_t m;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w0), "n"(y * 4 + x));
                px[y * 4 + x] = (c[qd] & m) | (c[qd + 1] & ~m);
            }
        return;
    }
    uint32_t c[8], flags;
    {
        const uint32_t w = (uint32_t)code[0] | ((uint32_t)code[1] << 8), b = w >> 8;
        if (b < 0x80u) {
            flags = w;
            // first index byte is the colour of SET bits (p2[1]), second of clear bits (p2[0])
            const uint32_t i0 = avail > 2u ? s_pal[code[2]] : 0u;
            const uint32_t i1 = avail > 3u ? s_pal[code[3]] : 0u;
            c[0] = c[2] = c[4] = c[6] = i1;
            c[1] = c[3] = c[5] = c[7] = i0;
        } else if (b >= 0x90u) {
            flags = w ^ 0xFFFFu;
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = avail > 2u + k ? s_pal[code[2 + k]] : 0u;
        } else {
            flags = 0;
            const uint32_t v = s_pal[w & 0xFFu];
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = v;
        }
    }
    // pixel (x,y): quadrant q = ((y&2)<<1) + (x&2) is static, only the flag bit is dynamic
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int q = ((y & 2) << 1) + (x & 2);
            px[y * 4 + x] = ((flags >> (y * 4 + x)) & 1u) ? c[q + 1] : c[q];
        }
}

}  // namespace
}  // namespace jsp
