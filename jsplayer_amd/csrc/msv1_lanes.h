// Per-lane part of the on-GPU MSVideo1 parse (msv1_fused_kernel): the slots (16-bit words) of a lane as BIT MASKS.
//
// The code walk "offset += length(code)" (MSVideo1.hx:128-181 / 311-364) only needs, per slot, the length class of the
// code that would start there — short (1 slot), mid (3 slots; 8-bit: 2), long (9 slots; 8-bit: 5) — and, for the few
// slots that are not plain one-block codes, a block count.  Both are pure functions of one or two 16-bit words, so a
// lane classifies all of its LS slots side by side:
//   * a predicate over the words of a dword is brought to bit 15 / bit 31 with packed 16-bit arithmetic;
//   * v_perm_b32's sign selectors (8..11: "bit 7 of byte 1 / 3 / 5 / 7, replicated") turn those bits of two dwords
//     into four bytes of 0x00 / 0xFF, and one v_dot4 with the weights -1, -2, -4, -8 (-16 … -128 for the next four)
//     packs them into consecutive mask bits: 1 instruction per slot pair instead of 4 per slot;
//   * the 9-entry table "entry slot -> (exit slot, blocks)" is one reverse pass of 5 instructions per slot over the
//     masks (two sign-extended bit fields, two bit-field inserts, one saturating add); the rare special slots (skip
//     codes, slots past the end of the data, the 8-bit end marker) are handled under a wave-uniform test per slot;
//   * the slots the real chain visits are the closure of the entry slot under "slot + length", grown as a bit mask.
// Everything here is lane-private and free of HIP runtime calls; with JSP_LANES_HOST defined the few instructions
// used are emulated in plain C++ so that tests/ can check the masks, tables and visited sets against the sequential
// walk on a machine without a GPU (tests/hoststage/shim.cpp, tests/test_msv1_lanes_cpu.py).
#pragma once
#include <cstdint>

#if defined(JSP_LANES_HOST)
#define JSP_LANE_FN inline
#else
#define JSP_LANE_FN __device__ __forceinline__
#endif

namespace jsp {
namespace lanes {

// ---- the handful of instructions the pass is written in ---------------------------------------------------------
#if defined(JSP_LANES_HOST)
inline uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) {   // V_PERM_B32: bytes of {s0, s1} (s1 = bytes 0..3)
    const uint64_t in = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t s = (sel >> (8 * k)) & 0xFFu;
        uint32_t b;
        if (s >= 13) b = 0xFF;
        else if (s == 12) b = 0;
        else if (s >= 8) b = ((in >> (8 * (2 * (s - 8) + 1) + 7)) & 1u) ? 0xFFu : 0u;   // sign of byte 1, 3, 5, 7
        else b = (uint32_t)(in >> (8 * s)) & 0xFFu;
        out |= b << (8 * k);
    }
    return out;
}
inline int32_t sdot4(uint32_t a, uint32_t b, int32_t c) {
    for (int k = 0; k < 4; ++k) c += (int32_t)(int8_t)(a >> (8 * k)) * (int32_t)(int8_t)(b >> (8 * k));
    return c;
}
inline uint32_t sign_mask(uint32_t v, int bit) { return (uint32_t)((int32_t)(v << (31 - bit)) >> 31); }   // 0 / ~0
inline uint32_t sat_add(uint32_t a, uint32_t b) { const uint64_t s = (uint64_t)a + b; return s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s; }
inline uint32_t pk_add_sat_u16(uint32_t a, uint32_t b) {
    uint32_t lo = (a & 0xFFFFu) + (b & 0xFFFFu), hi = (a >> 16) + (b >> 16);
    return (lo > 0xFFFFu ? 0xFFFFu : lo) | ((hi > 0xFFFFu ? 0xFFFFu : hi) << 16);
}
inline uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    const uint32_t lo = (a & 0xFFFFu) > (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu), hi = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}
inline uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return (((a & 0xFFFFu) - (b & 0xFFFFu)) & 0xFFFFu) | (((a >> 16) - (b >> 16)) << 16); }
inline int first_bit(uint32_t v) { return __builtin_ctz(v); }
#else
typedef unsigned short lanes_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lanes_us2 lanes_as_us2(uint32_t v) { lanes_us2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t lanes_as_u32(lanes_us2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ int32_t sdot4(uint32_t a, uint32_t b, int32_t c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
template <int BIT> __device__ __forceinline__ uint32_t sign_mask_c(uint32_t v) {
    uint32_t m;   // (asm: the compiler turns the shift pair back into and + compare + select)
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(v), "n"(BIT));
    return m;
}
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ uint32_t pk_add_sat_u16(uint32_t a, uint32_t b) { return lanes_as_u32(__builtin_elementwise_add_sat(lanes_as_us2(a), lanes_as_us2(b))); }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) { return lanes_as_u32(__builtin_elementwise_max(lanes_as_us2(a), lanes_as_us2(b))); }
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return lanes_as_u32((lanes_us2)(lanes_as_us2(a) - lanes_as_us2(b))); }
__device__ __forceinline__ int first_bit(uint32_t v) { return __builtin_ctz(v); }
#endif
#if defined(JSP_LANES_HOST)
template <int BIT> inline uint32_t sign_mask_c(uint32_t v) { return sign_mask(v, BIT); }
#endif

constexpr uint32_t REST_OF_FRAME = 0xFFFFFu;   // what a skip code with count 0 covers (MSVideo1.hx:131-133: skip = -1 never returns to 0)

// bit s of the result = bit 15 of 16-bit word s of d[0 .. ND) (ND a multiple of 2, at most 16)
template <int ND>
JSP_LANE_FN uint32_t plane15(const uint32_t* d) {
    static_assert(ND % 2 == 0 && ND <= 16, "pairs of dwords, at most 32 words");
    uint32_t out = 0;
#pragma unroll
    for (int g = 0; g < ND / 4; ++g) {          // eight words -> one byte
        const uint32_t m0 = perm_b32(d[4 * g + 1], d[4 * g], 0x0B0A0908u), m1 = perm_b32(d[4 * g + 3], d[4 * g + 2], 0x0B0A0908u);
        const uint32_t byte = (uint32_t)sdot4(m0, 0xF8FCFEFFu, sdot4(m1, 0x80C0E0F0u, 0));
        out |= byte << (8 * g);
    }
    if (ND % 4) {                                // a last pair of dwords: four words
        const uint32_t m0 = perm_b32(d[ND - 1], d[ND - 2], 0x0B0A0908u);
        out |= (uint32_t)sdot4(m0, 0xF8FCFEFFu, 0) << (2 * (ND - 2));
    }
    return out;
}

// The lane's LS slots as masks (bit s = slot s).  A slot is exactly one of: short (neither M nor L), mid (M), long (L).
// Z marks the special slots — always short — whose block count is not 1: skip codes (K: count from the word), slots
// past the end of the frame's data (count 0), the 8-bit end marker (EM: count 0).
struct Masks {
    uint32_t M, L, Z, K, EM, valid;
};

// `w`: the lane's LS/2 dwords + one dword of halo (bytes past the frame's data are zero); `nvalid`: how many of the
// lane's slots lie inside the frame's code units (>= LS: all of them).
template <int BITS, int LS>
JSP_LANE_FN Masks build_masks(const uint32_t (&w)[LS / 2 + 1], uint32_t nvalid) {
    constexpr int ND = LS / 2;
    constexpr uint32_t ALL = LS == 32 ? 0xFFFFFFFFu : ((1u << (LS & 31)) - 1u);
    Masks m;
    m.valid = nvalid >= (uint32_t)LS ? ALL : ((1u << (nvalid & 31u)) - 1u);
    const uint32_t n15 = plane15<ND>(w);                                   // high byte >= 0x80
    uint32_t t[ND];
    // skip codes: high byte 0x84..0x87 (both depths: MSVideo1.hx:131, 315)
#pragma unroll
    for (int i = 0; i < ND; ++i) t[i] = pk_add_sat_u16((w[i] ^ 0x84008400u) & 0xFC00FC00u, 0x7FFF7FFFu);   // bit 15: NOT a skip code
    m.K = ~plane15<ND>(t) & m.valid;
    if (BITS == 16) {
        // pattern code (high byte < 0x80): 8 colours when bit 15 of the NEXT word is set (MSVideo1.hx:142), else 2
        const uint32_t next15 = (n15 >> 1) | (((w[ND] >> 15) & 1u) << (LS - 1));
        const uint32_t pattern = ~n15 & m.valid;
        m.L = pattern & next15;
        m.M = pattern & ~next15;
        m.EM = 0;
    } else {
        // 8-bit: word 0 = end of data (MSVideo1.hx:313); high byte < 0x80: 2 colours (2 slots); >= 0x90: 8 colours (5 slots)
#pragma unroll
        for (int i = 0; i < ND; ++i) t[i] = pk_add_sat_u16(w[i], 0x7FFF7FFFu);                              // bit 15: word != 0
        m.EM = ~plane15<ND>(t) & m.valid;
#pragma unroll
        for (int i = 0; i < ND; ++i) t[i] = pk_sub_u16(pk_max_u16(w[i], 0x8FFF8FFFu), 0x90009000u);         // bit 15: word < 0x9000
        m.L = ~plane15<ND>(t) & m.valid;
        m.M = ~n15 & ~m.EM & m.valid;
    }
    m.Z = (m.K | m.EM | ~m.valid) & ALL;
    return m;
}

// blocks a special slot covers, << 4 (the tables keep the exit slot in their low 4 bits)
// (`w`: the lane's words WHERE THEY LIE — LDS in the kernel —, read only here: special slots are one word in a few hundred, and the table pass need
// not keep all LS / 2 + 1 words of the lane in registers for them)
template <int LS>
JSP_LANE_FN uint32_t special_count16(const uint32_t* w, const Masks& m, int s) {
    const uint32_t word = (w[s >> 1] >> (16 * (s & 1))) & 0xFFFFu, n = word & 0x3FFu;
    const uint32_t cnt = n ? n : REST_OF_FRAME;
    return ((m.K >> s) & 1u) ? cnt << 4 : 0u;
}

// Left alone the compiler computes every step's two sign masks and its addend FIRST — 3 x 32 values alive at once — and the dependent chain afterwards: that
// alone makes msv1_fused_kernel a 111-VGPR kernel (four workgroups per CU).  An empty volatile asm that "rewrites" a step's result and the three masks the next
// step reads keeps every step to itself: 42 VGPRs (52 fenced every 8 steps), five workgroups per CU (LDS then bounds it) — and no faster: on a quiet box
// all-8-colour 0.680 / 0.675 -> 0.667 / 0.669, all-solid 0.893 / 0.857 -> 0.858 / 0.827, M1 and inter frames the same, 8-bit +1.5 % (profiles/
// r05_fused_lane_table_fence_ab.txt); on two noisy boxes nothing consistent (r05_fused_fence_variants_noisy_box*.txt).  What the hoisting buys in
// instruction-level parallelism is worth what the fifth workgroup is.  JSP_LANE_FENCE_EVERY = steps between fences; 64 = none (the product).
#ifndef JSP_LANE_FENCE_EVERY
#define JSP_LANE_FENCE_EVERY 64
#endif
#if defined(JSP_LANES_HOST)
#define JSP_LANE_SCHED_FENCE(v, a, b, c) do { } while (0)
#else
#define JSP_LANE_SCHED_FENCE(v, a, b, c) asm volatile("" : "+v"(v), "+v"(a), "+v"(b), "+v"(c))
#endif
// The 9-entry table of the lane: tab[e] = exit slot (0..8, relative to the next lane's first slot) | blocks << 4 of the
// chain that enters at slot e.  `zw`: the OR of Z over the wave (any superset works: it only gates the special path).
template <int BITS, int LS>
JSP_LANE_FN void lane_table(const uint32_t* w, const Masks& m, uint32_t zw, uint32_t (&tab)[9]) {
    constexpr int MID = BITS == 16 ? 2 : 1, LONG = BITS == 16 ? 8 : 4;     // window index of slot + 3 / + 9 (8-bit: + 2 / + 5)
    uint32_t dw[9];                                                         // dw[k] = value of slot s + 1 + k
    uint32_t fM = m.M, fL = m.L, fZ = m.Z;                                  // (the masks the steps read, threaded through the fences)
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[k] = (uint32_t)k;
#define JSP_LANE_STEP(S)                                                                                     \
    {                                                                                                        \
        const uint32_t mm = sign_mask_c<S>(fM), ml = sign_mask_c<S>(fL);                                     \
        uint32_t nx = (dw[MID] & mm) | (dw[0] & ~mm);                                                        \
        nx = (dw[LONG] & ml) | (nx & ~ml);                                                                   \
        uint32_t add = 16u;                                                                                  \
        if (zw & (1u << S)) {                                                                                \
            const uint32_t mz = sign_mask_c<S>(fZ);                                                          \
            add = (special_count16<LS>(w, m, S) & mz) | (16u & ~mz);                                         \
        }                                                                                                    \
        const uint32_t v = sat_add(nx, add);                                                                 \
        dw[8] = dw[7]; dw[7] = dw[6]; dw[6] = dw[5]; dw[5] = dw[4]; dw[4] = dw[3]; dw[3] = dw[2]; dw[2] = dw[1]; dw[1] = dw[0]; \
        dw[0] = v;                                                                                           \
        if (((S) & (JSP_LANE_FENCE_EVERY - 1)) == 0) JSP_LANE_SCHED_FENCE(dw[0], fM, fL, fZ);                \
    }
    if (LS == 32) {
        JSP_LANE_STEP(31) JSP_LANE_STEP(30) JSP_LANE_STEP(29) JSP_LANE_STEP(28) JSP_LANE_STEP(27) JSP_LANE_STEP(26) JSP_LANE_STEP(25) JSP_LANE_STEP(24)
        JSP_LANE_STEP(23) JSP_LANE_STEP(22) JSP_LANE_STEP(21) JSP_LANE_STEP(20) JSP_LANE_STEP(19) JSP_LANE_STEP(18) JSP_LANE_STEP(17) JSP_LANE_STEP(16)
    }
    JSP_LANE_STEP(15) JSP_LANE_STEP(14) JSP_LANE_STEP(13) JSP_LANE_STEP(12) JSP_LANE_STEP(11) JSP_LANE_STEP(10) JSP_LANE_STEP(9) JSP_LANE_STEP(8)
    JSP_LANE_STEP(7) JSP_LANE_STEP(6) JSP_LANE_STEP(5) JSP_LANE_STEP(4) JSP_LANE_STEP(3) JSP_LANE_STEP(2) JSP_LANE_STEP(1) JSP_LANE_STEP(0)
#undef JSP_LANE_STEP
#pragma unroll
    for (int e = 0; e < 9; ++e) tab[e] = dw[e];
}

// The slots of the lane that the chain entering at `entry_slot` (0..8) visits.
template <int BITS, int LS>
JSP_LANE_FN uint32_t visited(const Masks& m, uint32_t entry_slot) {
    constexpr int MIDL = BITS == 16 ? 3 : 2, LONGL = BITS == 16 ? 9 : 5;
    constexpr uint32_t ALL = LS == 32 ? 0xFFFFFFFFu : ((1u << (LS & 31)) - 1u);
    const uint32_t S = ~(m.M | m.L);
    uint32_t V = (1u << entry_slot) & ALL;      // (an entry slot past a 16-slot lane's end: nothing of this lane is visited)
    for (;;) {
        const uint32_t V2 = (V | ((V & S) << 1) | ((V & m.M) << MIDL) | ((V & m.L) << LONGL)) & ALL;
        if (V2 == V) break;
        V = V2;
    }
    return V;
}

}  // namespace lanes
}  // namespace jsp
