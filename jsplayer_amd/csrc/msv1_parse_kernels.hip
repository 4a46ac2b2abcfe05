// On-GPU parse of MSVideo1 code streams: raw frame bytes in HBM -> the per-block
// descriptor table msv1_blocks_kernel consumes, without the sequential host walk.
//
// The walk "offset += length(code at offset)" (MSVideo1.hx:128-181 / 311-364) is a chain through the
// even byte offsets ("slots").  A code is 1, 3 or 9 slots long (16-bit; 1, 2 or 5 for 8-bit), so a
// chain that enters a run of slots does so at one of its first 9 slots.  For any run of slots the
// function   entry slot (0..8) -> (exit slot into the next run (0..8), blocks covered)
// is a 9-entry table, and tables compose associatively.  Three kernels:
//   K1 msv1_parse_tiles   : one workgroup per 16 KiB tile; each lane builds the table of its 32 slots
//                           with a register-resident reverse DP (static indices only), the workgroup
//                           reduces the 256 tables by composition -> one table per tile;
//   K2 msv1_parse_chain   : one workgroup per frame walks its <= few hundred tile tables from entry 0
//                           (in LDS) -> true entry slot and first block index of every tile;
//   K3 msv1_parse_emit    : K1's work again + an inclusive scan of the lane tables, so every lane
//                           knows where the real chain enters its slots and which block comes first;
//                           it then replays its 32 slots and writes one descriptor per coded block.
// Every tile owns the block range its codes cover and writes that whole range — coded blocks and
// MSV1_DESC_SKIP for skipped ones — through an LDS staging buffer, so descriptor writes are coalesced.
// Frames the reference treats specially (stream ends before all blocks are covered, 8-bit end
// marker, skip code with no previous frame, 16-bit early-outs) are detected from the counters these
// kernels return and re-done by the host parser (msv1_host.cpp), which is exact for every input.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "msv1.h"
#include "msv1_lanes.h"
#include "msv1_decode.h"
#include <msv1_fused_hooks.h>   // (angle brackets: a lab build puts its own in front on the include path, see the Makefile)

namespace jsp {
namespace {

constexpr int PWG = 256;                 // lanes per workgroup
constexpr int LSLOTS = JSP_BATCH_LS;     // slots (2 bytes each) per lane (msv1_fused_hooks.h)
constexpr int TSLOTS = PWG * LSLOTS;     // slots per tile
constexpr uint32_t BSAT = (1u << 28) - 1;

__device__ __forceinline__ uint32_t pack(uint32_t exit_slot, uint32_t blocks) { return exit_slot | (blocks << 4); }
__device__ __forceinline__ uint32_t add_blocks(uint32_t v, uint32_t n) {
    const uint32_t b = (v >> 4) + n;
    return (v & 15u) | ((b > BSAT ? BSAT : b) << 4);
}
// table composition: first `a` (value for one entry), then table `tb`
__device__ __forceinline__ uint32_t compose(uint32_t a, const uint32_t* tb) {
    return add_blocks(tb[a & 15u], a >> 4);
}

// Per-slot classification, packed: bits 0..3 length in slots, bit 4 coded block (not a skip code),
// bit 5 end marker (8-bit), bits 8..27 blocks covered (BSAT-clamped later)
template <int BITS>
__device__ __forceinline__ uint32_t classify(uint32_t a, uint32_t b, uint32_t hi, bool hi_ok) {
    if ((b & 0xFCu) == 0x84u) {
        const uint32_t n = ((b - 0x84u) << 8) + a;          // 0 = "the rest of the frame"
        return 1u | ((n ? n : 0xFFFFFu) << 8);
    }
    if (BITS == 16) {
        if (b < 0x80u) return ((hi_ok && (hi & 0x80u)) ? 9u : 3u) | 16u | (1u << 8);
        return 1u | 16u | (1u << 8);
    }
    if (a == 0u && b == 0u) return 1u | 32u;                 // end marker: covers nothing, flagged
    if (b < 0x80u) return 2u | 16u | (1u << 8);
    if (b >= 0x90u) return 5u | 16u | (1u << 8);
    return 1u | 16u | (1u << 8);
}

// The lane's 32 slots: classification of every slot (cls[]) and the 9-entry table tab[e].
// `w` = the lane's 64 bytes + 4 bytes of halo as 17 dwords; `p0` = absolute byte offset of slot 0;
// `end` = absolute end of the frame's bytes.  Everything indexes registers statically.
template <int BITS, int LS = LSLOTS>
__device__ __forceinline__ void lane_table(const uint32_t (&w)[LS / 2 + 1], uint32_t p0, uint32_t end, uint32_t (&cls)[LS],
                                           uint32_t (&tab)[9]) {
#pragma unroll
    for (int s = 0; s < LS; ++s) {
        const uint32_t a = (w[(2 * s) >> 2] >> (8 * ((2 * s) & 3))) & 0xFFu;
        const uint32_t b = (w[(2 * s + 1) >> 2] >> (8 * ((2 * s + 1) & 3))) & 0xFFu;
        const uint32_t hi = (w[(2 * s + 3) >> 2] >> (8 * ((2 * s + 3) & 3))) & 0xFFu;
        const uint32_t p = p0 + 2u * s;
        // slots past the end of the data cover nothing (frames are even-padded: a present => b present)
        cls[s] = p < end ? classify<BITS>(a, b, hi, p + 3u < end) : 1u;
    }
    // reverse DP with a 9-deep window: d[k] = value of slot s+1+k
    uint32_t d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = pack(k, 0);
#pragma unroll
    for (int s = LS - 1; s >= 0; --s) {
        const uint32_t len = cls[s] & 15u;
        uint32_t nx;
        if (BITS == 16) nx = len == 1u ? d[0] : (len == 3u ? d[2] : d[8]);
        else nx = len == 1u ? d[0] : (len == 2u ? d[1] : d[4]);
        const uint32_t v = add_blocks(nx, cls[s] >> 8);
#pragma unroll
        for (int k = 8; k > 0; --k) d[k] = d[k - 1];
        d[0] = v;
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) tab[e] = d[e];
}

// lane_table for a lane whose 32 slots and their 4-byte halo lie wholly inside the frame's data (every tile of a frame
// but the last): no end-of-data tests, the two slots of a dword classified straight from its halves, block counts
// added with the hardware's saturating add.  A saturated value is all ones: its exit-slot bits (15) then point past the
// 9 table entries, which is harmless — every consumer sees "more blocks than the frame has" first, and the LDS rows
// read through such an index lie inside the kernels' own arenas.
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { return __builtin_elementwise_add_sat(a, b); }

template <int BITS, int LS = LSLOTS>
__device__ __forceinline__ void lane_table_fast(const uint32_t (&w)[LS / 2 + 1], uint32_t (&cls)[LS], uint32_t (&tab)[9]) {
    // one reverse pass: a slot is classified and folded into the 9-deep window (dw[k] = value of slot s+1+k) while
    // its predicates are still in flight
    uint32_t dw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[k] = pack(k, 0);
#pragma unroll
    for (int s = LS - 1; s >= 0; --s) {
        const uint32_t d = w[s >> 1], dn = w[(s + 1) >> 1];
        const int sh = (s & 1) * 16, shn = ((s + 1) & 1) * 16;       // where the slot's / the next slot's 16-bit word sits
        const bool skip = (d & (0xFC00u << sh)) == (0x8400u << sh);
        const uint32_t n = (d >> sh) & 0x3FFu;                         // skip count (valid when `skip`)
        const uint32_t nblk = n ? n : 0xFFFFFu;                        // 0 = "the rest of the frame"
        uint32_t c, nx;
        if (BITS == 16) {
            const bool pattern = (d & (0x8000u << sh)) == 0u;          // high byte < 0x80
            const bool eight = (dn & (0x8000u << shn)) != 0u;          // bit 15 of the first colour
            c = pattern ? (eight ? 9u | 0x110u : 3u | 0x110u) : 1u | 0x110u;
            nx = pattern ? (eight ? dw[8] : dw[2]) : dw[0];
        } else {
            const uint32_t cw = (d >> sh) & 0xFFFFu;
            const bool two = cw < 0x8000u, eight = cw >= 0x9000u;
            c = two ? 2u | 0x110u : (eight ? 5u | 0x110u : 1u | 0x110u);
            nx = two ? dw[1] : (eight ? dw[4] : dw[0]);
            if (cw == 0u) c = 1u | 32u;                                // end marker: covers nothing
        }
        c = skip ? ((nblk << 8) | 1u) : c;
        cls[s] = c;
        const uint32_t v = sat_add(nx, (c >> 4) & 0xFFFFFFF0u);
#pragma unroll
        for (int k = 8; k > 0; --k) dw[k] = dw[k - 1];
        dw[0] = v;
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) tab[e] = dw[e];
}

// Stage a tile through LDS and hand every lane its 17 dwords.
__device__ __forceinline__ void load_lane_bytes(const uint8_t* __restrict__ stream, uint32_t tile_byte0, uint32_t end,
                                                uint32_t* lds_bytes /* TSLOTS*2/4 + 4 dwords */, uint32_t (&w)[LSLOTS / 2 + 1]) {
    constexpr uint32_t tile_bytes = TSLOTS * 2;
    constexpr int NLOAD = (tile_bytes + 16u + PWG * 16u - 1u) / (PWG * 16u);
    uint4 v[NLOAD];   // every load is out before the first is waited for (one by one: NLOAD memory round trips in a row)
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
        const uint32_t o = threadIdx.x * 16u + (uint32_t)q * (PWG * 16u);
        v[q] = make_uint4(0, 0, 0, 0);
        if (o < tile_bytes + 16u && tile_byte0 + o < end) v[q] = *reinterpret_cast<const uint4*>(stream + tile_byte0 + o);  // buffers are padded
    }
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) {
        const uint32_t o = threadIdx.x * 16u + (uint32_t)q * (PWG * 16u);
        if (o < tile_bytes + 16u) *reinterpret_cast<uint4*>(lds_bytes + o / 4) = v[q];
    }
    __syncthreads();
    const uint32_t* mine = lds_bytes + threadIdx.x * (LSLOTS * 2 / 4);
#pragma unroll
    for (int k = 0; k < LSLOTS / 2 + 1; ++k) w[k] = mine[k];
}

template <int BITS>
__global__ __launch_bounds__(PWG) void msv1_parse_tiles(const uint8_t* __restrict__ stream,
                                                        const Msv1ParseFrame* __restrict__ frames,
                                                        const uint32_t* __restrict__ tile_frame,
                                                        uint32_t* __restrict__ tile_tab) {
    __shared__ __align__(16) uint32_t lds_bytes[TSLOTS * 2 / 4 + 8];
    __shared__ uint32_t tabs[2][PWG][9];
    const uint32_t t = blockIdx.x;
    const Msv1ParseFrame fr = frames[tile_frame[t]];
    if (fr.host_parsed) return;
    const uint32_t tile_byte0 = fr.beg + (t - fr.first_tile) * (TSLOTS * 2);
    uint32_t w[LSLOTS / 2 + 1], cls[LSLOTS], tab[9];
    load_lane_bytes(stream, tile_byte0, fr.end, lds_bytes, w);
    lane_table<BITS>(w, tile_byte0 + threadIdx.x * (LSLOTS * 2), fr.end, cls, tab);
#pragma unroll
    for (int e = 0; e < 9; ++e) tabs[0][threadIdx.x][e] = tab[e];
    __syncthreads();
    // reduce by composition, every (node, entry) pair one work item, ping-pong between two buffers
    int cur = 0;
    for (int l = 1; l <= 8; ++l) {
        const int nodes = PWG >> l;
        for (int k = threadIdx.x; k < nodes * 9; k += PWG) {
            const int j = k / 9, e = k - j * 9;
            tabs[cur ^ 1][j][e] = compose(tabs[cur][2 * j][e], tabs[cur][2 * j + 1]);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (threadIdx.x < 9) tile_tab[t * 9 + threadIdx.x] = tabs[cur][0][threadIdx.x];
}

// One workgroup per frame: chain the tile tables from entry slot 0.
__global__ __launch_bounds__(64) void msv1_parse_chain(const Msv1ParseFrame* __restrict__ frames,
                                                       const uint32_t* __restrict__ tile_tab,
                                                       uint32_t* __restrict__ tile_entry, uint32_t* __restrict__ tile_block0,
                                                       Msv1FrameInfo* __restrict__ info) {
    extern __shared__ uint32_t tt[];  // ntiles * 9
    const Msv1ParseFrame fr = frames[blockIdx.x];
    if (fr.host_parsed) return;
    for (uint32_t k = threadIdx.x; k < fr.ntiles * 9u; k += 64) tt[k] = tile_tab[fr.first_tile * 9u + k];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t e = 0, blocks = 0;
        for (uint32_t t = 0; t < fr.ntiles; ++t) {
            tile_entry[fr.first_tile + t] = e;
            tile_block0[fr.first_tile + t] = blocks;
            const uint32_t v = tt[t * 9u + e];
            e = v & 15u;
            blocks = blocks + (v >> 4) > BSAT ? BSAT : blocks + (v >> 4);
        }
        info[blockIdx.x].total_blocks = blocks;
    }
}

// Composition tree over the 256 lane tables, kept whole in LDS: level 0 = lane tables, level l node j
// covers lanes [j*2^l, (j+1)*2^l).  Node (l, j) lives at row tree_row(l) + j.
__device__ __forceinline__ int tree_row(int level) { return 2 * PWG - (2 * PWG >> level); }  // 0,256,384,...,510

template <int BITS>
__global__ __launch_bounds__(PWG) void msv1_parse_emit(const uint8_t* __restrict__ stream,
                                                       const Msv1ParseFrame* __restrict__ frames,
                                                       const uint32_t* __restrict__ tile_frame,
                                                       const uint32_t* __restrict__ tile_entry,
                                                       const uint32_t* __restrict__ tile_block0,
                                                       uint32_t* __restrict__ desc, Msv1FrameInfo* __restrict__ info,
                                                       uint32_t nblocks, uint32_t s1_first_block) {
    // one LDS arena: [tile bytes | composition tree | per-node entry]; once every lane knows where
    // the chain enters its slots, the front of the arena is reused as the descriptor staging buffer
    constexpr int BYTES_W = TSLOTS * 2 / 4 + 8, TREE_W = 2 * PWG * 9, STAGE = TSLOTS;
    static_assert(BYTES_W + TREE_W >= STAGE, "staging overlay must fit in front of `enter`");
    __shared__ __align__(16) uint32_t arena[BYTES_W + TREE_W + 2 * PWG];
    uint32_t* lds_bytes = arena;
    uint32_t (*tree)[9] = reinterpret_cast<uint32_t (*)[9]>(arena + BYTES_W);
    uint32_t* enter = arena + BYTES_W + TREE_W;
    uint32_t* stage = arena;
    const uint32_t t = blockIdx.x;
    const uint32_t f = tile_frame[t];
    const uint32_t te = tile_entry[t], tb0 = tile_block0[t];   // asked for now: needed after the up-sweep, a round trip away
    const Msv1ParseFrame fr = frames[f];
    if (fr.host_parsed) return;
    const uint32_t tile_byte0 = fr.beg + (t - fr.first_tile) * (TSLOTS * 2);
    uint32_t w[LSLOTS / 2 + 1], cls[LSLOTS], tab[9];
    load_lane_bytes(stream, tile_byte0, fr.end, lds_bytes, w);
    const uint32_t p0 = tile_byte0 + threadIdx.x * (LSLOTS * 2);
    lane_table<BITS>(w, p0, fr.end, cls, tab);
#pragma unroll
    for (int e = 0; e < 9; ++e) tree[threadIdx.x][e] = tab[e];
    __syncthreads();
    // up-sweep: every (node, entry) pair of a level is one work item
    for (int l = 1; l <= 8; ++l) {
        const int nodes = PWG >> l, lo = tree_row(l - 1), hi = tree_row(l);
        for (int k = threadIdx.x; k < nodes * 9; k += PWG) {
            const int j = k / 9, e = k - j * 9;
            tree[hi + j][e] = compose(tree[lo + 2 * j][e], tree[lo + 2 * j + 1]);
        }
        __syncthreads();
    }
    // down-sweep of ONE value per node: where the real chain enters the node and with which block.
    // packed as entry | block << 4 (block saturates at BSAT like everything else)
    if (threadIdx.x == 0) enter[tree_row(8)] = pack(te, tb0);
    __syncthreads();
    for (int l = 8; l >= 1; --l) {
        const int nodes = PWG >> l, hi = tree_row(l), lo = tree_row(l - 1);
        if ((int)threadIdx.x < nodes) {
            const uint32_t v = enter[hi + threadIdx.x];
            enter[lo + 2 * threadIdx.x] = v;
            enter[lo + 2 * threadIdx.x + 1] = add_blocks(tree[lo + 2 * threadIdx.x][v & 15u], v >> 4);
        }
        __syncthreads();
    }
    const uint32_t mine = enter[threadIdx.x];
    // blocks this tile is responsible for: [tb0, span_end)
    const uint32_t whole = add_blocks(tree[tree_row(8)][te], tb0) >> 4;
    const uint32_t span_end = whole < nblocks ? whole : nblocks;
    const uint32_t span = span_end > tb0 ? span_end - tb0 : 0u;
    const bool staged = span <= (uint32_t)STAGE;
    __syncthreads();                                   // everyone has read tree/enter: the arena is free
    uint32_t* gdesc = desc + fr.desc_base;
    if (staged) {
        for (uint32_t i = threadIdx.x; i < span; i += PWG) stage[i] = MSV1_DESC_SKIP;
    } else {
        for (uint32_t i = tb0 + threadIdx.x; i < span_end; i += PWG) gdesc[i] = MSV1_DESC_SKIP;
    }
    __syncthreads();                                   // (vmcnt(0) too: the global fill has landed)
    uint32_t pos = mine & 15u, blk = mine >> 4;
    uint32_t ncoded = 0, nskipcodes = 0, flags = 0, s1 = 0, consumed = 0;
#pragma unroll
    for (int s = 0; s < LSLOTS; ++s) {
        if (pos == (uint32_t)s && blk < nblocks) {
            const uint32_t c = cls[s];
            if (p0 + 2u * s < fr.end) {
                if (c & 16u) {
                    if (staged) stage[blk - tb0] = p0 + 2u * s;
                    else gdesc[blk] = p0 + 2u * s;
                    ++ncoded;
                    s1 |= blk >= s1_first_block ? 1u : 0u;
                } else if (c & 32u) flags |= MSV1_INFO_END_MARKER;
                else ++nskipcodes;
                const uint32_t nb = blk + (c >> 8);
                blk = nb > BSAT ? BSAT : nb;
                if (blk >= nblocks) consumed = p0 + 2u * s + 2u * (c & 15u) - fr.beg;
            }
            pos += c & 15u;
        }
    }
    if (staged) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < span; i += PWG) gdesc[tb0 + i] = stage[i];   // coalesced write-out
    }
    // per-frame counters: one atomic per wave and counter
    const unsigned long long any = __ballot(ncoded | nskipcodes | flags | s1 | consumed);
    if (any) {
        uint32_t a = ncoded, b = nskipcodes, c2 = flags | (s1 ? MSV1_INFO_S1 : 0u);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor((int)a, o);
            b += __shfl_xor((int)b, o);
            c2 |= (uint32_t)__shfl_xor((int)c2, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (a) atomicAdd(&info[f].n_coded, a);
            if (b) atomicAdd(&info[f].n_skip_codes, b);
            if (c2) atomicOr(&info[f].flags, c2);
        }
        if (consumed) info[f].consumed = consumed;   // exactly one lane covers the last block
    }
}


// ---------------------------------------------------------------------------------------------------------
// Fused parse + reconstruction: raw stream bytes -> pixels in ONE launch, no descriptor table in HBM.
//
// One workgroup per 16 KiB tile of a frame's stream, in stream order (workgroups are dispatched in increasing
// index, so every tile with a lower number is already resident or done: what the look-back below relies on; a
// tile that waits too long raises a fault word: the batch is then reported as failed, never silently wrong).  Per tile:
//   1. the tile's bytes go to LDS once (bytes past the frame's end zeroed); every lane builds the table of
//      its 32 slots (register-resident reverse DP) and the 256 tables are reduced by composition -> the
//      tile's table "entry slot 0..8 -> (exit slot, blocks covered)";
//   2. the tile publishes its table: 9 words of {launch epoch, value}, each a single 8-byte agent-scope
//      atomic store — a word is its own flag, there is no payload to order behind it;
//   3. look-back: the tile reads the tables of ALL earlier tiles of its frame (contiguous, 72 B per tile, polled
//      with agent-scope atomic loads until their epoch matches) and chains them from entry slot 0 — the first
//      tile of a frame starts immediately, so no tile ever waits for a prefix, only for tables that every
//      resident tile publishes a few microseconds after it starts;
//   4. down-sweep of one value per tree node: where the real chain enters each lane's slots and which block
//      comes first; lanes replay their slots and drop, per coded block, the offset of its code into a
//      16-bit staging window in LDS (skipped blocks keep the SKIP mark) — the descriptor table of the 3-kernel
//      path, but only ever in LDS;
//   5. reconstruction straight from the staged bytes: lane = block, raster order, so a wave still writes 1 KiB
//      contiguous per pixel row; a code is two LDS reads (4 + 16 bytes at 2-byte alignment).
// The whole frame batch is one launch; the stream is read from HBM exactly once.
// Frames the host parser has to settle (Msv1ParseFrame::host_parsed) are skipped here and take the
// descriptor path.  Requires X % 4 == 0 and 16-byte aligned frame buffers (msv1_codec.cpp checks).
constexpr uint32_t F_SKIP = 0xFFFFu;    // staging window: the block is skipped (a copy of the previous frame's)
constexpr uint32_t F_NONE = 0xFFFEu;    //   ... the block belongs to another tile (code offsets are even and below 16 KiB)
// look-back: LBW words per lane hold the tables of the nearest LBW * PWG / 9 earlier tiles of the frame in one poll (3 words = 85
// tiles with 16 KiB tiles, 5 words = 142 with 8 KiB tiles: a 1080p M1 frame is 64 / 127 tiles); the chain through a batch is
// walked in LBW * 7 / 3 segments side by side
constexpr int LOOKBACK_SPIN_LIMIT = 1 << 18;   // polls before the tile gives up and raises the fault word
constexpr int VERDICT_SPIN_LIMIT = 1 << 15;    // MODE 3: polls for the frame's other tiles before the frame goes to the host path

// OR of a 32-bit value over the wave, as a wave-uniform (scalar) value: four DPP steps leave every row's OR in all 16 of
// its lanes, then one lane of each row is read.
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// One launch: grid = tiles, in stream order.
// Measured and dropped (profiles/archive/r02_fused_notes.txt): a persistent variant (rounds of tiles, the next tile's bytes
// prefetched into registers) hid the tile load but ran the workgroups of a CU in lockstep — every phase then competes
// for the same issue slots — and was 10 % slower.
// The asynchronous per-frame path (one frame per launch):
//   MODE 3: one launch — every tile parses and reports into `info`, waits for the whole frame's verdict, then writes its
//           pixels or nothing (see msv1.h); what frames of up to MSV1_MERGED_MAX_TILES tiles get;
//   MODE 1, "scout": everything up to the replay, no pixel is written; reports in `info` what the host stage of the
//           descriptor path learns from its parse — stream too short, coded block in a significant block row (stage-1
//           significance), 8-bit end marker or skip code on the chain;
//   MODE 2: the decode proper, which returns at once when the scout found one of the conditions in `bad_mask` — such a
//           frame is re-done by the synchronous path, and must find `dst` exactly as the caller left it.
// MODE 4: the batch form writing block tables instead of pixels (the descriptor parse of inter-frame batches).
// MODE 0 is the batch form (no report); its frames never read a previous frame (msv1_codec.cpp only fuses frames without
// skipped blocks and without a stage-2 compare), so it is compiled without the copy and compare paths: its decode loop
// then holds no load at all, and the row stores of consecutive blocks are never waited for.
// LS = slots (2 bytes) per lane: 32 (16 KiB tiles) for the batch forms, 16 (8 KiB tiles) for the one-frame-per-launch forms,
// whose few dozen workgroups have the GPU to themselves: half the serial work per tile, twice the tiles.
template <int BITS, int MODE, int LS>
__global__ __launch_bounds__(PWG, (MODE == 4 || MODE == 5) ? JSP_FUSED_WAVES_TABLES : JSP_FUSED_WAVES) void msv1_fused_kernel(const uint8_t* __restrict__ stream_p,
                                                            const Msv1TileRec* __restrict__ recs,
                                                            const int32_t* __restrict__ palette,
                                                            unsigned long long* __restrict__ agg_p, uint32_t epoch_p,
                                                            uint32_t tile0, uint32_t* __restrict__ fault_p,
                                                            uint32_t nblocks, int nbx, int X,
                                                            Msv1AsyncInfo* __restrict__ info_p, uint32_t s1_first_block,
                                                            uint32_t bad_mask_p, uint32_t* __restrict__ poison,
                                                            Msv1TileRec one_rec_p, Msv1AsyncInfo* __restrict__ host_info_p, uint32_t want_p,
                                                            uint8_t* __restrict__ keep_p, Msv1Riders riders) {
    // MODE 3 may carry SEVERAL frames in one launch (msv1.h, Msv1Riders): the first workgroups are the first frame's tiles, the rest belong to
    // the riders, with everything a frame calls its own — bytes, tables, report, record — taken from the rider's entry.  All frames parse
    // side by side; a rider paints when the frame in front is through (it may read its pixels).  Every other mode: the parameters as they are.
    uint32_t ride = 0;                                         // 0: the launch's first frame; i: rider i - 1
    if (MODE == 3)
        for (uint32_t q = 0; q < riders.count; ++q) ride = blockIdx.x >= riders.f[q].first_wg ? q + 1u : ride;
    const bool rider = MODE == 3 && ride != 0u;
    const Msv1Rider& rd = riders.f[rider ? ride - 1u : 0u];
    const uint32_t bid = rider ? blockIdx.x - rd.first_wg : blockIdx.x;
    const uint8_t* __restrict__ stream = rider ? rd.stream : stream_p;
    unsigned long long* __restrict__ agg = rider ? rd.agg : agg_p;
    const uint32_t epoch = rider ? rd.epoch : epoch_p;
    Msv1AsyncInfo* __restrict__ info = rider ? rd.info : info_p;
    uint32_t* __restrict__ fault = rider ? &rd.info->fault : fault_p;
    const uint32_t bad_mask = rider ? rd.bad_mask : bad_mask_p;
    Msv1AsyncInfo* __restrict__ host_info = rider ? rd.host_info : host_info_p;
    const uint32_t want = rider ? rd.want : want_p;
    uint8_t* __restrict__ keep = rider ? rd.keep : keep_p;
    // the frame in front of a rider: its report and the value its `finished` counter reaches when all its tiles are through
    const Msv1AsyncInfo* info_before = ride >= 2u ? riders.f[ride - 2u].info : info_p;
    const uint32_t want_before = ride >= 2u ? riders.f[ride - 2u].want : want_p;
    const uint32_t epoch_before = ride >= 2u ? riders.f[ride - 2u].epoch : epoch_p;
    constexpr bool INFO = MODE == 1 || MODE == 3;
    constexpr bool USES_PREV = MODE != 0;                      // copies of skipped blocks, stage-2 compare (see above)
    constexpr int TSLOTS = PWG * LS;                           // (shadow the file's 32-slot constants)
    constexpr int FSTAGE = 4096;                               // blocks per staging window (a 16 KiB tile of solid codes: two windows)
    constexpr uint32_t TILE_BYTES = TSLOTS * 2;
    constexpr int LSLOTS = LS;
    constexpr int LBW = LS == 32 ? 3 : 5, LOOKBACK_BATCH = LBW * PWG / 9, LOOKBACK_SEGS = LBW * 7 / 3;
    // one LDS arena: [tile bytes | lane tables | look-back scratch]; the tables' space becomes the staging window once every
    // lane knows where the chain enters its slots
    // The batch forms take their tiles tile-major over the batch's frames (tile j of every frame before tile j + 1 of any,
    // see msv1_launch_order): when a tile starts, its predecessor in the frame finished about a thousand workgroups ago and has
    // left the one value this tile needs — where the chain stands at the tile's first slot — so the look-back is ONE word,
    // asked for before anything else and there when it is needed (PREFIX).  The one-frame launches have all of a frame's
    // tiles in flight together: those publish their tables and every tile chains the tables of all tiles before it.
    constexpr bool PREFIX = MODE == 0 || MODE == 4 || MODE == 5;
    constexpr int BYTES_W = TSLOTS * 2 / 4 + 8, TAB_W = PWG * 9 + 16, ENTER_W = PREFIX ? 4 : LOOKBACK_BATCH * 9 + 4;
    static_assert(TAB_W * 2 >= FSTAGE, "staging window must fit in the tables' space");
    __shared__ __align__(16) uint32_t arena[BYTES_W + TAB_W + ENTER_W + JSP_FUSED_LDS_PAD];
    __shared__ uint32_t s_pal[BITS == 8 ? 256 : 1];
    __shared__ uint32_t s_entry, s_root[16], s_seg[PREFIX ? 1 : LOOKBACK_SEGS][9];
    __shared__ uint32_t s_grp[PWG / 64][7][16], s_wav[PWG / 64][16];   // (16 per row: a saturated value indexes entry 15, see msv1_lanes.h)
    uint8_t* lds_bytes = reinterpret_cast<uint8_t*>(arena);
    uint32_t* tabs = arena + BYTES_W;
    uint32_t* enter = arena + BYTES_W + TAB_W;
    uint16_t* stage = reinterpret_cast<uint16_t*>(arena + BYTES_W);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto wave_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    // Tiles are taken in grid order: workgroups are dispatched in increasing index, so every earlier tile of the
    // frame is resident (or done) when this one polls for its table.  Everything a tile needs to know comes in ONE
    // record (a uniform, scalar load); its bytes sit at t * TILE_BYTES (frames start on tile boundaries), so the
    // stream loads do not wait for the record: under a saturated write stream every dependent global round trip
    // costs microseconds.
    JSP_CLOCK_BEGIN();
    Msv1TileRec r = MODE == 3 ? (rider ? rd.rec : one_rec_p) : recs[tile0 + blockIdx.x];   // MODE 3: the record is a kernel argument
    if (MODE == 3) r.k = bid;
    const uint32_t t = PREFIX ? r.first_tile + r.k : tile0 + bid;   // the tile's number in stream order (its slot in `agg`)
    if (MODE != 3 && (r.flags & MSV1_TILE_SKIP)) return;
    // MODE 3 bookkeeping (thread 0): `arrived` counts the workgroups whose findings are in info->flags, `finished` those
    // that have written their last pixel; both run on from launch to launch (`want` = their value once this launch is
    // through), so nothing has to be zeroed between frames.  The last workgroup to finish hands the report to the host
    // (pinned memory) and clears the words for the next launch.
    auto arrive = [&] { __threadfence(); atomicAdd(&info->arrived, 1u); };
    auto finish = [&] {
        __threadfence();
        if (atomicAdd(&info->finished, 1u) + 1u == want) {
            __threadfence();
            host_info->flags = __hip_atomic_load(&info->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_info->signif = __hip_atomic_load(&info->signif, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_info->fault = __hip_atomic_load(&info->fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&info->flags, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&info->signif, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&info->fault, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&info->verdict, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (every tile has obeyed it by now)
        }
    };
    if (MODE == 3) {
        // (Set by earlier launches on this stream — then every tile of this launch sees it — or by a frame of THIS launch after its verdict:
        // with several frames per launch a tile dispatched late may see the word while an earlier tile of its frame did not.  A tile that
        // leaves here therefore marks its frame "left unpainted by launch `epoch`" first: the frame behind reads that mark once all tiles of
        // this frame have finished, and stays unpainted with it, whichever of this frame's tiles got as far as the verdict.)
        if (__hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (threadIdx.x == 0) {
                __hip_atomic_store(&info->pad[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                arrive();
                finish();
            }
            return;
        }
    }
    if (MODE == 2) {
        // (uniform) the scout handed this frame — or an earlier one still in flight — to the host path: nothing may be
        // written from here on (later frames reuse, as destination, buffers the re-run still needs to read), until the
        // host has re-run them and cleared the word
        const bool bad = (info->flags & bad_mask) != 0u;
        if (bad && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(poison, 1u);
        if (bad || __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    if (BITS == 8) s_pal[tid] = (uint32_t)palette[tid];
    const uint32_t k = r.k;                                    // which tile of its frame
    const uint32_t tile_byte0 = PREFIX ? r.byte0 : t * TILE_BYTES;   // (frames start on tile boundaries: r.byte0 == t * TILE_BYTES)
    const uint32_t data_end = r.data_end;                      // 16-bit: whole code units only; 8-bit: every byte

    // ---- 0. look-back loads go out first: the tables of the nearest earlier tiles of the frame (up to LOOKBACK_BATCH
    //         of them, 3 words per lane) travel while this tile builds its own -----------------------------------
    const uint32_t nlook = k < (uint32_t)LOOKBACK_BATCH ? k : (uint32_t)LOOKBACK_BATCH;
    const uint32_t look0 = k - nlook;                          // first tile (within the frame) of that batch
    unsigned long long lv[LBW];
#pragma unroll
    for (int q = 0; q < LBW; ++q) lv[q] = 0ull;
    if (PREFIX) {
        if (tid == 0 && k) lv[0] = __hip_atomic_load(agg + (size_t)(t - 1u) * 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const unsigned long long* look_src = agg + (size_t)(r.first_tile + look0) * 9u;
#pragma unroll
        for (int q = 0; q < LBW; ++q)
            if ((uint32_t)(tid + q * PWG) < nlook * 9u)
                lv[q] = __hip_atomic_load(look_src + tid + q * PWG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- 1. tile bytes -> LDS (zero past the end of the frame's data), lane masks and tables -------------------
    // All of a lane's loads go out before the first is waited for: one after the other they were five memory round trips in
    // a row, ~3 us each under a saturated write stream — the longest phase of a tile.
    {
        constexpr int NLOAD = (BYTES_W * 4 + PWG * 16 - 1) / (PWG * 16);
        uint4 v[NLOAD];
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const uint32_t o = tid * 16u + (uint32_t)q * (PWG * 16u), at = tile_byte0 + o;
            v[q] = make_uint4(0, 0, 0, 0);
            // (MODE 3 reads the caller's own memory: the frame's last, partial 16 bytes are fetched byte by byte below)
            if (o < (uint32_t)BYTES_W * 4u && at < data_end && !(MODE == 3 && data_end - at < 16u))
                v[q] = *reinterpret_cast<const uint4*>(stream + at);   // buffers are padded
        }
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const uint32_t o = tid * 16u + (uint32_t)q * (PWG * 16u), at = tile_byte0 + o;
            if (o >= (uint32_t)BYTES_W * 4u) continue;
            if (at < data_end) {
                const uint32_t nvalid = data_end - at;
                if (nvalid < 16u) {
                    if (MODE == 3) {
                        uint32_t wds[4] = {0, 0, 0, 0};
                        for (uint32_t i = 0; i < nvalid; ++i) wds[i >> 2] |= (uint32_t)stream[at + i] << (8u * (i & 3u));
                        v[q] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                    }
                    auto keep_bytes = [&](uint32_t word, uint32_t first) -> uint32_t {
                        if (nvalid >= first + 4u) return word;
                        if (nvalid <= first) return 0u;
                        return word & ((1u << (8u * (nvalid - first))) - 1u);
                    };
                    v[q].x = keep_bytes(v[q].x, 0); v[q].y = keep_bytes(v[q].y, 4); v[q].z = keep_bytes(v[q].z, 8); v[q].w = keep_bytes(v[q].w, 12);
                }
            }
            *reinterpret_cast<uint4*>(lds_bytes + o) = v[q];
            if (MODE == 3 && keep && o < TILE_BYTES && at < data_end) *reinterpret_cast<uint4*>(keep + at) = v[q];   // the frame's bytes stay in HBM
        }
    }
    __syncthreads();
    JSP_CLOCK(0);   // bytes in LDS
    // The lane's slots as bit masks (msv1_lanes.h), kept in registers from here to the replay: the table pass and the replay
    // both read them, nothing is classified twice.
    const uint32_t p0 = tile_byte0 + tid * (LSLOTS * 2);
    lanes::Masks masks;
    {
        uint32_t w[LS / 2 + 1], tab[9];
        const uint32_t* mine_w = arena + tid * (LSLOTS * 2 / 4);
#pragma unroll
        for (int i = 0; i < LS / 2 + 1; ++i) w[i] = mine_w[i];
        masks = lanes::build_masks<BITS, LS>(w, r.frame_end > p0 ? (r.frame_end - p0) >> 1 : 0u);
        lanes::lane_table<BITS, LS>(mine_w, masks, wave_or(masks.Z), tab);   // (special slots read their word from LDS: `w` is dead from here)
#pragma unroll
        for (int e = 0; e < 9; ++e) tabs[tid * 9 + e] = tab[e];
    }
    // ---- 2. the tile's table by chaining, not by a tree: within a wave 7 groups of lane tables (10 + 6 x 9), each walked
    //         by 9 lanes (one per entry slot) table after table; then the 7 group tables, then the 4 wave tables.  Every
    //         walk leaves its running value in place of the table it has just passed, so afterwards row i holds "where the
    //         chain stands after table i, per entry slot of the walk" — what the down-sweep reads, once, per lane.
    //         (A wave's LDS accesses execute in order: no workgroup barrier inside a wave's walks.) -------------------
    wave_sync();
    const int grp = lane < 10 ? 0 : (lane - 1) / 9;            // group of this lane's TABLE (down-sweep)
    {
        const int g1 = lane / 9, e1 = lane - g1 * 9;           // the walk this lane takes part in (lane 63: none)
        if (lane < 63) {
            const int start = g1 == 0 ? 0 : 1 + 9 * g1, cnt = g1 == 0 ? 10 : 9;
            uint32_t* row = tabs + (wave * 64 + start) * 9;
            uint32_t v = (uint32_t)e1;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                if (q < cnt) {
                    v = lanes::sat_add(row[v & 15u], v & ~15u);
                    row[e1] = v;                               // (all 9 lanes have read the row: same instruction)
                    row += 9;
                }
            }
            s_grp[wave][g1][e1] = v;
        }
        wave_sync();
        if (lane < 9) {
            uint32_t v = (uint32_t)lane;
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                v = lanes::sat_add(s_grp[wave][g][v & 15u], v & ~15u);
                s_grp[wave][g][lane] = v;
            }
            s_wav[wave][lane] = v;
        }
    }
    __syncthreads();                                           // the four wave tables are in place
    JSP_CLOCK(1);   // lane tables + walks
    if (tid < 9) {
        uint32_t v = (uint32_t)tid;
#pragma unroll
        for (int u = 0; u < PWG / 64; ++u) {
            v = lanes::sat_add(s_wav[u][v & 15u], v & ~15u);
            s_wav[u][tid] = v;
        }
        s_root[tid] = v;
        // the published form keeps the block count within 28 bits and the exit slot intact
        const uint32_t pv = (v >> 4) >= BSAT ? (BSAT << 4) : v;
        // ---- publish the tile's table (the last tile of a frame has no reader): a word is its own flag --------
        if (!PREFIX && k + 1u < r.ntiles)
            __hip_atomic_store(agg + (size_t)t * 9u + tid, ((unsigned long long)epoch << 32) | pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- 3. look-back --------------------------------------------------------------------------------------------
    if (PREFIX) {
        // the predecessor's word {launch epoch, where the chain stands after it}; this tile's own goes out as soon as it is known
        if (tid == 0) {
            uint32_t ent = 0;
            bool ok = true;
            if (k) {
                for (int spin = 0; (uint32_t)(lv[0] >> 32) != epoch; ++spin) {
                    if (spin > LOOKBACK_SPIN_LIMIT) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(8);
                    lv[0] = __hip_atomic_load(agg + (size_t)(t - 1u) * 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ent = (uint32_t)lv[0];
            }
            if (ok) {
                const uint32_t after = lanes::sat_add(s_root[ent & 15u], ent & ~15u);
                if (k + 1u < r.ntiles)
                    __hip_atomic_store(agg + (size_t)t * 9u, ((unsigned long long)epoch << 32) | ((after >> 4) >= BSAT ? (BSAT << 4) : after),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                atomicOr(fault, 1u);                           // reported by jsp_staged_results / the call as an error
            }
            s_entry = ok ? ent : 0xFFFFFFFFu;                  // (a published value never has all its bits set)
        }
        __syncthreads();
        if (s_entry == 0xFFFFFFFFu) return;
    } else {
        uint32_t e = 0, blocks = 0;                            // thread 0 carries the chain
        // frames with more than LOOKBACK_BATCH earlier tiles: the far ones first, batch by batch (frames of more than LOOKBACK_BATCH tiles)
        for (uint32_t j0 = 0; j0 < k; j0 += LOOKBACK_BATCH) {
            const bool last = j0 + LOOKBACK_BATCH >= k;        // the batch whose loads are already in flight
            const uint32_t b0 = last ? look0 : j0, nj = last ? nlook : (uint32_t)LOOKBACK_BATCH;
            const unsigned long long* src = agg + (size_t)(r.first_tile + b0) * 9u;
            // (the loads of a round all go out before any is looked at: checked one by one, each cost a round trip of its own)
            unsigned long long v[LBW];
            bool have = true;
#pragma unroll
            for (int q = 0; q < LBW; ++q) {
                const bool wanted = (uint32_t)(tid + q * PWG) < nj * 9u;
                v[q] = last ? lv[q] : (wanted ? __hip_atomic_load(src + tid + q * PWG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull);
            }
#pragma unroll
            for (int q = 0; q < LBW; ++q) {
                const bool wanted = (uint32_t)(tid + q * PWG) < nj * 9u;
                have &= !wanted || (uint32_t)(v[q] >> 32) == epoch;
            }
            for (int spin = 0; !__syncthreads_and(have); ++spin) {
                if (spin > LOOKBACK_SPIN_LIMIT) {              // uniform: every lane counts the same rounds
                    // Giving up says something about the GPU's timing, nothing about the stream: the frame — and the frames in
                    // flight behind it — go to the host path, exactly as when the verdict barrier times out (STUCK): the flag
                    // is in the report before this tile arrives, so no tile of the frame gets past the verdict.
                    if (tid == 0) {
                        atomicOr(fault, 1u);
                        if (MODE >= 1 && MODE <= 3) atomicOr(&info->flags, MSV1_ASYNC_STUCK);
                        if (MODE == 2 || MODE == 3) atomicOr(poison, 1u);
                        if (MODE == 3) { arrive(); finish(); }
                    }
                    return;
                }
                __builtin_amdgcn_s_sleep(8);
                have = true;
                bool stale[LBW];
#pragma unroll
                for (int q = 0; q < LBW; ++q) stale[q] = (uint32_t)(tid + q * PWG) < nj * 9u && (uint32_t)(v[q] >> 32) != epoch;
#pragma unroll
                for (int q = 0; q < LBW; ++q)
                    if (stale[q]) v[q] = __hip_atomic_load(src + tid + q * PWG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int q = 0; q < LBW; ++q) {
                    const bool wanted = (uint32_t)(tid + q * PWG) < nj * 9u;
                    have &= !wanted || (uint32_t)(v[q] >> 32) == epoch;
                }
            }
#pragma unroll
            for (int q = 0; q < LBW; ++q)
                if ((uint32_t)(tid + q * PWG) < nj * 9u) enter[tid + q * PWG] = (uint32_t)v[q];
            __syncthreads();
            // chain tables [skip, nj) of the batch: LOOKBACK_SEGS segments walked side by side for all 9 entry slots
            // (lane = segment x entry), then thread 0 hops over the segment results
            const uint32_t skip = last ? j0 - b0 : 0u;         // already chained by the batch before
            const uint32_t todo = nj - skip, per = (todo + LOOKBACK_SEGS - 1) / LOOKBACK_SEGS;
            if (tid < LOOKBACK_SEGS * 9) {
                const uint32_t g = tid / 9u, e0 = tid - g * 9u;
                uint32_t se = e0, sb = 0;
                const uint32_t lo = skip + g * per, hi = lo + per < nj ? lo + per : nj;
                for (uint32_t j = lo; j < hi; ++j) {
                    const uint32_t tv = enter[j * 9u + se];
                    se = tv & 15u;
                    sb = sb + (tv >> 4) > BSAT ? BSAT : sb + (tv >> 4);
                }
                s_seg[g][e0] = pack(se, sb);
            }
            __syncthreads();
            if (tid == 0) {
                for (uint32_t g = 0; g < (uint32_t)LOOKBACK_SEGS; ++g) {
                    const uint32_t tv = s_seg[g][e];
                    e = tv & 15u;
                    blocks = blocks + (tv >> 4) > BSAT ? BSAT : blocks + (tv >> 4);
                }
            }
            __syncthreads();
        }
        if (tid == 0) s_entry = pack(e, blocks);
        __syncthreads();
    }
    // ---- 4. down-sweep: where the chain enters this lane's slots and with which block — the tile's entry taken through
    //         the walks' running values: waves before this one, groups before this lane's, tables before this lane's ----
    const uint32_t entry = s_entry;
    JSP_CLOCK(2);   // publish + look-back
    const uint32_t tb0 = entry >> 4;
    uint32_t mine;
    {
        const uint32_t wv = wave == 0 ? entry : lanes::sat_add(s_wav[wave - 1][entry & 15u], entry & ~15u);
        const uint32_t gv = grp == 0 ? wv : lanes::sat_add(s_grp[wave][grp - 1][wv & 15u], wv & ~15u);
        const int start = grp == 0 ? 0 : 1 + 9 * grp;
        mine = lane == start ? gv : lanes::sat_add(tabs[(tid - 1) * 9 + (gv & 15u)], gv & ~15u);
    }
    const uint32_t whole = lanes::sat_add(s_root[entry & 15u], entry & ~15u) >> 4;   // blocks covered once this tile is done
    const uint32_t span_end = whole < nblocks ? whole : nblocks;
    if (INFO && tid == 0 && k + 1u == r.ntiles && whole < nblocks) atomicOr(&info->flags, MSV1_ASYNC_SHORT);   // the stream ends early
    // the slots of this lane the chain visits
    const uint32_t visits = lanes::visited<BITS, LS>(masks, mine & 15u);
    const uint32_t blk0 = mine >> 4;
    __syncthreads();                                           // the tables are dead: the staging window takes their place
    JSP_CLOCK(3);   // down-sweep

    uint32_t seen = 0;                                         // INFO: what this lane's codes on the chain were
    bool arrived = false;                                      // MODE 3 (uniform): this workgroup has been through the verdict
    const uint32_t* __restrict__ prev = reinterpret_cast<const uint32_t*>(r.prev);
    uint32_t* __restrict__ dstf = reinterpret_cast<uint32_t*>(r.dst);
    // Staging windows start on multiples of 256 blocks, whatever block the tile starts with: lane i of the workgroup then always
    // has block (multiple of 256) + i, so a wave's row store is 1 KiB starting on a 512-byte boundary of the frame and writes whole
    // memory lines.  (Windows that started at the tile's first block made every row store of every wave begin and end inside a
    // line that the neighbouring wave completes: two partial-line writes per store, see profiles/archive/r03_fused_notes.txt.)
    for (uint32_t w0 = JSP_FUSED_ALIGN ? tb0 & ~255u : tb0; w0 < span_end; w0 += FSTAGE) {   // more than one window only behind long skip runs / in tiles of short codes
        const uint32_t wn = span_end - w0 < (uint32_t)FSTAGE ? span_end - w0 : (uint32_t)FSTAGE;   // window entries [0, wn) ...
        const uint32_t wlo = tb0 > w0 ? tb0 - w0 : 0u;                 // ... of which [wlo, wn) are this tile's blocks (wlo < 256)
        const bool first_window = w0 <= tb0;
        if (MODE != 1) {
            for (uint32_t i = tid; i < (wn + 1u) / 2u; i += PWG)
                reinterpret_cast<uint32_t*>(stage)[i] = (2u * i < wlo ? F_NONE : F_SKIP) | ((2u * i + 1u < wlo ? F_NONE : F_SKIP) << 16);
            __syncthreads();
        }
        {
            // replay: every visited slot in order; a plain code leaves the offset of its code and is one block, a special slot
            // (rare) is a skip code — its count comes from the word —, a slot past the data or the 8-bit end marker.
            // (Every window walks all of the lane's codes again.  Going on where the window before stopped — round 5, bit-exact — changes nothing
            // for tiles of two or three windows: inter frames 1.023 / 1.027 against 1.016 / 1.013 ms, all-solid 0.660 against 0.656,
            // profiles/r05_fused_replay_resume_ab.txt, although thread 0's clocks show 15 k of a table-writing tile's 34 k cycles here
            // (r05_fused_clocks_inter70.txt): it waits at the window's barriers, it does not walk.)
            uint32_t blk = blk0;
            for (uint32_t left = visits; left; left &= left - 1u) {
                const uint32_t sl = (uint32_t)lanes::first_bit(left), bit = left & (0u - left);
                if (masks.Z & bit) {
                    if (masks.valid & bit) {
                        if (INFO && first_window && blk < nblocks) seen |= (masks.K & bit) ? MSV1_ASYNC_SKIPCODE : MSV1_ASYNC_END;
                        if (masks.K & bit) {
                            const uint32_t n = (uint32_t)reinterpret_cast<const uint16_t*>(lds_bytes)[tid * LSLOTS + sl] & 0x3FFu;
                            blk += n ? n : lanes::REST_OF_FRAME;
                        }
                    }
                } else {
                    if (INFO && first_window && blk >= s1_first_block && blk < nblocks) seen |= MSV1_ASYNC_S1;
                    if (MODE != 1 && blk - w0 < wn) stage[blk - w0] = (uint16_t)(tid * (LSLOTS * 2) + 2u * sl);
                    ++blk;
                }
            }
        }
        if (MODE == 1) break;                                  // the scout only needed the replay of the first window
        if (MODE == 3 && first_window) {
            // ---- the frame's verdict: every tile's findings are in before any tile writes a pixel (all tiles of a
            //      frame this small are resident together: the launcher only takes frames of a few hundred tiles) ----
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) seen |= (uint32_t)__shfl_xor((int)seen, o);
            if (lane == 0 && seen) atomicOr(&info->flags, seen);
            __syncthreads();
            if (tid == 0) {
                // ONE verdict per launch, whoever pronounces it: a tile that sees every report in proposes GO or VETO from the
                // (then complete) flags, a tile that gives up waiting proposes VETO; the first compare-and-swap wins and every
                // tile obeys the word it then reads — no tile can write pixels of a frame another tile vetoes a moment later.
                arrive();
                uint32_t verdict = 0;
                for (int spin = 0;; ++spin) {
                    verdict = __hip_atomic_load(&info->verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (verdict) break;
                    uint32_t propose = 0;
                    if (!(bad_mask & MSV1_LAB_DEAF) && (int32_t)(__hip_atomic_load(&info->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0)
                        propose = (__hip_atomic_load(&info->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (bad_mask | MSV1_ASYNC_STUCK)) ? MSV1_VERDICT_VETO : MSV1_VERDICT_GO;
                    else if (spin >= VERDICT_SPIN_LIMIT) {
                        atomicOr(&info->flags, MSV1_ASYNC_STUCK);      // the host re-runs the frame synchronously
                        propose = MSV1_VERDICT_VETO;
                    }
                    if (propose) atomicCAS(&info->verdict, 0u, propose);
                    else __builtin_amdgcn_s_sleep(8);
                }
                s_entry = verdict == MSV1_VERDICT_VETO ? 1u : 0u;
                if (rider) {
                    // The frame in front, in this very launch: every tile of it must be through — painted, or left unpainted — before a
                    // pixel of this one is written (it may copy from that frame, and be compared with it) AND before this frame may set the
                    // veto word: all frames of a launch reach their verdicts side by side, and a word set now could still meet a tile of a
                    // frame in front at its first instruction.  If the frame in front was left unpainted (vetoed, or behind a vetoed one
                    // itself: its tiles say so in its report, tagged with its launch epoch, before they finish) this frame goes to the host
                    // with it, unpainted too.
                    for (int spin = 0;; ++spin) {
                        if ((int32_t)(__hip_atomic_load(&info_before->finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want_before) >= 0) break;
                        if (spin >= VERDICT_SPIN_LIMIT) {
                            atomicOr(&info->flags, MSV1_ASYNC_STUCK);
                            s_entry = 1u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    __threadfence();                           // (what the frame in front wrote is what this one reads)
                    if (!s_entry && __hip_atomic_load(&info_before->pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch_before) s_entry = 1u;
                }
                if (s_entry) {
                    atomicOr(poison, 1u);                      // ... and every later frame in flight with it
                    __hip_atomic_store(&info->pad[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // "launch `epoch` leaves this frame unpainted"
                }
            }
            arrived = true;
            __syncthreads();
            if (s_entry) break;                                // vetoed: `dst` stays exactly as the caller left it
        }
        __syncthreads();
        JSP_CLOCK(4);   // replay into the staging window
        if (MODE == 5) {
            // ---- 5''. the compact descriptor form (msv1.h): two bytes per block and a base per group of 256 blocks.  The window starts on a multiple of
            //      256 blocks, so its groups are whole and its entries pair up into aligned words; an entry another tile owns (only in front of the
            //      tile's first block) is left alone, and so is a group's base when this tile has no code in the group — the tile that has writes it,
            //      and when both have, either value will do ----
            static_assert(JSP_FUSED_ALIGN, "compact tables need staging windows that start on multiples of 256 blocks");
            uint16_t* __restrict__ tab16 = reinterpret_cast<uint16_t*>(dstf) + w0;
            uint32_t* __restrict__ bases = reinterpret_cast<uint32_t*>(const_cast<int32_t*>(r.prev)) + (w0 >> 8);
            auto entry = [&](uint32_t o) { return o == F_SKIP ? MSV1_TAB16_SKIP : ((tile_byte0 + o) & 0x7FFFu); };
            for (uint32_t i = 2u * tid; i < wn; i += 2u * PWG) {
                const uint32_t a = stage[i], b = i + 1u < wn ? (uint32_t)stage[i + 1u] : F_NONE;
                if (a != F_NONE && b != F_NONE) *reinterpret_cast<uint32_t*>(tab16 + i) = entry(a) | (entry(b) << 16);
                else {
                    if (a != F_NONE) tab16[i] = (uint16_t)entry(a);
                    if (b != F_NONE) tab16[i + 1u] = (uint16_t)entry(b);
                }
            }
            for (uint32_t g = wave; g * 256u < wn; g += PWG / 64) {
                uint32_t m = 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t idx = g * 256u + k * 64u + lane;
                    const uint32_t o = idx < wn ? (uint32_t)stage[idx] : F_NONE;
                    m = o < F_NONE ? min(m, o) : m;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
                if (lane == 0 && m != 0xFFFFFFFFu) bases[g] = tile_byte0 + m;
            }
        } else if (MODE == 4) {
            // ---- 5'. the descriptor form: the window goes out as the frame's block table (byte offset of each block's code
            //      in the stream buffer, or "copy from the previous frame"), which msv1_blocks_temporal_kernel /
            //      msv1_blocks_kernel read — what msv1_parse_tiles + _chain + _emit build in three launches ----
            uint32_t* __restrict__ table = dstf;               // (the record's `dst` is the frame's table)
            for (uint32_t i = tid; i < wn; i += PWG) {
                const uint32_t o = stage[i];
                if (o != F_NONE) table[w0 + i] = o == F_SKIP ? MSV1_DESC_SKIP : tile_byte0 + o;
            }
        } else {
        // ---- 5. reconstruction: lane = block, raster order; block coordinates advance by PWG blocks per round ----
        // 5a. skipped blocks are copies from the previous frame: U blocks per lane at a time, all their row loads out before
        //     the first store (one at a time, every block waits a memory round trip of its own: an inter frame of a single
        //     stream has a dozen workgroups on the whole GPU and nothing else to hide it behind).
        if (USES_PREV) {
            constexpr int U = 4;
            uint32_t by = (w0 + tid) / (uint32_t)nbx, bx = (w0 + tid) - by * (uint32_t)nbx;
            for (uint32_t i = tid; i < wn; i += PWG * (uint32_t)U) {
                uint32_t di[U];
                bool sk[U], any = false;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t idx = i + (uint32_t)u * PWG;
                    sk[u] = idx < wn && stage[idx] == F_SKIP;
                    any |= sk[u];
                    di[u] = (by * (uint32_t)X + bx) * 4u;
                    bx += PWG;
                    while (bx >= (uint32_t)nbx) { bx -= (uint32_t)nbx; ++by; }
                }
                if (!any) continue;
                fu32x4 q[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (sk[u]) {
#pragma unroll
                        for (int y = 0; y < 4; ++y) q[u][y] = *(fcgu32x4*)(prev + di[u] + (size_t)y * X);
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (sk[u]) {
#pragma unroll
                        for (int y = 0; y < 4; ++y) __builtin_nontemporal_store(q[u][y], (fgu32x4*)(dstf + di[u] + (size_t)y * X));
                    }
            }
        }
        // 5b. coded blocks
        uint32_t by = (w0 + tid) / (uint32_t)nbx, bx = (w0 + tid) - by * (uint32_t)nbx;
        bool compare = USES_PREV && r.cmp_row_lo != 0xFFFFFFFFu;       // (wave-uniform) stage-2 significance still open
        for (uint32_t i = tid; i < wn; i += PWG) {
            const uint32_t o = stage[i];
            const uint32_t di = (by * (uint32_t)X + bx) * 4u;          // pixel index: a frame has fewer than 2^28 pixels
            uint32_t* __restrict__ dst = dstf + di;
            const uint32_t by_now = by;
            bx += PWG;
            while (bx >= (uint32_t)nbx) { bx -= (uint32_t)nbx; ++by; }
            const bool coded = o < F_NONE;
            fu32x4 q[4];
            if (USES_PREV && compare && coded) {                       // the previous frame's rows travel while the block is decoded
                const uint32_t* __restrict__ pv = prev + di;
#pragma unroll
                for (int y = 0; y < 4; ++y) q[y] = *(fcgu32x4*)(pv + (size_t)y * X);
            }
            bool diff = false;
            if (coded) {
                uint32_t px[16];
                decode_block<BITS>(lds_bytes + o, data_end - (tile_byte0 + o), s_pal, px);
                // Store throttle.  A CU's loads and stores share one in-order queue: every row store a wave leaves in flight is
                // something the other workgroups' loads (tile bytes, look-back word) wait behind.  Keeping at most a few rows per
                // wave in flight keeps the memory pipe full without that queue growing: profiles/archive/r03_fused_notes.txt.
                if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JSP_FUSED_VMCNT) : "memory");
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    __builtin_nontemporal_store(fu32x4{px[y * 4], px[y * 4 + 1], px[y * 4 + 2], px[y * 4 + 3]}, (fgu32x4*)(dst + (size_t)y * X));
                if (USES_PREV && compare) {                            // stage-2 significance, MSVideo1.hx:195-204
                    // (all four rows are looked at, rows below the first compared one masked out: no load is left pending
                    // behind a condition, so the loop's next round need not wait for this round's stores)
                    uint32_t d4 = 0;
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        const uint32_t dy = (q[y].x ^ px[y * 4]) | (q[y].y ^ px[y * 4 + 1]) | (q[y].z ^ px[y * 4 + 2]) | (q[y].w ^ px[y * 4 + 3]);
                        d4 |= by_now * 4u + y >= r.cmp_row_lo ? dy : 0u;
                    }
                    diff = d4 != 0u;
                    if (diff && __hip_atomic_load(r.signif, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(r.signif, 1u);
                }
            }
            if (USES_PREV && compare && __any(diff)) compare = false;  // one differing pixel settles it: no more rows of the previous frame
        }
        }
        if (w0 + FSTAGE < span_end) __syncthreads();           // the window is refilled by the next round
    }
    JSP_CLOCK(5);   // decode + store issue
    if (MODE == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) seen |= (uint32_t)__shfl_xor((int)seen, o);
        if (lane == 0 && seen) atomicOr(&info->flags, seen);
    }
    if (MODE == 3) {
        __syncthreads();                                       // every lane's stores and significance reports are out
        if (tid == 0) {
            if (!arrived) arrive();                            // a tile behind the last block: nothing to report, nothing to write
            finish();
        }
    }
}

}  // namespace

uint32_t msv1_parse_tile_bytes() { return TSLOTS * 2; }
uint32_t msv1_small_tile_bytes() { return PWG * 16 * 2; }   // `small_tiles` of msv1_launch_fused: 16 slots per lane

void msv1_launch_parse(const Msv1Geometry& geo, const uint8_t* d_stream, const Msv1ParseFrame* d_frames, int nframes,
                       const uint32_t* d_tile_frame, int ntiles, int max_tiles_per_frame, uint32_t* d_tile_tab,
                       uint32_t* d_tile_entry, uint32_t* d_tile_block0, uint32_t* d_desc, Msv1FrameInfo* d_info,
                       int insignificant_blocks, hipStream_t stream) {
    if (nframes <= 0 || ntiles <= 0) return;
    (void)hipMemsetAsync(d_info, 0, sizeof(Msv1FrameInfo) * nframes, stream);
    const uint32_t s1_first = (uint32_t)(insignificant_blocks < 0 ? 0 : insignificant_blocks) * (uint32_t)geo.nbx;
    if (geo.bits == 16) {
        hipLaunchKernelGGL(msv1_parse_tiles<16>, dim3(ntiles), dim3(PWG), 0, stream, d_stream, d_frames, d_tile_frame, d_tile_tab);
        hipLaunchKernelGGL(msv1_parse_chain, dim3(nframes), dim3(64), sizeof(uint32_t) * 9 * max_tiles_per_frame, stream,
                           d_frames, d_tile_tab, d_tile_entry, d_tile_block0, d_info);
        hipLaunchKernelGGL(msv1_parse_emit<16>, dim3(ntiles), dim3(PWG), 0, stream, d_stream, d_frames, d_tile_frame, d_tile_entry,
                           d_tile_block0, d_desc, d_info, (uint32_t)geo.nblocks, s1_first);
    } else {
        hipLaunchKernelGGL(msv1_parse_tiles<8>, dim3(ntiles), dim3(PWG), 0, stream, d_stream, d_frames, d_tile_frame, d_tile_tab);
        hipLaunchKernelGGL(msv1_parse_chain, dim3(nframes), dim3(64), sizeof(uint32_t) * 9 * max_tiles_per_frame, stream,
                           d_frames, d_tile_tab, d_tile_entry, d_tile_block0, d_info);
        hipLaunchKernelGGL(msv1_parse_emit<8>, dim3(ntiles), dim3(PWG), 0, stream, d_stream, d_frames, d_tile_frame, d_tile_entry,
                           d_tile_block0, d_desc, d_info, (uint32_t)geo.nblocks, s1_first);
    }
}


void msv1_launch_fused(const Msv1Geometry& geo, const uint8_t* d_stream, const Msv1TileRec* d_recs, const int32_t* d_palette,
                       unsigned long long* d_agg, uint32_t epoch, uint32_t tile0, int ntiles, uint32_t* d_fault,
                       hipStream_t stream, Msv1AsyncInfo* d_info, int insignificant_blocks, int mode, uint32_t bad_mask,
                       uint32_t* d_poison, const Msv1TileRec* one_rec, Msv1AsyncInfo* h_info, uint32_t want, uint8_t* d_keep,
                       bool small_tiles, const Msv1Riders* riders) {
    if (ntiles <= 0) return;
    Msv1Riders two{};
    if (riders && mode == 3) {
        two = *riders;
        for (uint32_t q = 0; q < two.count; ++q) { two.f[q].first_wg = (uint32_t)ntiles; ntiles += (int)two.f[q].rec.ntiles; }
    }
    const uint32_t s1_first = (uint32_t)(insignificant_blocks < 0 ? 0 : insignificant_blocks) * (uint32_t)geo.nbx;
    const Msv1TileRec rec = one_rec ? *one_rec : Msv1TileRec{};
    if (mode == 0) small_tiles = false;                        // the batch form lays frames out on 16 KiB boundaries (the table-writing form, mode 4, may take them in 8 KiB tiles: its records say where each begins)
#define JSP_FUSED(BITS, MODE, LS)                                                                                            \
    hipLaunchKernelGGL((msv1_fused_kernel<BITS, MODE, LS>), dim3(ntiles), dim3(PWG), 0, stream, d_stream, d_recs, d_palette, \
                       d_agg, epoch, tile0, d_fault, (uint32_t)geo.nblocks, geo.nbx, geo.X, d_info, s1_first, bad_mask, d_poison, rec, h_info, want, d_keep, two)
#define JSP_FUSED_LS(BITS, MODE) do { if (small_tiles) JSP_FUSED(BITS, MODE, 16); else JSP_FUSED(BITS, MODE, JSP_BATCH_LS); } while (0)
#define JSP_FUSED_MODES(BITS)                                                                                                \
    switch (mode) { case 1: JSP_FUSED_LS(BITS, 1); break; case 2: JSP_FUSED_LS(BITS, 2); break; case 3: JSP_FUSED_LS(BITS, 3); break; \
                    case 4: JSP_FUSED_LS(BITS, 4); break; case 5: JSP_FUSED_LS(BITS, 5); break; default: JSP_FUSED(BITS, 0, JSP_BATCH_LS); }
    if (geo.bits == 16) { JSP_FUSED_MODES(16) } else { JSP_FUSED_MODES(8) }
#undef JSP_FUSED_MODES
#undef JSP_FUSED_LS
#undef JSP_FUSED
}

}  // namespace jsp
