// ScreenPressor reconstruction kernels for gfx950 (MI355X).  Integer work, HBM-bound, no MFMA.
//
// Key frames.  The run table resolves every pixel to either a constant or "the pixel one row up (same column or one
// to the left) plus a per-run addend" (ScreenPressor.hx:242-273; the gradient predictor telescopes inside a run), so
// the only dependency left is on the row above; the host stage cuts the frame into bands of rows and hands each band
// the row above it ("seed"), so bands are independent.  Two kernels:
//   sp_iframe_tile_kernel        X % 4 == 0, 16-byte aligned frame buffers (every real frame): one WAVE per tile
//                                (band x 256-column span), row above in registers, run words scattered through a
//                                wave-private LDS row, no barrier at all;
//   sp_iframe_rows_search_kernel any width / alignment: one workgroup per band, per-lane run search, row above in LDS.
// Inter frames:
//   sp_pframe_kernel             one frame: a workgroup covers 4 horizontally adjacent 16x16 blocks (256
//                                contiguous bytes per row), lane = 16-byte chunk of a row: unchanged / base
//                                copy / motion from the previous frame in HBM, literal payload for data
//                                rectangles (ScreenPressor.hx:361-475);
//   sp_pframe_group_kernel       consecutive inter frames in one launch, pixels carried in registers: 8 blocks per workgroup,
//                                four worker waves and a loader wave that brings 16 frames' block records and literals
//                                into LDS at a time (X % 4 == 0, aligned buffers);
//   sp_pframe_group1_kernel      the same for any width / alignment: the workgroup stages its own chunks.
#include <cstddef>
#include <cstdlib>
#include <mutex>

#include "sp.h"

namespace jsp::sp {
namespace {

constexpr int IWG = 512;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for this wave's
// outstanding global stores (vmcnt(0)); the row loop would then pay a full HBM/L2 write latency per
// image row although no other wave ever reads those stores back.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Pointers that reach a kernel inside a struct read from memory have no known address space, so
// plain accesses through them become FLAT instructions — which count on lgkmcnt as well as vmcnt and
// would tie every LDS wait of the row loop to the latency of the frame stores.  Say "global".
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) u32x4 gu32x4;
__device__ __forceinline__ void store4_global(uint32_t* p, const uint4& v) {
    *(gu32x4*)p = u32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void store1_global(uint32_t* p, uint32_t v) { *(gu32*)p = v; }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) uint32_t cgu32;
typedef const __attribute__((address_space(1))) u32x2 cgu32x2;
__device__ __forceinline__ uint32_t load1_global(const uint32_t* p) { return *(cgu32*)p; }
__device__ __forceinline__ uint2 load2_global(const uint2* p) { const u32x2 v = *(cgu32x2*)p; return make_uint2(v.x, v.y); }

__device__ __forceinline__ uint32_t add_bytes(uint32_t u, uint32_t d) {  // per byte, bytes 0..2
    return (((u & 0x00FF00FFu) + (d & 0x00FF00FFu)) & 0x00FF00FFu) | (((u & 0x0000FF00u) + (d & 0x0000FF00u)) & 0x0000FF00u);
}

__global__ __launch_bounds__(IWG) void sp_iframe_rows_search_kernel(const IFrameArgs* __restrict__ args, int X, int Y,
                                                                    int run_cap, int band_rows) {
    extern __shared__ __align__(16) uint32_t lds[];
    const IFrameArgs fa = args[blockIdx.x];
    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(fa.dst);
    const int tid = threadIdx.x;
    const int yb = (int)blockIdx.y * band_rows;       // this workgroup's rows: [yb, ye)
    if (yb >= Y) return;
    const int ye = yb + band_rows < Y ? yb + band_rows : Y;
    if (fa.flat) {  // flat key frame: one colour (ScreenPressor.hx:132-155)
        for (size_t i = (size_t)yb * X + tid; i < (size_t)ye * X; i += IWG) store1_global(dst + i, fa.colour);
        return;
    }
    // LDS plan: two row buffers, the whole row index, and a window of run records {start, word}
    // covering as many rows as fit (one global round trip per window instead of one per row)
    const int rowcap = (X + 4 + 3) & ~3;
    uint32_t* rowbuf0 = lds;
    uint32_t* rowbuf1 = lds + rowcap;
    uint32_t* rowidx_lds = lds + 2 * rowcap;          // ye - yb + 1 entries: rows yb .. ye
    uint32_t* lastpix = rowidx_lds + ((ye - yb + 1 + 3) & ~3);  // last pixel of each of the 4 most recent rows
    uint2* win = reinterpret_cast<uint2*>(lastpix + 4);   // run_cap records
    const uint32_t* rowidx = rowidx_lds - yb;         // indexed by absolute row
    for (int k = tid; k <= ye - yb; k += IWG) rowidx_lds[k] = load1_global(fa.row_run + yb + k);
    if (yb > 0) {  // the row above the band and the two wrap pixels, from the host stage's seed
        const uint32_t* sd = fa.seeds + (size_t)(blockIdx.y - 1) * ((size_t)X + 1);
        uint32_t* upbuf = (yb & 1) ? rowbuf0 : rowbuf1;
        for (int k = tid; k < X; k += IWG) upbuf[k] = load1_global(sd + 1 + k);
        if (tid == 0) {
            lastpix[yb & 3] = 0; lastpix[(yb + 1) & 3] = 0;
            lastpix[(yb + 2) & 3] = load1_global(sd);        // (yb-2) & 3
            lastpix[(yb + 3) & 3] = load1_global(sd + X);    // (yb-1) & 3
        }
    } else if (tid < 4) lastpix[tid] = 0;
    __syncthreads();
    const bool vec = (X & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    int hint = 0;  // this lane's run index relative to the row start, carried from row to row:
                   // neighbouring rows of screen content are cut into runs almost identically
    int y = yb;
    while (y < ye) {
        // window: rows y .. y_end-1 whose runs rowidx[y] .. rowidx[y_end] fit in run_cap records
        // (a single row always fits: run_cap >= X + 2)
        const uint32_t w0 = rowidx[y];
        int y_end = y + 1;
        while (y_end < ye && (int)(rowidx[y_end + 1] - w0) + 1 <= run_cap) ++y_end;
        const int wn = (int)(rowidx[y_end] - w0) + 1;
        const uint2* __restrict__ gruns = reinterpret_cast<const uint2*>(fa.runs) + w0;
        for (int k = tid; k < wn; k += IWG) win[k] = load2_global(gruns + k);
        __syncthreads();  // the run records arrive through vmcnt: full barrier once per window
        for (; y < y_end; ++y) {
            uint32_t* cur = (y & 1) ? rowbuf1 : rowbuf0;
            const uint32_t* up = (y & 1) ? rowbuf0 : rowbuf1;
            const uint2* rr = win + (int)(rowidx[y] - w0);          // first run of this row
            const int nr = (int)(rowidx[y + 1] - rowidx[y]) + 1;    // runs intersecting the row
            // pixel (X-1, y-2) for the x == 0 case of the above-left predictor (linear index i-X-1)
            const uint32_t wrap_left = y >= 2 ? lastpix[(y - 2) & 3] : 0u;
            const uint32_t row0 = (uint32_t)((size_t)y * X);
            for (int x0 = tid * 4; x0 < X; x0 += IWG * 4) {
                const uint32_t i0 = row0 + x0;
                // the row above, fetched before the run lookup so both LDS latencies overlap
                uint32_t u[5];  // u[0] = pixel x0-1 of the row above, u[1..4] = pixels x0..x0+3
                if (vec) {
                    const uint4 q = *reinterpret_cast<const uint4*>(up + x0);
                    u[1] = q.x; u[2] = q.y; u[3] = q.z; u[4] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[1 + j] = x0 + j < X ? up[x0 + j] : 0u;
                }
                u[0] = x0 > 0 ? up[x0 - 1] : wrap_left;
                // run holding pixel x0: try the hint, fall back to a binary search
                int k = hint < nr ? hint : nr - 1;
                uint2 e = rr[k];
                uint2 nx = k + 1 < nr ? rr[k + 1] : make_uint2(0xFFFFFFFFu, 0u);
                if (!(e.x <= i0 && i0 < nx.x)) {
                    int lo = 0, hi = nr - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (rr[mid].x <= i0) lo = mid; else hi = mid - 1;
                    }
                    k = lo;
                    e = rr[k];
                    nx = k + 1 < nr ? rr[k + 1] : make_uint2(0xFFFFFFFFu, 0u);
                }
                hint = k;
                uint32_t word = e.y;
                uint32_t px[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t i = i0 + j;
                    while (nx.x <= i) {   // next run starts at or before this pixel
                        ++k;
                        word = nx.y;
                        nx = k + 1 < nr ? rr[k + 1] : make_uint2(0xFFFFFFFFu, 0u);
                    }
                    const uint32_t kind = word >> 24, val = word & 0xFFFFFFu;
                    uint32_t v;
                    if (kind == RUN_CONST) v = val;
                    else if (y == 0) v = 0;                                  // above the buffer: undefined -> 0
                    else if (kind == RUN_ABOVE) v = add_bytes(u[1 + j], val);
                    else v = u[j];                                           // RUN_ABOVE_LEFT
                    px[j] = v;
                }
                if (vec) {
                    const uint4 q = make_uint4(px[0], px[1], px[2], px[3]);
                    *reinterpret_cast<uint4*>(cur + x0) = q;
                    store4_global(dst + row0 + x0, q);
                    if (x0 + 4 == X) lastpix[y & 3] = px[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (x0 + j < X) {
                            cur[x0 + j] = px[j];
                            store1_global(dst + row0 + x0 + j, px[j]);
                            if (x0 + j == X - 1) lastpix[y & 3] = px[j];
                        }
                }
            }
            lds_barrier();
        }
    }
}

// Tile path: one WAVE per tile = a band of rows x a span of 256 columns (4 pixels per lane).  The host stage hands every
// tile what crosses its borders — the row above the band (seeds) and, per row, the pixel left of the span's
// first pixel one row up (`left`) — orders the run records tile by tile and never lets one cross a span, so
// a wave needs nothing another wave produces: no workgroup barrier, no LDS shared between waves, the row
// loop is one wave's private instruction stream: row above in registers, run words scattered through an LDS row
// of the span, branch-free predictor.
//
// The kernel is bound by the instructions it issues (a wave64 VALU instruction holds its SIMD for four cycles; traffic is 1.003 x the bytes
// moved), so the row loop is written instruction by instruction (round 6; ~45 -> ~25 VALU on a row that repeats the layout of the row above,
// ~75 -> ~50 on one that does not, a third of the scalar branches):
//   * everything that is the same for all 64 lanes lives on the scalar unit: a window's index entries, left pixels and kind counts sit one per
//     lane in three registers and each row takes its own with v_readlane — unconditionally: the entries past the window's last row are copies
//     of its end, so "no next row" is "a next row with no records";
//   * a record arrives as 4 bytes {column inside the span, 24-bit value} (tile_record32, sp.h), its kind is its place among the row's records
//     (sorted by kind, two scalar counts per row), and one ds_write puts `value | kTileHead | kind nibble` at the column;
//   * a pixel is `start value + record value`, byte by byte, for ALL kinds: ONE v_perm_b32 picks the start value out of {pixel above-left,
//     pixel above} with a per-pixel byte selector (bytes of the one, of the other, or 0xFF for a constant — whose record carries colour + 1),
//     three SDWA byte adds put the record's bytes on top (v_add_u32_sdwa ... dst_unused:UNUSED_PRESERVE: byte 3 stays the selector's zero):
//     4 VALU per pixel where the mask-and-carry form took 7, and the selector is one register where the two lane masks were two;
//   * the row is computed IN PLACE, pixel 3 first (pixel j needs the old pixel j - 1): no copy of the finished row into the "row above";
//   * the row store takes a scalar base that advances by the pitch and a per-lane offset that never changes.
__device__ __forceinline__ uint32_t ffbh(uint32_t v) {   // leading zeros; 0xFFFFFFFF for 0
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// Launch shape (round 5): the waves of a band's neighbouring spans share a WORKGROUP — up to eight waves side by side in the row, nothing between them but
// the CU they run on: no barrier, no shared LDS, each wave its own slice of the workgroup's allocation.  A row's 7 680 bytes then leave one CU at about the
// same time; the store shape alone takes 6.6 TB/s that way against 6.3 with the waves dealt out one per workgroup (profiles/r05_front_lab_band_workgroups.txt).
// (1920 columns are 7.5 spans: the eighth wave of a 1080p workgroup runs with half its lanes masked, 6 % of the launch's issue slots.  Giving that wave the half spans
// of two bands would need two window states on the scalar unit — every scalar of the row loop twice — and is not done.)
constexpr int TILE_WAVES = 8;                         // at most; tall bands get fewer (tile_plan: the workgroup's LDS is waves x a tile's index)
__global__ __launch_bounds__(64 * TILE_WAVES, 8) void sp_iframe_tile_kernel(const IFrameArgs* __restrict__ args, int X, int Y,
                                                            int band_rows, int nspans, int win_cap, int span_groups, int lds_words_per_wave, int waves_per_group) {
    constexpr int PPL = 4, SPAN = 64 * PPL;
    extern __shared__ __align__(16) uint32_t lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint32_t* lds = lds_all + (size_t)wave * lds_words_per_wave;
    const IFrameArgs fa = args[blockIdx.x];
    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(fa.dst);
    const int band = (int)blockIdx.y / span_groups, span = ((int)blockIdx.y - band * span_groups) * waves_per_group + wave;
    if (span >= nspans) return;
    const int tile = band * nspans + span;
    const int yb = band * band_rows;
    if (yb >= Y) return;
    const int ye = yb + band_rows < Y ? yb + band_rows : Y;
    const int lane = (int)threadIdx.x & 63;
    const int xs = span * SPAN;                       // first column of the span
    const int x0 = xs + lane * PPL;
    const bool active = x0 < X;                       // X % 4 == 0: an active lane owns 4 pixels
    if (fa.flat) {
        if (active)
            for (int y = yb; y < ye; ++y) store4_global(dst + (size_t)y * X + x0, make_uint4(fa.colour, fa.colour, fa.colour, fa.colour));
        return;
    }
    uint32_t* head = lds;                             // SPAN words: 4 x a record's column is its byte offset
    uint32_t* idx = lds + SPAN;                       // band_rows + 1 offsets (relative to the frame's records)
    uint2* left = reinterpret_cast<uint2*>(idx + ((band_rows + 1 + 3) & ~3));   // band_rows pairs {left pixel, kind counts}
    uint32_t* win = reinterpret_cast<uint32_t*>(left + ((band_rows + 1) & ~1));   // win_cap + 2 records (8-byte aligned) + 64 words that may be read, never used
    const uint32_t* gidx = fa.tile_idx + (size_t)tile * (band_rows + 1);
    const uint2* gleft = reinterpret_cast<const uint2*>(fa.left) + (size_t)tile * band_rows;
    for (int k = lane; k <= band_rows; k += 64) idx[k] = load1_global(gidx + k);
    for (int k = lane; k < band_rows; k += 64) left[k] = load2_global(gleft + k);
    *reinterpret_cast<uint4*>(head + lane * PPL) = make_uint4(0, 0, 0, 0);
    u32x4 pv = {0, 0, 0, 0};                          // this lane's pixels of the row above; after the row's arithmetic, of the row (the row loop's asm names its registers)
    if (yb > 0 && active) {
        const uint32_t* sd = fa.seeds + (size_t)(band - 1) * ((size_t)X + 1) + 1 + x0;
        pv.x = load1_global(sd); pv.y = load1_global(sd + 1); pv.z = load1_global(sd + 2); pv.w = load1_global(sd + 3);
    }
    const unsigned long long active_lanes = __ballot(active);
    // everything fetched so far must have landed before the row loop: a vmcnt wait inside it would also wait for
    // the frame stores (loads and stores share the counter)
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), expcnt/lgkmcnt untouched
    __builtin_amdgcn_wave_barrier();

    const uint32_t* __restrict__ grecs = reinterpret_cast<const uint32_t*>(fa.runs);   // 4-byte records (sp.h: tile_record32), 8-byte aligned per frame
    constexpr uint32_t OFF = ~kRowRepeats;            // an index entry = record offset | kRowRepeats ("same words as the row above")
    // What a pixel's run word says, unpacked once per CHANGE of layout (not once per row): the word itself (its low three bytes are what is
    // added) and the byte selector of its starting value.  A row that repeats the layout of the row above keeps both.
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, sel0 = 0x0C0C0C0Cu, sel1 = 0x0C0C0C0Cu, sel2 = 0x0C0C0C0Cu, sel3 = 0x0C0C0C0Cu;
    // window: rows [y, y + n) whose records fit in win_cap (at most 62 rows: a window's index entries live one per lane, and a row reads the entry
    // two past its own).  A single row with more records than that (more than one run every other pixel) is scattered straight from global memory.
    constexpr int WMAX = 4;                                    // win_cap <= 128 * WMAX - 2 (tile_plan): a lane fetches two records per load
    // The window's loads are written as asm and waited for BY COUNT.  A wave's loads and stores share one in-order counter
    // (vmcnt): the compiler, seeing loads whose results are used after a loop of row stores, waits for vmcnt(0) — every window
    // then also waited for the acknowledgement of all its own row stores, 2-3 us a dozen times per tile.  Issued as asm the
    // loads are invisible to that bookkeeping; exactly WMAX load instructions go out per window (lanes past the window's end
    // re-read its last pair) and every row issues exactly one row store, so after R rows `s_waitcnt vmcnt(R)` says precisely
    // "the window's records have landed" while the R stores behind them stay in flight.
    unsigned long long wva[WMAX];
    // (Tried and not kept, round 4: touching the window AFTER the next as well — one LDS-DMA load per window into a sink, so that the
    // records are in the caches when the real request comes.  With every window read out of the same 8 KB the launch takes 0.39 - 0.42 ms
    // instead of 0.47 (profiles/r04_sp_tile_parts.txt), but the touch made it 0.49 - 0.51 through the builtin (the compiler then waits for
    // vmcnt(0) at the next LDS access) and still 0.47 - 0.49 issued as asm behind the window's loads (profiles/
    // r04_sp_tile_touch_ab.txt): what the cached build saves is the requests, not the wait for them.)
    // ve: lane r = index entry of row (first + r), lanes past n = the entry of lane n (the window's end); wb = the window's first record rounded
    // down to an even one (two-record loads stay 8-byte aligned); wn2 = records from wb to the window's end
    struct Window { uint32_t ve; uint32_t w0, wb; int n, wn2, from; bool direct; };   // (left pixels and kind counts are read again when the window begins: they need not travel a window ahead)
    auto plan_and_fetch = [&](int from) {
        Window w;
        const int k = from + lane;
        const uint32_t ve = idx[(k < ye ? k : ye) - yb];
        w.from = from;
        w.w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ve) & OFF;
        w.wb = w.w0 & ~1u;
        const bool fits = lane >= 1 && k <= ye && (int)((ve & OFF) - w.w0) <= win_cap;   // entry r fits: rows first .. first + r - 1 do
        const unsigned long long m = (__ballot(fits) >> 1) & 0x3FFFFFFFFFFFFFFFull;          // (62 rows at most)
        int n = __builtin_ctzll(~m);
        w.direct = n == 0;                                     // the first row alone is too much for the window
        if (n == 0) n = 1;
        w.n = n;
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)ve, n);
        w.ve = lane > n ? end : ve;
        w.wn2 = w.direct ? 0 : (int)((end & OFF) - w.wb);
        const int npairs = (w.wn2 + 1) >> 1;
#pragma unroll
        for (int q = 0; q < WMAX; ++q) {
            const int kq = lane + 64 * q, kk = kq < npairs ? kq : (npairs > 0 ? npairs - 1 : 0);
            const uint32_t* src = grecs + w.wb + 2 * kk;       // (a pair of this frame's records: its table is padded to an even count)
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(wva[q]) : "v"(src) : "memory");
        }
        return w;
    };
    // wait until at most `stores_behind` vector-memory operations of this wave are outstanding (the newest ones)
    auto settle_window = [&](int stores_behind) {
        switch (stores_behind < 16 ? stores_behind : 16) {     // (more than 16 rows per window: waiting down to 16 is just as exact)
#define JSP_VM(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
            JSP_VM(0) JSP_VM(1) JSP_VM(2) JSP_VM(3) JSP_VM(4) JSP_VM(5) JSP_VM(6) JSP_VM(7) JSP_VM(8)
            JSP_VM(9) JSP_VM(10) JSP_VM(11) JSP_VM(12) JSP_VM(13) JSP_VM(14) JSP_VM(15) JSP_VM(16)
#undef JSP_VM
        }
#pragma unroll
        for (int q = 0; q < WMAX; ++q) asm volatile("" : "+v"(wva[q]));   // (the values are defined from here on)
    };
    // a record -> the word the row's resolver reads.  The kind is the record's place among the row's records: the first `nc` are
    // constants, the next up to `nca` start from the pixel above, the rest from the pixel above and to the left (scalar counts)
    auto scatter = [&](uint32_t rec, int k, uint32_t nc, uint32_t nca) {
        const uint32_t kind = (uint32_t)k < nc ? (kTileHead | kTileConst) : ((uint32_t)k < nca ? (kTileHead | kTileAbove) : (kTileHead | kTileAboveLeft));
        *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(head) + ((rec >> 22) & 0x3FCu)) = (rec & 0x00FFFFFFu) | kind;
    };
    // the row store: a scalar base that walks down the band + this lane's 16 bytes of the row
    typedef __attribute__((address_space(1))) char gchar;
    const uint32_t voff = (uint32_t)x0 * 4u;
    Window nw = plan_and_fetch(yb);
    int rows_since_fetch = 0;                                  // row stores issued after the loads in flight
    int y = yb;
    while (y < ye) {
        // The window's records were asked for a window ago — all of them at once, and BEFORE the rows of the window in
        // between were stored: in a load -> wait -> LDS-write loop each 64 records cost a memory round trip of their own, and
        // every wait also waited for the frame stores in front of it (loads and stores share the counter); a tile has a
        // dozen windows.  Now a window costs one wait for the stores behind its loads.
        const Window cw = nw;
        uint32_t cw_vl, cw_vk;                                 // lane r = left pixel / kind counts of row (first + r)
        {
            const int k = cw.from + lane;
            const uint2 lk = left[(k < ye ? k : ye - 1) - yb];
            cw_vl = lk.x;
            cw_vk = lk.y;
        }
        settle_window(rows_since_fetch);
        {
            const int npairs = (cw.wn2 + 1) >> 1;
#pragma unroll
            for (int q = 0; q < WMAX; ++q) {
                const int k = lane + 64 * q;
                if (k < npairs) *reinterpret_cast<uint2*>(win + 2 * k) = make_uint2((uint32_t)wva[q], (uint32_t)(wva[q] >> 32));
            }
        }
        if (y + cw.n < ye) nw = plan_and_fetch(y + cw.n);      // the next window's records start travelling now
        rows_since_fetch = 0;
        uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)cw.ve, 0);   // this row's entry (its flag matters)
        uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)cw.ve, 1);
        {
            const int nfirst = (int)((e1 & OFF) - cw.w0);
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cw_vk, 0);
            const uint32_t nc = c0 & 0xFFFFu, nca = nc + (c0 >> 16);
            if (cw.direct) {
                for (int r = lane; r < nfirst; r += 64) scatter(load1_global(grecs + cw.w0 + r), r, nc, nca);
                __builtin_amdgcn_s_waitcnt(0x0F70);            // (its scatter read straight from memory)
            }
            __builtin_amdgcn_wave_barrier();
            if (!cw.direct) {
                const int at = (int)(cw.w0 - cw.wb);
                for (int r = lane; r < nfirst; r += 64) scatter(win[at + r], r, nc, nca);
            }
        }
        gchar* rowp = (gchar*)(dst + (size_t)y * X);           // (uniform: a scalar pair)
        for (int r = 0; r < cw.n; ++r, ++y) {
            const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)cw.ve, r + 2);   // (r + 2 <= n + 1 <= 63; past the window's end: the end again)
            const bool repeat = (e0 & kRowRepeats) != 0u;                  // (uniform) no records: the words of the row above stay
            const uint32_t eg = (uint32_t)__builtin_amdgcn_readlane((int)cw_vl, r);
            const int n_next = (int)((e2 & OFF) - (e1 & OFF));             // records of the next row (0 past the window's last)
            const int next_at = (int)((e1 & OFF) - cw.wb);
            const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)cw_vk, r + 1);   // the next row's kind counts (not used when it has no records)
            const uint32_t nrec = win[next_at + lane];                     // (every lane reads: past the row's records whatever is there, within the wave's slice)
            // the pixel left of this lane's first, one row up: the last pixel of the lane to the left; lane 0 keeps the window's word
            const uint32_t u0 = (uint32_t)__builtin_amdgcn_update_dpp((int)eg, (int)pv.w, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            if (!repeat) {
                const uint4 hv = *reinterpret_cast<const uint4*>(head + lane * PPL);
                *reinterpret_cast<uint4*>(head + lane * PPL) = make_uint4(0, 0, 0, 0);
                uint32_t last = hv.x;
                last = hv.y ? hv.y : last;
                last = hv.z ? hv.z : last;
                last = hv.w ? hv.w : last;
                // the word in force at this lane's first pixel: the last one of the nearest lane to the left that has any (lane 0
                // always has its own first: a record starts every span; its `below` is empty and whatever it fetches is not used)
                const unsigned long long seen = __ballot(last != 0u);
                // the lanes BELOW this one that have a word: the set shifted up by (64 - lane), so that this lane's own bit and everything above fall out
                // (lane 0: its result is not used; a shift by 64 leaves whatever it leaves)
                const unsigned long long below = seen << ((64 - lane) & 63);
                const uint32_t below_lo = (uint32_t)below, below_hi = (uint32_t)(below >> 32);
                const uint32_t lead = min(ffbh(below_hi), ffbh(below_lo) + 32u);   // leading zeros of the 64-bit set (v_ffbh: all ones for 0, which loses the min)
                uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - 1 - (int)lead) << 2, (int)last);   // the nearest lane below with a word: lane - 1 - lead (the instruction looks at address bits 7:2 only)
                w = hv.x ? hv.x : w; d0 = w; sel0 = ((w >> 24) & 0xFu) * kTileSelMul + kTileSelAdd;
                w = hv.y ? hv.y : w; d1 = w; sel1 = ((w >> 24) & 0xFu) * kTileSelMul + kTileSelAdd;
                w = hv.z ? hv.z : w; d2 = w; sel2 = ((w >> 24) & 0xFu) * kTileSelMul + kTileSelAdd;
                w = hv.w ? hv.w : w; d3 = w; sel3 = ((w >> 24) & 0xFu) * kTileSelMul + kTileSelAdd;
            }
            // pixel = start value + the word's low three bytes, byte by byte, in place and from the right (pixel j's start may be the OLD pixel j - 1).
            // v_perm_b32 D, S0, S1, sel: selector 0..3 = bytes of S1 (the pixel above), 4..7 = bytes of S0 (above-left), 12 = 0x00, >= 13 = 0xFF.
            // A partial (SDWA) write must not be read by the very next VALU instruction: the four pixels' chains are interleaved.  The row lives in
            // v[60:63] by name (inline asm cannot name the parts of a register tuple, and the row store wants the four pixels side by side).
#define JSP_ADD_BYTE(n) \
            "v_add_u32_sdwa v63, v63, %4 dst_sel:BYTE_" #n " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #n " src1_sel:BYTE_" #n "\n\t" \
            "v_add_u32_sdwa v62, v62, %3 dst_sel:BYTE_" #n " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #n " src1_sel:BYTE_" #n "\n\t" \
            "v_add_u32_sdwa v61, v61, %2 dst_sel:BYTE_" #n " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #n " src1_sel:BYTE_" #n "\n\t" \
            "v_add_u32_sdwa v60, v60, %1 dst_sel:BYTE_" #n " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #n " src1_sel:BYTE_" #n "\n\t"
            asm("v_perm_b32 v63, v62, v63, %8\n\t"
                "v_perm_b32 v62, v61, v62, %7\n\t"
                "v_perm_b32 v61, v60, v61, %6\n\t"
                "v_perm_b32 v60, %9, v60, %5\n\t"
                JSP_ADD_BYTE(0) JSP_ADD_BYTE(1) JSP_ADD_BYTE(2)
                "s_nop 0"
                : "+{v[60:63]}"(pv)
                : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(sel0), "v"(sel1), "v"(sel2), "v"(sel3), "v"(u0));
#undef JSP_ADD_BYTE
            // the row store, lanes past the frame's right edge switched off for its duration (no branch); the two instructions behind it are the wait
            // states a 16-byte store's data registers need before anything may write them.  (Not counted by the compiler: settle_window counts.)
            {
                unsigned long long keep;
                asm volatile("s_and_saveexec_b64 %0, %4\n\t"
                             "global_store_dwordx4 %1, %2, %3\n\t"
                             "s_mov_b64 exec, %0\n\t"
                             "s_nop 0"
                             : "=&s"(keep) : "v"(voff), "v"(pv), "s"(rowp), "s"(active_lanes) : "scc");   // (s_and_saveexec writes SCC)
            }
            rowp += (size_t)X * 4;
            ++rows_since_fetch;                                // (one row store per row, issued by every wave with an active lane)
            if (n_next > 0) {                                  // (uniform) the next row brings records: put them into the head row
                const uint32_t nc = cn & 0xFFFFu, nca = nc + (cn >> 16);
                if (lane < n_next) scatter(nrec, lane, nc, nca);
                if (n_next > 64)
                    for (int k = lane + 64; k < n_next; k += 64) scatter(win[next_at + k], k, nc, nca);   // rows with more records than lanes
            }
            __builtin_amdgcn_wave_barrier();
            e0 = e1;
            e1 = e2;
        }
    }
}

// (Round 4 tried two workgroup forms of the tile kernel above — tile waves + a LOADER wave that brings their record windows into LDS,
// commit ee3facd; RESOLVER waves that put finished rows into an LDS ring + a STORER wave that does nothing but issue row stores, commit
// 6651615 — both bit-exact, both slower in every configuration measured (505 - 553 us and 554 - 601 us against 450 - 474 us per 256 frames):
// profiles/r04_sp_tile_loader_ab.txt, r04_sp_tile_storer_ab.txt.  One wave per tile stays.)

constexpr int PWG = 256;  // 16 rows x 16 chunks of 4 pixels = 4 blocks side by side

__global__ __launch_bounds__(PWG) void sp_pframe_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ prev,
                                                        const PBlock* __restrict__ blocks,
                                                        const uint32_t* __restrict__ payload, int X, int Y, int nbx,
                                                        int vec) {
    const int ly = threadIdx.x >> 4;          // row inside the block row
    const int chunk = threadIdx.x & 15;       // 4-pixel chunk inside the 64-pixel span
    const int bx = blockIdx.x * 4 + (chunk >> 2);
    const int by = blockIdx.y;
    if (bx >= nbx) return;
    const int y = by * 16 + ly;
    const int x0 = bx * 16 + (chunk & 3) * 4;
    if (y >= Y || x0 >= X) return;
    const size_t npx = (size_t)X * Y;
    const size_t i0 = (size_t)y * X + x0;
    const bool full = vec && x0 + 4 <= X;
    // the block record and the co-located pixels of the previous frame are fetched together: most
    // blocks are unchanged, and a dependent second round trip would double this short kernel
    const PBlock pb = blocks[(size_t)by * nbx + bx];
    uint4 same = make_uint4(0, 0, 0, 0);
    if (full) same = *reinterpret_cast<const uint4*>(prev + i0);
    uint32_t px[4];
    const int cx0 = (chunk & 3) * 4;          // chunk origin relative to the block
    const bool row_in = ly >= pb.y1 && ly < pb.y2;
    const bool touched = pb.flags != 0 && row_in && cx0 < pb.x2 && cx0 + 4 > pb.x1;
    if (!touched) {
        if (full) {
            *reinterpret_cast<uint4*>(dst + i0) = same;
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < X) dst[i0 + j] = prev[i0 + j];
        return;
    }
    const int w = pb.x2 - pb.x1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rx = cx0 + j;               // column relative to the block
        uint32_t v = 0;
        if (x0 + j < X) {
            if (rx >= pb.x1 && rx < pb.x2) {
                if (pb.flags & PB_MOTION) {
                    // linear index into the previous frame, no per-axis clipping (ScreenPressor.hx:400-405);
                    // outside the buffer reads as 0
                    const long jdx = (long)(y + pb.my) * X + (x0 + j + pb.mx);
                    v = (jdx >= 0 && (size_t)jdx < npx) ? prev[jdx] : 0u;
                } else {
                    v = payload[pb.payload + (uint32_t)((ly - pb.y1) * w + (rx - pb.x1))];
                }
            } else
                v = full ? (j == 0 ? same.x : j == 1 ? same.y : j == 2 ? same.z : same.w)
                         : prev[i0 + j];      // base copy around a sub-rectangle
        }
        px[j] = v;
    }
    if (vec && x0 + 4 <= X) *reinterpret_cast<uint4*>(dst + i0) = make_uint4(px[0], px[1], px[2], px[3]);
    else
        for (int j = 0; j < 4; ++j)
            if (x0 + j < X) dst[i0 + j] = px[j];
}

// Consecutive inter frames, one launch: same lane <-> pixel mapping as sp_pframe_kernel, frames iterated
// inside the kernel with the lane's 4 pixels carried in registers.  No block of these frames is
// motion-compensated (the host stage turned such rectangles into literal ones), so a lane never needs a
// pixel another lane produced: no synchronisation between frames, the previous frame is read once, every
// frame of the group costs its block records, its literal pixels and one 16-byte store per lane.
// Loads and stores share vmcnt, so a load waited for inside the frame loop would queue behind the
// acknowledgement of every frame store issued before it.  The frame loop therefore touches LDS only: for a
// chunk of frames the workgroup first stages (a) the records of its 4 blocks and each frame's destination
// and (b) the literal pixels of its changed rectangles — as many frames as fit the literal buffer, at
// least one — and only then walks those frames.  (Fetching literals from HBM inside the loop cost a store
// round trip in 28 % of the wave-frames of the test clip: 2.1 us per frame instead of 1.6.)
constexpr int GROUP_LITERALS_MIN = 1024;  // one frame needs at most 4 x 256 words of literal pixels
struct GroupSlot {       // LDS image of one frame of the chunk, 80 bytes
    PBlock pb[4];
    uint32_t* dst;
    uint32_t payload_off;
    uint32_t pad;
};
static_assert(sizeof(GroupSlot) == 80, "GroupSlot layout");

__global__ __launch_bounds__(PWG) void sp_pframe_group1_kernel(const PGroupFrame* __restrict__ frames, int nframes,
                                                              const uint32_t* __restrict__ prev,
                                                              const PBlock* __restrict__ blocks,
                                                              const uint32_t* __restrict__ payload, int X, int Y, int nbx,
                                                              int vec, int chunk_frames, int literal_words, int stagger) {
    // LDS: [slots: chunk_frames x 80 B][lit_at: chunk_frames x 4 words][lits: literal_words][wave_tot 4][wave_over 4]
    extern __shared__ __align__(16) uint32_t group_lds[];
    GroupSlot* slots = reinterpret_cast<GroupSlot*>(group_lds);
    uint32_t* lit_at = group_lds + (size_t)chunk_frames * (sizeof(GroupSlot) / 4);   // where in `lits` the rectangle of (frame, block) starts
    uint32_t* lits = lit_at + (size_t)chunk_frames * 4;
    uint32_t* wave_tot = lits + literal_words;       // 4 words
    int* wave_over = reinterpret_cast<int*>(wave_tot + 4);   // 4 words
    const int ly = threadIdx.x >> 4;
    const int chunk = threadIdx.x & 15;
    const int kb = chunk >> 2;                // which of the workgroup's 4 blocks
    const int bx = blockIdx.x * 4 + kb;
    const int by = blockIdx.y;
    const int y = by * 16 + ly;
    const int x0 = bx * 16 + (chunk & 3) * 4;
    const bool mine = bx < nbx && y < Y && x0 < X;   // lanes past the frame edge still help staging
    const size_t i0 = (size_t)y * X + x0;
    const bool full = vec && x0 + 4 <= X;
    const int cx0 = (chunk & 3) * 4;          // chunk origin relative to the block
    uint32_t px[4] = {0, 0, 0, 0};
    if (mine) {
        if (full) {
            const uint4 q = *reinterpret_cast<const uint4*>(prev + i0);
            px[0] = q.x; px[1] = q.y; px[2] = q.z; px[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < X) px[j] = prev[i0 + j];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the previous-frame pixels are settled before any store is issued
    const int nb_here = nbx - (int)blockIdx.x * 4 < 4 ? nbx - (int)blockIdx.x * 4 : 4;   // blocks this workgroup covers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int f0 = 0;
    while (f0 < nframes) {
        int nf = nframes - f0 < chunk_frames ? nframes - f0 : chunk_frames;
        // the first chunk is shortened by a per-workgroup amount, so that workgroups do not all stop storing at
        // the same time to stage their next chunk (the staging loads wait for the store queue to drain)
        if (f0 == 0 && stagger) {
            const int first = 1 + (int)((blockIdx.x * 5u + blockIdx.y * 3u) % (unsigned)chunk_frames);
            nf = nf < first ? nf : first;
        }
        __syncthreads();                      // the previous chunk's slots and literals are no longer read
        // one round trip: the frames' destinations / literal bases and the records of this workgroup's blocks (the
        // block tables of a group's frames follow each other in memory: PGroupFrame::block_off advances by nblocks)
        const uint32_t block_off0 = frames[0].block_off, nblocks_frame = (uint32_t)nbx * (uint32_t)gridDim.y;
        for (int t = threadIdx.x; t < nf * 5; t += PWG) {
            const int f = t / 5, k = t - f * 5;
            if (k < 4) {
                PBlock pb{};
                if (k < nb_here)
                    pb = blocks[(size_t)block_off0 + (size_t)(f0 + f) * nblocks_frame + (size_t)by * nbx + blockIdx.x * 4 + k];
                slots[f].pb[k] = pb;
            } else {
                const PGroupFrame gf = frames[f0 + f];
                slots[f].dst = reinterpret_cast<uint32_t*>(gf.dst);
                slots[f].payload_off = gf.payload_off;
            }
        }
        __syncthreads();
        // where each changed rectangle's literals go in `lits`, and how many frames fit: exclusive prefix sum over the
        // (frame, block) items — one item per lane (chunk_frames <= 64), wave scans joined through LDS
        {
            const int item = threadIdx.x;             // = frame * 4 + block
            uint32_t need = 0;
            if (item < nf * 4) {
                const PBlock pb = slots[item >> 2].pb[item & 3];
                if (pb.flags & PB_DATA) need = (uint32_t)(pb.x2 - pb.x1) * (uint32_t)(pb.y2 - pb.y1);
            }
            uint32_t incl = need;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += up;
            }
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            uint32_t base = 0;
            for (int w2 = 0; w2 < wave; ++w2) base += wave_tot[w2];
            const uint32_t excl = base + incl - need;
            if (item < nf * 4) lit_at[item] = excl;
            // a frame fits if the literals up to and including its last block do; frame 0 always does (<= 1024 words)
            const bool over = item < nf * 4 && (item & 3) == 3 && base + incl > (uint32_t)literal_words && item >= 4;
            const unsigned long long m = __ballot(over);
            if (lane == 0) wave_over[wave] = m ? (wave * 64 + __ffsll((long long)m) - 1) >> 2 : 0x7FFFFFFF;
            __syncthreads();
            int fit = nf;
            for (int w2 = 0; w2 < 4; ++w2) fit = wave_over[w2] < fit ? wave_over[w2] : fit;
            nf = fit;
        }
        // fetch the literals: wave k takes block k's rectangles, lanes run along the rectangle's pixels
        for (int f = 0; f < nf; ++f) {
            const PBlock pb = slots[f].pb[wave];
            if (pb.flags & PB_DATA) {
                const uint32_t n = (uint32_t)(pb.x2 - pb.x1) * (uint32_t)(pb.y2 - pb.y1);
                const uint32_t* src = payload + slots[f].payload_off + pb.payload;
                uint32_t* to = lits + lit_at[f * 4 + wave];
                for (uint32_t i = lane; i < n; i += 64) to[i] = load1_global(src + i);
            }
        }
        __syncthreads();                      // literals are in place (the barrier waits for vmcnt and LDS)
        if (mine) {
            for (int f = 0; f < nf; ++f) {
                const PBlock pb = slots[f].pb[kb];
                uint32_t* out = slots[f].dst + i0;
                const bool touched = pb.flags != 0 && ly >= pb.y1 && ly < pb.y2 && cx0 < pb.x2 && cx0 + 4 > pb.x1;
                if (touched) {
                    const int w = pb.x2 - pb.x1;
                    const uint32_t* lit = lits + lit_at[f * 4 + kb] + (uint32_t)((ly - pb.y1) * w) - pb.x1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rx = cx0 + j;
                        if (rx >= pb.x1 && rx < pb.x2 && x0 + j < X) px[j] = lit[rx];
                    }
                }
                if (full) store4_global(out, make_uint4(px[0], px[1], px[2], px[3]));   // (nontemporal: no gain, measured)
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (x0 + j < X) store1_global(out + j, px[j]);
                }
            }
        }
        f0 += nf;
    }
}


// ---------------------------------------------------------------------------------------------------------
// Inter-frame groups, second form: 8 pixels per lane and a LOADER wave.
//
// sp_pframe_group1_kernel above staged every chunk of frames itself: all four waves stopped at workgroup barriers — which wait
// for the wave's row stores in flight — and then paid two load round trips behind the CU's store queue (54 % of its wave cycles
// parked, profiles/archive/r03_sp_pclip300_group_sq_counters.txt).  A loader wave cannot simply be added to it: with one wave per 16x16
// block a 1080p frame needs 8 160 of the chip's 8 192 wave slots.  So a lane carries TWO rows of 4 pixels here (a workgroup = 8
// adjacent blocks = 128 x 16 pixels, 256 worker lanes; a wave's store is two 512-byte row segments), half the waves do the
// same work, and the fifth wave of every workgroup fetches the next chunk — block records, frame records, literal pixels —
// straight into LDS (global_load_lds) while the workers walk the current one.  The workers' frame loop holds no load and no
// workgroup barrier; chunks change hands through two LDS counters.
constexpr int G2_BLOCKS = 8;                  // blocks per workgroup
constexpr int G2_CF = 16;                     // frames per chunk, at most
constexpr int G2_LW = 3072;                   // literal words per chunk (one frame needs at most 8 x 256)
constexpr int G2_WG = 5 * 64;
constexpr int G2_SPIN = 1 << 24;
struct G2Chunk {
    PBlock pb[G2_CF * G2_BLOCKS];             // (frame, block) records, frame-major: what one LDS-DMA per 64 records writes
    PGroupFrame gf[G2_CF];
    uint32_t lit_at[G2_CF * G2_BLOCKS];       // where in `lits` the rectangle of (frame, block) starts
    uint32_t lits[G2_LW];
    int nf, next;
    int pad[2];                               // (sizeof a multiple of 16: the second chunk's tables stay 16-byte aligned for the LDS-DMAs)
};
static_assert(sizeof(G2Chunk) % 16 == 0 && offsetof(G2Chunk, lits) % 16 == 0, "G2Chunk tables are 16-byte aligned");
static_assert(sizeof(PGroupFrame) == 16, "one LDS-DMA lane per frame record");

typedef __attribute__((address_space(1))) const void g2_gvoid;
typedef __attribute__((address_space(3))) void g2_lvoid;

__global__ __launch_bounds__(G2_WG) void sp_pframe_group_kernel(const PGroupFrame* __restrict__ frames, int nframes,
                                                                const uint32_t* __restrict__ prev,
                                                                const PBlock* __restrict__ blocks,
                                                                const uint32_t* __restrict__ payload, int X, int Y, int nbx) {
    extern __shared__ __align__(16) uint8_t g2_lds[];
    G2Chunk* chunks = reinterpret_cast<G2Chunk*>(g2_lds);                       // [2]
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;       // (LDS spelled out: generic pointers would poll with FLAT loads, which count on vmcnt too)
    lds_vu32* s_ready = (lds_vu32*)(g2_lds + 2 * sizeof(G2Chunk));          // chunks the loader has handed over
    lds_vu32* s_done = s_ready + 1;                                          // [4] chunks each worker wave is through with
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 5) s_ready[tid] = 0u;
    __syncthreads();
    const int by = blockIdx.y;
    const int nb_here = nbx - (int)blockIdx.x * G2_BLOCKS < G2_BLOCKS ? nbx - (int)blockIdx.x * G2_BLOCKS : G2_BLOCKS;

    if (wave == 4) {
        // ---------------------------------------- loader ----------------------------------------
        const uint32_t block_off0 = frames[0].block_off, nblocks_frame = (uint32_t)nbx * (uint32_t)gridDim.y;
        int f0 = 0, c = 0;
        while (f0 < nframes) {
            G2Chunk& ck = chunks[c & 1];
            // the buffer was chunk c - 2's: EVERY worker wave must be through with that chunk (a count summed over the waves
            // would let a wave that is a chunk ahead stand in for one that is still reading)
            if (c >= 2)
                for (int spin = 0; spin < G2_SPIN; ++spin) {
                    const uint32_t slowest = min(min(s_done[0], s_done[1]), min(s_done[2], s_done[3]));
                    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)slowest) >= (uint32_t)(c - 1)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            const int nf_try = nframes - f0 < G2_CF ? nframes - f0 : G2_CF;
            // 1. records: (frame, block) items 64 at a time, frame records one per lane
#pragma unroll
            for (int q = 0; q < G2_CF * G2_BLOCKS / 64; ++q) {
                const int item = q * 64 + lane, f = item >> 3, k = item & 7;
                if (q * 64 < nf_try * G2_BLOCKS && f < nf_try && k < nb_here)
                    __builtin_amdgcn_global_load_lds((g2_gvoid*)(blocks + (size_t)block_off0 + (size_t)(f0 + f) * nblocks_frame + (size_t)by * nbx + blockIdx.x * G2_BLOCKS + k),
                                                     (g2_lvoid*)&ck.pb[q * 64], 16, 0, 0);
            }
            if (lane < nf_try) __builtin_amdgcn_global_load_lds((g2_gvoid*)(frames + f0 + lane), (g2_lvoid*)&ck.gf[0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // 2. where each changed rectangle's literals go, and how many frames fit: two items per lane, exclusive scan
            uint32_t need[2] = {0, 0}, from[2] = {0, 0};    // literal words of the lane's two items, and where they start in `payload`
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int item = lane * 2 + h;
                if (item < nf_try * G2_BLOCKS && (item & 7) < nb_here) {
                    const PBlock pb = ck.pb[item];
                    if (pb.flags & PB_DATA) {
                        // (rounded up to 16 bytes: the host stage starts every rectangle's literals on a 16-byte boundary of the table — so does
                        // its place in `lits` — and the fetch below moves 16 bytes per lane; what it reads past a rectangle's end is table too)
                        need[h] = ((uint32_t)(pb.x2 - pb.x1) * (uint32_t)(pb.y2 - pb.y1) + 3u) & ~3u;
                        from[h] = ck.gf[item >> 3].payload_off + pb.payload;
                    }
                }
            }
            uint32_t incl = need[0] + need[1];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += up;
            }
            const uint32_t excl = incl - need[0] - need[1];
            ck.lit_at[lane * 2] = excl;
            ck.lit_at[lane * 2 + 1] = excl + need[0];
            // a frame fits if the literals up to and including its last block do; frame 0 always does (at most 2048 words)
            const bool over = (lane & 3) == 3 && lane * 2 + 1 < nf_try * G2_BLOCKS && incl > (uint32_t)G2_LW && lane >= 4;
            const unsigned long long om = __ballot(over);
            const int nf = om ? (__ffsll((long long)om) - 1) >> 2 : nf_try;
            // 3. the literals: rectangle by rectangle, 256 words per LDS-DMA, every request out before any is waited for
            unsigned long long want = __ballot(need[0] != 0u && lane * 2 < nf * G2_BLOCKS) ;
            unsigned long long want1 = __ballot(need[1] != 0u && lane * 2 + 1 < nf * G2_BLOCKS);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                unsigned long long m = h ? want1 : want;
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)need[h], l);
                    const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)(h ? excl + need[0] : excl), l);
                    // (everything the loop needs comes out of registers: an LDS read here would be made to wait for the LDS-DMAs
                    // already in flight, one round trip per rectangle)
                    const uint32_t* src = payload + (uint32_t)__builtin_amdgcn_readlane((int)from[h], l);
                    for (uint32_t i = 0; i < n; i += 256)          // 1 KB per instruction (4-byte lanes until round 5: four times the requests for the same bytes)
                        if (i + lane * 4 < n) __builtin_amdgcn_global_load_lds((g2_gvoid*)(src + i + lane * 4), (g2_lvoid*)&ck.lits[at + i], 16, 0, 0);
                }
            }
            if (lane == 0) { ck.nf = nf; ck.next = f0 + nf; }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) *s_ready = (uint32_t)(c + 1);
            f0 += nf;
            ++c;
        }
        return;
    }

    // ---------------------------------------- workers ----------------------------------------
    // lane = 16-byte chunk of two rows: rows r and r + 8 of the block row, chunk ch of the workgroup's 128 pixels; a wave's 64
    // lanes are two rows x 32 chunks, so one store instruction writes two 512-byte row segments
    const int r = tid >> 5, ch = tid & 31;
    const int kb = ch >> 2;                                  // which of the workgroup's 8 blocks
    const int cx0 = (ch & 3) * 4;                            // chunk origin relative to the block
    const int bx = blockIdx.x * G2_BLOCKS + kb;
    const int x0 = bx * 16 + cx0;
    const int ya = by * 16 + r, yb2 = ya + 8;
    const bool col = bx < nbx && x0 < X;
    const bool mine_a = col && ya < Y, mine_b = col && yb2 < Y;
    const size_t ia = (size_t)ya * X + x0, ib = (size_t)yb2 * X + x0;
    uint32_t pa[4] = {0, 0, 0, 0}, pb4[4] = {0, 0, 0, 0};
    if (mine_a) { const uint4 q = *reinterpret_cast<const uint4*>(prev + ia); pa[0] = q.x; pa[1] = q.y; pa[2] = q.z; pa[3] = q.w; }
    if (mine_b) { const uint4 q = *reinterpret_cast<const uint4*>(prev + ib); pb4[0] = q.x; pb4[1] = q.y; pb4[2] = q.z; pb4[3] = q.w; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the previous-frame pixels are settled before any store is issued
    for (int c = 0;; ++c) {
        int spin = 0;
        for (; *s_ready < (uint32_t)(c + 1) && spin < G2_SPIN; ++spin) __builtin_amdgcn_s_sleep(1);
        if (spin >= G2_SPIN) return;                         // (cannot happen: every wait here is bounded so that a mistake ends the launch)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const G2Chunk& ck = chunks[c & 1];
        const int nf = ck.nf, next = ck.next;
        if (col) {
            // (Asking for frame f + 1's record and destination before frame f's rows go out — the loop software-pipelined — changes nothing:
            // 0.950 against 0.956 ms per 598 frames, profiles/r05_sp_group_pipelined_ab.txt; the simple loop stays.)
            // (Round 6, measured and NOT kept: a per-wave "this frame touches none of my rows" bit from the loader, so that such frames — half of all
            // workgroup-frames — cost two stores and no record read, and the two row stores issued with a scalar base and constant lane offsets.  Each
            // bit-exact, each SLOWER in one process on the same frames: 0.9323 ms as it is | mask alone 0.9388 | scalar-base stores alone 0.9455 | both
            // 0.9587; the same with an s_sleep 8 per frame 0.9443 (profiles/r06_sp_group_touched_mask_variants.txt).  The fewer instructions stand between
            // a wave's stores the worse the memory side takes them; the records' reads pace the walk.)
            for (int f = 0; f < nf; ++f) {
                const PBlock pb = ck.pb[f * G2_BLOCKS + kb];
                uint32_t* out = reinterpret_cast<uint32_t*>(ck.gf[f].dst);
                if (pb.flags != 0 && cx0 < pb.x2 && cx0 + 4 > pb.x1) {
                    const int w = pb.x2 - pb.x1;
                    const uint32_t* lit0 = ck.lits + ck.lit_at[f * G2_BLOCKS + kb] - pb.x1;
                    if (r >= pb.y1 && r < pb.y2) {
                        const uint32_t* lit = lit0 + (uint32_t)((r - pb.y1) * w);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int rx = cx0 + j; if (rx >= pb.x1 && rx < pb.x2) pa[j] = lit[rx]; }
                    }
                    if (r + 8 >= pb.y1 && r + 8 < pb.y2) {
                        const uint32_t* lit = lit0 + (uint32_t)((r + 8 - pb.y1) * w);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int rx = cx0 + j; if (rx >= pb.x1 && rx < pb.x2) pb4[j] = lit[rx]; }
                    }
                }
                if (mine_a) store4_global(out + ia, make_uint4(pa[0], pa[1], pa[2], pa[3]));
                if (mine_b) store4_global(out + ib, make_uint4(pb4[0], pb4[1], pb4[2], pb4[3]));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) s_done[wave] = (uint32_t)(c + 1);
        if (next >= nframes) break;
    }
}


// (Round 4 also split the group launch along the TIME axis — sp_pframe_chunk_kernel: a workgroup emits 4 / 8 / 16 frames and exits, its starting
// pixels found by a last-writer look-back over block records the host stage linked: bit-exact, 0.39 - 0.52 of peak against 0.60,
// profiles/r04_sp_chunk_ab.txt; round 5's store-shape lab (profiles/r05_front_lab_instep.txt) confirms that fresh workgroups per chunk get
// what looping ones get from this shape.  Removed in round 5; commit history has kernel, host linking and tests.)

}  // namespace

namespace {
int rows_in_band(const Geometry& g, int band_rows) { return band_rows > 0 && band_rows < g.Y ? band_rows : g.Y; }

// LDS plan of sp_iframe_rows_search_kernel: two row buffers, the band's row index, 4 wrap pixels, then the window of
// run records — what a 52 KB budget (three workgroups per CU) leaves, at least one row's worth
size_t search_fixed_words(const Geometry& g, int band_rows) {
    const size_t rowidx = ((size_t)rows_in_band(g, band_rows) + 1 + 3) & ~size_t(3);
    return 2 * (((size_t)g.X + 4 + 3) & ~size_t(3)) + rowidx + 4;
}
int search_run_cap(const Geometry& g, int band_rows) {
    const size_t fixed = search_fixed_words(g, band_rows), budget_words = 52 * 1024 / 4;
    size_t cap = budget_words > fixed ? (budget_words - fixed) / 2 : 0;
    if (cap < (size_t)g.X + 2) cap = (size_t)g.X + 2;
    return (int)cap;
}
}  // namespace
size_t iframe_lds_bytes(const Geometry& g, int band_rows) {
    return sizeof(uint32_t) * (search_fixed_words(g, band_rows) + 2 * (size_t)search_run_cap(g, band_rows));
}

int choose_band_rows(const Geometry& g, int nframes) {
    if (nframes <= 0) return 0;
    // ~3000 workgroups per launch (256 CUs x 4 resident, three times over, so the last partial round is
    // short) when the batch allows it; bands of at least 24 rows keep the seed rows (one per band) near 4 %
    // of the frame
    const int want = (3072 + nframes - 1) / nframes;
    int rows = (g.Y + want - 1) / (want > 0 ? want : 1);
    if (rows < 24) rows = 24;
    return rows >= g.Y ? 0 : rows;
}

// Row-major layout: any width, any alignment (the tile kernel needs X % 4 == 0 and 16-byte aligned frame buffers).
void launch_iframes(const Geometry& g, const IFrameArgs* d_args, int nframes, int band_rows, hipStream_t stream) {
    if (nframes <= 0) return;
    if (band_rows <= 0 || band_rows >= g.Y) band_rows = g.Y;
    const int bands = (g.Y + band_rows - 1) / band_rows;
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sp_iframe_rows_search_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(sp_iframe_rows_search_kernel, dim3(nframes, bands), dim3(IWG), iframe_lds_bytes(g, band_rows), stream,
                       d_args, g.X, g.Y, search_run_cap(g, band_rows), band_rows);
}

bool iframe_tiles_ok(const Geometry& g) { return (g.X & 3) == 0 && g.aligned16; }
// 4 pixels per lane, 256-column spans (8 per lane measured 189 us against 147 us at 64 x 1080p: half as many waves,
// each with a longer serial row step)
int iframe_tile_span(const Geometry&) { return 256; }   // (8 pixels per lane, 512-column spans: 0.71 vs 0.54 ms, profiles/archive/r03_fused_notes.txt)
namespace {
constexpr size_t kTileGroupLds = 64 * 1024;            // what a workgroup of tile waves may take (hipFuncAttributeMaxDynamicSharedMemorySize below)
struct TilePlan { int rows, span, nspans, win_cap, waves; size_t lds; };
TilePlan tile_plan(const Geometry& g, int band_rows) {
    TilePlan t;
    t.rows = rows_in_band(g, band_rows);
    t.span = iframe_tile_span(g);
    t.nspans = (g.X + t.span - 1) / t.span;
    // per wave: head row + row index + {left pixel, kind counts} per row + record window (4-byte records) + 64 words a row's unconditional read of "the next
    // row's records" may run into; ~4.5 KB keeps 32 waves on a CU
    const size_t fixed = (size_t)t.span + (((size_t)t.rows + 1 + 3) & ~size_t(3)) + 2 * (((size_t)t.rows + 1) & ~size_t(1)) + 64;
    const size_t budget = 4608 / 4;
    size_t cap = budget > fixed + 2 ? (budget - fixed - 2) & ~size_t(1) : 0;
    if (cap < 128) cap = 128;                                  // (a row with more records is scattered from global memory)
    if (cap > 510) cap = 510;                                  // the kernel fetches a window with four two-record loads per lane (512 records, one may be the pad in front)
    t.win_cap = (int)cap;
    t.lds = 4 * ((fixed + cap + 2 + 3) & ~size_t(3));          // (a multiple of 16 bytes: every wave's head row is read and cleared 16 bytes per lane)
    // A tile's index and left pixels grow with the band (12 bytes per row): a workgroup takes as many neighbouring spans as fit its LDS — eight up to ~550 rows,
    // four at 1080 (one band per frame), one at 4096, the tallest band there is (sp_codec.cpp cuts taller frames).
    size_t waves = kTileGroupLds / t.lds;
    t.waves = (int)(waves < 1 ? 1 : (waves > (size_t)TILE_WAVES ? (size_t)TILE_WAVES : waves));
    return t;
}
}  // namespace
int iframe_tile_max_band_rows() { return 4096; }
void launch_iframe_tiles(const Geometry& g, const IFrameArgs* d_args, int nframes, int band_rows, hipStream_t stream) {
    if (nframes <= 0) return;
    const TilePlan t = tile_plan(g, band_rows);
    const int bands = (g.Y + t.rows - 1) / t.rows;
    const int groups = (t.nspans + t.waves - 1) / t.waves;     // workgroups per band: up to eight neighbouring spans each (1080p: one)
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sp_iframe_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileGroupLds);
    });
    // frames fastest in the launch order (0.616 of 8 TB/s against 0.573 with tiles fastest, same buffers, round 3)
    hipLaunchKernelGGL(sp_iframe_tile_kernel, dim3(nframes, bands * groups), dim3(64 * t.waves), t.lds * t.waves, stream, d_args, g.X, g.Y, t.rows, t.nspans,
                       t.win_cap, groups, (int)(t.lds / 4), t.waves);
}

void launch_pframe(const Geometry& g, int32_t* dst, const int32_t* prev, const PBlock* d_blocks,
                   const uint32_t* d_payload, hipStream_t stream) {
    const int vec = ((g.X & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(prev) & 15) == 0) ? 1 : 0;
    dim3 grid((g.nbx + 3) / 4, g.nby);
    hipLaunchKernelGGL(sp_pframe_kernel, grid, dim3(PWG), 0, stream, reinterpret_cast<uint32_t*>(dst),
                       reinterpret_cast<const uint32_t*>(prev), d_blocks, d_payload, g.X, g.Y, g.nbx, vec);
}

void launch_pframe_group(const Geometry& g, const PGroupFrame* d_frames, int nframes, const int32_t* prev,
                         const PBlock* d_blocks, const uint32_t* d_payload, bool aligned16, hipStream_t stream) {
    if (nframes <= 0) return;
    const int vec = ((g.X & 3) == 0 && aligned16 && (reinterpret_cast<uintptr_t>(prev) & 15) == 0) ? 1 : 0;
    dim3 grid((g.nbx + 3) / 4, g.nby);
    // A chunk of 32 frames and 2048 literal words is 11 KB of LDS per workgroup (8 workgroups per CU still fit).
    // Same-run A/B on one box, 299 1080p frames: 16 frames / 1024 words 682 us, 32 / 2048 640 us; on another box
    // 16 / 1024 gave 625 us, 32 / 1024 647 us (the literals no longer fit: chunks get cut short), 64 / 2048 627 us.
    // Also measured, no gain: block records regrouped per tile so that a chunk is one contiguous read; all literal
    // pixels of a chunk fetched by a flat index space in one round trip (3 % slower than rectangle by rectangle).
    // With the block records withheld (every block "unchanged", nothing staged but the destinations) the same loop
    // takes 483 us — the temporal fill ceiling; without the literal fetches 539 us.
    static const bool old_form = std::getenv("JSP_SP_GROUP_OLD") != nullptr;   // lab: the kernel that stages its own chunks
    if (vec && (g.X & 3) == 0 && !old_form) {   // whole 16-byte chunks everywhere: the loader-wave kernel
        static std::once_flag attr_once;
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sp_pframe_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        });
        const dim3 grid2((g.nbx + G2_BLOCKS - 1) / G2_BLOCKS, g.nby);
        hipLaunchKernelGGL(sp_pframe_group_kernel, grid2, dim3(G2_WG), 2 * sizeof(G2Chunk) + 32, stream, d_frames, nframes,
                           reinterpret_cast<const uint32_t*>(prev), d_blocks, d_payload, g.X, g.Y, g.nbx);
        return;
    }
    constexpr int chunk = 32, lit_words = 2048, stagger = 1;   // (chunk <= 64: one (frame, block) item per lane in the kernel's scan)
    static_assert(lit_words >= GROUP_LITERALS_MIN, "one frame's literals must fit");
    const size_t lds = (size_t)chunk * sizeof(GroupSlot) + (size_t)chunk * 16 + (size_t)lit_words * 4 + 32;
    hipLaunchKernelGGL(sp_pframe_group1_kernel, grid, dim3(PWG), lds, stream, d_frames, nframes,
                       reinterpret_cast<const uint32_t*>(prev), d_blocks, d_payload, g.X, g.Y, g.nbx, vec, chunk, lit_words, stagger);
}

}  // namespace jsp::sp
