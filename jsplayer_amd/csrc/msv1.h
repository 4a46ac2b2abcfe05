// MSVideo1 (CRAM) path: host parse -> per-block descriptor table -> HIP block reconstruction.
// Reference behaviour: MSVideo1.hx (see include/jsplayer_amd.h for the per-call citations).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

namespace jsp {

// Descriptor = one uint32 per 4x4 block, raster order over the (bottom-up) buffer:
// byte offset of the block's code word inside the batch's stream buffer, or one of:
constexpr uint32_t MSV1_DESC_SKIP = 0xFFFFFFFFu;       // copy the block from the previous frame
constexpr uint32_t MSV1_DESC_UNTOUCHED = 0xFFFFFFFEu;  // leave dst as it is (end marker / abort)
// The COMPACT block table (round 6: what the replays of an inter-frame batch write and read): 2 bytes per block — the code's offset in the stream buffer
// modulo 32 768, or a sentinel with bit 15 set — plus ONE 32-bit base per group of 256 blocks: the offset of some code of that group.  A group's codes lie
// within 256 x 18 bytes of each other, so base and the 15 bits give the offset back (msv1_tab16_offset); any code of the group may serve as its base, which
// is what lets two tiles that share a group each write one without agreeing on it.  Half the table bytes, written and read.
constexpr uint32_t MSV1_TAB16_SKIP = 0xFFFEu, MSV1_TAB16_UNTOUCHED = 0xFFFFu;
inline int msv1_tab16_pitch(int nblocks) { return (nblocks + 7) & ~7; }          // entries per frame row (rows start on 16-byte boundaries)
inline int msv1_tab16_groups(int nblocks) { return (nblocks + 255) / 256; }

// Per-frame launch record, read by every workgroup of that frame (grid.y = frame).
struct Msv1FrameArgs {
    int32_t* dst;
    const int32_t* prev;   // may be null when no block is a skip and no compare is requested
    uint32_t* signif;      // device word OR-ed with 1 when a compared pixel differs
    uint32_t stream_end;   // absolute offset one past this frame's last stream byte
    uint32_t desc_base;    // index of this frame's first descriptor
    uint32_t cmp_row_lo;   // first pixel row taking part in the stage-2 compare; ~0u = none
    uint32_t pad;          // flags: MSV1_FRAME_USES_PREV
};

constexpr uint32_t MSV1_FRAME_USES_PREV = 1u;  // some block copies from / compares with the previous frame
constexpr uint32_t MSV1_FRAME_NOOP = 2u;       // early-out frame: no block is written

struct Msv1Geometry {
    int bits;  // 16 or 8
    int X, Y;
    int nbx, nby;
    int nblocks;
};

struct Msv1Parse {
    bool early_out = false;   // 16-bit only: empty / all-skip short stream -> prevFrame untouched
    bool changes = false;     // at least one coded block (dst gets adopted)
    bool s1 = false;          // stage-1 significance (coded block in a significant block row)
    bool aborted = false;     // skip block with no previous frame: the reference raises
    uint64_t n_coded = 0, n_skipped = 0, n_untouched = 0;
    uint64_t consumed = 0;    // stream bytes walked over
};

// Host parse (sequential: code lengths are data dependent, MSVideo1.hx:128-181,311-364).
// Writes geo.nblocks descriptors.  `base` is added to every offset (position of this frame's
// bytes in the batch stream buffer).  `block_changes` is the codec's persistent per-row state.
void msv1_parse(const Msv1Geometry& geo, const uint8_t* src, size_t n, bool have_prev,
                size_t size_of_just_skips, int insignificant_blocks, uint32_t base,
                uint32_t* desc, std::vector<uint8_t>& block_changes, Msv1Parse& out);

bool msv1_just_skip_blocks(const Msv1Geometry& geo, const uint8_t* src, size_t n);
int msv1_is_key_frame(const Msv1Geometry& geo, const uint8_t* src, size_t n);

// ---- on-GPU parse (msv1_parse_kernels.hip) ----------------------------------------------------
struct Msv1ParseFrame {   // one per frame of a batch
    uint32_t beg, end;        // byte range of the frame in the batch stream buffer (beg 16-byte aligned)
    uint32_t desc_base;       // index of the frame's first descriptor
    uint32_t first_tile, ntiles;
    uint32_t host_parsed;     // 1: descriptors come from the host parser, the GPU parse skips the frame
    uint32_t pad0, pad1;
};
constexpr uint32_t MSV1_INFO_END_MARKER = 1u;  // an 8-bit end-of-data marker sits on the code chain
constexpr uint32_t MSV1_INFO_S1 = 2u;          // a coded block lies in a significant block row
struct Msv1FrameInfo {    // counters the parse kernels return per frame
    uint32_t n_coded;         // coded (non-skip) blocks
    uint32_t n_skip_codes;
    uint32_t total_blocks;    // blocks covered by the whole stream (saturating); < nblocks = stream too short
    uint32_t flags;
    uint32_t consumed;        // bytes up to the end of the code covering the last block
    uint32_t pad[3];
};
uint32_t msv1_parse_tile_bytes();    // 16 KiB: staged batches (frames start on tile boundaries of the stream buffer)
uint32_t msv1_small_tile_bytes();    // 8 KiB: `small_tiles` launches (one frame per launch: modes 1, 2, 3 of msv1_launch_fused)
void msv1_launch_parse(const Msv1Geometry& geo, const uint8_t* d_stream, const Msv1ParseFrame* d_frames, int nframes,
                       const uint32_t* d_tile_frame, int ntiles, int max_tiles_per_frame, uint32_t* d_tile_tab,
                       uint32_t* d_tile_entry, uint32_t* d_tile_block0, uint32_t* d_desc, Msv1FrameInfo* d_info,
                       int insignificant_blocks, hipStream_t stream);

// Fused parse + reconstruction (msv1_fused_kernel): everything one 16 KiB tile of a frame's stream needs, in one
// 64-byte record the workgroup reads with a single scalar load.
constexpr uint32_t MSV1_TILE_SKIP = 1u;   // the frame takes the descriptor path (host-parsed)
struct alignas(16) Msv1TileRec {
    uint32_t byte0;        // first byte of the tile in the batch stream buffer
    uint32_t frame_end;    // end of the frame's whole code units (even)
    uint32_t data_end;     // end of the bytes a code may read: frame_end (16-bit) or the frame's true end (8-bit)
    uint32_t k;            // index of the tile within its frame
    uint32_t first_tile;   // index of the frame's first tile within the batch
    uint32_t ntiles;       // tiles of the frame
    uint32_t cmp_row_lo;   // first pixel row of the stage-2 compare; ~0u = none
    uint32_t flags;        // MSV1_TILE_SKIP
    int32_t* dst;
    const int32_t* prev;
    uint32_t* signif;
    uint64_t pad;
};
static_assert(sizeof(Msv1TileRec) == 64, "one record = 64 bytes");
// Tiles [tile0, tile0 + ntiles) in one launch (no descriptor table).  `d_agg`: 9 words per tile of the batch, zeroed
// once; `epoch` (> 0) must differ from launch to launch on the same `d_agg`.  *d_fault becomes non-zero if a tile
// gave up waiting for the tables of the tiles before it (the caller reports the batch as failed).
// mode 1 / 2 (one frame per launch, `d_info` != null): the scout pass that reports what the descriptor path's parse would
// have told the host (Msv1AsyncInfo), then the decode pass, which does nothing when d_info->flags & bad_mask; the
// frame's significance word is d_info->signif; *d_poison (one word per codec instance) is set by a vetoed decode pass
// and vetoes every later one until the host clears it.  mode 0: the batch form.
constexpr uint32_t MSV1_ASYNC_SHORT = 1u;     // the stream does not cover every block: the host parser has to settle it
constexpr uint32_t MSV1_ASYNC_S1 = 2u;        // a coded block lies in a significant block row (MSVideo1.hx:187-194)
constexpr uint32_t MSV1_ASYNC_END = 4u;       // an 8-bit end-of-data marker sits on the code chain
constexpr uint32_t MSV1_ASYNC_SKIPCODE = 8u;  // a skip code sits on the code chain
constexpr uint32_t MSV1_ASYNC_STUCK = 16u;    // mode 3: the frame's tiles did not all report in time (the GPU is shared with something that
                                              // keeps them from being resident together): the host path settles the frame
struct Msv1AsyncInfo {
    uint32_t flags;
    uint32_t signif;   // stage-2 significance word (OR-ed with 1 when a compared pixel differs)
    uint32_t fault;    // look-back gave up
    uint32_t arrived;  // mode 3: workgroups whose findings are in `flags` / that have written their last pixel; both run
    uint32_t finished; //         on from launch to launch (see `want`)
    uint32_t verdict;  // mode 3: 0 undecided, MSV1_VERDICT_GO / _VETO — set ONCE per launch (compare-and-swap), obeyed by every tile
    uint32_t pad[2];
};
constexpr uint32_t MSV1_VERDICT_GO = 1u, MSV1_VERDICT_VETO = 2u;
constexpr uint32_t MSV1_LAB_DEAF = 0x80000000u;   // in `bad_mask` (tests): mode 3 tiles never see all reports in — the verdict is the time-out's
// mode 4 (batch form): nothing is rebuilt; every tile writes its blocks' entries of the frame's descriptor table (the record's
// `dst` points at it) — the on-GPU descriptor parse in ONE launch, for the batches whose frames depend on each other.
// mode 5: the same, writing the COMPACT table (above): the record's `dst` points at the frame's 2-byte entries, its `prev` at the frame's group bases.
// mode 3 (one frame per launch, at most MSV1_MERGED_MAX_TILES tiles): scout and decode in ONE launch — every tile parses
// and reports, waits until all `ntiles` reports are in (`want` = the value of d_info->arrived / finished once this launch is
// through), and only then writes, or does not.  All tiles share `*one_rec` (k = the tile's index); the last workgroup copies
// flags / signif / fault to `h_info` (pinned host memory) and clears them in d_info.  `d_stream` may be pinned HOST memory
// (any alignment, nothing read past the frame's last byte): each tile reads its bytes once, over the bus, and leaves a copy
// in `d_keep` (HBM, the frame's size rounded up to 16 bytes).
constexpr int MSV1_MERGED_MAX_TILES = 128;
// mode 3 with SEVERAL frames in one launch: the frames submitted right behind the first one (up to 3) ride along — workgroups
// [0, f[0].first_wg) are the first frame's tiles, [f[i].first_wg, f[i].first_wg + f[i].rec.ntiles) those of rider i.  All frames load, parse
// and reach their verdicts side by side (none of that touches a pixel); a rider's tiles then wait until every tile of the frame in front
// has finished (that frame's `finished` has reached its `want`) before they write — a frame may copy from the pixels of the one before it
// and be compared with them — and leave their frame unwritten when the one in front was vetoed.  One launch and one kernel-to-kernel gap
// per group, and every frame's parse hidden behind the painting of the frames before it: what bounds ONE player stream is the chain of
// its frames' kernels.  count = 0: one frame, as ever.
struct Msv1Rider {
    const uint8_t* stream = nullptr;
    unsigned long long* agg = nullptr;
    Msv1AsyncInfo* info = nullptr;
    Msv1AsyncInfo* host_info = nullptr;
    uint8_t* keep = nullptr;
    Msv1TileRec rec{};
    uint32_t epoch = 0, bad_mask = 0, want = 0, first_wg = 0;
};
constexpr int MSV1_MAX_RIDERS = 3;
struct Msv1Riders {
    uint32_t count = 0, pad = 0;
    Msv1Rider f[MSV1_MAX_RIDERS];
};
// One-frame launches of a small frame (an inter frame, an 8-bit frame: a few hundred KB) use 8 KiB tiles — `small_tiles` —: twice the
// workgroups, half the serial work in each; the kernel's time is a chain of dependent steps per tile, and such a launch has the
// GPU to itself.  (1080p inter frames with 70 % skipped blocks: 28.8 -> 37.9 Gpixels/s on one stream, 8-bit key frames 42.4 -> 52.8;
// frames of a megabyte gain nothing on one stream and lose a fifth on sixteen: they keep 16 KiB tiles — with their uploads gone, jsp_prefetch,
// they lose an eighth on one stream as well: profiles/r04_msv1_small_tiles_e2e.txt.)
constexpr size_t MSV1_SMALL_TILE_FRAME_BYTES = 640 * 1024;
void msv1_launch_fused(const Msv1Geometry& geo, const uint8_t* d_stream, const Msv1TileRec* d_recs, const int32_t* d_palette,
                       unsigned long long* d_agg, uint32_t epoch, uint32_t tile0, int ntiles, uint32_t* d_fault,
                       hipStream_t stream, Msv1AsyncInfo* d_info = nullptr, int insignificant_blocks = 0, int mode = 0,
                       uint32_t bad_mask = 0, uint32_t* d_poison = nullptr, const Msv1TileRec* one_rec = nullptr,
                       Msv1AsyncInfo* h_info = nullptr, uint32_t want = 0, uint8_t* d_keep = nullptr, bool small_tiles = false,
                       const Msv1Riders* riders = nullptr);   // (mode 3 only: more frames in the same launch; first_wg is filled in here)

// Kernel launchers (msv1_kernels.hip).  All asynchronous on `stream`.
void msv1_launch_blocks(const Msv1Geometry& geo, const uint8_t* d_stream, const uint32_t* d_desc,
                        const Msv1FrameArgs* d_frames, int nframes, const int32_t* d_palette,
                        bool vec_ok, hipStream_t stream);
// Inter-frame group in one launch: workgroup = spatial tile, frames walked in registers (needs X%4==0,
// 16-byte aligned buffers, and frames that write all of their blocks or none).
void msv1_launch_blocks_temporal(const Msv1Geometry& geo, const uint8_t* d_stream, const uint32_t* d_desc,
                                 const Msv1FrameArgs* d_frames, int nframes, const int32_t* d_palette,
                                 hipStream_t stream, const uint16_t* d_tab16 = nullptr, const uint32_t* d_bases = nullptr);   // d_tab16 != null: the compact table instead of d_desc
// The compact form of `count` frames' 4-byte tables (frames listed in d_list): for the frames of a batch whose tables the replays do not rewrite.
void msv1_launch_tables_compact(const Msv1Geometry& geo, const uint32_t* d_desc, uint16_t* d_tab16, uint32_t* d_bases, const uint32_t* d_list, int count,
                                hipStream_t stream);
// Stage-2 compare over the pixels no block covers (X&3 / Y&3 remainders), MSVideo1.hx:197-203.
void msv1_launch_edge_compare(const Msv1Geometry& geo, const Msv1FrameArgs* d_frames, int nframes,
                              hipStream_t stream);

}  // namespace jsp
