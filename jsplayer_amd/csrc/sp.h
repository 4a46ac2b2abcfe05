// ScreenPressor path: host entropy stage -> descriptor tables -> HIP reconstruction.
//
// The entropy contexts are derived from reconstructed pixels (ScreenPressor.hx:274-275,462-463),
// so the host stage keeps a shadow of the current and previous frame while it decodes symbols;
// what it hands to the GPU is a compact description from which the frame is materialised in HBM:
//   I-frame  : run records (8 B each), cut into tiles (band of rows x 256-column span) with a per-tile row
//              index; one wave rebuilds a tile row by row, the row above in registers.  The host stage —
//              which holds every reconstructed pixel anyway — supplies the row above each band ("seed") and
//              the pixel left of each span, so tiles do not wait for each other;
//   P-frame  : one 16-byte record per 16x16 block (unchanged / motion / sub-rectangle / data) and
//              literal pixels for the data rectangles only; the kernel copies, motion-compensates
//              and patches against the previous frame in HBM.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include <hip/hip_runtime.h>

#include "sp_entropy.h"

namespace jsp::sp {

// ---- I-frame descriptors --------------------------------------------------------------------
// kind: how the pixels of the run are produced
enum RunKind : uint32_t {   // bit 0: the pixel starts from the row above; bit 1: ... from one column to the left of it
    RUN_CONST = 0,       // colour in the low 24 bits (literal runs, "repeat previous pixel" runs)
    RUN_ABOVE = 1,       // pixel i takes pixel i-X plus the byte-wise addend in the low 24 bits: 0 for the plain
                         // "copy above" predictor, the per-run delta for the gradient predictor, which
                         // telescopes inside a run: p[i]-p[i-X] = p[i0-1]-p[i0-1-X]
    RUN_ABOVE_LEFT = 3,  // pixel i takes pixel i-X-1 (addend 0)
};
constexpr uint32_t kRowRepeats = 0x80000000u;   // tile layout: flag in a row's index entry — same words as the row above, no records stored
constexpr long kRunSplit = 256;  // row-major layout: records never cross a multiple of this many columns (= one wave x 4 pixels)
struct IRun {
    uint32_t start;  // linear pixel index of the first pixel
    uint32_t word;   // low 24 bits: colour or addend; bits 24..25: RunKind; bits 26..31 zero
};
// Tile layout (sp_iframe_tile_kernel): 4 bytes per record — bits 31..24 the column of the run's first pixel inside its 256-column span,
// bits 23..0 the colour / addend.  The kind travels beside it: a row's records are stored sorted by kind (constants, "above",
// "above-left") and the row's counts — n_const | n_above << 16 — sit next to its left pixel (FrameOut::left).  In the kernel a
// scattered record becomes the word `value | kTileHead | kind nibble << 24`: kTileHead tells "a run starts here" from "nothing", and the
// nibble is what ONE multiply-add turns into the byte selector of the v_perm_b32 that picks the pixel's starting value out of
// {pixel above-left, pixel above}: selector = nibble * 0x010101 + 0x0C020100 — bytes 0..2 of the pixel above (nibble 0), bytes 4..6 = the
// pixel above-left (nibble 4), or 0xFF 0xFF 0xFF (nibble 13: selectors >= 13 read as 0xFF), byte 3 always zero (selector 12).  A constant
// therefore starts from 0xFFFFFF, and its record carries the colour PLUS ONE in every byte (mod 256): 0xFF + (c + 1) = c.  All three kinds
// are then "start value + record value, byte by byte", which is three SDWA byte adds.
constexpr uint32_t kTileHead = 0x80000000u;
constexpr uint32_t kTileAbove = 0x0u << 24;       // the pixel starts from the pixel above (RUN_ABOVE)
constexpr uint32_t kTileAboveLeft = 0x4u << 24;   // ... from the pixel above and one to the left (RUN_ABOVE_LEFT)
constexpr uint32_t kTileConst = 0xDu << 24;       // ... from 0xFFFFFF (RUN_CONST; the record holds colour + 0x010101 byte-wise)
constexpr uint32_t kTileSelMul = 0x00010101u, kTileSelAdd = 0x0C020100u;
inline uint32_t tile_record32(const IRun& r, uint32_t span_row_origin) {   // span_row_origin: linear index of (row, first column of the span)
    uint32_t v = r.word & 0x00FFFFFFu;
    if ((r.word >> 24) == RUN_CONST) v = ((v & 0x007F7F7Fu) + 0x00010101u) ^ (v & 0x00808080u);   // + 1 in every byte, no carry across bytes
    return ((r.start - span_row_origin) << 24) | v;
}

// ---- P-frame descriptors --------------------------------------------------------------------
enum : uint8_t { PB_SUBRECT = 1, PB_MOTION = 2, PB_DATA = 4 };  // 0 = unchanged block
struct PBlock {        // 16 bytes
    uint8_t flags;
    uint8_t x1, y1, x2, y2;  // changed rectangle relative to the block origin, x2/y2 exclusive
    uint8_t pad;
    uint16_t pad2;
    int16_t mx, my;
    uint32_t payload;  // index of the rectangle's first literal pixel (group launches: inside the batch's payload, not the frame's)
};
static_assert(sizeof(PBlock) == 16, "PBlock is 16 bytes");

struct Geometry {
    int X, Y, bpp, nbx, nby;
    bool aligned16 = true;  // launch-time: every frame buffer of the launch is 16-byte aligned
};

enum class FrameKind { None, Flat, Intra, Inter };

// What the host stage produced for one frame.
struct FrameOut {
    FrameKind kind = FrameKind::None;
    int status = 0;            // DecoderState
    bool adopted = false;      // dst becomes the previous frame
    bool prev_cleared = false; // the call ended with prevFrame == null (RenewI ran, decode failed)
    bool significant = false;
    uint32_t flat_colour = 0;
    std::vector<IRun> runs;          // Intra: run pieces (see kRunSplit), with a sentinel at start = X*Y
    uint64_t stream_runs = 0;        // Intra: runs the stream coded (what A = 8R + 4P counts)
    std::vector<uint32_t> row_run;   // Intra: Y+1 entries, index of the record that starts at pixel y*X
    std::vector<uint32_t> seeds;     // Intra: per band after the first, seed_stride(X) words (see IFrameArgs)
    int band_rows = 0;               // Intra: rows per band the seeds were cut for
    // Intra, tile layout (set_iframe_layout with a span): `runs` is then ordered tile by tile (band-major,
    // then column span, then row, then column; no sentinel) and these two tables describe the tiles
    int span_px = 0;
    std::vector<uint32_t> tile_idx;  // per tile band_rows+1 offsets into the 4-byte records (which travel in `runs`' memory): first record of each of its rows (| kRowRepeats), then the end
    std::vector<uint32_t> left;      // per tile band_rows PAIRS: the pixel left of the span's first pixel, one row up; the row's kind counts (n_const | n_above << 16)
    std::vector<PBlock> blocks;      // Inter
    std::vector<uint32_t> payload;   // Inter: literal pixels of the data rectangles
    uint64_t prev_pixels = 0;        // Inter: pixels the stream takes from the previous frame
    uint64_t data_pixels = 0;        // Inter: pixels the stream codes (data rectangles)
    uint64_t motion_pixels = 0;      // Inter: part of prev_pixels that is motion-compensated
    uint64_t stream_bytes = 0;
    bool literalised = false;        // Inter: literalise_motion() has been applied
    int key_differs = -2;            // key frames with HostDecoder::set_key_compare_row: 1 / 0 against the picture before, -1 there was none; -2 not worked out
    const char* error = nullptr;
    // Back to the empty state WITHOUT giving the tables' memory back: a stream's frames need about the same room one
    // after the other, and megabyte-sized allocations per frame (page faults, allocator locks shared by the host
    // threads of other streams) were a measurable part of the host stage.
    void reset() {
        kind = FrameKind::None; status = 0; adopted = prev_cleared = significant = false; flat_colour = 0;
        runs.clear(); stream_runs = 0; row_run.clear(); seeds.clear(); band_rows = 0; span_px = 0; tile_idx.clear(); left.clear();
        blocks.clear(); payload.clear(); prev_pixels = data_pixels = motion_pixels = stream_bytes = 0; literalised = false; key_differs = -2; error = nullptr;
    }
};

// Host entropy stage for one stream (one codec instance): ScreenPressor.hx state + shadow frames.
class HostDecoder {
public:
    HostDecoder(int width, int height, int bpp);
    void preinit(int lines);  // ScreenPressor.hx:86-89
    static bool is_key_frame(const uint8_t* src, size_t n);  // :96-101
    // :117-295.  `have_prev` mirrors prevFrame != null (it becomes null inside RenewI).
    void decode_i(const uint8_t* src, size_t n, FrameOut& out);
    // :302-484
    void decode_p(const uint8_t* src, size_t n, FrameOut& out);
    const Geometry& geo() const { return g_; }
    bool has_prev() const { return has_prev_; }
    // rows per band of the following key frames (0 = whole frame is one band, no seeds)
    void set_band_rows(int rows) { band_rows_ = rows < 0 ? 0 : rows; span_px_ = 0; }
    // bands AND column spans: key frames come out as independent tiles (see FrameOut::tile_idx); span 0 = row-major
    // (a tile record's column has 8 bits: spans of at most 256 columns)
    void set_iframe_layout(int band_rows, int span_px) { band_rows_ = band_rows < 0 ? 0 : band_rows; span_px_ = span_px < 0 ? 0 : (span_px > 256 ? 256 : span_px); }
    // Rewrites the motion rectangles of the inter frame just decoded as literal rectangles (pixels from
    // the shadow frame appended to the payload): no block of `out` then reads the previous frame anywhere
    // but at its own position, which is what lets consecutive inter frames share one launch.
    void literalise_motion(FrameOut& out) const;
    // ---- decoding the GOPs of a batch side by side (decode_frames below) ----
    HostDecoder(HostDecoder&&) = default;
    HostDecoder& operator=(HostDecoder&&) = default;
    // The ONE place where an inter frame reads its destination before writing it (ScreenPressor.hx:436-449): a data rectangle in
    // block column 0 whose predictor looks "left" / "above-left" of x = 0 reads, through the linear index, the LAST pixel of the row
    // above — a block this frame has not reached yet, i.e. whatever the caller's buffer held.  `column()` answers with column X - 1
    // of the destination as it stands before the frame (Y entries), or null when that is not known (the decoder then reads its own
    // shadow of the position: the picture two frames back, what a caller rotating two buffers would have there).  Asked lazily,
    // at most once per frame.
    void set_destination_column(std::function<const int32_t*()> column) { dst_column_ = std::move(column); }
    void last_column(int32_t* out) const;                       // column X - 1 of the picture the last decoded frame left
    // Key frames are also compared with the picture before them, rows from `row` on (the pixel loop of Manager.hx:413-419): the stage
    // holds both pictures.  Only meaningful on the decoder that decoded the frame before (the stream's own, frame after frame).  -1: off.
    void set_key_compare_row(int row) { key_compare_row_ = row; }
    int pinned_version() const { return version_; }            // 0: no coded key frame has chosen the entropy coder yet
    bool pin_version(int version) { return ec_ ? version_ == version : init_entropy(version); }
    struct Settings { int insignificant_blocks = 0, band_rows = 0, span_px = 0; };   // Preinit and key-frame layout: what a group's own decoder takes over from the stream's
    Settings settings() const { return {insignificant_blocks_, band_rows_, span_px_}; }
    void adopt_settings(const Settings& s) {
        insignificant_blocks_ = s.insignificant_blocks;
        band_rows_ = s.band_rows;
        span_px_ = s.span_px;
    }
    void adopt_settings(const HostDecoder& o) { adopt_settings(o.settings()); }

private:
    void note_key_compare(FrameOut& out, bool had_prev) const;
    int32_t literal();
    bool init_entropy(int version);
    void renew_i();
    Geometry g_;
    int cx_ = 0, cx1_ = 0, cxshift_;
    int version_ = 0;               // which entropy coder ec_ is (2, 3, 4); pinned by the first coded key frame
    std::unique_ptr<EntropyDecoder> ec_;
    BigVector<int32_t> shadow_[2];    // [cur_] is being written, [1-cur_] is the previous frame (huge-page candidates: sp_models.h)
    int cur_ = 0;
    bool has_prev_ = false;     // prevFrame != null
    bool decoded_i_ = false;
    bool last_flat_ = false;    // last_one_was_flat != null
    bool use_bool_ = false;
    int insignificant_blocks_ = 0;
    std::vector<int32_t> bts_;
    std::vector<uint8_t> stale_;     // per 16x16 block: shadow_[cur_] may differ from shadow_[1-cur_] there
    int stall_ = 0;
    std::function<const int32_t*()> dst_column_;
    int key_compare_row_ = -1;
    int band_rows_ = 0, span_px_ = 0;
    std::vector<IRun> tiled_;        // scratch of the tile regrouping (kept: no per-frame allocation)
    std::vector<uint32_t> cursor_;
    std::vector<uint32_t> recs32_;   // the 4-byte tile records of the frame being regrouped
    std::vector<uint32_t> slot_of_;  // tile slot of every record, in emission order
    std::vector<uint32_t> row_slot_; // tile slot of (row, span 0) for the current layout
    int row_slot_rows_ = 0, row_slot_span_ = 0;
};

// A run of a stream's frames through the host stage, groups of pictures side by side: a coded key frame renews every bit of
// decoder state (ScreenPressor.hx:108-115 + :117-295), so the frames from one coded key frame up to the next depend on
// nothing before them and are decoded by a decoder of their own on a host thread of their own.  outs[i] is what
// `stream_decoder` would have produced for frames[i] decoding them one after the other, and `stream_decoder` ends in the
// state that run would leave it in (the decoder that took the last group takes its place).  A group whose key frame does
// not decode (its failure leaves older state showing through) is re-run in order.  `literalise`: inter frames that move at
// most a quarter of their pixels get literalise_motion() applied (what the staged batch's group launches need).
struct HostFrame { const uint8_t* src; size_t n; bool key; const void* dst = nullptr; const int32_t* dst_host = nullptr; };
// What the caller's destination buffers hold in their last column (see HostDecoder::set_destination_column): asked before a frame is
// decoded into `dst`, told after.  Implemented by the codec (which knows the buffers); may be called from several host threads.
struct DstColumns {
    virtual ~DstColumns() = default;
    virtual const int32_t* before(const HostFrame& f) = 0;
    virtual void after(const HostFrame& f, const HostDecoder& d, const FrameOut& out) = 0;
};
bool starts_group(const HostFrame& f);   // a coded key frame
void decode_single(HostDecoder& d, const HostFrame& f, FrameOut& out, bool literalise, DstColumns* cols = nullptr);   // one frame through decoder `d` (what decode_frames does per frame)
void decode_frames(HostDecoder& stream_decoder, std::vector<std::unique_ptr<HostDecoder>>& spare, const HostFrame* frames,
                   int count, FrameOut* outs, int threads, bool literalise, DstColumns* cols = nullptr);

// ---- kernels (sp_kernels.hip), asynchronous on `stream` -------------------------------------
struct IFrameArgs {    // one per frame of an intra launch (grid.x = frame, grid.y = band)
    int32_t* dst;
    const IRun* runs;
    const uint32_t* row_run;
    // band b >= 1 starts at row y0 = b * band_rows and finds at seeds + (b-1) * seed_stride(X):
    // word 0 = pixel (X-1, y0-2) (what "above-left" of column 0 reads, linear index i-X-1), words 1..X = row y0-1
    const uint32_t* seeds;
    // tile layout only (launch_iframe_tiles): FrameOut::tile_idx / FrameOut::left of this frame
    const uint32_t* tile_idx;
    const uint32_t* left;
    uint32_t nruns;
    uint32_t flat;     // 1: fill with `colour`
    uint32_t colour;
    uint32_t pad;
};
inline size_t seed_stride(int X) { return (size_t)X + 1; }
inline int band_count(int Y, int band_rows) { return band_rows > 0 && band_rows < Y ? (Y + band_rows - 1) / band_rows : 1; }
// Rows per band for a launch of `nframes` key frames: enough workgroups to fill the chip several
// times over, but bands tall enough that the seed rows stay a few percent of the frame.
int choose_band_rows(const Geometry& g, int nframes);
// Tile layout: one WAVE per tile (band x 256-column span), no workgroup barrier at all.  Usable when
// iframe_tiles_ok(g): X % 4 == 0 and every frame buffer of the launch 16-byte aligned.
bool iframe_tiles_ok(const Geometry& g);
int iframe_tile_span(const Geometry& g);   // columns per tile (256 or 512): what set_iframe_layout must be given
void launch_iframe_tiles(const Geometry& g, const IFrameArgs* d_args, int nframes, int band_rows, hipStream_t stream);
int iframe_tile_max_band_rows();        // the tallest band one wave's LDS slice can index (taller frames must be cut into bands)
// band_rows <= 0 or >= Y: one band per frame
void launch_iframes(const Geometry& g, const IFrameArgs* d_args, int nframes, int band_rows, hipStream_t stream);
void launch_pframe(const Geometry& g, int32_t* dst, const int32_t* prev, const PBlock* d_blocks,
                   const uint32_t* d_payload, hipStream_t stream);
// A run of consecutive inter frames without motion blocks (see literalise_motion) in ONE launch: a lane
// keeps its 4 pixels in registers from frame to frame, so the previous frame is read from HBM once.
struct PGroupFrame {   // one per frame of the group, in decode order
    int32_t* dst;
    uint32_t block_off;    // first PBlock of the frame in d_blocks
    uint32_t payload_off;  // base of the frame's literal pixels in d_payload
};
void launch_pframe_group(const Geometry& g, const PGroupFrame* d_frames, int nframes, const int32_t* prev,
                         const PBlock* d_blocks, const uint32_t* d_payload, bool aligned16, hipStream_t stream);
constexpr int kGroupMaxFrames = 65535;        // frames one group launch may walk
size_t iframe_lds_bytes(const Geometry& g, int band_rows = 0);
constexpr int kMaxIntraWidth = 8192;  // LDS plan of the row-wavefront kernel

}  // namespace jsp::sp
