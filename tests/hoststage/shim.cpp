// TEST-ONLY shim: exposes the product's ScreenPressor HOST stage (jsplayer_amd/csrc/sp_host.cpp,
// sp_entropy.cpp, sp_models.cpp — compiled from the same sources, no HIP runtime) so that the
// descriptor tables it emits can be checked on a machine without a GPU.  Not part of the product
// and not linked into libjsplayer_amd.so.
#include <cstring>
#include <memory>
#include <vector>
#include "../../jsplayer_amd/csrc/msv1.h"
#include "../../jsplayer_amd/csrc/sp.h"

using namespace jsp::sp;

struct Shim {
    std::vector<int32_t> dst_column;                     // hs_set_dst_column: what the destination holds in its last column (empty: not known)
    HostDecoder host;
    FrameOut out;
    std::vector<FrameOut> outs;                          // hs_decode_batch
    std::vector<std::unique_ptr<HostDecoder>> spare;
    Shim(int w, int h, int bpp) : host(w, h, bpp) {}
};

extern "C" {
void* hs_create(int w, int h, int bpp) { return new Shim(w, h, bpp); }
void hs_destroy(void* p) { delete (Shim*)p; }
void hs_preinit(void* p, int lines) { ((Shim*)p)->host.preinit(lines); }
const char* hs_error(void* p) { const char* e = ((Shim*)p)->out.error; return e ? e : ""; }
void hs_set_band_rows(void* p, int rows) { ((Shim*)p)->host.set_band_rows(rows); }
void hs_set_iframe_layout(void* p, int rows, int span) { ((Shim*)p)->host.set_iframe_layout(rows, span); }
size_t hs_tile_words(void* p, int which) { const FrameOut& o = ((Shim*)p)->out; return which ? o.left.size() : o.tile_idx.size(); }
void hs_fetch_tiles(void* p, uint32_t* idx, uint32_t* left) {
    const FrameOut& o = ((Shim*)p)->out;
    if (idx && !o.tile_idx.empty()) std::memcpy(idx, o.tile_idx.data(), o.tile_idx.size() * 4);
    if (left && !o.left.empty()) std::memcpy(left, o.left.data(), o.left.size() * 4);
}
size_t hs_seed_words(void* p) { return ((Shim*)p)->out.seeds.size(); }
void hs_fetch_seeds(void* p, uint32_t* seeds) {
    const FrameOut& o = ((Shim*)p)->out;
    if (seeds && !o.seeds.empty()) std::memcpy(seeds, o.seeds.data(), o.seeds.size() * 4);
}
// returns status; fills meta: [kind, adopted, significant, prev_cleared, nruns, nrows, nblocks, npayload, flat_colour]
void hs_set_dst_column(void* p, const int32_t* col, int n) {
    auto* s = (Shim*)p;
    s->dst_column.assign(col, col + (col ? n : 0));
}
int hs_decode(void* p, int key, const uint8_t* src, size_t n, uint64_t* meta) {
    auto* s = (Shim*)p;
    s->host.set_destination_column([s]() -> const int32_t* { return s->dst_column.empty() ? nullptr : s->dst_column.data(); });
    if (key) s->host.decode_i(src, n, s->out); else s->host.decode_p(src, n, s->out);
    const FrameOut& o = s->out;
    meta[0] = (uint64_t)o.kind; meta[1] = o.adopted; meta[2] = o.significant; meta[3] = o.prev_cleared;
    meta[4] = o.runs.size(); meta[5] = o.row_run.size(); meta[6] = o.blocks.size(); meta[7] = o.payload.size();
    meta[8] = o.flat_colour; meta[9] = o.prev_pixels; meta[10] = o.data_pixels; meta[11] = o.stream_bytes;
    return o.status;
}
static void fill_meta(const FrameOut& o, uint64_t* meta) {
    meta[0] = (uint64_t)o.kind; meta[1] = o.adopted; meta[2] = o.significant; meta[3] = o.prev_cleared;
    meta[4] = o.runs.size(); meta[5] = o.row_run.size(); meta[6] = o.blocks.size(); meta[7] = o.payload.size();
    meta[8] = o.flat_colour; meta[9] = o.prev_pixels; meta[10] = o.data_pixels; meta[11] = o.stream_bytes;
}
// decode_frames(): a run of frames, groups of pictures side by side on `threads` host threads; results kept for hs_select
void hs_decode_batch(void* p, int n, const uint8_t* const* srcs, const size_t* lens, const uint8_t* keys, int threads, int literalise) {
    auto* s = (Shim*)p;
    std::vector<HostFrame> hf(n);
    for (int i = 0; i < n; ++i) hf[i] = HostFrame{srcs[i], lens[i], keys[i] != 0};
    s->outs.assign(n, FrameOut{});
    decode_frames(s->host, s->spare, hf.data(), n, s->outs.data(), threads, literalise != 0);
}
// make frame i of the last batch "the frame just decoded" for the hs_fetch* calls; returns its status, meta as hs_decode
int hs_select(void* p, int i, uint64_t* meta) {
    auto* s = (Shim*)p;
    s->out = s->outs[i];
    fill_meta(s->out, meta);
    meta[12] = s->out.literalised;
    return s->out.status;
}
// rewrite the motion rectangles of the inter frame just decoded as literal ones; meta as hs_decode
void hs_literalise_motion(void* p, uint64_t* meta) {
    auto* s = (Shim*)p;
    s->host.literalise_motion(s->out);
    meta[6] = s->out.blocks.size(); meta[7] = s->out.payload.size();
}
void hs_fetch(void* p, uint32_t* runs /*2 per run*/, uint32_t* rows, uint8_t* blocks /*16 B each*/, uint32_t* payload) {
    const FrameOut& o = ((Shim*)p)->out;
    if (runs && !o.runs.empty()) std::memcpy(runs, o.runs.data(), o.runs.size() * sizeof(IRun));
    if (rows && !o.row_run.empty()) std::memcpy(rows, o.row_run.data(), o.row_run.size() * 4);
    if (blocks && !o.blocks.empty()) std::memcpy(blocks, o.blocks.data(), o.blocks.size() * sizeof(PBlock));
    if (payload && !o.payload.empty()) std::memcpy(payload, o.payload.data(), o.payload.size() * 4);
}

// MSVideo1 host parser (msv1_host.cpp): descriptors + the facts it settles without pixels.
// out: [early_out, changes, s1, aborted, n_coded, n_skipped, n_untouched, consumed]
void hs_msv1_parse(int bits, int w, int h, const uint8_t* src, size_t n, int have_prev, int lines, uint32_t* desc,
                   uint8_t* block_changes, uint64_t* out) {
    jsp::Msv1Geometry g{bits, w, h, w >> 2, h >> 2, (w >> 2) * (h >> 2)};
    std::vector<uint8_t> bc(block_changes, block_changes + (g.nby > 0 ? g.nby : 0));
    jsp::Msv1Parse pr;
    const size_t sjs = (size_t)(g.nblocks / 1023) * 2 + 10;
    jsp::msv1_parse(g, src, n, have_prev != 0, sjs, (lines + 3) >> 2, 0, desc, bc, pr);
    if (!bc.empty()) std::memcpy(block_changes, bc.data(), bc.size());
    out[0] = pr.early_out; out[1] = pr.changes; out[2] = pr.s1; out[3] = pr.aborted;
    out[4] = pr.n_coded; out[5] = pr.n_skipped; out[6] = pr.n_untouched; out[7] = pr.consumed;
}
int hs_msv1_is_key(int bits, int w, int h, const uint8_t* src, size_t n) {
    jsp::Msv1Geometry g{bits, w, h, w >> 2, h >> 2, (w >> 2) * (h >> 2)};
    return jsp::msv1_is_key_frame(g, src, n);
}
}
