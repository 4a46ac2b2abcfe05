// ScreenPressor host entropy stage: walks the symbol stream exactly as ScreenPressor.hx does
// (DecompressI :117-295, DecompressP :302-484), keeps the shadow frames the context derivation
// needs, and emits the descriptor tables the HIP kernels materialise the frame from.
#include "sp.h"

#include <cstring>
#include <exception>
#include <thread>

namespace jsp::sp {

namespace {
constexpr int kStallLimit = 65536;  // consecutive zero-length runs before we call it a hang
}

HostDecoder::HostDecoder(int width, int height, int bpp) {
    g_.X = width;
    g_.Y = height;
    g_.bpp = bpp;
    g_.nbx = (width + 15) / 16;
    g_.nby = (height + 15) / 16;
    cxshift_ = bpp == 16 ? 0 : 2;  // ScreenPressor.hx:59
    bts_.assign((size_t)g_.nbx * g_.nby, 0);
    stale_.assign(bts_.size(), 1);
    for (auto& s : shadow_) s.assign((size_t)width * height, 0);
}

void HostDecoder::preinit(int lines) { insignificant_blocks_ = g_.nbx * ((lines + 15) / 16); }

bool HostDecoder::is_key_frame(const uint8_t* src, size_t n) {
    if (!src || n == 0) return false;
    const int b = src[0];
    return b == 0x12 || b == 0x11 || b == 0x22 || b == 0x21 || b == 0x32 || b == 0x31;
}

bool HostDecoder::init_entropy(int version) {  // ScreenPressor.hx:66-79
    version_ = version;
    switch (version) {
        case 2: ec_ = make_range_decoder(); break;
        case 3: ec_ = make_rans_decoder(64); cxshift_ = 2; break;
        case 4: ec_ = make_rans_decoder(32); cxshift_ = 2; break;
        default: version_ = 0; return false;
    }
    use_bool_ = ec_->has_bool();
    return true;
}

void HostDecoder::renew_i() {  // ScreenPressor.hx:108-115
    has_prev_ = false;
    if (last_flat_) return;
    if (!ec_) throw DecodeAbort{"flat key frame before any coded key frame: the reference dereferences a null coder"};
    ec_->renewI();
}

int32_t HostDecoder::literal() {  // ScreenPressor.hx:173-189 / 224-235 / 419-430 (one call into the coder for the three components)
    const int64_t v = ec_->literal(cx_, cx1_, cxshift_);
    if (v < 0) throw DecodeAbort{"colour context outside the table"};
    return (int32_t)v;
}

void HostDecoder::decode_i(const uint8_t* src, size_t n, FrameOut& out) {
    out.reset();
    const long X = g_.X, end = (long)g_.X * g_.Y;
    int32_t* dst = shadow_[cur_].data();
    const int head = n ? src[0] : 0;
    const int version = (head >> 4) + 1;
    const bool had_prev = has_prev_;
    try {
        if ((head & 0xF) == 1) {  // flat, :132-155
            renew_i();
            uint32_t c;
            if (g_.bpp == 16) {
                const int v = n >= 2 ? src[0] + src[1] * 256 : 0;
                c = (uint32_t)((((v >> 10) & 0x1F) << 3) << 16) + (uint32_t)((((v >> 5) & 0x1F) << 3) << 8) +
                    (uint32_t)((v & 0x1F) << 3);
            } else {
                const int b = n > 1 ? src[1] : -1, gg = n > 2 ? src[2] : 0, r = n > 3 ? src[3] : 0;
                c = b < 0 ? 0u : (uint32_t)((r << 16) + (gg << 8) + b);
            }
            std::fill(dst, dst + end, (int32_t)c);
            out.kind = FrameKind::Flat;
            out.flat_colour = c;
            out.adopted = true;
            out.stream_bytes = n < 4 ? n : 4;
            has_prev_ = true;
            last_flat_ = true;
            decoded_i_ = true;
            cur_ ^= 1;
            note_key_compare(out, had_prev);
            std::fill(stale_.begin(), stale_.end(), 1);
            return;
        }
        last_flat_ = false;
        if ((head & 0xF) != 2) { out.status = 2; out.error = "unknown ScreenPressor frame header"; return; }
        if (!ec_ && !init_entropy(version)) { out.status = 2; out.error = "unknown ScreenPressor stream version"; return; }
        renew_i();
        ec_->begin(src, n, 1);
        cx_ = cx1_ = 0;
        stall_ = 0;
        auto progress = [&](bool advanced) {
            if (advanced) stall_ = 0;
            else if (++stall_ > kStallLimit) throw DecodeAbort{"no progress: the reference would never return"};
        };
        auto& runs = out.runs;
        runs.clear();
        const bool tile_layout = span_px_ > 0;
        const long split = tile_layout ? span_px_ : kRunSplit;
        const int split_shift = (split & (split - 1)) == 0 ? __builtin_ctzl((unsigned long)split) : -1;
        const int rows_per = band_rows_ > 0 && band_rows_ < g_.Y ? band_rows_ : g_.Y;
        const int nspans = tile_layout ? (int)((X + split - 1) / split) : 0;
        const size_t stride = (size_t)rows_per + 1;   // index entries per tile: its rows, then its end
        if (tile_layout && (row_slot_.size() != (size_t)g_.Y || row_slot_rows_ != rows_per || row_slot_span_ != span_px_)) {
            row_slot_.resize(g_.Y);   // tile slot of (row y, span 0): ((band * nspans) * stride + row inside the band)
            for (int y = 0; y < g_.Y; ++y) row_slot_[y] = (uint32_t)(((size_t)(y / rows_per) * nspans) * stride + (size_t)(y % rows_per));
            row_slot_rows_ = rows_per;
            row_slot_span_ = span_px_;
        }
        slot_of_.clear();
        long di = 0;
        int row = 0, col = 0;   // di == row * X + col, kept without dividing
        // One record per piece of a run: runs are cut at row starts and at every `split`-th column, so a
        // row's records are exactly those that start inside it and none crosses the span of one wave of
        // the row kernel (which then scatters each record with a single store, no clamping, no loop).  In the
        // tile layout the tile slot of every piece is noted on the way: the regrouping below never divides.
        auto emit = [&](int cnt, uint32_t kind, uint32_t payload) {   // the run starts at di
            if (cnt <= 0 || di >= end) return;
            ++out.stream_runs;
            const uint32_t word = (payload & 0xFFFFFFu) | (kind << 24);
            long left = di + cnt < end ? cnt : end - di;
            long i = di;
            int y = row, c = col;
            long sp = split_shift >= 0 ? c >> split_shift : c / split;
            while (left > 0) {
                runs.push_back({(uint32_t)i, word});
                if (tile_layout) slot_of_.push_back(row_slot_[y] + (uint32_t)(sp * stride));
                const long next_col = (sp + 1) * split < X ? (sp + 1) * split : X;
                const long step = next_col - c;
                i += step;
                left -= step;
                c = (int)next_col;
                ++sp;
                if (c >= X) { c = 0; sp = 0; ++y; }
            }
        };
        auto advance = [&](int cnt) {
            di += cnt;
            col += cnt;
            while (col >= X) { col -= (int)X; ++row; }
        };
        auto rd = [&](long i) -> int32_t { return (i >= 0 && i < end) ? dst[i] : 0; };
        auto inside = [&](int cnt) -> long { return di >= end ? 0 : (di + cnt < end ? cnt : end - di); };
        auto fill = [&](int cnt, int32_t v) { std::fill_n(dst + di, inside(cnt), v); };
        // pixel i takes pixel i - back (+ the byte-wise addend d): forward order matters only when the run is longer
        // than `back`, so it goes in pieces of at most X pixels, each of them a plain non-overlapping copy
        auto from_above = [&](int cnt, long back, uint32_t d) {
            const long n = inside(cnt);
            for (long off = 0; off < n; off += X) {
                const long len = n - off < X ? n - off : X;
                int32_t* to = dst + di + off;
                const int32_t* from = to - back;
                if (d == 0) std::memcpy(to, from, (size_t)len * 4);
                else
                    for (long j = 0; j < len; ++j) {
                        const uint32_t u = (uint32_t)from[j];
                        to[j] = (int32_t)((((u & 0x7F7F7Fu) + (d & 0x7F7F7Fu)) ^ ((u ^ d) & 0x808080u)) & 0xFFFFFFu);
                    }
            }
        };
        long k = 0;
        int32_t clr = 0;
        while (k < X + 1) {  // phase 1: literal runs until a full row + 1 exists, :170-197
            clr = literal();
            const int cnt = ec_->run(0);
            k += cnt;
            progress(cnt > 0);
            emit(cnt, RUN_CONST, (uint32_t)clr);
            fill(cnt, clr);
            advance(cnt);
        }
        int mask1 = 0xFC00, shift1 = 4, shiftc = 18;
        if (g_.bpp == 16 && ec_->rc_16bpp_constants()) { mask1 = 0xFF00; shift1 = 2; shiftc = 16; }
        int pt = 0;
        while (di < end) {  // phase 2, :218-286.  di >= X + 1 here: the row above and its left neighbour exist
            pt = ec_->ptype(pt);
            if (pt >= 6) throw DecodeAbort{"predictor type outside its tables"};
            if (pt == 0) clr = literal();
            const int cnt = ec_->run(pt);
            progress(cnt > 0 && pt != 3);
            switch (pt) {
                case 0:
                    emit(cnt, RUN_CONST, (uint32_t)clr);
                    fill(cnt, clr);
                    advance(cnt);
                    break;
                case 1:  // repeat the pixel before the run
                    clr = rd(di - 1);
                    emit(cnt, RUN_CONST, (uint32_t)clr);
                    fill(cnt, clr);
                    advance(cnt);
                    break;
                case 2:
                case 5: {
                    const long back = pt == 2 ? X : X + 1;
                    emit(cnt, pt == 2 ? RUN_ABOVE : RUN_ABOVE_LEFT, 0);
                    if (cnt > 0) {
                        from_above(cnt, back, 0);
                        clr = rd(di + cnt - 1 - back);   // the last pixel of the run, also when the run overshoots the frame
                        advance(cnt);
                    }
                    break;
                }
                case 4: {  // left + above - aboveleft per byte; constant offset from the row above
                    if (cnt > 0) {
                        const uint32_t a = (uint32_t)rd(di - 1), b = (uint32_t)rd(di - 1 - X);
                        const uint32_t d = (((a & 0xFF) - (b & 0xFF)) & 0xFF) | (((a & 0xFF00) - (b & 0xFF00)) & 0xFF00) |
                                           (((a & 0xFF0000) - (b & 0xFF0000)) & 0xFF0000);
                        emit(cnt, RUN_ABOVE, d);
                        from_above(cnt, X, d);
                        const uint32_t u = (uint32_t)rd(di + cnt - 1 - X);
                        clr = (int32_t)((((u & 0xFF) + (d & 0xFF)) & 0xFF) | (((u & 0xFF00) + (d & 0xFF00)) & 0xFF00) |
                                        (((u & 0xFF0000) + (d & 0xFF0000)) & 0xFF0000));
                        advance(cnt);
                    }
                    break;
                }
                default: break;  // 3 has no case in the I-frame switch: nothing written
            }
            cx1_ = (clr & mask1) >> shift1;
            cx_ = clr >> shiftc;
        }
        runs.push_back({(uint32_t)end, 0});  // sentinel
        out.row_run.assign(g_.Y + 1, 0);
        size_t r = 0;
        for (int y = 0; y <= g_.Y; ++y) {
            const uint32_t first = (uint32_t)((long)y * X);
            while (r + 1 < runs.size() && runs[r + 1].start <= first) ++r;
            out.row_run[y] = (uint32_t)r;
        }
        if (tile_layout) {  // records regrouped tile by tile, plus what each tile needs from its left
            const int nbands = (g_.Y + rows_per - 1) / rows_per;
            const size_t ntiles = (size_t)nbands * nspans;
            runs.pop_back();                              // the sentinel has no place in a tile
            std::vector<uint32_t>& idx = out.tile_idx;
            idx.assign(ntiles * stride + 1, 0);           // counts first (shifted by one), then prefix sums
            const uint32_t* slot_of = slot_of_.data();    // per record, noted when it was emitted
            const size_t nrec = runs.size();
            for (size_t i = 0; i < nrec; ++i) ++idx[slot_of[i] + 1];
            for (size_t i = 1; i < idx.size(); ++i) idx[i] += idx[i - 1];
            std::vector<IRun>& tiled = tiled_;
            tiled.resize(runs.size());
            std::vector<uint32_t>& cursor = cursor_;
            cursor.assign(idx.begin(), idx.end() - 1);
            for (size_t i = 0; i < nrec; ++i) tiled[cursor[slot_of[i]]++] = runs[i];   // stable: row-major order survives inside a tile
            idx.pop_back();                               // ntiles * stride entries: the slot after a tile's last row is its end
            // Screen content repeats itself from row to row: a row of a tile whose records equal, column for column, those
            // of the row above it in the same tile is not stored at all — bit 31 of its index entry (kRowRepeats) says "the
            // same words as the row above", and the wave keeps using the words it has in registers.
            // What is stored is 4 bytes per record (tile_record32, sp.h): column inside the span | 24-bit colour / addend.  The record's KIND is
            // not in it: a row's records are stored sorted by kind — constants, then "above", then "above-left" — and the row's two counts sit
            // beside its left pixel (out.left: two words per row).  Scattering records into the row is order-independent, so the kernel
            // loses nothing by the sort, and a key frame's records are half the bytes they were (8-byte {offset, word} pairs until round 4).
            runs.clear();
            cursor.assign(idx.begin(), idx.end());        // the index into `tiled`; idx is rewritten to index the packed records
            std::vector<uint32_t>& recs = recs32_;
            recs.clear();
            out.left.assign(ntiles * rows_per * 2, 0);
            for (size_t t = 0; t < ntiles; ++t) {
                const size_t base = t * stride;
                for (int r = 0; r < rows_per; ++r) {
                    const uint32_t lo = cursor[base + r], hi = cursor[base + r + 1];
                    bool same = r > 0 && hi > lo && hi - lo == lo - cursor[base + r - 1];
                    const uint32_t plo = same ? cursor[base + r - 1] : 0;
                    for (uint32_t k = 0; same && k < hi - lo; ++k)
                        same = tiled[lo + k].word == tiled[plo + k].word && tiled[lo + k].start - tiled[plo + k].start == (uint32_t)X;
                    idx[base + r] = (uint32_t)recs.size() | (same ? kRowRepeats : 0u);
                    if (!same) {
                        const size_t b = t / (size_t)nspans, sp = t % (size_t)nspans;
                        const uint32_t origin = (uint32_t)((b * (size_t)rows_per + (size_t)r) * (size_t)X + sp * (size_t)span_px_);
                        uint32_t n_const = 0, n_above = 0;
                        for (uint32_t k = lo; k < hi; ++k) if ((tiled[k].word >> 24) == RUN_CONST) { recs.push_back(tile_record32(tiled[k], origin)); ++n_const; }
                        for (uint32_t k = lo; k < hi; ++k) if ((tiled[k].word >> 24) == RUN_ABOVE) { recs.push_back(tile_record32(tiled[k], origin)); ++n_above; }
                        for (uint32_t k = lo; k < hi; ++k) if ((tiled[k].word >> 24) == RUN_ABOVE_LEFT) recs.push_back(tile_record32(tiled[k], origin));
                        out.left[(t * rows_per + (size_t)r) * 2 + 1] = n_const | (n_above << 16);
                    }
                }
                idx[base + rows_per] = (uint32_t)recs.size();   // the tile's end
            }
            // (the records travel in `runs`' memory, two to an element, padded to an even count: every frame's records then start on an
            // 8-byte boundary in the batch's table, which the kernel's two-record loads rely on)
            if (recs.size() & 1) recs.push_back(0);
            runs.resize(recs.size() / 2);
            if (!recs.empty()) std::memcpy(static_cast<void*>(runs.data()), recs.data(), recs.size() * 4);
            for (int b = 0; b < nbands; ++b)
                for (int sp = 0; sp < nspans; ++sp)
                    for (int r = 0; r < rows_per && b * rows_per + r < g_.Y; ++r) {
                        const long y = (long)b * rows_per + r;
                        // what "above-left" of the span's first pixel reads: one row up, one column left;
                        // for column 0 the linear index wraps to the last pixel two rows up
                        const long i = y * X + (long)sp * span_px_ - X - 1;
                        out.left[(((size_t)b * nspans + sp) * rows_per + r) * 2] = i >= 0 ? (uint32_t)dst[i] : 0u;
                    }
            out.span_px = span_px_;
        }
        if (band_rows_ > 0 && band_rows_ < g_.Y) {  // the row above every band after the first, from the shadow
            out.band_rows = band_rows_;
            for (long y0 = band_rows_; y0 < g_.Y; y0 += band_rows_) {
                out.seeds.push_back(y0 >= 2 ? (uint32_t)dst[(y0 - 1) * X - 1] : 0u);
                out.seeds.insert(out.seeds.end(), dst + (y0 - 1) * X, dst + y0 * X);
            }
        }
        out.kind = FrameKind::Intra;
        out.adopted = true;
        out.stream_bytes = ec_->consumed();
        has_prev_ = true;
        decoded_i_ = true;
        cur_ ^= 1;
        note_key_compare(out, had_prev);
        std::fill(stale_.begin(), stale_.end(), 1);
    } catch (const DecodeAbort& a) {
        out.reset();
        out.status = 2;
        out.error = a.why;
        out.prev_cleared = had_prev && !has_prev_;  // RenewI nulled prevFrame and it stays null
        std::fill(stale_.begin(), stale_.end(), 1);
    }
}

void HostDecoder::decode_p(const uint8_t* src, size_t n, FrameOut& out) {
    out.reset();
    last_flat_ = false;
    if (n == 0 || !decoded_i_) return;  // :308-309
    if (src[0] == 0) return;            // :311-313 "no changes"
    const long X = g_.X, end = (long)g_.X * g_.Y;
    int32_t* dst = shadow_[cur_].data();
    const int32_t* prev = shadow_[cur_ ^ 1].data();
    try {
        if (!ec_) throw DecodeAbort{"no entropy coder"};
        int mask1 = 0xFC00, shift1 = 4, shiftc = 18;
        if (ec_->rc_16bpp_constants() && g_.bpp == 16) { mask1 = 0xFF00; shift1 = 2; shiftc = 16; }
        ec_->begin(src, n, 1);
        stall_ = 0;
        auto progress = [&](bool advanced) {
            if (advanced) stall_ = 0;
            else if (++stall_ > kStallLimit) throw DecodeAbort{"no progress: the reference would never return"};
        };
        int t = ec_->xx();
        const int xx1 = (ec_->xx() << 8) + t;
        t = ec_->xx();
        const int xx2 = (ec_->xx() << 8) + t;
        std::fill(bts_.begin(), bts_.end(), 0);
        const long nb = (long)bts_.size();
        for (long x = xx1; x <= xx2;) {  // block-type runs, :335-344
            const int bt = ec_->bt();
            const int cnt = ec_->bn();
            for (int i = 0; i < cnt; ++i, ++x)
                if (x >= 0 && x < nb) bts_[x] = bt;
            progress(cnt > 0);
        }
        bool signif = false;
        for (long i = insignificant_blocks_ < 0 ? 0 : insignificant_blocks_; i < nb; ++i)
            if (bts_[i] > 0) { signif = true; break; }

        out.blocks.assign((size_t)nb, PBlock{});
        out.payload.clear();
        auto rdp = [&](long i) -> int32_t { return (i >= 0 && i < end) ? prev[i] : 0; };
        auto rdd = [&](long i) -> int32_t { return (i >= 0 && i < end) ? dst[i] : 0; };
        // a read "left of x = 0": linear index i = pixel (X - 1, ry).  Rows of the block row above are this frame's already; rows of
        // the current block row belong to a block not decoded yet: what the caller's buffer holds (set_destination_column)
        const int32_t* dst_col = nullptr;
        bool dst_col_asked = false;
        auto wrapped = [&](long i, int ry, int y16) -> int32_t {
            if (ry >= y16 && g_.nbx > 1 && dst_column_) {
                if (!dst_col_asked) { dst_col = dst_column_(); dst_col_asked = true; }
                if (dst_col) return dst_col[ry];
            }
            return rdd(i);
        };
        auto need_prev = [&] { if (!has_prev_) throw DecodeAbort{"block copied from a previous frame that does not exist"}; };
        int32_t clr = 0;
        cx_ = cx1_ = 0;
        int lastmx = 0, lastmy = 0;
        for (int by = 0; by < g_.nby; ++by)
            for (int bx = 0; bx < g_.nbx; ++bx) {
                const int x16 = bx * 16, y16 = by * 16;
                int x1 = x16, y1 = y16, x2 = x16 + 16 > g_.X ? g_.X : x16 + 16, y2 = y16 + 16 > g_.Y ? g_.Y : y16 + 16;
                const int bw = x2 - x1;
                const size_t bi = (size_t)by * g_.nbx + bx;
                const int bt = bts_[bi];
                PBlock& pb = out.blocks[bi];
                // The two shadow frames differ only where the frame before this one changed something (stale_): the rest of
                // an unchanged block is already in place.
                auto copy_block = [&] {
                    need_prev();
                    if (!stale_[bi]) return;
                    for (int y = y1; y < y2; ++y) std::memcpy(dst + (long)y * X + x1, prev + (long)y * X + x1, sizeof(int32_t) * bw);
                    stale_[bi] = 0;
                };
                if (bt <= 0) {  // unchanged, :468-474
                    copy_block();
                    out.prev_pixels += (uint64_t)bw * (y2 - y1);
                    continue;
                }
                const int tb = bt - 1;
                if (tb & 1) {  // sub-rectangle: whole block from prev first, :375-386
                    copy_block();
                    out.prev_pixels += (uint64_t)bw * (y2 - y1);
                    const int bx2 = x2, by2 = y2;
                    x1 = ec_->sxy(0) + x16;
                    y1 = ec_->sxy(1) + y16;
                    x2 = ec_->sxy(2) + x16 + 1;
                    y2 = ec_->sxy(3) + y16 + 1;
                    if (x1 >= x2 || y1 >= y2 || x2 > bx2 || y2 > by2)
                        throw DecodeAbort{"sub-rectangle empty or outside its block/frame (not produced by any encoder)"};
                    pb.flags |= PB_SUBRECT;
                }
                pb.x1 = (uint8_t)(x1 - x16);
                pb.y1 = (uint8_t)(y1 - y16);
                pb.x2 = (uint8_t)(x2 - x16);
                pb.y2 = (uint8_t)(y2 - y16);
                if (tb & 2) {  // motion, :388-405
                    int mx, my;
                    if (use_bool_ && ec_->flag()) { mx = lastmx; my = lastmy; }
                    else { mx = ec_->mx() - 256; my = ec_->my() - 256; }
                    lastmx = mx;
                    lastmy = my;
                    need_prev();
                    for (int y = y1; y < y2; ++y) {
                        const long i = (long)y * X + x1, j = (long)(y + my) * X + (x1 + mx);
                        for (int x = 0; x < x2 - x1; ++x) dst[i + x] = rdp(j + x);
                    }
                    pb.flags |= PB_MOTION;
                    pb.mx = (int16_t)mx;
                    pb.my = (int16_t)my;
                    out.prev_pixels += (uint64_t)(x2 - x1) * (y2 - y1);
                    out.motion_pixels += (uint64_t)(x2 - x1) * (y2 - y1);
                } else {  // data: run stream confined to the rectangle, :406-466
                    int x = x1, y = y1, pt = 0;
                    while (y < y2) {
                        pt = ec_->ptype(pt);
                        if (pt >= 6) throw DecodeAbort{"predictor type outside its tables"};
                        if (pt == 0) clr = literal();
                        const int cnt = ec_->run(pt);
                        progress(cnt > 0);
                        for (int c = 0; c < cnt; ++c) {
                            // the reference keeps writing below the rectangle when a run is longer
                            // than what is left of it; no encoder does that, and it cannot be
                            // reproduced block-parallel: refuse the stream
                            if (y >= y2) throw DecodeAbort{"run crosses the end of its rectangle"};
                            const long di = (long)y * X + x;
                            switch (pt) {
                                case 1: clr = x == 0 ? wrapped(di - 1, y - 1, y16) : rdd(di - 1); break;
                                case 2: clr = rdd(di - X); break;
                                case 3: need_prev(); clr = rdp(di); break;
                                case 4: {
                                    const long l = di - 1, u = di - X, ul = di - X - 1;
                                    if (l < 0 || u < 0 || ul < 0) { clr = 0; break; }  // NaN bytes
                                    const uint32_t a = (uint32_t)(x == 0 ? wrapped(l, y - 1, y16) : dst[l]), b = (uint32_t)dst[u],
                                                   cc = (uint32_t)(x == 0 ? wrapped(ul, y - 2, y16) : dst[ul]);
                                    clr = (int32_t)((((a & 0xFF) + (b & 0xFF) - (cc & 0xFF)) & 0xFF) |
                                                    ((((a >> 8) & 0xFF) + ((b >> 8) & 0xFF) - ((cc >> 8) & 0xFF)) & 0xFF) << 8 |
                                                    ((((a >> 16) & 0xFF) + ((b >> 16) & 0xFF) - ((cc >> 16) & 0xFF)) & 0xFF) << 16);
                                    break;
                                }
                                case 5: clr = x == 0 ? wrapped(di - X - 1, y - 2, y16) : rdd(di - X - 1); break;
                                default: break;
                            }
                            dst[di] = clr;  // inside the frame: the rectangle was checked above
                            if (++x >= x2) { x = x1; ++y; }
                        }
                        cx1_ = (clr & mask1) >> shift1;
                        cx_ = clr >> shiftc;
                    }
                    pb.flags |= PB_DATA;
                    out.payload.resize((out.payload.size() + 3) & ~size_t(3), 0u);   // a rectangle's literals start on a 16-byte boundary (the group kernel's loader fetches 16 bytes per lane)
                    pb.payload = (uint32_t)out.payload.size();
                    for (int yy = y1; yy < y2; ++yy)
                        for (int xx = x1; xx < x2; ++xx) out.payload.push_back((uint32_t)dst[(long)yy * X + xx]);
                    out.data_pixels += (uint64_t)(x2 - x1) * (y2 - y1);
                }
            }
        out.kind = FrameKind::Inter;
        out.adopted = true;
        out.significant = signif;
        out.stream_bytes = ec_->consumed();
        has_prev_ = true;
        cur_ ^= 1;
        for (size_t i = 0; i < stale_.size(); ++i) stale_[i] = bts_[i] > 0;   // what the other shadow frame lacks now
    } catch (const DecodeAbort& a) {
        out.reset();
        out.status = 2;
        out.error = a.why;
        std::fill(stale_.begin(), stale_.end(), 1);   // the frame being written is in an unknown state
    }
}

// (after a key frame has decoded and the shadows have swapped: shadow_[cur_ ^ 1] is the new picture, shadow_[cur_] the one before it,
// complete — it is what the inter frames of the group before read)
void HostDecoder::note_key_compare(FrameOut& out, bool had_prev) const {
    if (key_compare_row_ < 0) return;
    if (!had_prev) { out.key_differs = -1; return; }
    const size_t first = (size_t)key_compare_row_ * (size_t)g_.X, end = (size_t)g_.X * (size_t)g_.Y;
    out.key_differs = first < end && std::memcmp(shadow_[cur_ ^ 1].data() + first, shadow_[cur_].data() + first, (end - first) * sizeof(int32_t)) != 0 ? 1 : 0;
}

void HostDecoder::last_column(int32_t* out) const {
    const int32_t* pic = shadow_[cur_ ^ 1].data();            // (the decoders swap after a frame: this is the picture just decoded)
    for (int y = 0; y < g_.Y; ++y) out[y] = pic[(size_t)y * g_.X + g_.X - 1];
}

void HostDecoder::literalise_motion(FrameOut& out) const {
    if (out.kind != FrameKind::Inter || out.motion_pixels == 0) return;
    const int32_t* pic = shadow_[cur_ ^ 1].data();   // decode_p swapped: this is the frame just decoded
    const long X = g_.X;
    for (int by = 0; by < g_.nby; ++by)
        for (int bx = 0; bx < g_.nbx; ++bx) {
            PBlock& pb = out.blocks[(size_t)by * g_.nbx + bx];
            if (!(pb.flags & PB_MOTION)) continue;
            pb.flags = (uint8_t)((pb.flags & ~PB_MOTION) | PB_DATA);
            pb.mx = pb.my = 0;
            out.payload.resize((out.payload.size() + 3) & ~size_t(3), 0u);   // (16-byte boundary, as for the coded rectangles)
            pb.payload = (uint32_t)out.payload.size();
            for (int y = by * 16 + pb.y1; y < by * 16 + pb.y2; ++y)
                for (int x = bx * 16 + pb.x1; x < bx * 16 + pb.x2; ++x) out.payload.push_back((uint32_t)pic[(long)y * X + x]);
        }
}


// ---- groups of pictures side by side ---------------------------------------------------------------------------------
namespace {
void decode_one(HostDecoder& d, const HostFrame& f, FrameOut& out, bool literalise, DstColumns* cols) {
    if (cols && !f.key && (f.dst || f.dst_host)) d.set_destination_column([cols, &f] { return cols->before(f); });
    else d.set_destination_column(nullptr);
    if (f.key) d.decode_i(f.src, f.n, out);
    else d.decode_p(f.src, f.n, out);
    d.set_destination_column(nullptr);
    if (cols && (f.dst || f.dst_host)) cols->after(f, d, out);
    const Geometry& g = d.geo();
    if (literalise && out.kind == FrameKind::Inter && out.motion_pixels * 4 <= (uint64_t)g.X * g.Y) {
        d.literalise_motion(out);
        out.literalised = true;
    }
}
}  // namespace
void decode_single(HostDecoder& d, const HostFrame& f, FrameOut& out, bool literalise, DstColumns* cols) { decode_one(d, f, out, literalise, cols); }
bool starts_group(const HostFrame& f) { return f.key && f.n > 0 && (f.src[0] & 0xF) == 2; }   // a CODED key frame (flat ones renew nothing)

void decode_frames(HostDecoder& stream_decoder, std::vector<std::unique_ptr<HostDecoder>>& spare, const HostFrame* frames,
                   int count, FrameOut* outs, int threads, bool literalise, DstColumns* cols) {
    // groups: [0, first coded key frame) continues whatever the stream decoder holds; then one group per coded key frame
    std::vector<int> begin;
    begin.push_back(0);
    for (int i = 1; i < count; ++i)
        if (starts_group(frames[i])) begin.push_back(i);
    begin.push_back(count);
    const int ngroups = (int)begin.size() - 1;
    // which entropy coder the later groups must start with: the stream's, or the one its first coded key frame picks
    int version = stream_decoder.pinned_version();
    for (int i = 0; i < count && version == 0; ++i)
        if (starts_group(frames[i])) version = (frames[i].src[0] >> 4) + 1;
    const bool side_by_side = threads > 1 && ngroups > 1 && version >= 2 && version <= 4;
    if (!side_by_side) {
        for (int i = 0; i < count; ++i) decode_one(stream_decoder, frames[i], outs[i], literalise, cols);
        return;
    }
    const Geometry g = stream_decoder.geo();
    while ((int)spare.size() < ngroups - 1) spare.push_back(std::make_unique<HostDecoder>(g.X, g.Y, g.bpp));
    auto decoder_of = [&](int grp) -> HostDecoder& { return grp == 0 ? stream_decoder : *spare[grp - 1]; };
    std::vector<char> usable(ngroups, 1);
    for (int grp = 1; grp < ngroups; ++grp) {
        const HostDecoder& d = decoder_of(grp);
        if (d.pinned_version() != 0 && d.pinned_version() != version) spare[grp - 1] = std::make_unique<HostDecoder>(g.X, g.Y, g.bpp);   // it served another coder
        HostDecoder& e = decoder_of(grp);
        e.adopt_settings(stream_decoder);
        usable[grp] = e.pin_version(version) ? 1 : 0;
    }
    auto run_group = [&](int grp) {
        if (!usable[grp]) return;
        HostDecoder& d = decoder_of(grp);
        try {
            for (int i = begin[grp]; i < begin[grp + 1]; ++i) decode_one(d, frames[i], outs[i], literalise, cols);
        } catch (...) {            // (out of memory on a thread of its own must not end the process: the group is decoded again, in order)
            if (grp > 0) usable[grp] = 0;
            else throw;
        }
    };
    {   // groups are dealt to the threads round robin; the calling thread takes its share
        const int nthreads = threads < ngroups ? threads : ngroups;
        // (whatever is thrown on this thread — group 0's failure, or a thread that cannot be started — is kept until every
        // thread that did start has been joined: unwinding past a joinable std::thread would end the process)
        std::vector<std::thread> pool;
        std::exception_ptr failed;
        int started = 1;
        try {
            for (int t = 1; t < nthreads; ++t, ++started)
                pool.emplace_back([&, t] { for (int grp = t; grp < ngroups; grp += nthreads) run_group(grp); });
        } catch (...) {
            for (int t = started; t < nthreads; ++t)           // the shares of the threads that never started: not decoded side by side
                for (int grp = t; grp < ngroups; grp += nthreads) usable[grp] = 0;
        }
        try {
            for (int grp = 0; grp < ngroups; grp += nthreads) run_group(grp);
        } catch (...) {
            failed = std::current_exception();
        }
        for (auto& th : pool) th.join();
        if (failed) std::rethrow_exception(failed);
    }
    // A group stands if its key frame decoded (then nothing older shows through it).  The first that does not — and
    // everything behind it — is decoded again in order, by the decoder holding the state in front of it.
    int last_good = 0;
    for (int grp = 1; grp < ngroups; ++grp) {
        if (!usable[grp] || outs[begin[grp]].status != 0) break;
        last_good = grp;
    }
    if (last_good > 0) std::swap(stream_decoder, *spare[last_good - 1]);   // the stream goes on from the last group that stands
    for (int i = begin[last_good + 1]; i < count; ++i) decode_one(stream_decoder, frames[i], outs[i], literalise, cols);
}

}  // namespace jsp::sp
