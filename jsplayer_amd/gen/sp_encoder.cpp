// ScreenPressor stream ENCODER — input generator for tests and benchmarks (the reference ships no
// encoder and no sample media).  Lossless: a decoder fed with the output reproduces
// the given frames exactly.
//
// Bitstream as the reference decodes it: ScreenPressor.hx:117-295 (I), :302-484 (P), SURVEY.md
// Appendix B/C.  The adaptive models are the decoder's own objects (csrc/sp_models.h) driven
// through `locate()` + `take()`, so the two sides cannot disagree about an interval; only the
// bit-level coders are written here:
//   v2    carry-propagating range encoder matching RangeCoder.hx (32-bit range, byte renormalisation
//         below 2^24, first byte = the carry cache the decoder skips, RangeCoder.hx:29-33)
//   v3/v4 byte-wise rANS, 12-bit probabilities, state in [2^23, 2^31), encoded backwards per block of
//         131072 symbols with raw bytes interleaved in decode order (ANS.hx:5-49, EntroCoders.hx:249-253)
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../csrc/sp_models.h"

namespace {
using namespace jsp::sp;
std::atomic<uint64_t> g_census[8];   // how often colour contexts entered each model stage (all encoders of the process)

// ---------------------------------------------------------------- bit-level coders ------------
class RangeEncoder {
public:
    void begin(std::vector<uint8_t>* out) { out_ = out; low_ = 0; range_ = 0xFFFFFFFFu; cache_ = 0; cache_size_ = 1; }
    void encode(uint32_t cum, uint32_t freq, uint32_t total) {
        range_ /= total;
        low_ += (uint64_t)cum * range_;
        range_ *= freq;
        while (range_ < (1u << 24)) { range_ <<= 8; shift_low(); }
    }
    void finish() { for (int i = 0; i < 5; ++i) shift_low(); }
private:
    void shift_low() {
        if ((uint32_t)low_ < 0xFF000000u || (low_ >> 32) != 0) {
            const uint8_t carry = (uint8_t)(low_ >> 32);
            uint8_t temp = cache_;
            do { out_->push_back((uint8_t)(temp + carry)); temp = 0xFF; } while (--cache_size_);
            cache_ = (uint8_t)((low_ >> 24) & 0xFF);
        }
        ++cache_size_;
        low_ = (low_ & 0x00FFFFFFu) << 8;
    }
    std::vector<uint8_t>* out_ = nullptr;
    uint64_t low_ = 0;
    uint32_t range_ = 0;
    uint8_t cache_ = 0;
    uint64_t cache_size_ = 1;
};

struct RansEvent { uint16_t cum, freq; uint8_t raw; };  // freq == 0: raw byte

void rans_flush(const std::vector<RansEvent>& ev, std::vector<uint8_t>& out) {
    constexpr size_t B = 131072;
    std::vector<uint8_t> buf;
    for (size_t b0 = 0; b0 < ev.size(); b0 += B) {
        const size_t b1 = std::min(ev.size(), b0 + B);
        buf.clear();
        uint32_t x = 1u << 23;
        for (size_t i = b1; i-- > b0;) {
            const RansEvent& e = ev[i];
            if (e.freq == 0) { buf.push_back(e.raw); continue; }
            const uint32_t f = e.freq == 0xFFFF ? 4096u : e.freq;
            const uint64_t x_max = ((uint64_t)(1u << 23) >> 12 << 8) * f;
            while (x >= x_max) { buf.push_back((uint8_t)(x & 0xFF)); x >>= 8; }
            x = ((x / f) << 12) + (x % f) + e.cum;
        }
        buf.push_back((uint8_t)(x >> 24)); buf.push_back((uint8_t)(x >> 16));
        buf.push_back((uint8_t)(x >> 8)); buf.push_back((uint8_t)x);
        out.insert(out.end(), buf.rbegin(), buf.rend());
    }
    if (!ev.empty() && ev.size() % B == 0) {  // the decoder re-reads a state after a full block
        const uint8_t st[4] = {0, 0, 0x80, 0};
        out.insert(out.end(), st, st + 4);
    }
}

// ---------------------------------------------------------------- symbol-level encoders -------
class SymbolEncoder {  // mirror of EntroCoders.hx:8-24 on the encoding side
public:
    virtual ~SymbolEncoder() = default;
    virtual void renewI() = 0;
    virtual void begin(std::vector<uint8_t>* out) = 0;
    virtual void finish() = 0;
    virtual void clr(int ctx, int c) = 0;
    virtual void run(int ptype, int n) = 0;
    virtual void ptype(int prev, int pt) = 0;
    virtual void xx(int v) = 0;
    virtual void bt(int v) = 0;
    virtual void bn(int v) = 0;
    virtual void sxy(int k, int v) = 0;
    virtual void mx(int v) = 0;
    virtual void my(int v) = 0;
    virtual bool has_bool() const = 0;
    virtual void flag(bool f) = 0;
    virtual bool rc_16bpp_constants() const = 0;
};

class RangeSymbolEncoder final : public SymbolEncoder {
public:
    RangeSymbolEncoder() {
        for (auto& t : ntab_) t.init(256, 400);
        for (auto& t : ptab_) t.init(6, 1000);
        xx_.init(256, 1); bn_.init(256, 20); bt_.init(5, 10);
        for (auto& t : sxy_) t.init(16, 100);
        for (auto& t : mv_) t.init(512, 100);
    }
    void renewI() override {
        clr_.reset_changed();
        for (auto& t : ntab_) t.reset();
        for (auto& t : ptab_) t.reset();
        xx_.reset(); bn_.reset(); bt_.reset();
        for (auto& t : sxy_) t.reset();
        for (auto& t : mv_) t.reset();
    }
    void begin(std::vector<uint8_t>* out) override { rc_.begin(out); }
    void finish() override { rc_.finish(); }
    void clr(int ctx, int c) override {
        const uint32_t tot = clr_.total(ctx);
        const Interval iv = clr_.take(ctx, clr_.cum_of(ctx, c));
        check(iv.sym == c);
        rc_.encode(iv.cum, iv.freq, tot);
    }
    void run(int ptype, int n) override { plain(ntab_[ptype], n); }
    void ptype(int prev, int pt) override { plain(ptab_[prev], pt); }
    void xx(int v) override { plain(xx_, v); }
    void bt(int v) override { plain(bt_, v); }
    void bn(int v) override { plain(bn_, v); }
    void sxy(int k, int v) override { plain(sxy_[k], v); }
    void mx(int v) override { plain(mv_[0], v); }
    void my(int v) override { plain(mv_[1], v); }
    bool has_bool() const override { return false; }
    void flag(bool) override {}
    bool rc_16bpp_constants() const override { return true; }
private:
    static void check(bool ok) { if (!ok) throw std::logic_error("encoder/model disagreement (range)"); }
    void plain(RcTable& t, int c) {
        const uint32_t tot = t.total();
        const Interval iv = t.take(t.cum_of(c));
        check(iv.sym == c);
        rc_.encode(iv.cum, iv.freq, tot);
    }
    RangeEncoder rc_;
    RcColourTables clr_;
    RcTable ntab_[6], ptab_[6], xx_, bn_, bt_, sxy_[4], mv_[2];
};

class RansSymbolEncoder final : public SymbolEncoder {
public:
    explicit RansSymbolEncoder(int f0) : clr_(f0) {
        for (auto& m : ntab_) m.init(256);
        for (auto& m : ptab_) m.init(6);
        xx_.init(256); bn_.init(256); bt_.init(5);
        for (auto& m : sxy_) m.init(16);
        for (auto& m : mv_) m.init(512);
    }
    void renewI() override {
        clr_.renew();
        for (auto& m : ntab_) m.renew();
        for (auto& m : ptab_) m.renew();
        xx_.renew(); bn_.renew(); bt_.renew();
        for (auto& m : sxy_) m.renew();
        for (auto& m : mv_) m.renew();
    }
    void begin(std::vector<uint8_t>* out) override { out_ = out; ev_.clear(); }
    void finish() override { rans_flush(ev_, *out_); }
    void clr(int ctx, int c) override {
        if (clr_.coded(ctx)) {
            const Interval iv = clr_.take(ctx, clr_.locate(ctx, c));
            if (iv.sym != c) throw std::logic_error("encoder/model disagreement (colour context)");
            push(iv);
        } else {
            ev_.push_back({0, 0, (uint8_t)c});
            clr_.learn(ctx, c);
        }
        for (int k = 0; k < 8; ++k) { g_census[k] += clr_.census[k]; clr_.census[k] = 0; }
    }
    void run(int ptype, int n) override { fixed(ntab_[ptype], n); }
    void ptype(int prev, int pt) override { fixed(ptab_[prev], pt); }
    void xx(int v) override { fixed(xx_, v); }
    void bt(int v) override { fixed(bt_, v); }
    void bn(int v) override { fixed(bn_, v); }
    void sxy(int k, int v) override { fixed(sxy_[k], v); }
    void mx(int v) override { fixed(mv_[0], v); }
    void my(int v) override { fixed(mv_[1], v); }
    bool has_bool() const override { return true; }
    void flag(bool f) override { ev_.push_back({(uint16_t)(f ? 2048 : 0), 2048, 0}); }
    bool rc_16bpp_constants() const override { return false; }
private:
    void push(const Interval& iv) {
        if (iv.freq == 0 || iv.freq > 4096) throw std::logic_error("bad interval");
        // The reference's 40-entry table built from a 64-symbol list (Cx6.createFrom2, ANS.hx:494-506) assigns
        // 256 - d + (d + 1) * f0 slots: with f0 = 64 (version 3) and d >= 60 distinct symbols before the first
        // repeat that is more than the 4096 there are, and the symbols pushed past the end cannot be coded by
        // any encoder (the decoder only ever sees slots 0..4095).  Version 4 halved f0.  Refuse such input.
        if (iv.cum + iv.freq > 4096)
            throw std::logic_error("symbol interval outside the 12-bit code space (version-3 model overflow: a colour context "
                                   "saw 60 or more distinct values before its first repeat); this content cannot be coded as v3");
        ev_.push_back({(uint16_t)iv.cum, (uint16_t)(iv.freq == 4096 ? 0xFFFF : iv.freq), 0});
    }
    void fixed(FixedModel& m, int c) {
        const Interval iv = m.take(m.locate(c));
        if (iv.sym != c) throw std::logic_error("encoder/model disagreement (fixed model)");
        push(iv);
    }
    std::vector<uint8_t>* out_ = nullptr;
    std::vector<RansEvent> ev_;
    ColourModels clr_;
    FixedModel ntab_[6], ptab_[6], xx_, bn_, bt_, sxy_[4], mv_[2];
};

// ---------------------------------------------------------------- frame encoder ----------------
inline uint32_t grad(uint32_t l, uint32_t u, uint32_t ul) {  // per byte 0..2: left + above - aboveleft
    const uint32_t r = ((l & 0xFF) + (u & 0xFF) - (ul & 0xFF)) & 0xFF;
    const uint32_t g = (((l >> 8) & 0xFF) + ((u >> 8) & 0xFF) - ((ul >> 8) & 0xFF)) & 0xFF;
    const uint32_t b = (((l >> 16) & 0xFF) + ((u >> 16) & 0xFF) - ((ul >> 16) & 0xFF)) & 0xFF;
    return (b << 16) | (g << 8) | r;
}

class FrameEncoder {
public:
    FrameEncoder(int w, int h, int bpp, int version) : X(w), Y(h), bpp_(bpp), version_(version) {
        if (version < 2 || version > 4) throw std::invalid_argument("version must be 2, 3 or 4");
        if (h < 2) throw std::invalid_argument("height must be at least 2");
        cxshift_ = (bpp == 16 && version == 2) ? 0 : 2;
        nbx_ = (w + 15) / 16;
        nby_ = (h + 15) / 16;
        prev_.assign((size_t)w * h, 0);
    }
    void encode_flat(uint32_t colour, std::vector<uint8_t>& out) {
        if (!se_) throw std::logic_error("a flat key frame cannot be the first frame (the reference crashes on it)");
        if (!last_flat_) se_->renewI();  // RenewI runs the model reset unless the previous I was flat too
        out.push_back((uint8_t)(((version_ - 1) << 4) | 1));
        uint32_t c;
        if (bpp_ == 16) {  // the decoder reads src[0] + src[1]*256: the header byte is the low byte
            const uint8_t hi = (uint8_t)(colour >> 8);
            out.push_back(hi);
            const int v = out[0] + hi * 256;
            c = (uint32_t)((((v >> 10) & 0x1F) << 3) << 16) + (uint32_t)((((v >> 5) & 0x1F) << 3) << 8) + (uint32_t)((v & 0x1F) << 3);
        } else {
            out.push_back((uint8_t)colour); out.push_back((uint8_t)(colour >> 8)); out.push_back((uint8_t)(colour >> 16));
            c = colour & 0xFFFFFF;
        }
        std::fill(prev_.begin(), prev_.end(), c);
        last_flat_ = true;
        have_i_ = true;
    }
    void encode_i(const uint32_t* t, std::vector<uint8_t>& out) {
        if (!se_) {
            if (version_ == 2) se_ = std::make_unique<RangeSymbolEncoder>();
            else se_ = std::make_unique<RansSymbolEncoder>(version_ == 3 ? 64 : 32);
        }
        last_flat_ = false;
        se_->renewI();
        out.push_back((uint8_t)(((version_ - 1) << 4) | 2));
        se_->begin(&out);
        cx_ = cx1_ = 0;
        const long end = (long)X * Y;
        long di = 0, k = 0;
        uint32_t clr = 0;
        while (k < X + 1) {
            clr = t[di];
            int n = 1;
            while (n < 255 && di + n < end && t[di + n] == clr) ++n;
            literal(clr);
            se_->run(0, n);
            di += n;
            k += n;
        }
        int mask1 = 0xFC00, shift1 = 4, shiftc = 18;
        if (bpp_ == 16 && se_->rc_16bpp_constants()) { mask1 = 0xFF00; shift1 = 2; shiftc = 16; }
        int pt = 0;
        while (di < end) {
            const long room = std::min<long>(255, end - di);
            int best = 0, bestn = 0;
            auto consider = [&](int type, int n) { if (n > bestn) { bestn = n; best = type; } };
            {   // 2: copy from above
                int n = 0;
                while (n < room && t[di + n] == t[di + n - X]) ++n;
                consider(2, n);
            }
            {   // 1: repeat previous pixel
                int n = 0;
                while (n < room && t[di + n] == t[di - 1]) ++n;
                consider(1, n);
            }
            {   // 5: above-left
                int n = 0;
                while (n < room && t[di + n] == t[di + n - X - 1]) ++n;
                consider(5, n);
            }
            {   // 4: gradient
                int n = 0;
                while (n < room && t[di + n] == grad(t[di + n - 1], t[di + n - X], t[di + n - X - 1])) ++n;
                consider(4, n);
            }
            if (bestn == 0) {
                int n = 1;
                while (n < room && t[di + n] == t[di]) ++n;
                best = 0;
                bestn = n;
            }
            se_->ptype(pt, best);
            pt = best;
            if (best == 0) literal(t[di]);
            se_->run(best, bestn);
            di += bestn;
            clr = t[di - 1];
            cx1_ = ((int)clr & mask1) >> shift1;
            cx_ = (int)clr >> shiftc;
        }
        se_->finish();
        std::memcpy(prev_.data(), t, sizeof(uint32_t) * (size_t)end);
        have_i_ = true;
    }
    // hints: per block (mx,my) or (INT16_MIN, *) for "no motion candidate"
    void encode_p(const uint32_t* t, const int16_t* hints, std::vector<uint8_t>& out) {
        if (!have_i_ || !se_) throw std::logic_error("P frame before a coded key frame");
        last_flat_ = false;
        const long end = (long)X * Y;
        const int nb = nbx_ * nby_;
        std::vector<uint8_t> bts(nb, 0);
        struct Rect { int x1, y1, x2, y2; int mx, my; };
        std::vector<Rect> rects(nb);
        int first = -1, last = -1;
        for (int by = 0; by < nby_; ++by)
            for (int bx = 0; bx < nbx_; ++bx) {
                const int bi = by * nbx_ + bx, x16 = bx * 16, y16 = by * 16;
                const int xe = std::min(x16 + 16, X), ye = std::min(y16 + 16, Y);
                int dx1 = 1 << 30, dy1 = 1 << 30, dx2 = -1, dy2 = -1;
                for (int y = y16; y < ye; ++y)
                    for (int x = x16; x < xe; ++x)
                        if (t[(long)y * X + x] != prev_[(long)y * X + x]) {
                            dx1 = std::min(dx1, x); dx2 = std::max(dx2, x);
                            dy1 = std::min(dy1, y); dy2 = std::max(dy2, y);
                        }
                if (dx2 < 0) continue;  // unchanged
                const bool sub = dx1 > x16 || dy1 > y16 || dx2 < xe - 1 || dy2 < ye - 1;
                Rect r{sub ? dx1 : x16, sub ? dy1 : y16, sub ? dx2 + 1 : xe, sub ? dy2 + 1 : ye, 0, 0};
                bool motion = false;
                if (hints && hints[2 * bi] != INT16_MIN) {
                    const int mx = hints[2 * bi], my = hints[2 * bi + 1];
                    motion = mx >= -256 && mx < 256 && my >= -256 && my < 256;
                    for (int y = r.y1; y < r.y2 && motion; ++y)
                        for (int x = r.x1; x < r.x2; ++x) {
                            const long j = (long)(y + my) * X + (x + mx);
                            const uint32_t s = (j >= 0 && j < end) ? prev_[j] : 0u;
                            if (s != t[(long)y * X + x]) { motion = false; break; }
                        }
                    r.mx = mx; r.my = my;
                }
                bts[bi] = (uint8_t)(1 + (sub ? 1 : 0) + (motion ? 2 : 0));
                rects[bi] = r;
                if (first < 0) first = bi;
                last = bi;
            }
        if (first < 0) { out.push_back(0); return; }  // "no changes" (ScreenPressor.hx:311-313)
        out.push_back(1);
        se_->begin(&out);
        se_->xx(first & 0xFF); se_->xx(first >> 8);
        se_->xx(last & 0xFF); se_->xx(last >> 8);
        for (int x = first; x <= last;) {
            int n = 1;
            while (n < 255 && x + n <= last && bts[x + n] == bts[x]) ++n;
            se_->bt(bts[x]);
            se_->bn(n);
            x += n;
        }
        int mask1 = 0xFC00, shift1 = 4, shiftc = 18;
        if (se_->rc_16bpp_constants() && bpp_ == 16) { mask1 = 0xFF00; shift1 = 2; shiftc = 16; }
        cx_ = cx1_ = 0;
        int lastmx = 0, lastmy = 0;
        for (int by = 0; by < nby_; ++by)
            for (int bx = 0; bx < nbx_; ++bx) {
                const int bi = by * nbx_ + bx, x16 = bx * 16, y16 = by * 16;
                if (!bts[bi]) continue;
                const int tb = bts[bi] - 1;
                const Rect& r = rects[bi];
                if (tb & 1) {
                    se_->sxy(0, r.x1 - x16); se_->sxy(1, r.y1 - y16);
                    se_->sxy(2, r.x2 - 1 - x16); se_->sxy(3, r.y2 - 1 - y16);
                }
                if (tb & 2) {
                    const bool same = r.mx == lastmx && r.my == lastmy;
                    if (se_->has_bool()) se_->flag(same);
                    if (!(se_->has_bool() && same)) { se_->mx(r.mx + 256); se_->my(r.my + 256); }
                    lastmx = r.mx; lastmy = r.my;
                    continue;
                }
                // data rectangle: raster inside the rectangle, run stream with predictor types
                const int w = r.x2 - r.x1, total = w * (r.y2 - r.y1);
                auto at = [&](int idx) -> long { return (long)(r.y1 + idx / w) * X + (r.x1 + idx % w); };
                int pos = 0, pt = 0;
                uint32_t clr = 0;
                while (pos < total) {
                    const int room = std::min(255, total - pos);
                    int best = 0, bestn = 0;
                    auto consider = [&](int type, int n) { if (n > bestn) { bestn = n; best = type; } };
                    auto scan = [&](auto&& ok) { int n = 0; while (n < room && ok(at(pos + n))) ++n; return n; };
                    // neighbours must already hold this frame's final pixels when the decoder reads
                    // them: column 0 has no usable left / above-left neighbour (it would be the end of
                    // the previous image row, decoded later), row 0 has nothing above
                    consider(3, scan([&](long i) { return t[i] == prev_[i]; }));
                    consider(2, scan([&](long i) { return i >= X && t[i] == t[i - X]; }));
                    // (tests only, set_stale: in column 0 the decoder's "left" is the last pixel of the row above — of a block
                    // this frame has not reached yet unless the row is a block row's first, i.e. whatever the destination
                    // buffer held: ScreenPressor.hx:436-444.  `stale_` says what that is, so that a stream exercising the read
                    // can be built; no real encoder emits it.)
                    consider(1, scan([&](long i) {
                        if (i % X != 0) return t[i] == t[i - 1];
                        return stale_ != nullptr && i > 0 && (i / X) % 16 != 0 && t[i] == stale_[i - 1];
                    }));
                    consider(5, scan([&](long i) { return i >= X && i % X != 0 && t[i] == t[i - X - 1]; }));
                    consider(4, scan([&](long i) { return i >= X && i % X != 0 && t[i] == grad(t[i - 1], t[i - X], t[i - X - 1]); }));
                    if (bestn == 0) {
                        const uint32_t c0 = t[at(pos)];
                        best = 0;
                        bestn = scan([&](long i) { return t[i] == c0; });
                    }
                    se_->ptype(pt, best);
                    pt = best;
                    if (best == 0) literal(t[at(pos)]);
                    se_->run(best, bestn);
                    pos += bestn;
                    clr = t[at(pos - 1)];
                    cx1_ = ((int)clr & mask1) >> shift1;
                    cx_ = (int)clr >> shiftc;
                }
            }
        se_->finish();
        std::memcpy(prev_.data(), t, sizeof(uint32_t) * (size_t)end);
    }
    const std::vector<uint32_t>& prev() const { return prev_; }
    void set_stale(const uint32_t* picture) {   // what the decoder's destination buffer holds before the next inter frame (null: unknown)
        if (!picture) { stale_store_.clear(); stale_ = nullptr; return; }
        stale_store_.assign(picture, picture + prev_.size());
        stale_ = stale_store_.data();
    }

private:
    std::vector<uint32_t> stale_store_;
    const uint32_t* stale_ = nullptr;
    void literal(uint32_t clr) {  // three components with the decoder's context chain
        const int comp[3] = {(int)(clr & 0xFF), (int)((clr >> 8) & 0xFF), (int)((clr >> 16) & 0xFF)};
        for (int ch = 0; ch < 3; ++ch) {
            const int ctx = ch * 4096 + cx_ + cx1_;
            if (ctx >= 3 * 4096) throw std::logic_error("colour context out of range (16bpp frames need 5-bit components)");
            se_->clr(ctx, comp[ch]);
            cx1_ = (cx_ << 6) & 0xFC0;
            cx_ = comp[ch] >> cxshift_;
        }
    }
    int X, Y, bpp_, version_, cxshift_, nbx_, nby_;
    int cx_ = 0, cx1_ = 0;
    bool last_flat_ = false, have_i_ = false;
    std::unique_ptr<SymbolEncoder> se_;
    std::vector<uint32_t> prev_;
};

struct Handle {
    FrameEncoder enc;
    std::vector<uint8_t> out;
    std::string err;
    Handle(int w, int h, int bpp, int v) : enc(w, h, bpp, v) {}
};

template <class F>
long guarded(Handle* h, uint8_t* out, size_t cap, F&& f) {
    try {
        h->out.clear();
        f();
        if (h->out.size() > cap) return -(long)h->out.size();
        std::memcpy(out, h->out.data(), h->out.size());
        return (long)h->out.size();
    } catch (const std::exception& e) {
        h->err = e.what();
        return -1;
    }
}
}  // namespace

extern "C" {
void* jspgen_sp_create(int w, int h, int bpp, int version) {
    try { return new Handle(w, h, bpp, version); } catch (...) { return nullptr; }
}
void jspgen_sp_destroy(void* p) { delete (Handle*)p; }
const char* jspgen_sp_error(void* p) { return ((Handle*)p)->err.c_str(); }
// Each call returns the number of bytes written, -1 on error (jspgen_sp_error), or -(needed size)
// when `cap` is too small (the encoder state has advanced: size `out` generously).
long jspgen_sp_encode_i(void* p, const uint32_t* frame, uint8_t* out, size_t cap) {
    auto* h = (Handle*)p;
    return guarded(h, out, cap, [&] { h->enc.encode_i(frame, h->out); });
}
long jspgen_sp_encode_flat(void* p, uint32_t colour, uint8_t* out, size_t cap) {
    auto* h = (Handle*)p;
    return guarded(h, out, cap, [&] { h->enc.encode_flat(colour, h->out); });
}
long jspgen_sp_encode_p(void* p, const uint32_t* frame, const int16_t* hints, uint8_t* out, size_t cap) {
    auto* h = (Handle*)p;
    return guarded(h, out, cap, [&] { h->enc.encode_p(frame, hints, h->out); });
}
// How often colour contexts entered each model stage since the library was loaded
// (index = ColourModels::Stage): test coverage evidence.
void jspgen_stage_census(uint64_t* out) { for (int k = 0; k < 8; ++k) out[k] = g_census[k].load(); }
// The frame a decoder holds after the last encoded frame (for flat frames the colour is derived
// from the emitted bytes).
void jspgen_sp_set_stale(void* p, const uint32_t* picture) { ((Handle*)p)->enc.set_stale(picture); }
void jspgen_sp_current(void* p, uint32_t* out) {
    auto* h = (Handle*)p;
    std::memcpy(out, h->enc.prev().data(), h->enc.prev().size() * sizeof(uint32_t));
}
}
