// ScreenPressor entropy decoders: bit-level state machines over the models of sp_models.h.
#include "sp_entropy.h"

namespace jsp::sp {
namespace {

// ---------------------------------------------------------------- v2: range decoder ----------
// RangeCoder.hx keeps `range`/`code` in JS doubles holding exact integers below 2^32 on valid
// streams; integers are used here.  A read past the end makes the reference's `code` NaN, after
// which Std.int(code / range) is 0 for ever: `poisoned_` reproduces that.
class RangeBits {
public:
    void begin(const uint8_t* src, size_t n, size_t pos0) {
        src_ = src; n_ = n;
        range_ = 0xFFFFFFFFu;
        code_ = 0;
        poisoned_ = false;
        for (size_t k = 1; k <= 4; ++k) code_ = code_ * 256 + byte_at(pos0 + k);
        pos_ = pos0 + 5;
    }
    uint32_t slot(uint32_t total) {  // get_freq, RangeCoder.hx:45-49
        range_ = total ? range_ / total : 0;
        if (poisoned_ || range_ == 0) return 0;
        const int32_t v = (int32_t)(uint32_t)(code_ / range_);  // Std.int wraps to int32
        return v < 0 ? 0u : (uint32_t)v;
    }
    void consume(uint32_t cum, uint32_t freq) {  // decode, RangeCoder.hx:36-43
        code_ -= (uint64_t)cum * range_;
        uint64_t r = (uint64_t)range_ * freq;
        int spins = 0;
        while (r < (1u << 24)) {
            if (++spins > 8) throw DecodeAbort{"range decoder cannot renormalise (zero range)"};
            code_ = code_ * 256 + byte_at(pos_++);
            r *= 256;
        }
        range_ = (uint32_t)r;  // < 2^32 on valid streams (freq <= total)
    }
    size_t pos() const { return pos_ < n_ ? pos_ : n_; }
private:
    uint32_t byte_at(size_t i) { if (i < n_) return src_[i]; poisoned_ = true; return 0; }
    const uint8_t* src_ = nullptr;
    size_t n_ = 0, pos_ = 0;
    uint64_t code_ = 0;
    uint32_t range_ = 0;
    bool poisoned_ = false;
};


// the shared shape of EntropyDecoder::literal(): three components, each in the context the one before left behind
template <class Coder>
inline int64_t literal_of(Coder& self, int& cx, int& cx1, int cxshift) {
    int comp[3];
#pragma GCC unroll 3
    for (int k = 0; k < 3; ++k) {
        const int ctx = k * 4096 + cx + cx1;
        if (ctx < 0 || ctx >= 3 * 4096) return -1;
        const int v = self.clr_inline(ctx);
        cx1 = (cx << 6) & 0xFC0;
        cx = v < 0 ? 0 : v >> cxshift;
        comp[k] = v;
    }
    if (comp[0] < 0) return 0;  // (b<<16)+(g<<8)+undefined is NaN, stored as 0
    return (int64_t)(((uint32_t)(comp[2] < 0 ? 0 : comp[2]) << 16) + ((uint32_t)(comp[1] < 0 ? 0 : comp[1]) << 8) + (uint32_t)comp[0]);
}

class RangeDecoder final : public EntropyDecoder {
public:
    RangeDecoder() {
        for (auto& t : ntab_) t.init(256, 400);      // SC_NSTEP
        for (auto& t : ptab_) t.init(6, 1000);       // SC_UNSTEP
        xx_.init(256, 1);                            // SC_XXSTEP
        bn_.init(256, 20);                           // SC_BTNSTEP
        bt_.init(5, 10);                             // SC_BTSTEP
        for (auto& t : sxy_) t.init(16, 100);        // SC_SXYSTEP
        for (auto& t : mv_) t.init(512, 100);        // SC_MSTEP
    }
    void renewI() override {
        clr_.reset_changed();
        for (auto& t : ntab_) t.reset();
        for (auto& t : ptab_) t.reset();
        xx_.reset(); bn_.reset(); bt_.reset();
        for (auto& t : sxy_) t.reset();
        for (auto& t : mv_) t.reset();
    }
    void begin(const uint8_t* src, size_t n, size_t pos0) override { rc_.begin(src, n, pos0); }
    int clr_inline(int ctx) {
        const Interval iv = clr_.take(ctx, rc_.slot(clr_.total(ctx)));
        rc_.consume(iv.cum, iv.freq);
        return iv.sym;
    }
    int clr(int ctx) override { return clr_inline(ctx); }
    int64_t literal(int& cx, int& cx1, int cxshift) override { return literal_of(*this, cx, cx1, cxshift); }
    int run(int ptype) override { return plain(ntab_[ptype]); }
    int ptype(int prev) override { return plain(ptab_[prev]); }
    int xx() override { return plain(xx_); }
    int bt() override { return plain(bt_); }
    int bn() override { return plain(bn_); }
    int sxy(int k) override { return plain(sxy_[k]); }
    int mx() override { return plain(mv_[0]); }
    int my() override { return plain(mv_[1]); }
    bool has_bool() const override { return false; }
    bool flag() override { return false; }
    bool rc_16bpp_constants() const override { return true; }
    size_t consumed() const override { return rc_.pos(); }
private:
    int plain(RcTable& t) {
        const Interval iv = t.take(rc_.slot(t.total()));
        rc_.consume(iv.cum, iv.freq);
        return iv.sym;
    }
    RangeBits rc_;
    RcColourTables clr_;
    RcTable ntab_[6], ptab_[6], xx_, bn_, bt_, sxy_[4], mv_[2];
};

// ---------------------------------------------------------------- v3/v4: rANS ----------------
class RansBits {  // ANS.hx:5-49 — 32-bit state, 12-bit probabilities, byte renormalisation at 2^23
public:
    void begin(const uint8_t* src, size_t n, size_t at) { src_ = src; n_ = n; load(at); }
    void reload() { load(pos_); }
    int slot() const { return x_ & (kProbScale - 1); }
    void advance(uint32_t cum, uint32_t freq) {
        int64_t x = (int64_t)freq * (x_ >> kProbBits) + (x_ & (kProbScale - 1)) - (int64_t)cum;
        int spins = 0;
        while (x < (1 << 23)) {
            if (++spins > 64) throw DecodeAbort{"rANS state cannot renormalise"};
            const int32_t lo = (int32_t)(uint32_t)x;           // the shift works on int32
            x = (int32_t)(((uint32_t)lo << 8) | byte());
        }
        x_ = (int32_t)(uint32_t)x;
    }
    int raw() { if (pos_ < n_) return src_[pos_++]; ++pos_; return -1; }
    size_t pos() const { return pos_ < n_ ? pos_ : n_; }
private:
    uint32_t byte() { return pos_ < n_ ? src_[pos_++] : (++pos_, 0u); }  // `undefined | x` is x
    void load(size_t at) {
        auto b = [&](size_t i) -> uint32_t { return i < n_ ? src_[i] : 0u; };
        x_ = (int32_t)(b(at) | b(at + 1) << 8 | b(at + 2) << 16 | b(at + 3) << 24);
        pos_ = at + 4;
    }
    const uint8_t* src_ = nullptr;
    size_t n_ = 0, pos_ = 0;
    int32_t x_ = 0;
};

class RansDecoder final : public EntropyDecoder {
public:
    explicit RansDecoder(int f0) : clr_(f0) {
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
    void begin(const uint8_t* src, size_t n, size_t pos0) override { bits_.begin(src, n, pos0); ndec_ = 0; }
    int clr_inline(int ctx) {
        int c;
        if (clr_.coded(ctx)) {
            const Interval iv = clr_.take(ctx, bits_.slot());
            bits_.advance(iv.cum, iv.freq);
            c = iv.sym;
        } else {
            c = bits_.raw();
            clr_.learn(ctx, c);
        }
        tick();
        return c;
    }
    int clr(int ctx) override { return clr_inline(ctx); }
    int64_t literal(int& cx, int& cx1, int cxshift) override { return literal_of(*this, cx, cx1, cxshift); }
    int run(int ptype) override { return fixed(ntab_[ptype]); }
    int ptype(int prev) override { return fixed(ptab_[prev]); }
    int xx() override { return fixed(xx_); }
    int bt() override { return fixed(bt_); }
    int bn() override { return fixed(bn_); }
    int sxy(int k) override { return fixed(sxy_[k]); }
    int mx() override { return fixed(mv_[0]); }
    int my() override { return fixed(mv_[1]); }
    bool has_bool() const override { return true; }
    bool flag() override {
        const bool f = bits_.slot() >= kProbScale / 2;
        bits_.advance(f ? kProbScale / 2 : 0, kProbScale / 2);
        tick();
        return f;
    }
    bool rc_16bpp_constants() const override { return false; }
    size_t consumed() const override { return bits_.pos(); }
private:
    void tick() { if (++ndec_ == 131072) { bits_.reload(); ndec_ = 0; } }  // Rans.B, EntroCoders.hx:249-253
    int fixed(FixedModel& m) {
        const Interval iv = m.take(bits_.slot());
        bits_.advance(iv.cum, iv.freq);
        tick();
        return iv.sym;
    }
    RansBits bits_;
    int ndec_ = 0;
    ColourModels clr_;
    FixedModel ntab_[6], ptab_[6], xx_, bn_, bt_, sxy_[4], mv_[2];
};

}  // namespace

std::unique_ptr<EntropyDecoder> make_range_decoder() { return std::make_unique<RangeDecoder>(); }
std::unique_ptr<EntropyDecoder> make_rans_decoder(int f0) { return std::make_unique<RansDecoder>(f0); }

}  // namespace jsp::sp
