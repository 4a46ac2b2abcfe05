// ScreenPressor adaptive models (host side), independent of the bit-level coder that drives them.
//
// Every model answers one question — "which symbol owns slot `value` of the current code space,
// and what is its interval?" — and then adapts, exactly as the reference's decoders do:
//   v2  range-coder tables : RangeCoder.hx:51-130 (DecodeVal / DecodeValUni), EntroCoders.hx:81-130
//   v3/4 rANS models       : ANS.hx:54-145 (FixedSizeRansCtx), :155-392 (Cx1..Cx5), :394-704 (Cx6),
//                            :706-772 (Cx7), :785-860 (Context)
// The stream encoder used to synthesise test/bench input drives the same objects through
// `locate(symbol)` (a slot inside the symbol's current interval) followed by the decoder's own
// lookup, so encoder and decoder cannot drift apart.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace jsp::sp {

struct Interval {
    int sym;
    uint32_t cum, freq;
};

// ------------------------------------------------------------------ v2 tables ----------------
// Plain table: `n` counts followed by their total.  (RangeCoder.hx:51-80)
struct RcTable {
    std::vector<uint32_t> cnt;  // n + 1
    int n = 0;
    uint32_t step = 0;
    void init(int nsym, uint32_t st) { n = nsym; step = st; cnt.assign(nsym + 1, 0); }
    void reset() { std::fill(cnt.begin(), cnt.begin() + n, 1u); cnt[n] = (uint32_t)n; }
    uint32_t total() const { return cnt[n]; }
    uint32_t cum_of(int c) const { uint32_t s = 0; for (int i = 0; i < c; ++i) s += cnt[i]; return s; }
    // symbol owning `value`, then the adaptive update.  No symbol owns value >= total: the scan
    // then ends at c == n with the last count read, as the reference's loop does.
    Interval take(uint32_t value) {
        uint32_t cum = 0, f = 0;
        int c = 0;
        for (; c < n; ++c) {
            f = cnt[c];
            if (value >= cum + f) cum += f; else break;
        }
        uint32_t tot = cnt[n];
        cnt[c] = f + step;
        tot += step;
        if (tot > 65536u) {
            tot = 0;
            for (int i = 0; i < n; ++i) { cnt[i] = (cnt[i] >> 1) + 1; tot += cnt[i]; }
        }
        cnt[n] = tot;
        return {c, cum, f};
    }
};

// Colour tables: 3*4096 rows of 273 words: [0..15] sums of 16-symbol groups, [16] total,
// [17..272] symbol counts.  (RangeCoder.hx:82-130, EntroCoders.hx:51,81-92)
struct RcColourTables {
    static constexpr int ROW = 273, ROWS = 3 * 4096;
    static constexpr uint32_t STEP = 400;
    std::vector<uint32_t> w;
    RcColourTables() : w((size_t)ROW * ROWS, 0u) {}
    void reset_changed() {  // only rows whose total moved away from 256 are rewritten
        for (int r = 0; r < ROWS; ++r) {
            uint32_t* p = &w[(size_t)r * ROW];
            if (p[16] != 256) {
                for (int i = 0; i < 16; ++i) p[i] = 16;
                p[16] = 256;
                for (int i = 0; i < 256; ++i) p[17 + i] = 1;
            }
        }
    }
    uint32_t total(int row) const { return w[(size_t)row * ROW + 16]; }
    uint32_t cum_of(int row, int c) const {
        const uint32_t* p = &w[(size_t)row * ROW];
        uint32_t s = 0;
        for (int g = 0; g < (c >> 4); ++g) s += p[g];
        for (int i = (c >> 4) << 4; i < c; ++i) s += p[17 + i];
        return s;
    }
    Interval take(int row, uint32_t value) {
        uint32_t* p = &w[(size_t)row * ROW];
        uint32_t cum = 0, fg = 0, f = 0;
        int g = 0;
        for (; g < 16; ++g) {
            fg = p[g];
            if (value >= cum + fg) cum += fg; else break;
        }
        int c = g * 16;
        for (; c < 256; ++c) {
            f = p[17 + c];
            if (value >= cum + f) cum += f; else break;
        }
        uint32_t tot = p[16];
        // c == 256 lands on the next row's first word (dropped past the end of the whole table),
        // g == 16 on this row's total, which is rewritten below — as in the reference
        if ((size_t)row * ROW + 17 + c < w.size()) p[17 + c] = f + STEP;
        p[g] = fg + STEP;
        tot += STEP;
        if (tot > 65536u) {
            tot = 0;
            for (int i = 0; i < 256; ++i) { p[17 + i] = (p[17 + i] >> 1) + 1; tot += p[17 + i]; }
            for (int k = 0; k < 16; ++k) {
                uint32_t s = 0;
                for (int j = 0; j < 16; ++j) s += p[17 + k * 16 + j];
                p[k] = s;
            }
        }
        p[16] = tot;
        return {c, cum, f};
    }
};

// ------------------------------------------------------------------ v3/v4 models -------------
constexpr int kProbBits = 12, kProbScale = 1 << kProbBits;

// Fixed alphabet, counts folded into the live intervals only when they fill the code space
// (deferred adaptation).  ANS.hx:54-145.  Also the last stage (Cx7) of a colour context.
class FixedModel {
public:
    explicit FixedModel(int nsym = 0) { if (nsym) init(nsym); }
    void init(int nsym) { n_ = nsym; fc_.assign(nsym, {0, 0}); cnt_.assign(nsym, 0); std::memset(start_, 0, sizeof start_); sum_ = 0; }
    void renew() {
        const int fr = kProbScale / n_, c0 = fr - (fr >> 1);
        sum_ = c0 * n_;
        int cf = 0;
        for (int i = 0; i < n_; ++i) { fc_[i] = {(uint16_t)fr, (uint16_t)cf}; cnt_[i] = (uint16_t)c0; mark(cf, fr, i); cf += fr; }
    }
    int locate(int c) const { return fc_[c].cum; }
    Interval take(int slot) {
        int j = start_[slot >> 7];
        while (j < n_ - 1 && fc_[j + 1].cum <= slot) ++j;
        Interval iv{j, fc_[j].cum, fc_[j].freq};
        bump(j);
        return iv;
    }
    int size() const { return n_; }
    // builders used by the colour-context upgrades (ANS.hx:711-771)
    struct FC { uint16_t freq, cum; };
    std::vector<FC>& fc() { return fc_; }
    std::vector<uint16_t>& cnt() { return cnt_; }
    int& sum() { return sum_; }
    void mark(int cf, int fr, int sym) {
        const int k0 = (cf + 127) >> 7, k1 = ((cf + fr - 1) >> 7) + 1;
        for (int k = std::max(k0, 0); k < k1 && k < 32; ++k) start_[k] = (uint8_t)sym;
    }
private:
    void bump(int c) {
        cnt_[c] = (uint16_t)(cnt_[c] + 16);
        sum_ += 16;
        if (sum_ + 16 > kProbScale) {
            sum_ = 0;
            int cf = 0;
            for (int j = 0; j < n_; ++j) {
                const int fr = cnt_[j];
                fc_[j] = {(uint16_t)fr, (uint16_t)cf};
                mark(cf, fr, j);
                cf += fr;
                cnt_[j] = (uint16_t)(fr - (fr >> 1));
                sum_ += cnt_[j];
            }
        }
    }
    int n_ = 0, sum_ = 0;
    std::vector<FC> fc_;
    std::vector<uint16_t> cnt_;
    uint8_t start_[32];
};

// State shared by the colour contexts of ONE coder (statics in the reference: ANS.hx:217,401-402,409).
struct AnsScratch {
    int tot = 0;   // SmallContext.totFr
    int f0 = 32;   // Cx6.f0: 64 for v3, 32 for v4
    uint16_t c256[256];
    uint16_t f512[512];
};

// Census of stage entries (how often a colour context reached each Stage) — lets the stream
// generator's tests prove that every model kind is exercised.  Not used by the decoder.
extern uint64_t g_stage_census[8];

// One colour context = a growing model (ANS.hx:785-860):
//   Empty -> List14 -> (repeat) Sparse4 / Sparse16 -> Table40 -> Full
//                   -> (15th new) List64 -> (repeat) Table40 | (65th new) List256 -> (repeat) Full
class ColourContext {
public:
    enum Stage : uint8_t { Empty, List14, List64, List256, Sparse4, Sparse16, Table40, Full };
    Stage stage() const { return stage_; }
    void renew() { stage_ = Empty; p_.reset(); }
    bool coded() const { return stage_ >= Sparse4; }  // false: the next symbol travels as a raw byte

    // Coded stages: interval of the symbol owning `slot`, then adapt (may upgrade the stage).
    Interval take(int slot, AnsScratch& sc);
    // Raw stages: learn symbol c (c < 0 = the reference's `undefined`: never equal to anything).
    void learn(int c, AnsScratch& sc);
    // Encoder side: a slot inside c's current interval (coded stages only).
    int locate(int c, const AnsScratch& sc) const;

private:
    struct Payload;
    Stage stage_ = Empty;
    std::shared_ptr<Payload> p_;
};

}  // namespace jsp::sp
