// ORACLE — TEST INFRASTRUCTURE ONLY (see msvideo1_oracle.cpp header).  PARITY UNPINNED.
//
// CPU restatement of jsplayer's entropy decoders used by ScreenPressor:
//   RangeCoder.hx:5-131          -> RangeDec           (v2 streams)
//   EntroCoders.hx:31-180        -> EntroRC
//   ANS.hx:5-49                  -> Rans
//   ANS.hx:54-145                -> FixedCtx           (FixedSizeRansCtx)
//   ANS.hx:155-208               -> SymbList / Cx1..Cx3
//   ANS.hx:210-392               -> SmallCtx / Cx4 / Cx5
//   ANS.hx:394-704               -> Cx6
//   ANS.hx:706-772               -> Cx7
//   ANS.hx:774-860               -> Context
//   EntroCoders.hx:182-313       -> EntroANS
// JavaScript number semantics are kept where they are observable:
//   * RangeCoder keeps `range` and `code` in doubles (RangeCoder.hx:22-25,38-48); a read past the
//     end of the data is `undefined`, which turns `code` into NaN, after which Std.int(code/range)
//     is 0 for ever.  Doubles are used here too, so the same thing happens.
//   * Rans combines bytes with `|` and `<<`, where `undefined` acts as 0 (ANS.hx:25-29,41);
//     Rans.raw() hands the `undefined` on (ANS.hx:46-48): modelled as symbol -1.
//   * Uint8Array / Uint16Array stores wrap (ANS.hx:88,99,257,290).
// Statics of the reference (Context.rcv, SmallContext.totFr, Cx6._cnts/_freqs, Cx6.f0) are
// per-coder state here so that several streams can be decoded side by side.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace orc {

struct ByteView {
    const uint8_t* p = nullptr;
    size_t n = 0;
    bool has(long i) const { return i >= 0 && (size_t)i < n; }
    int at(long i) const { return has(i) ? p[i] : -1; }  // -1 = undefined
};

inline int32_t js_to_int32(double d) {  // Std.int on the JS target: d | 0
    if (std::isnan(d) || std::isinf(d)) return 0;
    double t = std::trunc(d);
    double m = std::fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

// ---- v2: range decoder -------------------------------------------------------------------------
struct RangeDec {
    double range = 0, code = 0;
    ByteView data;
    long pos = 0;
    static constexpr double TOP = 16777216.0;  // 1<<24
    static constexpr uint32_t BOT = 65536;

    double byte_at(long i) const { return data.has(i) ? (double)data.p[i] : NAN; }

    void begin(ByteView src, long pos0) {  // RangeCoder.hx:19-34
        range = 65535.0 * 65536.0 + 65535.0;
        data = src;
        pos = pos0;
        code = 0;
        for (int k = 1; k <= 4; ++k) code = code * 256.0 + byte_at(pos + k);
        pos += 5;
    }
    int32_t get_freq(uint32_t tot) {  // :45-49
        range = (double)js_to_int32(range / (double)tot);
        return js_to_int32(code / range);
    }
    void consume(double cum, double freq) {  // :36-43
        code -= cum * range;
        range = range * freq;
        while (range < TOP) {
            code = code * 256.0 + byte_at(pos++);
            range *= 256.0;
        }
    }
    // DecodeVal :51-80 — cnt has maxc counts then the total
    int decode_val(uint32_t* cnt, int maxc, uint32_t step) {
        uint32_t tot = cnt[maxc];
        int32_t value = get_freq(tot);
        int c = 0;
        double cum = 0;
        uint32_t cnt_c = 0;
        while (c < maxc) {
            cnt_c = cnt[c];
            if ((double)value >= cum + (double)cnt_c) cum += cnt_c;
            else break;
            ++c;
        }
        consume(cum, (double)cnt_c);
        cnt[c] = cnt_c + step;  // c == maxc lands on the total slot, rewritten just below
        tot += step;
        if (tot > BOT) {
            tot = 0;
            for (int i = 0; i < maxc; ++i) {
                uint32_t nc = (cnt[i] >> 1) + 1;
                cnt[i] = nc;
                tot += nc;
            }
        }
        cnt[maxc] = tot;
        return c;
    }
    // DecodeValUni :82-130 — row: [0..15] group sums, [16] total, [17..272] symbol counts
    int decode_uni(uint32_t* row, uint32_t step, size_t row_room) {
        uint32_t tot = row[16];
        int32_t value = get_freq(tot);
        int x = 0;
        double cum = 0;
        uint32_t cnt_x = 0;
        while (x < 16) {
            cnt_x = row[x];
            if ((double)value >= cum + (double)cnt_x) cum += cnt_x;
            else break;
            ++x;
        }
        int c = x * 16;
        uint32_t cnt_c = 0;
        while (c < 256) {
            cnt_c = row[c + 17];
            if ((double)value >= cum + (double)cnt_c) cum += cnt_c;
            else break;
            ++c;
        }
        consume(cum, (double)cnt_c);
        // c == 256 / x == 16 write one slot past the row / onto the total (typed-array stores;
        // a store past the whole table is dropped)
        if ((size_t)(c + 17) < row_room) row[c + 17] = cnt_c + step;
        row[x] = cnt_x + step;
        tot += step;
        if (tot > BOT) {
            tot = 0;
            for (int i = 17; i < 17 + 256; ++i) {
                uint32_t nc = (row[i] >> 1) + 1;
                row[i] = nc;
                tot += nc;
            }
            for (int i = 0; i < 16; ++i) {
                uint32_t sum = 0;
                for (int j = 0; j < 16; ++j) sum += row[17 + i * 16 + j];
                row[i] = sum;
            }
        }
        row[16] = tot;
        return c;
    }
};

// Common interface, EntroCoders.hx:8-24
struct EntroCoder {
    virtual ~EntroCoder() = default;
    virtual void preinit() = 0;
    virtual void renewI() = 0;
    virtual void decodeBegin(ByteView src, long pos0) = 0;
    virtual int decodeClr(int cxi) = 0;  // may return -1 (`undefined`) for the ANS coder
    virtual int decodeN(int ptype) = 0;
    virtual int decodeP(int ptype) = 0;
    virtual int decodeX() = 0;
    virtual int decodeBT() = 0;
    virtual int decodeBN() = 0;
    virtual int decodeSXY(int n) = 0;
    virtual int decodeMX() = 0;
    virtual int decodeMY() = 0;
    virtual bool canDecodeBool() = 0;
    virtual bool decodeBool() = 0;
    virtual bool differentConstantsFor16bbp() = 0;
    virtual bool failed() { return false; }  // the reference would spin for ever on this stream
};

std::unique_ptr<EntroCoder> make_entro_rc();
std::unique_ptr<EntroCoder> make_entro_ans(int f0);

}  // namespace orc
