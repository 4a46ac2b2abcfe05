// Colour-context models of the v3/v4 (rANS) ScreenPressor streams.  Behaviour follows ANS.hx
// (SmallContext :210-309, Cx4 :312-327, Cx5 :329-392, Cx6 :394-704, Cx7 :706-772,
// Context :785-860); the data layout is this library's own (sp_models.h: a 64-byte record per context, pools
// for the big stages) — the host entropy stage is bound by cache misses on these models, not by arithmetic.
#include "sp_models.h"

namespace jsp::sp {

namespace {
constexpr int kSparseStep = 50;   // SmallContext.f0
constexpr int kTableStep = 25;    // Cx6.Step

inline int shift_for(int tot) {   // smallest shift with (tot << shift) > 2048
    int s = 0;
    while (tot <= kProbScale / 2) { tot <<= 1; ++s; }
    return s;
}
inline void sort_bytes(uint8_t* a, int n) {
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && a[j - 1] > a[j]; --j) std::swap(a[j], a[j - 1]);
}
}  // namespace

ColourModels::ColourModels(int f0) : small_(3 * 4096), f0_(f0) {
    std::memset(small_.data(), 0, small_.size() * sizeof(Small));
    std::memset(c256_, 0, sizeof c256_);
    std::memset(f512_, 0, sizeof f512_);
}

void ColourModels::renew() {
    for (Small& s : small_) s.stage = Empty;
    lists_.clear();      // (capacity stays: the next key frame needs about as much)
    tables_.clear();
    fulls_.clear();
}

void ColourModels::enter(Small& s, Stage st) {
    s.stage = st;
#ifdef JSP_MODEL_TOOLS
    ++census[st];
#endif
}

// ---- sparse -----------------------------------------------------------------------------------------------
int ColourModels::sparse_total(const Small& s) {
    int t = 256 - s.n;
    for (int i = 0; i < s.n; ++i) t += s.freq[i];
    return t;
}
void ColourModels::sparse_halve(Small& s) {
    int sum = 256 - s.n;
    for (int i = 0; i < s.n; ++i) { s.freq[i] = (uint16_t)(s.freq[i] - (s.freq[i] >> 1)); sum += s.freq[i]; }
    tot_ = sum;
}
bool ColourModels::sparse_insert(Small& s, int pos, int c) {
    if (s.n == s.cap) return false;
    for (int i = s.n - 1; i >= pos; --i) { s.sym[i + 1] = s.sym[i]; s.freq[i + 1] = s.freq[i]; }
    s.sym[pos] = (uint8_t)c;
    s.freq[pos] = kSparseStep;
    ++s.n;
    if (s.maxpos >= pos) ++s.maxpos;
    tot_ += kSparseStep;
    if (tot_ + kSparseStep > kProbScale) sparse_halve(s);
    return true;
}
// returns false when the symbol is new and there is no room (caller upgrades); `iv` is valid either way
bool ColourModels::sparse_take(Small& s, int slot, int tot0, Interval& iv) {
    tot_ = tot0;
    const int shift = shift_for(tot0);
    slot >>= shift;
    const int bonus = (kProbScale - (tot0 << shift)) >> shift;  // spare code space -> most frequent symbol
    const uint16_t keep = s.freq[s.maxpos];
    s.freq[s.maxpos] = (uint16_t)(s.freq[s.maxpos] + bonus);
    int cum = 0, last = 0;
    for (int pos = 0; pos < s.n; ++pos) {
        const int sy = s.sym[pos];
        const int start = cum + sy - last;
        if (slot < start) {  // an unseen symbol below sy
            iv = {slot - cum + last, (uint32_t)(slot << shift), (uint32_t)(1 << shift)};
            s.freq[s.maxpos] = keep;
            return sparse_insert(s, pos, iv.sym);
        }
        const int fr = s.freq[pos];
        if (start + fr > slot) {
            iv = {sy, (uint32_t)(start << shift), (uint32_t)(fr << shift)};
            s.freq[s.maxpos] = keep;
            s.freq[pos] = (uint16_t)(s.freq[pos] + kSparseStep);
            tot_ += kSparseStep;
            if (pos != s.maxpos && s.freq[pos] > s.freq[s.maxpos]) s.maxpos = (uint8_t)pos;
            if (tot_ + kSparseStep > kProbScale) sparse_halve(s);
            return true;
        }
        cum += sy - last + fr;
        last = sy + 1;
    }
    s.freq[s.maxpos] = keep;
    iv = {last + slot - cum, (uint32_t)(slot << shift), (uint32_t)(1 << shift)};
    return sparse_insert(s, s.n, iv.sym);
}
void ColourModels::sparse_from_list14(Small& s, int capacity, int c) {  // SmallContext.create
    s.cap = (uint8_t)capacity;
    s.maxpos = 0;
    sort_bytes(s.sym, s.n);
    for (int i = s.n; i < 16; ++i) s.sym[i] = 0;
    std::memset(s.freq, 0, sizeof s.freq);
    for (int i = 0; i < s.n; ++i) {
        if (s.sym[i] == c) { s.freq[i] = 2 * kSparseStep; s.maxpos = (uint8_t)i; }
        else s.freq[i] = kSparseStep;
    }
}
void ColourModels::sparse16_from_sparse4(Small& s, int c) {  // Cx5.createFrom4
    const Small old = s;
    s.cap = 16;
    std::memset(s.sym, 0, sizeof s.sym);
    std::memset(s.freq, 0, sizeof s.freq);
    s.maxpos = 0;  // the reference starts the wider context with maxpos 0, whatever it was before
    int i = 0, tot = 0;
    while (i < old.n && old.sym[i] < c) { s.sym[i] = old.sym[i]; tot += s.freq[i] = old.freq[i]; ++i; }
    int j = i;
    s.sym[j] = (uint8_t)c;
    tot += s.freq[j] = kSparseStep;
    ++j;
    while (i < old.n) { s.sym[j] = old.sym[i]; tot += s.freq[j] = old.freq[i]; ++i; ++j; }
    s.n = (uint8_t)(old.n + 1);
    if (tot > kProbScale) sparse_halve(s);
    s.cached_tot = (uint16_t)sparse_total(s);
}

// ---- table40 ----------------------------------------------------------------------------------------------
void ColourModels::table_calc_sum(Table& t) {
    const int sh = t.fshift > 0 ? t.fshift - 1 : 0;
    int sum = (256 - t.td) << sh;
    for (int i = 0; i < t.tcap; ++i) sum += t.cnt[i];
    t.tsum = (uint16_t)sum;
}
void ColourModels::table_rebuild(Table& t) {  // Cx6.rescaleDec
    const int sh = t.fshift > 0 ? t.fshift - 1 : 0;
    for (int i = 0; i < 256; ++i) c256_[i] = (uint16_t)(1 << sh);
    for (int i = 0; i < t.td; ++i) c256_[t.sym[i]] = t.cnt[i];
    int cum = 0;
    for (int i = 0; i < 256; ++i) { f512_[2 * i] = c256_[i]; f512_[2 * i + 1] = (uint16_t)cum; cum += c256_[i]; }
    if (t.fshift > 0) --t.fshift;
    const int sh2 = t.fshift > 0 ? t.fshift - 1 : 0;
    int sum = (256 - t.td) << sh2;
    for (int i = 0; i < t.td; ++i) {
        t.cnt[i] = (uint16_t)(t.cnt[i] - (t.cnt[i] >> 1));
        sum += t.cnt[i];
        t.freq[i] = f512_[2 * t.sym[i]];
        t.cum[i] = f512_[2 * t.sym[i] + 1];
    }
    t.tsum = (uint16_t)sum;
}
void ColourModels::table_bump(Table& t, int pos) {  // Cx6.incrCntDec
    const int step = kTableStep << t.fshift;
    t.cnt[pos] = (uint16_t)(t.cnt[pos] + step);
    t.tsum = (uint16_t)(t.tsum + step);
    if (pos > 0 && t.cnt[pos] > t.cnt[pos - 1]) table_swap(t, pos, pos - 1);
    if (t.tsum + step > kProbScale) table_rebuild(t);
}
int ColourModels::table_add(Table& t, int c, int freq, int cum) {
    if (t.td >= 40 || t.td >= t.tcap) return -1;
    t.set(t.td, cum, freq, freq - (freq >> 1), c);
    return t.td++;
}
// interval an unseen symbol c would get right now
int ColourModels::table_unseen_cum(const Table& t, int c) {
    int lower = -1, lfreq = 0, lcum = 0;
    for (int i = 0; i < t.td; ++i)
        if (t.sym[i] > lower && t.sym[i] < c) { lower = t.sym[i]; lfreq = t.freq[i]; lcum = t.cum[i]; }
    return lfreq > 0 ? lcum + lfreq + ((c - lower - 1) << t.fshift) : c << t.fshift;
}
uint32_t ColourModels::table_from_sparse16(const Small& s, int c) {  // Cx6.createFrom5
    tables_.emplace_back();
    Table& t = tables_.back();
    std::memset(&t, 0, sizeof t);
    t.tcap = 32;
    const int oldd = s.n;
    const int shift = shift_for(sparse_total(s));
    int cum = 0, last = 0;
    for (int pos = 0; pos < oldd; ++pos) {
        const int sy = s.sym[pos];
        cum += sy - last;
        const int fr = s.freq[pos] << shift;
        t.set(pos, cum << shift, fr, fr - (fr >> 1), sy);
        cum += s.freq[pos];
        last = sy + 1;
    }
    t.td = oldd;
    t.fshift = shift;
    const int f = 1 << t.fshift;
    const int cf = c > 0 ? table_unseen_cum(t, c) : 0;
    t.set(oldd, cf, f, f - (f >> 1), c);
    t.td = oldd + 1;
    const int step = kTableStep << t.fshift;
    t.cnt[oldd] = (uint16_t)(t.cnt[oldd] + step);
    t.tsum = (uint16_t)(t.tsum + step);
    if (t.tsum + step > kProbScale) table_rebuild(t);
    table_calc_sum(t);
    for (int i = 0; i < t.td - 1; ++i)  // most frequent first (exchange sort, as the reference)
        for (int j = i + 1; j < t.td; ++j)
            if (t.freq[j] > t.freq[i]) table_swap(t, i, j);
    return (uint32_t)(tables_.size() - 1);
}
uint32_t ColourModels::table_from_list(ListBig& l, int c) {  // Cx6.createFrom2
    tables_.emplace_back();
    Table& t = tables_.back();
    std::memset(&t, 0, sizeof t);
    const int oldd = l.ld;
    t.tcap = oldd <= 32 ? 32 : 64;
    const int shift = shift_for(256 - oldd + oldd * f0_ + f0_);
    // ascending symbols: read off the membership bits when the list holds each symbol once (always, but for the 0 a read
    // past the end of the stream stores), else sorted as the reference does
    if (__builtin_popcountll(l.seen[0]) + __builtin_popcountll(l.seen[1]) + __builtin_popcountll(l.seen[2]) + __builtin_popcountll(l.seen[3]) == oldd) {
        int k = 0;
        for (int w = 0; w < 4; ++w)
            for (uint64_t m = l.seen[w]; m; m &= m - 1) l.list[k++] = (uint8_t)(w * 64 + __builtin_ctzll(m));
    } else
        sort_bytes(l.list, oldd);
    int cum = 0, last = 0, at = 0;
    for (int pos = 0; pos < oldd; ++pos) {
        const int sy = l.list[pos];
        cum += sy - last;
        int cfr = f0_;
        if (sy == c) { at = pos; cfr = 2 * f0_; }
        const int fr = cfr << shift;
        t.set(pos, cum << shift, fr, fr - (fr >> 1), sy);
        cum += cfr;
        last = sy + 1;
    }
    t.td = oldd;
    t.fshift = shift;
    table_calc_sum(t);
    if (at > 0) table_swap(t, 0, at);  // the repeated symbol leads
    return (uint32_t)(tables_.size() - 1);
}
bool ColourModels::table_take(Table& t, int slot, Interval& iv) {  // Cx6.decode
    // The reference walks the entries in order: the first whose interval holds the slot wins; if none does, the entry
    // with the highest start at or below the slot (the later one on a tie) is the seen symbol below the new one.  Here:
    // eight entries per compare, one bit per entry, the two questions answered from the bit sets.
    uint64_t below = 0, holds = 0;   // bit i: cum[i] <= slot / ... and slot < cum[i] + freq[i]
    {
        const __m128i bias = _mm_set1_epi16((short)0x8000);
        const __m128i sl = _mm_set1_epi16((short)slot), sv = _mm_xor_si128(sl, bias);
        for (int v = 0; v * 8 < t.td; ++v) {
            const __m128i cu = _mm_load_si128(reinterpret_cast<const __m128i*>(t.cum + v * 8));
            const __m128i fr = _mm_load_si128(reinterpret_cast<const __m128i*>(t.freq + v * 8));
            const __m128i le = _mm_xor_si128(_mm_cmpgt_epi16(_mm_xor_si128(cu, bias), sv), _mm_set1_epi16(-1));   // cum <= slot (unsigned)
            const __m128i room = _mm_sub_epi16(sl, cu);                                                            // slot - cum, exact where le
            const __m128i in = _mm_cmpgt_epi16(_mm_xor_si128(fr, bias), _mm_xor_si128(room, bias));               // freq > slot - cum
            const uint32_t mle = (uint32_t)_mm_movemask_epi8(_mm_packs_epi16(le, _mm_setzero_si128()));
            const uint32_t min_ = (uint32_t)_mm_movemask_epi8(_mm_packs_epi16(_mm_and_si128(le, in), _mm_setzero_si128()));
            below |= (uint64_t)mle << (v * 8);
            holds |= (uint64_t)min_ << (v * 8);
        }
        const uint64_t live = t.td >= 64 ? ~0ull : (1ull << t.td) - 1;
        below &= live;
        holds &= live;
    }
    if (holds) {
        const int i = __builtin_ctzll(holds);
        iv = {t.sym[i], t.cum[i], t.freq[i]};
        table_bump(t, i);
        return true;
    }
    int lfreq = 0, lcum = 0, lower = 0;
    for (uint64_t m = below; m; m &= m - 1) {
        const int i = __builtin_ctzll(m);
        const int cf = t.cum[i];
        if (cf >= lcum) { lfreq = t.freq[i]; lcum = cf; lower = t.sym[i]; }
    }
    const int f = 1 << t.fshift;
    int c, cf;
    if (lfreq > 0) {
        const int x = (slot - (lcum + lfreq)) >> t.fshift;
        c = x + lower + 1;
        cf = lcum + lfreq + (x << t.fshift);
    } else {
        c = slot >> t.fshift;
        cf = c << t.fshift;
    }
    iv = {c, (uint32_t)cf, (uint32_t)f};
    int p = table_add(t, c, f, cf);
    if (p < 0) {
        if (t.tcap == 64) return false;
        t.tcap = 64;  // growDec: the arrays double (the entries beyond the old capacity are empty)
        p = table_add(t, c, f, cf);
    }
    table_bump(t, p);
    return true;
}

// ---- full -------------------------------------------------------------------------------------------------
uint32_t ColourModels::full_from_list(const ListBig& l, int c) {  // Cx7.createFrom3
    fulls_.emplace_back(256);
    Full256& m = fulls_.back();
    uint16_t* cnt = m.cnt();
    uint16_t* cum = m.cum();
    uint16_t freq[256];
    for (int i = 0; i < 256; ++i) { freq[i] = 1; cnt[i] = 1; }
    const int d = l.ld;
    const int f0 = (kProbScale - (256 - d)) / (d + 1), c0 = f0 - (f0 >> 1);
    for (int i = 0; i < d; ++i) { freq[l.list[i]] = (uint16_t)f0; cnt[l.list[i]] = (uint16_t)c0; }
    freq[c] = (uint16_t)(freq[c] + f0);
    cnt[c] = (uint16_t)(cnt[c] + 16);
    int sum = 0, cf = 0;
    for (int i = 0; i < 256; ++i) {
        sum += cnt[i];
        cum[i] = (uint16_t)cf;
        cf += freq[i];
    }
    cum[256] = (uint16_t)cf;
    m.sum() = sum;
    m.reindex();
    return (uint32_t)(fulls_.size() - 1);
}
uint32_t ColourModels::full_from_table(const Table& t) {  // Cx7.createFrom6
    fulls_.emplace_back(256);
    Full256& m = fulls_.back();
    uint16_t* cnt = m.cnt();
    uint16_t* cum = m.cum();
    m.sum() = t.tsum;
    uint16_t freq[256];
    std::memset(freq, 0, sizeof freq);
    for (int i = 0; i < t.tcap; ++i)
        if (t.cnt[i] > 0) { freq[t.sym[i]] = t.freq[i]; cnt[t.sym[i]] = t.cnt[i]; }
    // (a seen symbol keeps the start the table gave it — the same running sum: the table tiles the code space with its own
    // intervals and one of width f per unseen symbol)
    const int f = 1 << t.fshift, cu = f - (f >> 1);
    int cf = 0;
    for (int i = 0; i < 256; ++i) {
        int fr;
        if (freq[i] > 0) fr = freq[i];
        else { cnt[i] = (uint16_t)cu; fr = f; }
        cum[i] = (uint16_t)cf;
        cf += fr;
    }
    cum[256] = (uint16_t)cf;
    m.reindex();
    return (uint32_t)(fulls_.size() - 1);
}

// ---- the context state machine (Context, ANS.hx:785-860) -------------------------------------------------------
void ColourModels::learn_slow(int ctx, int c) {
    Small& s = small_[ctx];
    const uint8_t byte = (uint8_t)(c < 0 ? 0 : c);   // a missing byte is stored as 0 (and matches a later 0)
    auto big_find_or_add = [&](ListBig& l, int capacity) -> int {   // 0 found, 1 added, 2 no room
        if (c >= 0 && (l.seen[(c >> 6) & 3] >> (c & 63) & 1)) return 0;
        if (l.ld < capacity) { l.list[l.ld++] = byte; l.seen[byte >> 6] |= 1ull << (byte & 63); return 1; }
        return 2;
    };
    switch (s.stage) {
        case Empty:
            s.n = 1;
            s.sym[0] = byte;
            enter(s, List14);
            break;
        case List14: {
            const unsigned same = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(s.sym)), _mm_set1_epi8((char)byte)));
            const bool found = c >= 0 && (same & ((1u << s.n) - 1u));
            if (found) {
                if (s.n <= 4) { sparse_from_list14(s, 4, c); enter(s, Sparse4); }
                else { sparse_from_list14(s, 16, c); s.cached_tot = (uint16_t)sparse_total(s); enter(s, Sparse16); }
            } else if (s.n < 14) {
                s.sym[s.n++] = byte;
            } else {   // the 15th distinct symbol: the list moves to the pool
                lists_.emplace_back();
                ListBig& l = lists_.back();
                std::memset(&l, 0, sizeof l);
                for (int i = 0; i < 14; ++i) { l.list[i] = s.sym[i]; l.seen[s.sym[i] >> 6] |= 1ull << (s.sym[i] & 63); }
                l.list[14] = byte;
                l.seen[byte >> 6] |= 1ull << (byte & 63);
                l.ld = 15;
                s.big = (uint32_t)(lists_.size() - 1);
                enter(s, List64);
            }
            break;
        }
        case List64: {
            ListBig& l = lists_[s.big];
            switch (big_find_or_add(l, 64)) {
                case 0: s.big = table_from_list(l, c); enter(s, Table40); break;
                case 1: break;
                default: l.list[l.ld++] = byte; l.seen[byte >> 6] |= 1ull << (byte & 63); enter(s, List256); break;
            }
            break;
        }
        case List256: {
            ListBig& l = lists_[s.big];
            if (big_find_or_add(l, 256) == 0) { s.big = full_from_list(l, c); enter(s, Full); }
            break;
        }
        default: break;  // coded stages never learn from raw bytes
    }
}

#ifdef JSP_MODEL_TOOLS
int ColourModels::locate(int ctx, int c) const {
    const Small& s = small_[ctx];
    auto sparse_locate = [&](int tot0) {
        const int shift = shift_for(tot0);
        const int bonus = (kProbScale - (tot0 << shift)) >> shift;
        int cum = 0, last = 0;
        for (int pos = 0; pos < s.n; ++pos) {
            const int sy = s.sym[pos];
            if (c < sy) return (cum + c - last) << shift;
            if (c == sy) return (cum + sy - last) << shift;
            cum += sy - last + s.freq[pos] + (pos == s.maxpos ? bonus : 0);
            last = sy + 1;
        }
        return (cum + c - last) << shift;
    };
    switch (s.stage) {
        case Sparse4: return sparse_locate(s.freq[0] + s.freq[1] + s.freq[2] + s.freq[3] + 256 - s.n);
        case Sparse16: return sparse_locate(s.cached_tot);
        case Table40: {
            const Table& t = tables_[s.big];
            for (int i = 0; i < t.td; ++i)
                if (t.sym[i] == c) return t.cum[i];
            return c > 0 ? table_unseen_cum(t, c) : 0;
        }
        case Full: return fulls_[s.big].locate(c);
        default: return 0;
    }
}
#endif

}  // namespace jsp::sp
