// ScreenPressor adaptive models (host side), independent of the bit-level coder that drives them.
//
// Every model answers one question — "which symbol owns slot `value` of the current code space,
// and what is its interval?" — and then adapts, exactly as the reference's decoders do:
//   v2  range-coder tables : RangeCoder.hx:51-130 (DecodeVal / DecodeValUni), EntroCoders.hx:81-130
//   v3/4 rANS models       : ANS.hx:54-145 (FixedSizeRansCtx), :155-392 (Cx1..Cx5), :394-704 (Cx6),
//                            :706-772 (Cx7), :785-860 (Context)
// The stream encoder used to synthesise test/bench input (jsplayer_amd/gen, built with JSP_MODEL_TOOLS) drives the same
// objects through `locate(symbol)` (a slot inside the symbol's current interval) followed by the decoder's own lookup,
// so encoder and decoder cannot drift apart; the decoder's build carries none of those helpers.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <type_traits>
#include <vector>

#include <emmintrin.h>   // SSE2: part of every x86-64
#include <sys/mman.h>
#include <cstdlib>
#include <new>

namespace jsp::sp {

// A zero-filled array of `n` words for tables that are read and written at random: 2 MB-aligned and handed to the kernel as a huge-page
// candidate BEFORE it is first touched (transparent huge pages in `madvise` mode: what this image runs), so that a lookup costs a cache
// miss but not a TLB miss on top — the version-2 colour tables are 13.4 MB per decoder, 3 300 small pages, and sixteen decoders side
// by side share one second-level TLB per core pair.
struct HugeWords {
    uint32_t* p = nullptr;
    size_t n = 0;
    explicit HugeWords(size_t words) : n(words) {
        const size_t bytes = (words * sizeof(uint32_t) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void* m = std::aligned_alloc((size_t)2 << 20, bytes);
        if (!m) throw std::bad_alloc();
        (void)madvise(m, bytes, MADV_HUGEPAGE);          // (a hint: refused or unsupported, the table is simply made of small pages)
        p = static_cast<uint32_t*>(m);
        std::fill(p, p + words, 0u);
    }
    ~HugeWords() { std::free(p); }
    HugeWords(const HugeWords&) = delete;
    HugeWords& operator=(const HugeWords&) = delete;
    HugeWords(HugeWords&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    size_t size() const { return n; }
    uint32_t& operator[](size_t i) { return p[i]; }
    const uint32_t& operator[](size_t i) const { return p[i]; }
};

// The same for vectors (the grown model pools, the shadow frames): allocations of a megabyte or more are 2 MB-aligned huge-page candidates.
template <class T>
struct HugeAllocator {
    using value_type = T;
    HugeAllocator() = default;
    template <class U> HugeAllocator(const HugeAllocator<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < ((size_t)1 << 20)) {
            void* m = std::malloc(bytes ? bytes : 1);
            if (!m) throw std::bad_alloc();
            return static_cast<T*>(m);
        }
        const size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void* m = std::aligned_alloc((size_t)2 << 20, rounded);
        if (!m) throw std::bad_alloc();
        (void)madvise(m, rounded, MADV_HUGEPAGE);
        return static_cast<T*>(m);
    }
    void deallocate(T* p, size_t) { std::free(p); }
    template <class U> bool operator==(const HugeAllocator<U>&) const { return true; }
    template <class U> bool operator!=(const HugeAllocator<U>&) const { return false; }
};
template <class T> using BigVector = std::vector<T, HugeAllocator<T>>;

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
#ifdef JSP_MODEL_TOOLS
    uint32_t cum_of(int c) const { uint32_t s = 0; for (int i = 0; i < c; ++i) s += cnt[i]; return s; }
#endif
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
    HugeWords w;
    RcColourTables() : w((size_t)ROW * ROWS) {}
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
#ifdef JSP_MODEL_TOOLS
    uint32_t cum_of(int row, int c) const {
        const uint32_t* p = &w[(size_t)row * ROW];
        uint32_t s = 0;
        for (int g = 0; g < (c >> 4); ++g) s += p[g];
        for (int i = (c >> 4) << 4; i < c; ++i) s += p[17 + i];
        return s;
    }
#endif
    // running sums of 16 counts on top of `before` into run[]; returns bit k set where run[k] > value
    static unsigned scan16(const uint32_t* q, uint32_t before, uint32_t* run, __m128i vv) {
        auto sums4 = [](__m128i v) { v = _mm_add_epi32(v, _mm_slli_si128(v, 4)); return _mm_add_epi32(v, _mm_slli_si128(v, 8)); };
        __m128i a = _mm_add_epi32(sums4(_mm_loadu_si128(reinterpret_cast<const __m128i*>(q))), _mm_set1_epi32((int)before));
        __m128i b = _mm_add_epi32(sums4(_mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 4))), _mm_shuffle_epi32(a, 0xFF));
        __m128i c = _mm_add_epi32(sums4(_mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 8))), _mm_shuffle_epi32(b, 0xFF));
        __m128i d = _mm_add_epi32(sums4(_mm_loadu_si128(reinterpret_cast<const __m128i*>(q + 12))), _mm_shuffle_epi32(c, 0xFF));
        _mm_store_si128(reinterpret_cast<__m128i*>(run), a);
        _mm_store_si128(reinterpret_cast<__m128i*>(run + 4), b);
        _mm_store_si128(reinterpret_cast<__m128i*>(run + 8), c);
        _mm_store_si128(reinterpret_cast<__m128i*>(run + 12), d);
        return (unsigned)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpgt_epi32(a, vv))) | (unsigned)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpgt_epi32(b, vv))) << 4 |
               (unsigned)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpgt_epi32(c, vv))) << 8 | (unsigned)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpgt_epi32(d, vv))) << 12;
    }
    Interval take(int row, uint32_t value) {
        uint32_t* p = &w[(size_t)row * ROW];
        // The reference scans the 16 group sums, then the symbols of the group it stopped in (and on into the next ones
        // when the sums do not add up — never on streams an encoder wrote).  Same answer from running sums built four
        // at a time: the group / symbol is the first whose running sum exceeds the value.
        uint32_t cum = 0, fg = 0, f = 0;
        int g = 16, c = 256;
        if (value < 0x7FFFFFFFu) {
            alignas(16) uint32_t run[16];
            const __m128i vv = _mm_set1_epi32((int)value);
            unsigned above = scan16(p, 0, run, vv);
            if (above) {
                g = __builtin_ctz(above);
                fg = p[g];
                const uint32_t before = g ? run[g - 1] : 0;
                above = scan16(p + 17 + g * 16, before, run, vv);
                if (above) {
                    const int k = __builtin_ctz(above);
                    c = g * 16 + k;
                    f = p[17 + c];
                    cum = k ? run[k - 1] : before;
                }
            }
        }
        if (c == 256) {   // off the common path: walk as the reference does
            cum = 0; fg = 0; f = 0;
            for (g = 0; g < 16; ++g) {
                fg = p[g];
                if (value >= cum + fg) cum += fg; else break;
            }
            for (c = g * 16; c < 256; ++c) {
                f = p[17 + c];
                if (value >= cum + f) cum += f; else break;
            }
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

// Fixed alphabet of up to CAP symbols, counts folded into the live intervals only when they fill the code space
// (deferred adaptation).  ANS.hx:54-145.  Also the last stage (Cx7) of a colour context.
// The symbol owning a slot is the first j whose successor starts above the slot (ANS.hx:105-126: a scan from a hint that
// never lies above the answer — the reference keeps 32 hints, this model 128, the answer is the same).  Layout for that
// scan: only the interval STARTS are kept (the intervals tile the code space, so a frequency is the distance to the next
// start; the end of the last interval closes the array), with eight 0xFFFF sentinels behind, so eight candidates are
// compared at once and the scan needs no bound.  A lookup then depends on two cache lines in a row (hint, starts) — the
// models of a stream are tens of megabytes, every lookup misses — and the count it bumps is off that path.
template <int CAP>
class FixedModelT {
public:
    using Hint = typename std::conditional<(CAP > 256), uint16_t, uint8_t>::type;
    explicit FixedModelT(int nsym = 0) { init(nsym); }
    void init(int nsym) {
        n_ = nsym; sum_ = 0;
        std::memset(hint_, 0, sizeof hint_); std::memset(cum_, 0, sizeof cum_); std::memset(cnt_, 0, sizeof cnt_);
        for (int i = nsym + 1; i < CAP + 9; ++i) cum_[i] = 0xFFFF;
    }
    void renew() {
        const int fr = kProbScale / n_, c0 = fr - (fr >> 1);
        sum_ = c0 * n_;
        int cf = 0;
        for (int i = 0; i < n_; ++i) { cnt_[i] = (uint16_t)c0; cum_[i] = (uint16_t)cf; cf += fr; }
        cum_[n_] = (uint16_t)cf;
        reindex();
    }
    void prefetch() const {   // every line a lookup may read first
        const char* p = reinterpret_cast<const char*>(hint_);
        for (size_t o = 0; o < sizeof hint_ + sizeof cum_; o += 64) __builtin_prefetch(p + o);
    }
    Interval take(int slot) {
        int j = hint_[slot >> 5];
        const __m128i bias = _mm_set1_epi16((short)0x8000);
        const __m128i sv = _mm_set1_epi16((short)(slot ^ 0x8000));
        for (;;) {   // first of cum_[j+1 ..] above the slot (unsigned compare); the sentinels end it
            const __m128i c = _mm_xor_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(cum_ + j + 1)), bias);
            const int above = _mm_movemask_epi8(_mm_cmpgt_epi16(c, sv));
            if (above) { j += __builtin_ctz((unsigned)above) >> 1; break; }
            j += 8;
        }
        if (j > n_ - 1) j = n_ - 1;   // a slot past the last interval belongs to the last symbol
        Interval iv{j, cum_[j], (uint32_t)(uint16_t)(cum_[j + 1] - cum_[j])};
        bump(j);
        return iv;
    }
    int size() const { return n_; }
    // builders used by the colour-context upgrades (ANS.hx:711-771): fill cum()[0..n] / cnt() / sum(), then reindex()
    uint16_t* cum() { return cum_; }
    uint16_t* cnt() { return cnt_; }
    int& sum() { return sum_; }
    void reindex() {   // hint of bucket k: the symbol owning slot 32 k
        int j = 0;
        for (int k = 0; k < 128; ++k) {
            const int s = k << 5;
            while (j < n_ - 1 && cum_[j + 1] <= s) ++j;
            hint_[k] = (Hint)j;
        }
    }
#ifdef JSP_MODEL_TOOLS   // stream generator only: a slot inside c's current interval
    int locate(int c) const { return cum_[c]; }
#endif
private:
    void bump(int c) {
        cnt_[c] = (uint16_t)(cnt_[c] + 16);
        sum_ += 16;
        if (sum_ + 16 > kProbScale) {
            sum_ = 0;
            int cf = 0;
            for (int j = 0; j < n_; ++j) {
                const int fr = cnt_[j];
                cum_[j] = (uint16_t)cf;
                cf += fr;
                cnt_[j] = (uint16_t)(fr - (fr >> 1));
                sum_ += cnt_[j];
            }
            cum_[n_] = (uint16_t)cf;
            reindex();
        }
    }
    int n_ = 0, sum_ = 0;
    alignas(64) Hint hint_[128];
    uint16_t cum_[CAP + 9];   // starts of the n intervals, the end of the last one, sentinels
    uint16_t cnt_[CAP];
};
using FixedModel = FixedModelT<512>;   // run lengths, predictor types, block types, motion vectors, ...: up to 512 symbols

// The 3 x 4096 colour contexts of ONE coder, each a growing model (ANS.hx:785-860):
//   Empty -> List14 -> (repeat) Sparse4 / Sparse16 -> Table40 -> Full
//                   -> (15th new) List64 -> (repeat) Table40 | (65th new) List256 -> (repeat) Full
// Layout (this library's own): one 64-byte record per context — enough for the stages screen content lives in
// (List14, Sparse4, Sparse16), so the whole hot set is 768 KB — and pools for the big stages (long lists, 40-entry
// tables with their entries interleaved, full 256-symbol models), addressed by index and emptied at every key frame.
// The statics of the reference (ANS.hx:217,401-402,409) are members: coders of different streams run side by side.
class ColourModels {
public:
    enum Stage : uint8_t { Empty, List14, List64, List256, Sparse4, Sparse16, Table40, Full };
    explicit ColourModels(int f0);
    void renew();                                             // every context back to Empty (a coded key frame begins)
    Stage stage(int ctx) const { return (Stage)small_[ctx].stage; }
    bool coded(int ctx) const { return small_[ctx].stage >= Sparse4; }   // false: the next symbol travels as a raw byte
    // Coded stages: interval of the symbol owning `slot`, then adapt (may upgrade the stage).
    Interval take(int ctx, int slot) {
        Small& s = small_[ctx];
        Interval iv{0, 0, 0};
        switch (s.stage) {
            case Sparse4: {
                const int tot = s.freq[0] + s.freq[1] + s.freq[2] + s.freq[3] + 256 - s.n;
                if (!sparse_take(s, slot, tot, iv)) { sparse16_from_sparse4(s, iv.sym); enter(s, Sparse16); }
                break;
            }
            case Sparse16:
                if (!sparse_take(s, slot, s.cached_tot, iv)) {
                    s.cached_tot = (uint16_t)tot_;
                    s.big = table_from_sparse16(s, iv.sym);
                    enter(s, Table40);
                } else
                    s.cached_tot = (uint16_t)tot_;
                break;
            case Table40:
                if (!table_take(tables_[s.big], slot, iv)) { s.big = full_from_table(tables_[s.big]); enter(s, Full); }
                break;
            case Full: {
                Full256& m = fulls_[s.big];
                m.prefetch();   // hint and starts together: the second read does not wait for the first
                iv = m.take(slot);
                break;
            }
            default: break;
        }
        return iv;
    }
    // Raw stages: learn symbol c (c < 0 = the reference's `undefined`: stored as 0, never found by itself).
    // The two cases a noisy key frame spends its raw bytes on stay in the caller's loop: the first symbol of a context, and
    // a new symbol for a short list with room (sixteen list entries compared at once); the rest is learn_slow().
    void learn(int ctx, int c) {
#ifndef JSP_MODEL_TOOLS   // (the stream generator counts stage entries: it takes the one path that does)
        Small& s = small_[ctx];
        const uint8_t byte = (uint8_t)(c < 0 ? 0 : c);
        if (s.stage == Empty) { s.n = 1; s.sym[0] = byte; s.stage = List14; return; }
        if (s.stage == List14 && s.n < 14) {
            const unsigned same = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(s.sym)), _mm_set1_epi8((char)byte)));
            if (!(c >= 0 && (same & ((1u << s.n) - 1u)))) { s.sym[s.n++] = byte; return; }
        }
#endif
        learn_slow(ctx, c);
    }
    void learn_slow(int ctx, int c);
#ifdef JSP_MODEL_TOOLS   // stream generator only
    int locate(int ctx, int c) const;                         // a slot inside c's current interval (coded stages)
    uint64_t census[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // how often a context entered each Stage
#endif

private:
    struct Small {            // 64 bytes
        uint8_t stage, n, cap, maxpos;   // n: List14 symbols seen / sparse symbols held
        uint16_t cached_tot, pad;
        uint32_t big;                    // List64/List256 -> lists_, Table40 -> tables_, Full -> fulls_
        uint8_t sym[16];                 // List14: symbols in arrival order; sparse: sorted symbols
        uint16_t freq[16];               // sparse frequencies
        uint8_t pad2[4];
    };
    struct ListBig { uint16_t ld; uint8_t list[256]; uint64_t seen[4]; };
    struct Table {            // Cx6: up to 40 explicit intervals inside the full 256-symbol cumulative space
        int tcap, td, fshift;
        uint16_t tsum;   // a 16-bit slot in the reference's typed array: it wraps
        // entry i = (cum[i], freq[i], cnt[i], sym[i]); one array per field: the lookup compares eight starts at once
        alignas(16) uint16_t cum[64];
        alignas(16) uint16_t freq[64];
        uint16_t cnt[64];
        uint8_t sym[64];
        void set(int i, int c, int f, int n, int sy) { cum[i] = (uint16_t)c; freq[i] = (uint16_t)f; cnt[i] = (uint16_t)n; sym[i] = (uint8_t)sy; }
    };
    using Full256 = FixedModelT<256>;
    static_assert(sizeof(Small) == 64, "one cache line per context");

    void enter(Small& s, Stage st);
    // sparse (SmallContext / Cx4 / Cx5, ANS.hx:210-392)
    static int sparse_total(const Small& s);
    void sparse_halve(Small& s);
    bool sparse_insert(Small& s, int pos, int c);
    bool sparse_take(Small& s, int slot, int tot0, Interval& iv);
    void sparse_from_list14(Small& s, int capacity, int c);
    void sparse16_from_sparse4(Small& s, int c);
    // table (Cx6, ANS.hx:394-704)
    static void table_swap(Table& t, int a, int b) {
        std::swap(t.cum[a], t.cum[b]); std::swap(t.freq[a], t.freq[b]); std::swap(t.cnt[a], t.cnt[b]); std::swap(t.sym[a], t.sym[b]);
    }
    static void table_calc_sum(Table& t);
    void table_rebuild(Table& t);
    void table_bump(Table& t, int pos);
    static int table_add(Table& t, int c, int freq, int cum);
    static int table_unseen_cum(const Table& t, int c);
    uint32_t table_from_sparse16(const Small& s, int c);
    uint32_t table_from_list(ListBig& l, int c);
    bool table_take(Table& t, int slot, Interval& iv);
    // full (Cx7, ANS.hx:706-772)
    uint32_t full_from_list(const ListBig& l, int c);
    uint32_t full_from_table(const Table& t);

    std::vector<Small> small_;
    BigVector<ListBig> lists_;
    BigVector<Table> tables_;
    BigVector<Full256> fulls_;
    int tot_ = 0;              // SmallContext.totFr
    int f0_;                   // Cx6.f0: 64 for v3, 32 for v4
    uint16_t c256_[256], f512_[512];   // Cx6._cnts / _freqs
};

}  // namespace jsp::sp
