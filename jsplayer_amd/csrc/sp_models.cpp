// Colour-context models of the v3/v4 (rANS) ScreenPressor streams.  Behaviour follows ANS.hx
// (SmallContext :210-309, Cx4 :312-327, Cx5 :329-392, Cx6 :394-704, Cx7 :706-772,
// Context :785-860); the data layout is this library's own.
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

uint64_t g_stage_census[8] = {0, 0, 0, 0, 0, 0, 0, 0};

struct ColourContext::Payload {
    // --- List14 / List64 / List256: symbols seen so far, none twice
    uint8_t list[256];
    int ld = 0;
    // --- Sparse4 / Sparse16: sorted symbols with frequencies; unseen symbols have width 1
    int cap = 0, sd = 0, maxpos = 0, cached_tot = 0;
    uint8_t ssym[16];
    uint16_t sfreq[16];
    // --- Table40: up to 40 explicit intervals inside the full 256-symbol cumulative space
    int tcap = 0, td = 0, fshift = 0;
    uint8_t tsym[64];
    uint16_t tfreq[64], tcum[64], tcnt[65];
    // --- Full
    std::unique_ptr<FixedModel> full;

    // ---- lists --------------------------------------------------------------------------
    enum Find { Found, Added, NoRoom };
    Find find_or_add(int c, int capacity) {
        for (int i = 0; i < ld; ++i)
            if (c >= 0 && list[i] == c) return Found;
        if (ld < capacity) { list[ld++] = (uint8_t)(c < 0 ? 0 : c); return Added; }
        return NoRoom;
    }

    // ---- sparse -------------------------------------------------------------------------
    void sparse_from_list(int capacity, int c) {  // SmallContext.create
        cap = capacity;
        std::memset(ssym, 0, sizeof ssym);
        std::memset(sfreq, 0, sizeof sfreq);
        maxpos = 0;
        sd = ld;
        sort_bytes(list, ld);
        for (int i = 0; i < sd; ++i) {
            ssym[i] = list[i];
            if (ssym[i] == c) { sfreq[i] = 2 * kSparseStep; maxpos = i; }
            else sfreq[i] = kSparseStep;
        }
    }
    int sparse_total() const {
        int t = 256 - sd;
        for (int i = 0; i < sd; ++i) t += sfreq[i];
        return t;
    }
    void sparse_halve(AnsScratch& sc) {
        int s = 256 - sd;
        for (int i = 0; i < sd; ++i) { sfreq[i] = (uint16_t)(sfreq[i] - (sfreq[i] >> 1)); s += sfreq[i]; }
        sc.tot = s;
    }
    bool sparse_insert(int pos, int c, AnsScratch& sc) {
        if (sd == cap) return false;
        for (int i = sd - 1; i >= pos; --i) { ssym[i + 1] = ssym[i]; sfreq[i + 1] = sfreq[i]; }
        ssym[pos] = (uint8_t)c;
        sfreq[pos] = kSparseStep;
        ++sd;
        if (maxpos >= pos) ++maxpos;
        sc.tot += kSparseStep;
        if (sc.tot + kSparseStep > kProbScale) sparse_halve(sc);
        return true;
    }
    // returns false when the symbol is new and there is no room (caller upgrades); `iv` is valid
    // either way
    bool sparse_take(int slot, int tot0, AnsScratch& sc, Interval& iv) {
        sc.tot = tot0;
        const int shift = shift_for(tot0);
        slot >>= shift;
        const int bonus = (kProbScale - (tot0 << shift)) >> shift;  // spare code space -> most frequent symbol
        const uint16_t keep = sfreq[maxpos];
        sfreq[maxpos] = (uint16_t)(sfreq[maxpos] + bonus);
        int cum = 0, last = 0;
        for (int pos = 0; pos < sd; ++pos) {
            const int s = ssym[pos];
            const int start = cum + s - last;
            if (slot < start) {  // an unseen symbol below s
                iv = {slot - cum + last, (uint32_t)(slot << shift), (uint32_t)(1 << shift)};
                sfreq[maxpos] = keep;
                return sparse_insert(pos, iv.sym, sc);
            }
            const int fr = sfreq[pos];
            if (start + fr > slot) {
                iv = {s, (uint32_t)(start << shift), (uint32_t)(fr << shift)};
                sfreq[maxpos] = keep;
                sfreq[pos] = (uint16_t)(sfreq[pos] + kSparseStep);
                sc.tot += kSparseStep;
                if (pos != maxpos && sfreq[pos] > sfreq[maxpos]) maxpos = pos;
                if (sc.tot + kSparseStep > kProbScale) sparse_halve(sc);
                return true;
            }
            cum += s - last + fr;
            last = s + 1;
        }
        sfreq[maxpos] = keep;
        iv = {last + slot - cum, (uint32_t)(slot << shift), (uint32_t)(1 << shift)};
        return sparse_insert(sd, iv.sym, sc);
    }
    int sparse_locate(int c, int tot0) const {
        const int shift = shift_for(tot0);
        const int bonus = (kProbScale - (tot0 << shift)) >> shift;
        int cum = 0, last = 0;
        for (int pos = 0; pos < sd; ++pos) {
            const int s = ssym[pos];
            if (c < s) return (cum + c - last) << shift;
            if (c == s) return (cum + s - last) << shift;
            cum += s - last + sfreq[pos] + (pos == maxpos ? bonus : 0);
            last = s + 1;
        }
        return (cum + c - last) << shift;
    }
    void sparse16_from_sparse4(const Payload& s4, int c, AnsScratch& sc) {  // Cx5.createFrom4
        cap = 16;
        std::memset(ssym, 0, sizeof ssym);
        std::memset(sfreq, 0, sizeof sfreq);
        maxpos = 0;  // the reference starts the wider context with maxpos 0, whatever it was before
        int i = 0, tot = 0;
        while (i < s4.sd && s4.ssym[i] < c) { ssym[i] = s4.ssym[i]; tot += sfreq[i] = s4.sfreq[i]; ++i; }
        int j = i;
        ssym[j] = (uint8_t)c;
        tot += sfreq[j] = kSparseStep;
        ++j;
        while (i < s4.sd) { ssym[j] = s4.ssym[i]; tot += sfreq[j] = s4.sfreq[i]; ++i; ++j; }
        sd = s4.sd + 1;
        if (tot > kProbScale) sparse_halve(sc);
        cached_tot = sparse_total();
    }

    // ---- table40 ------------------------------------------------------------------------
    void table_alloc(int capacity) {
        tcap = capacity;
        td = 0;
        std::memset(tsym, 0, sizeof tsym);
        std::memset(tfreq, 0, sizeof tfreq);
        std::memset(tcum, 0, sizeof tcum);
        std::memset(tcnt, 0, sizeof tcnt);
    }
    uint16_t& tsum() { return tcnt[tcap]; }
    void table_swap(int a, int b) {
        std::swap(tsym[a], tsym[b]);
        std::swap(tfreq[a], tfreq[b]);
        std::swap(tcum[a], tcum[b]);
        std::swap(tcnt[a], tcnt[b]);
    }
    void table_calc_sum() {
        const int sh = fshift > 0 ? fshift - 1 : 0;
        int sum = (256 - td) << sh;
        for (int i = 0; i < tcap; ++i) sum += tcnt[i];
        tsum() = (uint16_t)sum;
    }
    void table_rebuild(AnsScratch& sc) {  // Cx6.rescaleDec
        const int sh = fshift > 0 ? fshift - 1 : 0;
        for (int i = 0; i < 256; ++i) sc.c256[i] = (uint16_t)(1 << sh);
        for (int i = 0; i < td; ++i) sc.c256[tsym[i]] = tcnt[i];
        int cum = 0;
        for (int i = 0; i < 256; ++i) { sc.f512[2 * i] = sc.c256[i]; sc.f512[2 * i + 1] = (uint16_t)cum; cum += sc.c256[i]; }
        if (fshift > 0) --fshift;
        const int sh2 = fshift > 0 ? fshift - 1 : 0;
        int sum = (256 - td) << sh2;
        for (int i = 0; i < td; ++i) {
            tcnt[i] = (uint16_t)(tcnt[i] - (tcnt[i] >> 1));
            sum += tcnt[i];
            tfreq[i] = sc.f512[2 * tsym[i]];
            tcum[i] = sc.f512[2 * tsym[i] + 1];
        }
        tsum() = (uint16_t)sum;
    }
    void table_bump(int pos, AnsScratch& sc) {  // Cx6.incrCntDec
        const int step = kTableStep << fshift;
        tcnt[pos] = (uint16_t)(tcnt[pos] + step);
        tsum() = (uint16_t)(tsum() + step);
        if (pos > 0 && tcnt[pos] > tcnt[pos - 1]) table_swap(pos, pos - 1);
        if (tsum() + step > kProbScale) table_rebuild(sc);
    }
    int table_add(int c, int freq, int cum) {
        if (td >= 40 || td >= tcap) return -1;
        tsym[td] = (uint8_t)c;
        tfreq[td] = (uint16_t)freq;
        tcum[td] = (uint16_t)cum;
        tcnt[td] = (uint16_t)(freq - (freq >> 1));
        return td++;
    }
    // interval an unseen symbol c would get right now
    int table_unseen_cum(int c) const {
        int lower = -1, lfreq = 0, lcum = 0;
        for (int i = 0; i < td; ++i)
            if (tsym[i] > lower && tsym[i] < c) { lower = tsym[i]; lfreq = tfreq[i]; lcum = tcum[i]; }
        return lfreq > 0 ? lcum + lfreq + ((c - lower - 1) << fshift) : c << fshift;
    }
    void table_from_sparse16(const Payload& s, int c, AnsScratch& sc) {  // Cx6.createFrom5
        table_alloc(32);
        const int oldd = s.sd;
        const int shift = shift_for(s.sparse_total());
        int cum = 0, last = 0;
        for (int pos = 0; pos < oldd; ++pos) {
            const int sy = s.ssym[pos];
            cum += sy - last;
            const int fr = s.sfreq[pos] << shift;
            tsym[pos] = (uint8_t)sy;
            tfreq[pos] = (uint16_t)fr;
            tcum[pos] = (uint16_t)(cum << shift);
            tcnt[pos] = (uint16_t)(fr - (fr >> 1));
            cum += s.sfreq[pos];
            last = sy + 1;
        }
        td = oldd;
        fshift = shift;
        const int f = 1 << fshift;
        const int cf = c > 0 ? table_unseen_cum(c) : 0;
        tsym[oldd] = (uint8_t)c;
        tfreq[oldd] = (uint16_t)f;
        tcum[oldd] = (uint16_t)cf;
        tcnt[oldd] = (uint16_t)(f - (f >> 1));
        td = oldd + 1;
        const int step = kTableStep << fshift;
        tcnt[oldd] = (uint16_t)(tcnt[oldd] + step);
        tsum() = (uint16_t)(tsum() + step);
        if (tsum() + step > kProbScale) table_rebuild(sc);
        table_calc_sum();
        for (int i = 0; i < td - 1; ++i)  // most frequent first (exchange sort, as the reference)
            for (int j = i + 1; j < td; ++j)
                if (tfreq[j] > tfreq[i]) table_swap(i, j);
    }
    void table_from_list64(int c, AnsScratch& sc) {  // Cx6.createFrom2
        const int oldd = ld;
        table_alloc(oldd <= 32 ? 32 : 64);
        const int f0 = sc.f0;
        const int shift = shift_for(256 - oldd + oldd * f0 + f0);
        sort_bytes(list, oldd);
        int cum = 0, last = 0, at = 0;
        for (int pos = 0; pos < oldd; ++pos) {
            const int sy = list[pos];
            cum += sy - last;
            int cfr = f0;
            if (sy == c) { at = pos; cfr = 2 * f0; }
            const int fr = cfr << shift;
            tsym[pos] = (uint8_t)sy;
            tfreq[pos] = (uint16_t)fr;
            tcum[pos] = (uint16_t)(cum << shift);
            tcnt[pos] = (uint16_t)(fr - (fr >> 1));
            cum += cfr;
            last = sy + 1;
        }
        td = oldd;
        fshift = shift;
        table_calc_sum();
        if (at > 0) table_swap(0, at);  // the repeated symbol leads
    }
    bool table_take(int slot, AnsScratch& sc, Interval& iv) {  // Cx6.decode
        int lfreq = 0, lcum = 0, lower = 0;
        for (int i = 0; i < td; ++i) {
            const int cf = tcum[i];
            if (cf <= slot) {
                const int fr = tfreq[i];
                if (cf + fr > slot) {
                    iv = {tsym[i], (uint32_t)cf, (uint32_t)fr};
                    table_bump(i, sc);
                    return true;
                }
                if (cf >= lcum) { lfreq = fr; lcum = cf; lower = tsym[i]; }
            }
        }
        const int f = 1 << fshift;
        int c, cf;
        if (lfreq > 0) {
            const int x = (slot - (lcum + lfreq)) >> fshift;
            c = x + lower + 1;
            cf = lcum + lfreq + (x << fshift);
        } else {
            c = slot >> fshift;
            cf = c << fshift;
        }
        iv = {c, (uint32_t)cf, (uint32_t)f};
        int p = table_add(c, f, cf);
        if (p < 0) {
            if (tcap == 64) return false;
            tcap = 64;  // growDec: arrays double, the running sum moves to the new last slot
            tcnt[64] = tcnt[32];
            tcnt[32] = 0;
            p = table_add(c, f, cf);
        }
        table_bump(p, sc);
        return true;
    }
    int table_locate(int c) const {
        for (int i = 0; i < td; ++i)
            if (tsym[i] == c) return tcum[i];
        return c > 0 ? table_unseen_cum(c) : 0;
    }

    // ---- full ---------------------------------------------------------------------------
    void full_from_list256(int c) {  // Cx7.createFrom3
        full = std::make_unique<FixedModel>(256);
        auto& fc = full->fc();
        auto& cnt = full->cnt();
        for (int i = 0; i < 256; ++i) { fc[i].freq = 1; cnt[i] = 1; }
        const int d = ld;
        const int f0 = (kProbScale - (256 - d)) / (d + 1), c0 = f0 - (f0 >> 1);
        for (int i = 0; i < d; ++i) { fc[list[i]].freq = (uint16_t)f0; cnt[list[i]] = (uint16_t)c0; }
        fc[c].freq = (uint16_t)(fc[c].freq + f0);
        cnt[c] = (uint16_t)(cnt[c] + 16);
        int sum = 0, cf = 0;
        for (int i = 0; i < 256; ++i) {
            sum += cnt[i];
            fc[i].cum = (uint16_t)cf;
            full->mark(cf, fc[i].freq, i);
            cf += fc[i].freq;
        }
        full->sum() = sum;
    }
    void full_from_table() {  // Cx7.createFrom6
        full = std::make_unique<FixedModel>(256);
        auto& fc = full->fc();
        auto& cnt = full->cnt();
        full->sum() = tsum();
        for (int i = 0; i < tcap; ++i)
            if (tcnt[i] > 0) { fc[tsym[i]] = {tfreq[i], tcum[i]}; cnt[tsym[i]] = tcnt[i]; }
        const int f = 1 << fshift, cu = f - (f >> 1);
        int cf = 0;
        for (int i = 0; i < 256; ++i) {
            int fr;
            if (fc[i].freq > 0) fr = fc[i].freq;
            else { fc[i] = {(uint16_t)f, (uint16_t)cf}; cnt[i] = (uint16_t)cu; fr = f; }
            full->mark(cf, fr, i);
            cf += fr;
        }
    }
};

void ColourContext::learn(int c, AnsScratch& sc) {
    if (!p_) p_ = std::make_shared<Payload>();
    Payload& p = *p_;
    switch (stage_) {
        case Empty:
            p.ld = 1;
            p.list[0] = (uint8_t)(c < 0 ? 0 : c);
            stage_ = List14; ++g_stage_census[List14];
            break;
        case List14:
            switch (p.find_or_add(c, 14)) {
                case Payload::Found:
                    if (p.ld <= 4) { p.sparse_from_list(4, c); stage_ = Sparse4; ++g_stage_census[Sparse4]; }
                    else { p.sparse_from_list(16, c); p.cached_tot = p.sparse_total(); stage_ = Sparse16; ++g_stage_census[Sparse16]; }
                    break;
                case Payload::Added: break;
                case Payload::NoRoom: p.list[p.ld++] = (uint8_t)(c < 0 ? 0 : c); stage_ = List64; ++g_stage_census[List64]; break;
            }
            break;
        case List64:
            switch (p.find_or_add(c, 64)) {
                case Payload::Found: p.table_from_list64(c, sc); stage_ = Table40; ++g_stage_census[Table40]; break;
                case Payload::Added: break;
                case Payload::NoRoom: p.list[p.ld++] = (uint8_t)(c < 0 ? 0 : c); stage_ = List256; ++g_stage_census[List256]; break;
            }
            break;
        case List256:
            if (p.find_or_add(c, 256) == Payload::Found) { p.full_from_list256(c); stage_ = Full; ++g_stage_census[Full]; }
            break;
        default: break;  // coded stages never learn from raw bytes
    }
}

Interval ColourContext::take(int slot, AnsScratch& sc) {
    Payload& p = *p_;
    Interval iv{0, 0, 0};
    switch (stage_) {
        case Sparse4: {
            const int tot = p.sfreq[0] + p.sfreq[1] + p.sfreq[2] + p.sfreq[3] + 256 - p.sd;
            if (!p.sparse_take(slot, tot, sc, iv)) {
                Payload old = std::move(p);
                p = Payload{};
                p.sparse16_from_sparse4(old, iv.sym, sc);
                stage_ = Sparse16; ++g_stage_census[Sparse16];
            }
            break;
        }
        case Sparse16:
            if (!p.sparse_take(slot, p.cached_tot, sc, iv)) {
                p.cached_tot = sc.tot;
                Payload old = std::move(p);
                p = Payload{};
                p.table_from_sparse16(old, iv.sym, sc);
                stage_ = Table40; ++g_stage_census[Table40];
            } else
                p.cached_tot = sc.tot;
            break;
        case Table40:
            if (!p.table_take(slot, sc, iv)) { p.full_from_table(); stage_ = Full; ++g_stage_census[Full]; }
            break;
        case Full: iv = p.full->take(slot); break;
        default: break;
    }
    return iv;
}

int ColourContext::locate(int c, const AnsScratch&) const {
    const Payload& p = *p_;
    switch (stage_) {
        case Sparse4: return p.sparse_locate(c, p.sfreq[0] + p.sfreq[1] + p.sfreq[2] + p.sfreq[3] + 256 - p.sd);
        case Sparse16: return p.sparse_locate(c, p.cached_tot);
        case Table40: return p.table_locate(c);
        case Full: return p.full->locate(c);
        default: return 0;
    }
}

}  // namespace jsp::sp
