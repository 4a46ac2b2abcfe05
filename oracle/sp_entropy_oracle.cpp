// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  See sp_entropy_oracle.h for the map of
// reference classes restated here.
#include "sp_entropy_oracle.h"

#include <algorithm>

namespace orc {

namespace {

constexpr int CXMAX = 4096;   // EntroCoders.hx:26-29
constexpr int NCXMAX = 6;
constexpr int MSR = 256;      // ScreenPressor.hx:21-22

// ================================ v2: EntroCoderRC (EntroCoders.hx:31-180) =======================
struct EntroRC final : EntroCoder {
    static constexpr uint32_t SC_STEP = 400, SC_NSTEP = 400, SC_BTSTEP = 10, SC_BTNSTEP = 20, SC_SXYSTEP = 100,
                              SC_MSTEP = 100, SC_UNSTEP = 1000, SC_XXSTEP = 1;
    static constexpr int ROW = 273;
    RangeDec rc;
    std::vector<uint32_t> cntab;
    uint32_t ptypetab[6][7];
    uint32_t ntab[NCXMAX][257];
    uint32_t xxtab[257], ntab2[257], bttab[6], sxytab[4][17], mvtab[2][2 * MSR + 1];

    EntroRC() : cntab((size_t)3 * CXMAX * ROW, 0u) {
        std::memset(ptypetab, 0, sizeof ptypetab);
        std::memset(ntab, 0, sizeof ntab);
        std::memset(xxtab, 0, sizeof xxtab);
        std::memset(ntab2, 0, sizeof ntab2);
        std::memset(bttab, 0, sizeof bttab);
        std::memset(sxytab, 0, sizeof sxytab);
        std::memset(mvtab, 0, sizeof mvtab);
    }
    bool differentConstantsFor16bbp() override { return true; }
    void preinit() override {  // :74-79
        for (int i = 0; i < 3 * CXMAX; ++i) cntab[(size_t)i * ROW + 16] = 0;
    }
    void renewI() override {  // :81-130
        for (int i = 0; i < 3 * CXMAX; ++i) {
            uint32_t* p = &cntab[(size_t)i * ROW];
            if (p[16] != 256) {
                for (int k = 0; k < 256; ++k) p[17 + k] = 1;
                for (int k = 0; k < 16; ++k) p[k] = 16;
                p[16] = 256;
            }
        }
        for (auto& t : ntab) { for (int i = 0; i < 256; ++i) t[i] = 1; t[256] = 256; }
        for (auto& t : ptypetab) { for (int i = 0; i < 6; ++i) t[i] = 1; t[6] = 6; }
        for (int i = 0; i < 256; ++i) xxtab[i] = ntab2[i] = 1;
        xxtab[256] = ntab2[256] = 256;
        for (int i = 0; i < 5; ++i) bttab[i] = 1;
        bttab[5] = 5;
        for (auto& t : sxytab) { for (int i = 0; i < 16; ++i) t[i] = 1; t[16] = 16; }
        for (auto& t : mvtab) { for (int i = 0; i < 2 * MSR; ++i) t[i] = 1; t[2 * MSR] = 2 * MSR; }
    }
    void decodeBegin(ByteView src, long pos0) override { rc.begin(src, pos0); }
    int decodeClr(int cxi) override {
        size_t off = (size_t)cxi * ROW;
        return rc.decode_uni(&cntab[off], SC_STEP, cntab.size() - off);
    }
    int decodeN(int ptype) override { return rc.decode_val(ntab[ptype], 256, SC_NSTEP); }
    int decodeP(int ptype) override { return rc.decode_val(ptypetab[ptype], 6, SC_UNSTEP); }
    int decodeX() override { return rc.decode_val(xxtab, 256, SC_XXSTEP); }
    int decodeBT() override { return rc.decode_val(bttab, 5, SC_BTSTEP); }
    int decodeBN() override { return rc.decode_val(ntab2, 256, SC_BTNSTEP); }
    int decodeSXY(int n) override { return rc.decode_val(sxytab[n], 16, SC_SXYSTEP); }
    int decodeMX() override { return rc.decode_val(mvtab[0], 2 * MSR, SC_MSTEP); }
    int decodeMY() override { return rc.decode_val(mvtab[1], 2 * MSR, SC_MSTEP); }
    bool canDecodeBool() override { return false; }
    bool decodeBool() override { return false; }
};

// ================================ v3/v4: rANS (ANS.hx) ===========================================
constexpr int PROB_SCALE = 4096;
constexpr int RANS_B = 131072;

struct Rans {  // ANS.hx:5-49
    int32_t r = 0;
    long pos = 0;
    ByteView data;
    void init(ByteView d, long i) {
        data = d;
        auto b = [&](long k) -> uint32_t { return d.has(k) ? d.p[k] : 0u; };  // undefined acts as 0 under |,<<
        r = (int32_t)(b(i) | (b(i + 1) << 8) | (b(i + 2) << 16) | (b(i + 3) << 24));
        pos = i + 4;
    }
    void reinit() { init(data, pos); }
    int decGet() const { return r & 4095; }
    bool hung = false;  // the reference's renormalisation loop would never end (corrupt state)
    void decAdvance(int start, int freq) {
        // int32 arithmetic as produced by the JS operators involved: * + - on doubles, then the
        // comparison; (x << 8) | byte wraps to int32
        double xd = (double)freq * (double)(r >> 12) + (double)(r & 4095) - (double)start;
        // all operands are < 2^31 in magnitude and freq <= 4096, so xd is exact
        int spins = 0;
        while (xd < 8388608.0) {
            if (++spins > 64) { hung = true; xd = 8388608.0; break; }
            int32_t xi = js_to_int32(xd);
            uint32_t byte = data.has(pos) ? data.p[pos] : 0u;
            ++pos;
            xi = (int32_t)(((uint32_t)xi << 8) | byte);
            xd = (double)xi;
        }
        r = js_to_int32(xd);
    }
    int raw() { int v = data.at(pos); ++pos; return v; }  // -1 = undefined
};

struct Rcv { int c = 0, freq = 0, cumFreq = 0; };

struct FixedCtx {  // FixedSizeRansCtx, ANS.hx:54-145
    static constexpr int STEP = 16, DSHIFT = 7, D = 1 << DSHIFT;
    std::vector<uint16_t> freqs, cnts;
    int cntsum = 0;
    uint8_t decTable[32];
    int NSym;
    explicit FixedCtx(int n) : freqs((size_t)n * 2, 0), cnts(n, 0), NSym(n) { std::memset(decTable, 0, sizeof decTable); }
    void setFreq(int i, int fr, int cf) { freqs[i * 2] = (uint16_t)fr; freqs[i * 2 + 1] = (uint16_t)cf; }
    void fillTable(int cf, int fr, int sym) {
        int k0 = (cf + D - 1) >> DSHIFT, k1 = ((cf + fr - 1) >> DSHIFT) + 1;
        for (int k = k0; k < k1; ++k)
            if (k >= 0 && k < 32) decTable[k] = (uint8_t)sym;  // Uint8Array(32): stores past the end are dropped
    }
    void incrCnt(int c) {  // :85-103
        cnts[c] = (uint16_t)(cnts[c] + STEP);
        cntsum += STEP;
        if (cntsum + STEP > PROB_SCALE) {
            cntsum = 0;
            int cf = 0;
            for (int j = 0; j < NSym; ++j) {
                int fr = cnts[j];
                setFreq(j, fr, cf);
                fillTable(cf, fr, j);
                cf += fr;
                cnts[j] = (uint16_t)(cnts[j] - (fr >> 1));
                cntsum += cnts[j];
            }
        }
    }
    bool decode(int someFreq, Rcv& rcv) {  // :105-126
        int c0 = decTable[someFreq >> DSHIFT];
        for (int j = c0; j < NSym - 1; ++j)
            if (freqs[(j + 1) * 2 + 1] > someFreq) {
                rcv.freq = freqs[j * 2];
                rcv.cumFreq = freqs[j * 2 + 1];
                rcv.c = j;
                incrCnt(j);
                return true;
            }
        rcv.freq = freqs[(NSym - 1) * 2];
        rcv.cumFreq = freqs[(NSym - 1) * 2 + 1];
        rcv.c = NSym - 1;
        incrCnt(NSym - 1);
        return true;
    }
    void renew() {  // :128-144
        int cf = 0, fr = PROB_SCALE / NSym, c0 = fr - (fr >> 1);
        cntsum = c0 * NSym;
        for (int i = 0; i < NSym; ++i) {
            setFreq(i, fr, cf);
            cnts[i] = (uint16_t)c0;
            fillTable(cf, fr, i);
            cf += fr;
        }
    }
};

enum class FindRes { Found, Added, NoRoom };

struct SymbList {  // ANS.hx:155-177 ; Cx1 (14) / Cx2 (64) / Cx3 (256)
    std::vector<uint8_t> symb;
    int d = 0;
    explicit SymbList(int n) : symb(n, 0) {}
    FindRes findOrAdd(int c) {  // c == -1 (undefined) never compares equal, is stored as 0
        for (int i = 0; i < d; ++i)
            if (c >= 0 && symb[i] == c) return FindRes::Found;
        if (d < (int)symb.size()) { symb[d] = (uint8_t)(c < 0 ? 0 : c); ++d; return FindRes::Added; }
        return FindRes::NoRoom;
    }
};

inline void insort(uint8_t* a, int n) {  // Sorter.insort, ANS.hx:862-872 (sorts the view in place)
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && a[j - 1] > a[j]; --j) std::swap(a[j], a[j - 1]);
}

struct AnsShared {     // the reference's statics, one set per coder instance
    int totFr = 0;     // SmallContext.totFr
    int f0 = 32;       // Cx6.f0
    uint16_t tmp_cnts[256];
    uint16_t tmp_freqs[512];
    Rcv rcv;           // Context.rcv
};

struct SmallCtx {  // SmallContext + Cx4 (S=4) + Cx5 (S=16), ANS.hx:210-392
    static constexpr int F0 = 50;
    int d = 0, maxpos = 0, S;
    std::vector<uint8_t> symbols;
    std::vector<uint16_t> freqs;
    int cntsum = 0;  // Cx5 only
    AnsShared* sh;
    SmallCtx(int size, AnsShared* s) : S(size), symbols(size, 0), freqs(size, 0), sh(s) {}

    void create(SymbList& c1, int c) {  // :226-238
        d = c1.d;
        insort(c1.symb.data(), d);
        for (int i = 0; i < d; ++i) {
            if (i < S) symbols[i] = c1.symb[i];
            int si = i < S ? symbols[i] : -2;  // reading past a typed array gives undefined
            if (si == c) { if (i < S) freqs[i] = 2 * F0; maxpos = i; }
            else if (i < S) freqs[i] = F0;
        }
    }
    bool addSymb(int pos, int c) {  // :240-252
        if (d == S) return false;
        for (int i = d - 1; i >= pos; --i) { symbols[i + 1] = symbols[i]; freqs[i + 1] = freqs[i]; }
        symbols[pos] = (uint8_t)c;
        freqs[pos] = F0;
        ++d;
        if (maxpos >= pos) ++maxpos;
        sh->totFr += F0;
        if (sh->totFr + F0 > PROB_SCALE) rescale();
        return true;
    }
    void rescale() {  // :254-261
        int s = 256 - d;
        for (int i = 0; i < d; ++i) { freqs[i] = (uint16_t)(freqs[i] - (freqs[i] >> 1)); s += freqs[i]; }
        sh->totFr = s;
    }
    uint16_t fr_at(int i) const { return (i >= 0 && i < S) ? freqs[i] : 0; }
    bool decodeSC(int someFreq, Rcv& rcv, int totFr0) {  // :263-309
        sh->totFr = totFr0;
        int shift = 0, tot = totFr0;
        while (tot <= PROB_SCALE / 2) { tot <<= 1; ++shift; }
        someFreq >>= shift;
        int bonus = (PROB_SCALE - tot) >> shift;
        const bool mp_ok = maxpos >= 0 && maxpos < S;
        int maxFreq = fr_at(maxpos);
        if (mp_ok) freqs[maxpos] = (uint16_t)(freqs[maxpos] + bonus);
        int cumFr = 0, lastSymb = 0, pos = 0;
        while (pos < d) {
            int s = symbols[pos];
            int startFr = cumFr + s - lastSymb;
            if (someFreq < startFr) {
                rcv.c = someFreq - cumFr + lastSymb;
                cumFr = someFreq;
                rcv.cumFreq = cumFr << shift;
                rcv.freq = 1 << shift;
                if (mp_ok) freqs[maxpos] = (uint16_t)maxFreq;
                return addSymb(pos, rcv.c);
            }
            int fr = freqs[pos];
            if (startFr + fr > someFreq) {
                rcv.c = s;
                cumFr += rcv.c - lastSymb;
                rcv.cumFreq = cumFr << shift;
                rcv.freq = fr << shift;
                if (mp_ok) freqs[maxpos] = (uint16_t)maxFreq;
                freqs[pos] = (uint16_t)(freqs[pos] + F0);
                sh->totFr += F0;
                if (pos != maxpos && freqs[pos] > fr_at(maxpos)) maxpos = pos;
                if (sh->totFr + F0 > PROB_SCALE) rescale();
                return true;
            }
            cumFr += s - lastSymb + fr;
            lastSymb = s + 1;
            ++pos;
        }
        if (mp_ok) freqs[maxpos] = (uint16_t)maxFreq;
        rcv.c = lastSymb + someFreq - cumFr;
        rcv.cumFreq = someFreq << shift;
        rcv.freq = 1 << shift;
        return addSymb(pos, rcv.c);
    }
    // Cx4.decode :319-322
    bool decode4(int someFreq, Rcv& rcv) {
        int tot = freqs[0] + freqs[1] + freqs[2] + freqs[3] + 256 - d;
        return decodeSC(someFreq, rcv, tot);
    }
    // Cx5 :329-392
    void calcSum() {
        int t = 256 - d;
        for (int i = 0; i < d; ++i) t += freqs[i];
        cntsum = t;
    }
    void createFrom4(const SmallCtx& c4, int c) {
        int i = 0, dd = c4.d, tot = 0;
        while (i < dd && c4.symbols[i] < c) { symbols[i] = c4.symbols[i]; tot += freqs[i] = c4.freqs[i]; ++i; }
        int j = i;
        symbols[j] = (uint8_t)c;
        tot += freqs[j] = F0;
        ++j;
        while (i < dd) { symbols[j] = c4.symbols[i]; tot += freqs[j] = c4.freqs[i]; ++i; ++j; }
        d = dd + 1;
        if (tot > PROB_SCALE) rescale();
        calcSum();
    }
    bool decode5(int someFreq, Rcv& rcv) {
        bool res = decodeSC(someFreq, rcv, cntsum);
        cntsum = sh->totFr;
        return res;
    }
};

struct Cx6 {  // ANS.hx:394-704
    static constexpr int STEP = 25;
    std::vector<uint8_t> symbols;
    std::vector<uint16_t> freqs, cnts;
    int d = 0, fshift = 0;
    AnsShared* sh;
    explicit Cx6(AnsShared* s) : sh(s) {}
    int S() const { return (int)symbols.size(); }
    void setFreq(int i, int fr, int cf) { freqs[i * 2] = (uint16_t)fr; freqs[i * 2 + 1] = (uint16_t)cf; }
    int readFreq(int i) const { return freqs[i * 2]; }
    int readCum(int i) const { return freqs[i * 2 + 1]; }
    void init(int n) { symbols.assign(n, 0); freqs.assign((size_t)n * 2, 0); cnts.assign(n + 1, 0); }

    void calcSum() {  // :571-578
        int shft = fshift > 0 ? fshift - 1 : 0;
        int sum = (256 - d) << shft;
        for (int i = 0; i < S(); ++i) sum += cnts[i];
        cnts[S()] = (uint16_t)sum;
    }
    void rescaleDec() {  // :580-604
        int sh0 = fshift > 0 ? fshift - 1 : 0, c0 = 1 << sh0;
        for (int i = 0; i < 256; ++i) sh->tmp_cnts[i] = (uint16_t)c0;
        for (int i = 0; i < d; ++i) sh->tmp_cnts[symbols[i]] = cnts[i];
        int cum = 0;
        for (int i = 0; i < 256; ++i) {
            sh->tmp_freqs[i * 2] = sh->tmp_cnts[i];
            sh->tmp_freqs[i * 2 + 1] = (uint16_t)cum;
            cum += sh->tmp_cnts[i];
        }
        if (fshift > 0) --fshift;
        int shft = fshift > 0 ? fshift - 1 : 0;
        int cntsum = (256 - d) << shft;
        for (int i = 0; i < d; ++i) {
            cnts[i] = (uint16_t)(cnts[i] - (cnts[i] >> 1));
            cntsum += cnts[i];
            int idx = symbols[i];
            setFreq(i, sh->tmp_freqs[idx * 2], sh->tmp_freqs[idx * 2 + 1]);
        }
        cnts[S()] = (uint16_t)cntsum;
    }
    void createFrom5(const SmallCtx& c5, int c) {  // :431-505
        init(32);
        const int Sz = 32;
        int oldd = c5.d, tot = 256 - oldd;
        for (int i = 0; i < oldd; ++i) tot += c5.freqs[i];
        int shift = 0, t = tot;
        while (t <= PROB_SCALE / 2) { t <<= 1; ++shift; }
        int cumFr = 0, lastSymb = 0;
        for (int pos = 0; pos < oldd; ++pos) {
            int s = c5.symbols[pos];
            cumFr += s - lastSymb;
            int cfr = c5.freqs[pos], fr = cfr << shift;
            setFreq(pos, fr, cumFr << shift);
            cnts[pos] = (uint16_t)(fr - (fr >> 1));
            symbols[pos] = (uint8_t)s;
            cumFr += cfr;
            lastSymb = s + 1;
        }
        fshift = shift;
        int fr_freq = 1 << fshift, fr_cum = 0;
        if (c > 0) {
            int lowerSym = -1, lfreq = 0, lcum = 0;
            for (int i = 0; i < oldd; ++i) {
                int s = symbols[i];
                if (s > lowerSym && s < c) { lowerSym = s; lfreq = readFreq(i); lcum = readCum(i); }
            }
            if (lfreq > 0) fr_cum = lcum + lfreq + ((c - lowerSym - 1) << fshift);
            else fr_cum = c << fshift;
        }
        setFreq(oldd, fr_freq, fr_cum);
        cnts[oldd] = (uint16_t)(fr_freq - (fr_freq >> 1));
        symbols[oldd] = (uint8_t)c;
        d = oldd + 1;
        int step = STEP << fshift;
        cnts[oldd] = (uint16_t)(cnts[oldd] + step);
        cnts[Sz] = (uint16_t)(cnts[Sz] + step);
        if (cnts[Sz] + step > PROB_SCALE) rescaleDec();
        calcSum();
        for (int i = 0; i < d - 1; ++i)
            for (int j = i + 1; j < d; ++j) {
                int fj = readFreq(j), fi = readFreq(i);
                if (fj > fi) {
                    int cfi = readCum(i), cfj = readCum(j);
                    setFreq(i, fj, cfj);
                    setFreq(j, fi, cfi);
                    std::swap(cnts[i], cnts[j]);
                    std::swap(symbols[i], symbols[j]);
                }
            }
    }
    void createFrom2(SymbList& cx, int c) {  // :507-555
        init(cx.d <= 32 ? 32 : 64);
        int f0 = sh->f0, oldd = cx.d;
        int tot = 256 - oldd + oldd * f0 + f0;
        int shift = 0, t = tot;
        while (t <= PROB_SCALE / 2) { t <<= 1; ++shift; }
        int cumFr = 0, cfr = 0, lastSymb = 0, newSymbPos = 0;
        insort(cx.symb.data(), oldd);
        for (int pos = 0; pos < oldd; ++pos) {
            int s = cx.symb[pos];
            cumFr += s - lastSymb;
            if (s == c) { newSymbPos = pos; cfr = f0 * 2; }
            else cfr = f0;
            int fr = cfr << shift;
            setFreq(pos, fr, cumFr << shift);
            symbols[pos] = (uint8_t)s;
            cnts[pos] = (uint16_t)(fr - (fr >> 1));
            cumFr += cfr;
            lastSymb = s + 1;
        }
        d = oldd;
        fshift = shift;
        calcSum();
        if (newSymbPos > 0) {
            int fr0 = readFreq(0), cf0 = readCum(0), frc = readFreq(newSymbPos), cfc = readCum(newSymbPos);
            setFreq(0, frc, cfc);
            setFreq(newSymbPos, fr0, cf0);
            uint8_t sym0 = symbols[0];
            std::swap(cnts[0], cnts[newSymbPos]);
            symbols[0] = (uint8_t)c;
            symbols[newSymbPos] = sym0;
        }
    }
    int addDec(int c, int freq, int cum) {  // :652-661
        if (d >= 40 || d >= S()) return -1;
        int pos = d;
        symbols[pos] = (uint8_t)c;
        setFreq(pos, freq, cum);
        cnts[pos] = (uint16_t)(freq - (freq >> 1));
        ++d;
        return pos;
    }
    void growDec() {  // :663-678
        int n = S() * 2;
        std::vector<uint8_t> sym(n, 0);
        std::vector<uint16_t> cs(n + 1, 0), fs((size_t)n * 2, 0);
        for (int i = 0; i < d; ++i) { sym[i] = symbols[i]; cs[i] = cnts[i]; fs[i * 2] = freqs[i * 2]; fs[i * 2 + 1] = freqs[i * 2 + 1]; }
        cs[n] = cnts[S()];
        symbols.swap(sym);
        cnts.swap(cs);
        freqs.swap(fs);
    }
    void incrCntDec(int pos) {  // :680-696
        int step = STEP << fshift, Sz = S();
        cnts[pos] = (uint16_t)(cnts[pos] + step);
        cnts[Sz] = (uint16_t)(cnts[Sz] + step);
        if (pos > 0 && cnts[pos] > cnts[pos - 1]) {
            std::swap(cnts[pos], cnts[pos - 1]);
            int fp = readFreq(pos), cfp = readCum(pos);
            setFreq(pos, readFreq(pos - 1), readCum(pos - 1));
            setFreq(pos - 1, fp, cfp);
            std::swap(symbols[pos], symbols[pos - 1]);
        }
        if (cnts[Sz] + step > PROB_SCALE) rescaleDec();
    }
    bool decode(int someFreq, Rcv& rcv) {  // :606-650
        int lfreq = 0, lcum = 0, lowerSym = 0;
        for (int i = 0; i < d; ++i) {
            int cf = readCum(i);
            if (cf <= someFreq) {
                int fr = readFreq(i);
                if (cf + fr > someFreq) {
                    rcv.c = symbols[i];
                    rcv.freq = fr;
                    rcv.cumFreq = cf;
                    incrCntDec(i);
                    return true;
                }
                if (cf >= lcum) { lfreq = fr; lcum = cf; lowerSym = symbols[i]; }
            }
        }
        int fr_freq = 1 << fshift, fr_cum = 0, c = 0;
        if (lfreq > 0) {
            int cumFr = lcum + lfreq;
            int x = (someFreq - cumFr) >> fshift;
            c = x + lowerSym + 1;
            fr_cum = lcum + lfreq + (x << fshift);
        } else {
            c = someFreq >> fshift;
            fr_cum = c << fshift;
        }
        rcv.freq = fr_freq;
        rcv.cumFreq = fr_cum;
        rcv.c = c;
        int p = addDec(c, fr_freq, fr_cum);
        if (p < 0) {
            if (S() == 64) return false;
            growDec();
            p = addDec(c, fr_freq, fr_cum);
        }
        incrCntDec(p);
        return true;
    }
};

struct Cx7 : FixedCtx {  // ANS.hx:706-772
    Cx7() : FixedCtx(256) {}
    void createFrom3(const SymbList& c3, int c) {
        for (int i = 0; i < 256; ++i) { freqs[i * 2] = 1; cnts[i] = 1; }
        int d = c3.d;
        int f0 = (PROB_SCALE - (256 - d)) / (d + 1);
        int c0 = f0 - (f0 >> 1);
        for (int i = 0; i < d; ++i) { int s = c3.symb[i]; freqs[s * 2] = (uint16_t)f0; cnts[s] = (uint16_t)c0; }
        freqs[c * 2] = (uint16_t)(freqs[c * 2] + f0);
        cnts[c] = (uint16_t)(cnts[c] + STEP);
        cntsum = 0;
        int cf = 0;
        for (int i = 0; i < 256; ++i) {
            cntsum += cnts[i];
            freqs[i * 2 + 1] = (uint16_t)cf;
            int fr = freqs[i * 2];
            fillTable(cf, fr, i);
            cf += fr;
        }
    }
    void createFrom6(const Cx6& c6, int /*c*/) {
        int Sz = c6.S();
        cntsum = c6.cnts[Sz];
        for (int i = 0; i < Sz; ++i)
            if (c6.cnts[i] > 0) {
                int x = c6.symbols[i];
                setFreq(x, c6.freqs[i * 2], c6.freqs[i * 2 + 1]);
                cnts[x] = c6.cnts[i];
            }
        int funmet = 1 << c6.fshift, cntUnmet = funmet - (funmet >> 1), cumFr = 0;
        for (int i = 0; i < 256; ++i) {
            int fr;
            if (freqs[i * 2] > 0) fr = freqs[i * 2];
            else { setFreq(i, funmet, cumFr); cnts[i] = (uint16_t)cntUnmet; fr = funmet; }
            fillTable(cumFr, fr, i);
            cumFr += fr;
        }
    }
};

struct Context {  // ANS.hx:785-860
    enum Kind { None, K1, K2, K3, K4, K5, K6, K7 } kind = None;
    std::unique_ptr<SymbList> list;   // K1..K3
    std::unique_ptr<SmallCtx> small;  // K4, K5
    std::unique_ptr<Cx6> c6;
    std::unique_ptr<Cx7> c7;
    void renew() { kind = None; list.reset(); small.reset(); c6.reset(); c7.reset(); }

    bool decode(int someFreq, AnsShared& sh) {
        Rcv& rcv = sh.rcv;
        switch (kind) {
            case K6:
                if (!c6->decode(someFreq, rcv)) {  // upgrade, :698-703
                    auto n = std::make_unique<Cx7>();
                    n->createFrom6(*c6, rcv.c);
                    c7 = std::move(n);
                    c6.reset();
                    kind = K7;
                }
                return true;
            case K7: c7->decode(someFreq, rcv); return true;
            case K4:
                if (!small->decode4(someFreq, rcv)) {  // Cx4.upgrade :324-326
                    auto n = std::make_unique<SmallCtx>(16, &sh);
                    n->createFrom4(*small, rcv.c);
                    small = std::move(n);
                    kind = K5;
                }
                return true;
            case K5:
                if (!small->decode5(someFreq, rcv)) {  // Cx5.upgrade :386-391
                    auto n = std::make_unique<Cx6>(&sh);
                    n->createFrom5(*small, rcv.c);
                    c6 = std::move(n);
                    small.reset();
                    kind = K6;
                }
                return true;
            default: return false;
        }
    }
    void update(int c, AnsShared& sh) {  // :812-859 ; c may be -1 (undefined)
        switch (kind) {
            case None:
                list = std::make_unique<SymbList>(14);
                list->d = 1;
                list->symb[0] = (uint8_t)(c < 0 ? 0 : c);
                kind = K1;
                break;
            case K1:
                switch (list->findOrAdd(c)) {
                    case FindRes::Found:
                        if (list->d <= 4) { small = std::make_unique<SmallCtx>(4, &sh); small->create(*list, c); kind = K4; }
                        else { small = std::make_unique<SmallCtx>(16, &sh); small->create(*list, c); small->calcSum(); kind = K5; }
                        list.reset();
                        break;
                    case FindRes::Added: break;
                    case FindRes::NoRoom: {  // Cx2(c1, c) :188-197
                        auto n = std::make_unique<SymbList>(64);
                        for (int i = 0; i < list->d; ++i) n->symb[i] = list->symb[i];
                        n->symb[list->d] = (uint8_t)(c < 0 ? 0 : c);
                        n->d = list->d + 1;
                        list = std::move(n);
                        kind = K2;
                        break;
                    }
                }
                break;
            case K2:
                switch (list->findOrAdd(c)) {
                    case FindRes::Found: {
                        auto n = std::make_unique<Cx6>(&sh);
                        n->createFrom2(*list, c);
                        c6 = std::move(n);
                        list.reset();
                        kind = K6;
                        break;
                    }
                    case FindRes::Added: break;
                    case FindRes::NoRoom: {  // Cx3(c2, c) :199-208
                        auto n = std::make_unique<SymbList>(256);
                        for (int i = 0; i < list->d; ++i) n->symb[i] = list->symb[i];
                        n->symb[list->d] = (uint8_t)(c < 0 ? 0 : c);
                        n->d = list->d + 1;
                        list = std::move(n);
                        kind = K3;
                        break;
                    }
                }
                break;
            case K3:
                if (list->findOrAdd(c) == FindRes::Found) {
                    auto n = std::make_unique<Cx7>();
                    n->createFrom3(*list, c);
                    c7 = std::move(n);
                    list.reset();
                    kind = K7;
                }
                break;
            default: break;  // "unexpected kind in Context.update"
        }
    }
};

struct EntroANS final : EntroCoder {  // EntroCoders.hx:182-313
    Rans rans;
    bool failed() override { return rans.hung; }
    int nDec = 0;
    AnsShared sh;
    std::vector<Context> cntab;
    std::vector<FixedCtx> ptypetab, ntab, sxytab, mvtab;
    FixedCtx xxtab{256}, ntab2{256}, bttab{5};
    Rcv myRcv;

    explicit EntroANS(int f0) : cntab((size_t)CXMAX * 3) {
        for (int i = 0; i < NCXMAX; ++i) ntab.emplace_back(256);
        for (int i = 0; i < 6; ++i) ptypetab.emplace_back(6);
        for (int i = 0; i < 4; ++i) sxytab.emplace_back(16);
        for (int i = 0; i < 2; ++i) mvtab.emplace_back(512);
        sh.f0 = f0;
    }
    void preinit() override {}
    bool differentConstantsFor16bbp() override { return false; }
    void renewI() override {
        for (auto& c : cntab) c.renew();
        for (auto& t : ntab) t.renew();
        for (auto& t : ptypetab) t.renew();
        xxtab.renew();
        ntab2.renew();
        bttab.renew();
        for (auto& t : sxytab) t.renew();
        for (auto& t : mvtab) t.renew();
    }
    void decodeBegin(ByteView src, long pos0) override { rans.init(src, pos0); nDec = 0; }
    void tick() { if (++nDec == RANS_B) { rans.reinit(); nDec = 0; } }
    int decodeClr(int cxi) override {
        Context& dcx = cntab[cxi];
        int c;
        if (dcx.decode(rans.decGet(), sh)) {
            c = sh.rcv.c;
            rans.decAdvance(sh.rcv.cumFreq, sh.rcv.freq);
        } else {
            c = rans.raw();
            dcx.update(c, sh);
        }
        tick();
        return c;
    }
    bool canDecodeBool() override { return true; }
    bool decodeBool() override {
        int f = rans.decGet();
        bool flag = f >= PROB_SCALE >> 1;
        rans.decAdvance(flag ? PROB_SCALE >> 1 : 0, PROB_SCALE >> 1);
        tick();
        return flag;
    }
    int decodeF(FixedCtx& dcx) {
        dcx.decode(rans.decGet(), myRcv);
        rans.decAdvance(myRcv.cumFreq, myRcv.freq);
        tick();
        return myRcv.c;
    }
    int decodeN(int ptype) override { return decodeF(ntab[ptype]); }
    int decodeP(int ptype) override { return decodeF(ptypetab[ptype]); }
    int decodeX() override { return decodeF(xxtab); }
    int decodeBT() override { return decodeF(bttab); }
    int decodeBN() override { return decodeF(ntab2); }
    int decodeSXY(int n) override { return decodeF(sxytab[n]); }
    int decodeMX() override { return decodeF(mvtab[0]); }
    int decodeMY() override { return decodeF(mvtab[1]); }
};

// Test hook (tests/test_js_semantics.py): drive the rANS state machine alone with an arbitrary list of operations, so
// that its int32 / out-of-range-read emulation can be held against a real JS engine.
//   op (start, freq): freq >= 0 decAdvance(start, freq); freq == -1 raw(); freq == -2 reinit()
// out[2k] = state r after op k (raw(): the byte or -1), out[2k+1] = pos; returns the number of ops done (stops at a hang)
int rans_trace(const uint8_t* src, size_t n, long pos0, const int32_t* ops, int nops, int64_t* out) {
    Rans rs;
    rs.init(ByteView{src, (long)n}, pos0);
    int k = 0;
    for (; k < nops; ++k) {
        const int32_t start = ops[2 * k], freq = ops[2 * k + 1];
        int64_t v;
        if (freq == -1) v = rs.raw();
        else if (freq == -2) { rs.reinit(); v = rs.r; }
        else { rs.decAdvance(start, freq); if (rs.hung) break; v = rs.r; }
        out[2 * k] = v;
        out[2 * k + 1] = rs.pos;
    }
    return k;
}

// Test hook (tests/test_js_semantics.py): decodeClr over a list of colour-context indices on an arbitrary byte stream,
// so that the whole model ladder (Context, Cx1..Cx7) can be held against a JS engine's typed arrays.
// out_syms[k] = symbol k (-1 = undefined); *out_pos = stream position afterwards; returns the calls done (stops at a hang)
int ans_clr_trace(int f0, const uint8_t* src, size_t n, long pos0, const int32_t* ctxs, int nctx, int32_t* out_syms, int64_t* out_pos) {
    EntroANS ec(f0);
    ec.renewI();
    ec.decodeBegin(ByteView{src, (long)n}, pos0);
    int k = 0;
    for (; k < nctx; ++k) {
        const int c = ec.decodeClr(ctxs[k]);
        if (ec.failed()) break;
        out_syms[k] = c;
    }
    *out_pos = ec.rans.pos;
    return k;
}

}  // namespace

std::unique_ptr<EntroCoder> make_entro_rc() { return std::make_unique<EntroRC>(); }
std::unique_ptr<EntroCoder> make_entro_ans(int f0) { return std::make_unique<EntroANS>(f0); }

}  // namespace orc

extern "C" int orc_ans_clr_trace(int f0, const uint8_t* src, size_t n, long pos0, const int32_t* ctxs, int nctx, int32_t* out_syms, int64_t* out_pos) {
    return orc::ans_clr_trace(f0, src, n, pos0, ctxs, nctx, out_syms, out_pos);
}
extern "C" int orc_rans_trace(const uint8_t* src, size_t n, long pos0, const int32_t* ops, int nops, int64_t* out) {
    return orc::rans_trace(src, n, pos0, ops, nops, out);
}
