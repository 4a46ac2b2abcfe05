// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference vectors exist).
//
// CPU restatement of ScreenPressor.hx (file:line relative to /root/reference/src):
//   ScreenPressor.hx:53-64    constructor          :66-79   initEntro
//   ScreenPressor.hx:86-89    Preinit              :96-101  IsKeyFrame
//   ScreenPressor.hx:108-115  RenewI               :117-295 DecompressI
//   ScreenPressor.hx:302-484  DecompressP
// Frame buffers are int32 per pixel, exactly X*Y long here: a read outside [0, X*Y) yields
// `undefined` in the reference (stored back as 0, or NaN -> 0 inside the byte arithmetic of the
// gradient predictor); a write outside is dropped.
//
// Status codes of the C entry points: 0 zero_state, 2 error_occured (the reference returns it),
// 3 the reference would raise / never return (null entropy coder on a flat first frame, ptype
// outside its tables, context index outside the table, rANS renormalisation that cannot end).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "sp_entropy_oracle.h"

namespace {
using namespace orc;

struct Abort {};  // thrown where the reference raises

struct SP {
    int X, Y, bpp;
    int cx = 0, cx1 = 0;
    std::unique_ptr<EntroCoder> ec;
    int SC_CXSHIFT;
    int32_t* prev = nullptr;
    int nbx, nby;
    std::vector<int32_t> bts;
    int insignificant_blocks = 0;
    bool decodedI = false;
    bool last_flat_set = false;  // last_one_was_flat != null
    bool decodingBools = false;

    SP(int w, int h, int bits) : X(w), Y(h), bpp(bits) {
        SC_CXSHIFT = bpp == 16 ? 0 : 2;
        nbx = (X + 15) / 16;
        nby = (Y + 15) / 16;
        bts.assign((size_t)nbx * nby, 0);
    }
    bool initEntro(int version) {
        switch (version) {
            case 2: ec = make_entro_rc(); break;
            case 3: ec = make_entro_ans(64); SC_CXSHIFT = 2; break;
            case 4: ec = make_entro_ans(32); SC_CXSHIFT = 2; break;
            default: return false;
        }
        decodingBools = ec->canDecodeBool();
        ec->preinit();
        return true;
    }
    void preinit(int lines) { insignificant_blocks = nbx * ((lines + 15) / 16); }
    static int is_key(const uint8_t* d, size_t n) {
        if (!d || n == 0) return 0;
        int b = d[0];
        return b == 0x12 || b == 0x11 || b == 0x22 || b == 0x21 || b == 0x32 || b == 0x31;
    }
    void RenewI() {
        prev = nullptr;
        if (last_flat_set) return;
        if (!ec) throw Abort{};  // null.renewI()
        ec->renewI();
    }

    long end() const { return (long)X * Y; }
    int32_t rd(const int32_t* f, long i) const { return (i >= 0 && i < end()) ? f[i] : 0; }
    bool inb(long i) const { return i >= 0 && i < end(); }
    void wr(int32_t* f, long i, int32_t v) const { if (i >= 0 && i < end()) f[i] = v; }
    // gradient predictor on bytes 0..2: any operand outside the buffer makes that byte NaN -> 0
    int32_t gradient(const int32_t* f, long left, long up, long upleft) const {
        if (!inb(left) || !inb(up) || !inb(upleft)) return 0;
        uint32_t a = (uint32_t)f[left], b = (uint32_t)f[up], c = (uint32_t)f[upleft];
        uint32_t r = ((a & 0xFF) + (b & 0xFF) - (c & 0xFF)) & 0xFF;
        uint32_t g = (((a >> 8) & 0xFF) + ((b >> 8) & 0xFF) - ((c >> 8) & 0xFF)) & 0xFF;
        uint32_t bl = (((a >> 16) & 0xFF) + ((b >> 16) & 0xFF) - ((c >> 16) & 0xFF)) & 0xFF;
        return (int32_t)((bl << 16) + (g << 8) + r);
    }
    int clrctx(int cxi) {
        if (cxi < 0 || cxi >= 3 * 4096) throw Abort{};
        return ec->decodeClr(cxi);
    }
    // three components with the context chain of ScreenPressor.hx:173-189 ; -1 = undefined
    int32_t literal() {
        int r = clrctx(cx + cx1);
        cx1 = (cx << 6) & 0xFC0;
        cx = r < 0 ? 0 : r >> SC_CXSHIFT;
        int g = clrctx(4096 + cx + cx1);
        cx1 = (cx << 6) & 0xFC0;
        cx = g < 0 ? 0 : g >> SC_CXSHIFT;
        int b = clrctx(2 * 4096 + cx + cx1);
        cx1 = (cx << 6) & 0xFC0;
        cx = b < 0 ? 0 : b >> SC_CXSHIFT;
        if (r < 0) return 0;  // (b<<16)+(g<<8)+undefined = NaN, stored as 0
        return (int32_t)(((uint32_t)(b < 0 ? 0 : b) << 16) + ((uint32_t)(g < 0 ? 0 : g) << 8) + (uint32_t)r);
    }
    // The reference's loops make no progress while the coder keeps returning zero-length runs
    // (e.g. a range coder poisoned by reading past the end): it would spin for ever.
    int stall = 0;
    void progress(bool advanced) {
        if (advanced) stall = 0;
        else if (++stall > 65536) throw Abort{};
    }
    int decN(int ptype) { if (ptype < 0 || ptype >= 6) throw Abort{}; return ec->decodeN(ptype); }
    int decP(int ptype) { if (ptype < 0 || ptype >= 6) throw Abort{}; return ec->decodeP(ptype); }

    int decompressI(const uint8_t* srcp, size_t n, int32_t* dst) {
        ByteView src{srcp, n};
        long di = 0;
        const long e = end();
        int32_t clr = 0;
        long lasti = 0;
        int maskcx1 = 0xFC00, shiftcx1 = 4, shiftcx = 18;
        int head = src.at(0);
        if (head < 0) head = 0;  // undefined >> 4 and undefined & 0xF are 0
        int version = (head >> 4) + 1;
        if ((head & 0xF) == 1) {
            RenewI();
            int32_t c;
            if (bpp == 16) {
                int lo = src.at(0), hi = src.at(1);
                int clr16 = (lo < 0 || hi < 0) ? 0 : lo + hi * 256;  // NaN & mask = 0
                int b = (clr16 & 0x1F) << 3, g = ((clr16 >> 5) & 0x1F) << 3, r = ((clr16 >> 10) & 0x1F) << 3;
                c = (r << 16) + (g << 8) + b;
            } else {
                int b = src.at(1), g = src.at(2), r = src.at(3);
                // (r<<16)+(g<<8)+b : only an undefined b poisons the sum
                c = b < 0 ? 0 : (((r < 0 ? 0 : r) << 16) + ((g < 0 ? 0 : g) << 8) + b);
            }
            for (long i = 0; i < e; ++i) dst[i] = c;
            prev = dst;
            last_flat_set = true;
            decodedI = true;
            return 0;
        }
        last_flat_set = false;
        if ((head & 0xF) != 2) return 2;
        if (!ec && !initEntro(version)) return 2;
        RenewI();
        ec->decodeBegin(src, 1);
        cx = cx1 = 0;
        long k = 0;
        lasti = di;
        while (k < X + 1) {
            clr = literal();
            int nn = decN(0);
            k += nn;
            progress(nn > 0);
            while (nn-- > 0) { wr(dst, di, clr); ++di; }
            lasti = di - 1;
            if (ec->failed()) throw Abort{};
        }
        if (bpp == 16 && ec->differentConstantsFor16bbp()) { maskcx1 = 0xFF00; shiftcx1 = 2; shiftcx = 16; }
        const long off = -(long)X - 1;
        int ptype = 0;
        while (di < e) {
            ptype = decP(ptype);
            if (ptype == 0) clr = literal();
            int nn = decN(ptype);
            progress(nn > 0 && ptype != 3);
            switch (ptype) {
                case 0:
                    while (nn-- > 0) { wr(dst, di, clr); ++di; }
                    lasti = di - 1;
                    break;
                case 1:
                    while (nn-- > 0) { wr(dst, di, rd(dst, lasti)); lasti = di; ++di; }
                    clr = rd(dst, lasti);
                    break;
                case 2:
                    while (nn-- > 0) { clr = rd(dst, di + off + 1); wr(dst, di, clr); ++di; }
                    lasti = di - 1;
                    break;
                case 4:
                    while (nn-- > 0) {
                        clr = gradient(dst, lasti, di + off + 1, di + off);
                        wr(dst, di, clr);
                        lasti = di;
                        ++di;
                    }
                    break;
                case 5:
                    while (nn-- > 0) { clr = rd(dst, di + off); wr(dst, di, clr); ++di; }
                    lasti = di - 1;
                    break;
                default: break;  // 3: no case in the I-frame switch — nothing is written
            }
            cx1 = (clr & maskcx1) >> shiftcx1;
            cx = clr >> shiftcx;
            if (ec->failed()) throw Abort{};
        }
        prev = dst;
        decodedI = true;
        return 0;
    }

    int decompressP(const uint8_t* srcp, size_t n, int32_t* dst, int32_t** data_pnt, int* signif_out) {
        ByteView src{srcp, n};
        last_flat_set = false;
        *data_pnt = prev;
        *signif_out = 0;
        if (n == 0 || !decodedI) return 0;
        if (src.at(0) == 0) return 0;
        int maskcx1 = 0xFC00, shiftcx1 = 4, shiftcx = 18;
        if (!ec) throw Abort{};  // decodedI after a flat-only history: ec is still null
        if (ec->differentConstantsFor16bbp() && bpp == 16) { maskcx1 = 0xFF00; shiftcx1 = 2; shiftcx = 16; }
        ec->decodeBegin(src, 1);
        int t = ec->decodeX();
        int xx1 = ec->decodeX();
        xx1 = (xx1 << 8) + t;
        t = ec->decodeX();
        int xx2 = ec->decodeX();
        xx2 = (xx2 << 8) + t;
        std::fill(bts.begin(), bts.end(), 0);
        const long nb = (long)bts.size();
        long x = xx1;
        while (x <= xx2) {
            int bt = ec->decodeBT();
            int cnt = ec->decodeBN();
            for (int i = 0; i < cnt; ++i) { if (x >= 0 && x < nb) bts[x] = bt; ++x; }
            progress(cnt > 0);
            if (ec->failed()) throw Abort{};
        }
        bool signif = false;
        for (long i = insignificant_blocks < 0 ? 0 : insignificant_blocks; i < nb; ++i)
            if (bts[i] > 0) { signif = true; break; }
        const long stride = X;
        int32_t clr = 0;
        const long off = -(long)X - 1;
        cx = cx1 = 0;
        int lastmx = 0, lastmy = 0;
        for (int by = 0; by < nby; ++by)
            for (int bx = 0; bx < nbx; ++bx) {
                const int y16 = by * 16, x16 = bx * 16;
                int x1 = x16, x2 = x16 + 16, y1 = y16, y2 = y16 + 16;
                if (x2 > X) x2 = X;
                if (y2 > Y) y2 = Y;
                const int bt = bts[(size_t)by * nbx + bx];
                if (bt > 0) {
                    if (((bt - 1) & 1) > 0) {
                        if (!prev) throw Abort{};
                        for (int y = y1; y < y2; ++y) {
                            long i = (long)y * stride + x1;
                            for (int xx = 0; xx < x2 - x1; ++xx) wr(dst, i + xx, rd(prev, i + xx));
                        }
                        x1 = ec->decodeSXY(0) + x16;
                        y1 = ec->decodeSXY(1) + y16;
                        x2 = ec->decodeSXY(2) + x16 + 1;
                        y2 = ec->decodeSXY(3) + y16 + 1;
                    }
                    if (((bt - 1) & 2) > 0) {
                        int mx, my;
                        if (decodingBools && ec->decodeBool()) { mx = lastmx; my = lastmy; }
                        else { mx = ec->decodeMX() - 256; my = ec->decodeMY() - 256; }
                        lastmx = mx;
                        lastmy = my;
                        if (!prev) throw Abort{};
                        for (int y = y1; y < y2; ++y) {
                            long i = (long)y * stride + x1;
                            long j = (long)(y + my) * stride + (x1 + mx);
                            for (int xx = 0; xx < x2 - x1; ++xx) wr(dst, i + xx, rd(prev, j + xx));
                        }
                    } else {
                        int xx = x1, y = y1;
                        int ptype = 0;
                        while (y < y2) {
                            long i = (long)y * stride + xx;
                            long di = i;
                            ptype = decP(ptype);
                            if (ptype == 0) clr = literal();
                            int cnt = decN(ptype);
                            progress(cnt > 0);
                            for (int c = 0; c < cnt; ++c) {
                                switch (ptype) {
                                    case 1: clr = rd(dst, di - 1); break;
                                    case 2: clr = rd(dst, di + off + 1); break;
                                    case 3: if (!prev) throw Abort{}; clr = rd(prev, i); break;
                                    case 4: clr = gradient(dst, di - 1, di + off + 1, di + off); break;
                                    case 5: clr = rd(dst, di + off); break;
                                    default: break;
                                }
                                wr(dst, di, clr);
                                ++xx;
                                if (xx >= x2) { xx = x1; ++y; i = (long)y * stride + xx; di = i; }
                                else { ++i; ++di; }
                            }
                            cx1 = (clr & maskcx1) >> shiftcx1;
                            cx = clr >> shiftcx;
                            if (ec->failed()) throw Abort{};
                        }
                    }
                } else {
                    if (!prev) throw Abort{};
                    for (int y = y1; y < y2; ++y) {
                        long i = (long)y * stride + x1;
                        for (int xx = 0; xx < x2 - x1; ++xx) wr(dst, i + xx, rd(prev, i + xx));
                    }
                }
            }
        prev = dst;
        *data_pnt = prev;
        *signif_out = signif ? 1 : 0;
        return 0;
    }
};
}  // namespace

extern "C" {
void* orc_sp_create(int w, int h, int bpp) { return (w > 0 && h > 0) ? new SP(w, h, bpp) : nullptr; }
void orc_sp_destroy(void* c) { delete (SP*)c; }
void orc_sp_preinit(void* c, int lines) { ((SP*)c)->preinit(lines); }
int32_t* orc_sp_previous_frame(void* c) { return ((SP*)c)->prev; }
int orc_sp_is_key_frame(void*, const uint8_t* src, size_t n) { return SP::is_key(src, n); }
int orc_sp_decompress_i(void* c, const uint8_t* src, size_t n, int32_t* dst) {
    try { return ((SP*)c)->decompressI(src, n, dst); } catch (const Abort&) { return 3; }
}
int orc_sp_decompress_p(void* c, const uint8_t* src, size_t n, int32_t* dst, int32_t** data_pnt, int* signif) {
    try { return ((SP*)c)->decompressP(src, n, dst, data_pnt, signif); } catch (const Abort&) { return 3; }
}
int orc_sp_needs_index(void*) { return 0; }
}
