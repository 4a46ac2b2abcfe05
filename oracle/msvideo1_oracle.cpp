// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of jsplayer's Microsoft Video 1 decoders, used as the checker
// for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
// Nothing under jsplayer_amd/ may include, link or call this file.
//
// PARITY UNPINNED: the reference ships no tests, no golden vectors, no sample
// media and no encoder, and its Haxe/OpenFL toolchain is not
// available, so this restatement cannot be checked against reference output.
// It is pinned only by (a) the publicly documented CRAM bit layout, (b) the
// hand-worked known-answer vectors in tests/golden/, and (c) agreement with the
// independently written host/HIP path.
//
// Follows (file:line relative to /root/reference/src):
//   MSVideo1.hx:20-31   constructor, size_of_just_skips
//   MSVideo1.hx:37-41   Preinit (16-bit)        MSVideo1.hx:281-291 Preinit (8-bit)
//   MSVideo1.hx:86-104  JustSkipBlocks
//   MSVideo1.hx:106-209 MSVideo1_16bit.DecompressP
//   MSVideo1.hx:293-393 MSVideo1_8bit.DecompressP
//   MSVideo1.hx:226-259 / 395-427 IsKeyFrame
// JS semantics reproduced on purpose (SURVEY.md §8a notes, Appendix D):
//   * a typed-array read past the end yields `undefined`; arithmetic on it gives
//     NaN, bit operations on NaN give 0, comparisons with NaN are false;
//   * `block_changes` persists across calls (rows after an early exit keep
//     their old value);
//   * the 8-bit class never sets insign_lines, so its stage-2 compare loop runs
//     zero times (NaN loop bound);
//   * a skip block while prevFrame is null raises a TypeError that the
//     reference's catch clauses do not match: the call aborts (reported here
//     as status 2 with dst written up to that block).
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>

namespace {

constexpr int UNDEF = -1;  // models JS `undefined` for a byte read

struct Bytes {
    const uint8_t* p;
    size_t n;
    int at(long i) const { return (i >= 0 && (size_t)i < n) ? p[i] : UNDEF; }
    // src[i] + src[i+1]*256 ; NaN (reported as -1) when either byte is missing
    int le16(long i) const {
        int lo = at(i), hi = at(i + 1);
        if (lo == UNDEF || hi == UNDEF) return UNDEF;
        return lo + hi * 256;
    }
};

inline int32_t rgb555_to_rgb32(int c) {  // MSVideo1.hx:211-214 ; NaN -> 0
    if (c == UNDEF) return 0;
    return ((c & 0x1F) << 3) + ((c & 0x3E0) << 6) + ((c & 0x7C00) << 9);
}

struct Msv1 {
    int bits;  // 16 or 8
    int X, Y;
    int nbx, nby;
    std::vector<uint8_t> block_changes;  // persistent, one per block row
    int insignificant_blocks = 0;
    bool insign_lines_set = false;  // 8-bit Preinit leaves it undefined
    int insign_lines = 0;
    size_t size_of_just_skips;
    int32_t* prev = nullptr;
    int32_t pal[256];
    std::vector<uint8_t> pal_bytes;

    Msv1(int bits_, int w, int h, const uint8_t* palette, int palette_len) : bits(bits_), X(w), Y(h) {
        nbx = X >> 2;
        nby = Y >> 2;
        block_changes.assign(nby > 0 ? nby : 0, 0);
        long nblocks = (long)nbx * nby;
        size_of_just_skips = (size_t)(nblocks / 1023) * 2 + 10;
        std::memset(pal, 0, sizeof pal);
        if (palette && palette_len > 0) pal_bytes.assign(palette, palette + palette_len);
    }

    void preinit(int lines) {
        insignificant_blocks = (lines + 3) >> 2;
        if (bits == 16) {
            insign_lines = lines;
            insign_lines_set = true;
        } else {
            // little-endian unsigned reads, 4 bytes at a time, while >=4 remain
            size_t avail = pal_bytes.size(), pos = 0;
            int i = 0;
            while (i < 256 && avail - pos >= 4) {
                uint32_t v = (uint32_t)pal_bytes[pos] | ((uint32_t)pal_bytes[pos + 1] << 8) |
                             ((uint32_t)pal_bytes[pos + 2] << 16) | ((uint32_t)pal_bytes[pos + 3] << 24);
                pal[i++] = (int32_t)v;
                pos += 4;
            }
        }
    }

    int32_t pal_lookup(int idx) const { return idx == UNDEF ? 0 : pal[idx & 0xFF]; }

    bool just_skip_blocks(const Bytes& s) const {
        long nblocks = (long)nbx * nby, n = 0;
        for (size_t si = 0; si < s.n; si += 2) {
            int a = s.at(si), b = s.at(si + 1);
            if (b != UNDEF && (b & 0xFC) == 0x84) {
                n += ((b - 0x84) << 8) + a;
                if (n >= nblocks) return true;
            } else
                return false;
        }
        return true;
    }

    // returns false when the JS TypeError (null prevFrame) aborts the call
    bool copy_block(long di, int32_t* dst) const {
        if (!prev) return false;
        for (int y = 0; y < 4; ++y)
            for (int x = 0; x < 4; ++x) dst[di + y * (long)X + x] = prev[di + y * (long)X + x];
        return true;
    }

    static void paint(int32_t* dst, long di, int X, const int32_t* colours, unsigned flags, bool eight) {
        for (int y = 0; y < 4; ++y) {
            int quad_y = (y & 2) << 1;
            for (int x = 0; x < 4; ++x) {
                int sel = eight ? quad_y + (x & 2) + (int)(flags & 1) : (int)(flags & 1);
                dst[di + y * (long)X + x] = colours[sel];
                flags >>= 1;
            }
        }
    }

    // status: 0 ok, 2 aborted by the uncaught TypeError
    int decompress_p(const uint8_t* src, size_t n, int32_t* dst, int32_t** data_pnt, int* signif_out) {
        Bytes s{src, n};
        *signif_out = 0;
        *data_pnt = prev;
        if (bits == 16) {
            if (n == 0 || (n < size_of_just_skips && just_skip_blocks(s))) return 0;
        }
        long si = 0;
        long skip = 0;
        bool changes = false;
        bool stop = false;  // 8-bit end-of-data marker
        for (int by = 0; by < nby && !stop; ++by) {
            block_changes[by] = 0;
            for (int bx = 0; bx < nbx; ++bx) {
                long di = (long)by * X * 4 + bx * 4;
                if (skip != 0) {
                    --skip;
                    if (!copy_block(di, dst)) return 2;
                    continue;
                }
                int a = s.at(si), b = s.at(si + 1);
                if (bits == 8 && a == 0 && b == 0) {  // a + b == 0 (NaN never equals 0)
                    stop = true;
                    break;
                }
                si += 2;
                if (b != UNDEF && (b & 0xFC) == 0x84) {
                    skip = (long)(((b - 0x84) << 8) + a) - 1;
                    if (!copy_block(di, dst)) return 2;
                    continue;
                }
                int32_t c[8];
                if (bits == 16) {
                    if (b != UNDEF && b < 0x80) {
                        unsigned flags = (unsigned)(((b << 8) + a) ^ 0xFFFF);
                        int clr0 = s.le16(si);
                        c[0] = rgb555_to_rgb32(clr0);
                        c[1] = rgb555_to_rgb32(s.le16(si + 2));
                        si += 4;
                        if (clr0 != UNDEF && (clr0 & 0x8000) != 0) {
                            for (int k = 0; k < 6; ++k) c[2 + k] = rgb555_to_rgb32(s.le16(si + 2 * k));
                            si += 12;
                            paint(dst, di, X, c, flags, true);
                        } else
                            paint(dst, di, X, c, flags, false);
                    } else {
                        // (b << 8) + a : undefined<<8 is 0, 0 + undefined is NaN
                        int v = (a == UNDEF) ? UNDEF : ((b == UNDEF ? 0 : (b << 8)) + a);
                        c[0] = c[1] = rgb555_to_rgb32(v);
                        paint(dst, di, X, c, 0, false);
                    }
                } else {
                    if (b != UNDEF && b < 0x80) {
                        unsigned flags = (unsigned)((b << 8) + a);
                        c[1] = pal_lookup(s.at(si));
                        c[0] = pal_lookup(s.at(si + 1));
                        si += 2;
                        paint(dst, di, X, c, flags, false);
                    } else if (b != UNDEF && b >= 0x90) {
                        unsigned flags = (unsigned)(((b << 8) + a) ^ 0xFFFF);
                        for (int k = 0; k < 8; ++k) c[k] = pal_lookup(s.at(si + k));
                        si += 8;
                        paint(dst, di, X, c, flags, true);
                    } else {
                        c[0] = c[1] = pal_lookup(a);
                        paint(dst, di, X, c, 0, false);
                    }
                }
                changes = true;
                block_changes[by] = 1;
            }
        }
        bool signif = false;
        if (changes)
            for (int i = insignificant_blocks < 0 ? 0 : insignificant_blocks; i < nby; ++i)
                if (block_changes[i]) {
                    signif = true;
                    break;
                }
        if (signif && prev) {
            signif = false;
            if (insign_lines_set) {
                long lo = (long)insign_lines * X, hi = (long)X * Y;
                for (long i = lo < 0 ? 0 : lo; i < hi; ++i)
                    if (dst[i] != prev[i]) {
                        signif = true;
                        break;
                    }
            }
        }
        if (changes) prev = dst;
        *data_pnt = prev;
        *signif_out = signif ? 1 : 0;
        return 0;
    }

    int is_key_frame(const uint8_t* src, size_t n) const {
        if (n == 0) return 0;
        Bytes s{src, n};
        long si = 0, skip = 0;
        bool key = true;
        for (int by = 0; by < nby; ++by)
            for (int bx = 0; bx < nbx; ++bx) {
                if (skip != 0) {
                    --skip;
                    continue;
                }
                int a = s.at(si), b = s.at(si + 1);
                if (bits == 8 && a == 0 && b == 0) return key ? 1 : 0;
                si += 2;
                if (b != UNDEF && (b & 0xFC) == 0x84) {
                    if (bits == 16) return 0;
                    skip = (long)(((b - 0x84) << 8) + a) - 1;
                    key = false;
                } else if (b != UNDEF && b < 0x80) {
                    if (bits == 16) {
                        int clr0 = s.le16(si);
                        si += (clr0 != UNDEF && (clr0 & 0x8000) != 0) ? 16 : 4;
                    } else
                        si += 2;
                } else if (bits == 8 && b != UNDEF && b >= 0x90)
                    si += 8;
            }
        return key ? 1 : 0;
    }
};

}  // namespace

extern "C" {

void* orc_msv1_create(int bits, int w, int h, const uint8_t* palette, int palette_bytes) {
    if ((bits != 16 && bits != 8) || w < 0 || h < 0) return nullptr;
    return new Msv1(bits, w, h, palette, palette_bytes);
}
void orc_msv1_destroy(void* c) { delete (Msv1*)c; }
void orc_msv1_preinit(void* c, int lines) { ((Msv1*)c)->preinit(lines); }
int32_t* orc_msv1_previous_frame(void* c) { return ((Msv1*)c)->prev; }
int orc_msv1_is_key_frame(void* c, const uint8_t* src, size_t n) { return ((Msv1*)c)->is_key_frame(src, n); }
int orc_msv1_decompress_p(void* c, const uint8_t* src, size_t n, int32_t* dst, int32_t** data_pnt, int* signif) {
    return ((Msv1*)c)->decompress_p(src, n, dst, data_pnt, signif);
}
// DecompressI just runs DecompressP and reports zero_state (MSVideo1.hx:62-67)
int orc_msv1_decompress_i(void* c, const uint8_t* src, size_t n, int32_t* dst) {
    int32_t* dp;
    int sg;
    return ((Msv1*)c)->decompress_p(src, n, dst, &dp, &sg);
}
int orc_msv1_needs_index(void*) { return 1; }

}  // extern "C"
