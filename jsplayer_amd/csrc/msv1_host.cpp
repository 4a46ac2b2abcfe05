// MSVideo1 host stage: walk the code stream once and emit one descriptor per 4x4 block.
// The walk is inherently sequential (a code's length depends on its own bytes,
// MSVideo1.hx:128-181 / 311-364); everything per-pixel happens on the GPU.
#include "msv1.h"

namespace jsp {

namespace {
// A byte read that may fall off the end of the stream.  The reference runs on JS typed
// arrays where such a read yields `undefined`; the callers below spell out what each
// expression of the reference makes of it (SURVEY.md 8a notes).
struct Src {
    const uint8_t* p;
    size_t n;
    bool has(uint64_t i) const { return i < n; }
};
}  // namespace

bool msv1_just_skip_blocks(const Msv1Geometry& geo, const uint8_t* src, size_t n) {
    // MSVideo1.hx:86-104
    long total = 0;
    for (size_t si = 0; si < n; si += 2) {
        if (si + 1 >= n) return false;  // high byte missing: not a skip code
        unsigned a = src[si], b = src[si + 1];
        if ((b & 0xFC) != 0x84) return false;
        total += (long)(((b - 0x84) << 8) + a);
        if (total >= geo.nblocks) return true;
    }
    return true;
}

void msv1_parse(const Msv1Geometry& geo, const uint8_t* src, size_t n, bool have_prev,
                size_t size_of_just_skips, int insignificant_blocks, uint32_t base,
                uint32_t* desc, std::vector<uint8_t>& block_changes, Msv1Parse& out) {
    out = Msv1Parse{};
    const bool is16 = geo.bits == 16;
    if (is16 && (n == 0 || (n < size_of_just_skips && msv1_just_skip_blocks(geo, src, n)))) {
        out.early_out = true;  // MSVideo1.hx:109-110
        return;
    }
    Src s{src, n};
    uint64_t si = 0;
    long skip = 0;  // may go to -1 and stay non-zero for the rest of the frame
    int blk = 0;
    bool stop = false;
    for (int by = 0; by < geo.nby && !stop; ++by) {
        block_changes[by] = 0;
        bool row_coded = false;
        for (int bx = 0; bx < geo.nbx; ++bx, ++blk) {
            if (skip != 0) {
                --skip;
                if (!have_prev) { out.aborted = true; stop = true; break; }
                desc[blk] = MSV1_DESC_SKIP;
                ++out.n_skipped;
                continue;
            }
            const bool a_ok = s.has(si), b_ok = s.has(si + 1);
            const unsigned a = a_ok ? src[si] : 0, b = b_ok ? src[si + 1] : 0;
            if (!is16 && b_ok && a == 0 && b == 0) {  // 8-bit end-of-data marker, MSVideo1.hx:313
                stop = true;
                break;
            }
            const uint64_t code_at = si;
            si += 2;
            if (b_ok && (b & 0xFC) == 0x84) {
                skip = (long)(((b - 0x84) << 8) + a) - 1;
                if (!have_prev) { out.aborted = true; stop = true; break; }
                desc[blk] = MSV1_DESC_SKIP;
                ++out.n_skipped;
                continue;
            }
            if (is16) {
                if (b_ok && b < 0x80) {
                    // 8-colour iff bit 15 of the first colour is set; a colour whose bytes are
                    // missing reads as NaN, whose bit 15 is clear
                    const bool eight = s.has(si + 1) && (src[si + 1] & 0x80);
                    si += eight ? 16 : 4;
                }
            } else {
                if (b_ok && b < 0x80) si += 2;
                else if (b_ok && b >= 0x90) si += 8;
            }
            desc[blk] = base + (uint32_t)code_at;
            ++out.n_coded;
            row_coded = true;
        }
        if (row_coded) { block_changes[by] = 1; out.changes = true; }
    }
    // blocks the walk never reached keep whatever dst held
    for (; blk < geo.nblocks; ++blk) { desc[blk] = MSV1_DESC_UNTOUCHED; ++out.n_untouched; }
    out.consumed = si < n ? si : n;
    if (out.aborted) return;
    if (out.changes)
        for (int i = insignificant_blocks < 0 ? 0 : insignificant_blocks; i < geo.nby; ++i)
            if (block_changes[i]) { out.s1 = true; break; }
}

int msv1_is_key_frame(const Msv1Geometry& geo, const uint8_t* src, size_t n) {
    // MSVideo1.hx:226-259 (16-bit: the first skip code answers "no") and :395-427 (8-bit: keeps
    // scanning, honours the a+b==0 terminator)
    if (n == 0) return 0;
    const bool is16 = geo.bits == 16;
    Src s{src, n};
    uint64_t si = 0;
    long skip = 0;
    bool key = true;
    for (int blk = 0; blk < geo.nblocks; ++blk) {
        if (skip != 0) { --skip; continue; }
        const bool b_ok = s.has(si + 1);
        const unsigned a = s.has(si) ? src[si] : 0, b = b_ok ? src[si + 1] : 0;
        if (!is16 && b_ok && a == 0 && b == 0) break;
        si += 2;
        if (b_ok && (b & 0xFC) == 0x84) {
            if (is16) return 0;
            skip = (long)(((b - 0x84) << 8) + a) - 1;
            key = false;
        } else if (b_ok && b < 0x80) {
            if (is16) si += (s.has(si + 1) && (src[si + 1] & 0x80)) ? 16 : 4;
            else si += 2;
        } else if (!is16 && b_ok && b >= 0x90)
            si += 8;
    }
    return key ? 1 : 0;
}

}  // namespace jsp
