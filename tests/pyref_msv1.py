"""Second, independent restatement of the MSVideo1 block layout for VALID streams, written in
plain Python straight from the documented CRAM format — no JS edge
semantics, no shared code with oracle/.  Used to cross-check the C++ oracle on small frames."""
import numpy as np


def rgb555(c):
    return ((c >> 10) & 31) << 19 | ((c >> 5) & 31) << 11 | (c & 31) << 3


def decode(bits, w, h, src, prev=None, palette=None, dst=None):
    """Returns (frame, coded_any, skip_codes_seen).  `palette` = list of 256 ints for 8-bit."""
    src = bytes(src)
    out = np.array(dst, dtype=np.int64).reshape(h, w).copy() if dst is not None else np.zeros((h, w), np.int64)
    si, skip, coded, nskips = 0, 0, False, 0
    for by in range(h // 4):
        for bx in range(w // 4):
            y0, x0 = by * 4, bx * 4
            if skip:
                skip -= 1
                out[y0:y0 + 4, x0:x0 + 4] = prev[y0:y0 + 4, x0:x0 + 4]
                continue
            a, b = src[si], src[si + 1]
            if bits == 8 and a == 0 and b == 0:
                return out, coded, nskips
            si += 2
            if 0x84 <= b <= 0x87:
                skip = ((b - 0x84) << 8) + a - 1
                nskips += 1
                out[y0:y0 + 4, x0:x0 + 4] = prev[y0:y0 + 4, x0:x0 + 4]
                continue
            coded = True
            flags = (b << 8) | a
            if bits == 16:
                if b < 0x80:
                    c0 = src[si] | src[si + 1] << 8
                    ncol = 8 if c0 & 0x8000 else 2
                    cols = [rgb555(src[si + 2 * k] | src[si + 2 * k + 1] << 8) for k in range(ncol)]
                    si += 2 * ncol
                    for y in range(4):
                        for x in range(4):
                            bit = (flags >> (4 * y + x)) & 1   # set => FIRST colour of the pair
                            q = (((y & 2) << 1) + (x & 2)) if ncol == 8 else 0
                            out[y0 + y, x0 + x] = cols[q + (0 if bit else 1)]
                else:
                    out[y0:y0 + 4, x0:x0 + 4] = rgb555(flags)
            else:
                if b < 0x80:
                    i0, i1 = src[si], src[si + 1]
                    si += 2
                    for y in range(4):
                        for x in range(4):
                            bit = (flags >> (4 * y + x)) & 1   # set => first index
                            out[y0 + y, x0 + x] = palette[i0 if bit else i1]
                elif b >= 0x90:
                    idx = list(src[si:si + 8])
                    si += 8
                    for y in range(4):
                        for x in range(4):
                            bit = (flags >> (4 * y + x)) & 1   # set => first index of the pair
                            q = ((y & 2) << 1) + (x & 2)
                            out[y0 + y, x0 + x] = palette[idx[q + (0 if bit else 1)]]
                else:
                    out[y0:y0 + 4, x0:x0 + 4] = palette[a]
    return out, coded, nskips


def palette_ints(pal_bytes):
    p = np.frombuffer(bytes(pal_bytes), dtype="<u4")
    out = [0] * 256
    for i, v in enumerate(p[:256]):
        out[i] = int(v)
    return out
