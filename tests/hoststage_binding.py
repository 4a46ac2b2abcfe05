"""Test-only access to the product's ScreenPressor HOST stage (tests/hoststage/shim.cpp) plus numpy
re-statements of what the HIP kernels do with its descriptor tables, so the tables can be validated
without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(HERE, "hoststage", "libhoststage.so")
_lib = None

RUN_CONST, RUN_ABOVE, RUN_ABOVE_LEFT = 0, 1, 3
TILE_ABOVE_LEFT = 2   # the model's own label for the third kind of a tile row (its place among the row's records tells a record's kind)
PB_SUBRECT, PB_MOTION, PB_DATA = 1, 2, 4
KIND_NONE, KIND_FLAT, KIND_INTRA, KIND_INTER = 0, 1, 2, 3


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "hoststage")])
        L = C.CDLL(_PATH)
        L.hs_create.restype = C.c_void_p
        L.hs_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.hs_destroy.argtypes = [C.c_void_p]
        L.hs_preinit.argtypes = [C.c_void_p, C.c_int]
        L.hs_decode.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_void_p]
        L.hs_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_set_band_rows.argtypes = [C.c_void_p, C.c_int]
        L.hs_error.restype = C.c_char_p
        L.hs_error.argtypes = [C.c_void_p]
        L.hs_literalise_motion.argtypes = [C.c_void_p, C.c_void_p]
        L.hs_set_iframe_layout.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.hs_decode_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.hs_select.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hs_tile_words.restype = C.c_size_t
        L.hs_tile_words.argtypes = [C.c_void_p, C.c_int]
        L.hs_fetch_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hs_seed_words.restype = C.c_size_t
        L.hs_seed_words.argtypes = [C.c_void_p]
        L.hs_fetch_seeds.argtypes = [C.c_void_p, C.c_void_p]
        L.hs_set_dst_column.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


class HostStage:
    def __init__(self, w, h, bpp):
        self.L = lib()
        self.X, self.Y = w, h
        self.h = self.L.hs_create(w, h, bpp)

    def preinit(self, lines):
        self.L.hs_preinit(self.h, lines)

    def set_band_rows(self, rows):
        self.band_rows, self.span_px = rows, 0
        self.L.hs_set_band_rows(self.h, rows)

    def set_iframe_layout(self, rows, span):
        self.band_rows, self.span_px = rows, span
        self.L.hs_set_iframe_layout(self.h, rows, span)

    def decode_batch(self, frames, keys, threads: int, literalise: bool = False):
        """HostDecoder-level decode_frames(): the frames of a stream, groups of pictures side by side; one dict per frame as
        decode() returns (plus `literalised`)."""
        n = len(frames)
        keep = [bytes(f) for f in frames]
        srcs = (C.c_char_p * n)(*keep)
        lens = (C.c_size_t * n)(*[len(f) for f in keep])
        self.L.hs_decode_batch(self.h, n, srcs, lens, bytes(bytearray(1 if k else 0 for k in keys)), threads, 1 if literalise else 0)
        outs = []
        for i in range(n):
            meta = np.zeros(16, dtype=np.uint64)
            status = self.L.hs_select(self.h, i, meta.ctypes.data)
            d = self._fetch(status, [int(v) for v in meta])
            d["literalised"] = bool(meta[12])
            outs.append(d)
        return outs

    def set_dst_column(self, column):
        """What the destination of the following frames holds in its last column (int32 per row; None: not known) —
        HostDecoder::set_destination_column."""
        if column is None:
            self.L.hs_set_dst_column(self.h, None, 0)
        else:
            col = np.ascontiguousarray(column, dtype=np.int32)
            self.L.hs_set_dst_column(self.h, col.ctypes.data, int(col.size))

    def decode(self, key: bool, src: bytes):
        meta = np.zeros(12, dtype=np.uint64)
        src = bytes(src)
        status = self.L.hs_decode(self.h, 1 if key else 0, src, len(src), meta.ctypes.data)
        return self._fetch(status, [int(v) for v in meta])

    def _fetch(self, status, m):
        out = dict(status=status, error=self.L.hs_error(self.h).decode(), kind=m[0], adopted=bool(m[1]), significant=bool(m[2]), prev_cleared=bool(m[3]),
                   flat_colour=m[8], prev_pixels=m[9], data_pixels=m[10], stream_bytes=m[11])
        runs = np.zeros((m[4], 2), dtype=np.uint32)
        rows = np.zeros(m[5], dtype=np.uint32)
        blocks = np.zeros((m[6], 16), dtype=np.uint8)
        payload = np.zeros(m[7], dtype=np.uint32)
        self.L.hs_fetch(self.h, runs.ctypes.data, rows.ctypes.data, blocks.ctypes.data, payload.ctypes.data)
        seeds = np.zeros(self.L.hs_seed_words(self.h), dtype=np.uint32)
        self.L.hs_fetch_seeds(self.h, seeds.ctypes.data)
        tile_idx = np.zeros(self.L.hs_tile_words(self.h, 0), dtype=np.uint32)
        left = np.zeros(self.L.hs_tile_words(self.h, 1), dtype=np.uint32)
        self.L.hs_fetch_tiles(self.h, tile_idx.ctypes.data, left.ctypes.data)
        out.update(runs=runs, rows=rows, blocks=blocks, payload=payload, seeds=seeds,
                   band_rows=getattr(self, "band_rows", 0), span_px=getattr(self, "span_px", 0),
                   tile_idx=tile_idx, left=left)
        return out

    def literalise_motion(self, desc):
        """HostDecoder::literalise_motion on the frame just decoded: new block table + payload."""
        meta = np.zeros(12, dtype=np.uint64)
        self.L.hs_literalise_motion(self.h, meta.ctypes.data)
        blocks = np.zeros((int(meta[6]), 16), dtype=np.uint8)
        payload = np.zeros(int(meta[7]), dtype=np.uint32)
        self.L.hs_fetch(self.h, None, None, blocks.ctypes.data, payload.ctypes.data)
        out = dict(desc)
        out.update(blocks=blocks, payload=payload)
        return out

    def close(self):
        if self.h:
            self.L.hs_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _add_bytes(u, d):
    u = u.astype(np.uint32)
    return ((((u & 0x00FF00FF) + (d & 0x00FF00FF)) & 0x00FF00FF) | (((u & 0xFF00) + (d & 0xFF00)) & 0xFF00)).astype(np.uint32)


def expand_iframe(desc, X, Y):
    """What sp_iframe_rows_kernel computes from the run table (row wavefront).  With band seeds every
    band is expanded on its own, from nothing but its seed row — as its workgroup does."""
    if desc["kind"] == KIND_FLAT:
        return np.full(X * Y, desc["flat_colour"], dtype=np.uint32)
    runs, rows = desc["runs"], desc["rows"]
    starts, words = runs[:, 0].astype(np.int64), runs[:, 1]
    out = np.zeros((Y, X), dtype=np.uint32)
    xs = np.arange(X)
    band_rows = desc.get("band_rows", 0) if len(desc.get("seeds", ())) else 0
    seeds = desc.get("seeds")
    for y in range(Y):
        r0, r1 = int(rows[y]), int(rows[y + 1])
        idx = y * X + xs
        k = np.searchsorted(starts[r0:r1 + 1], idx, side="right") - 1 + r0
        w = words[k]
        kind, val = w >> 24, w & 0xFFFFFF
        v = val.copy()
        if y > 0:
            band_first = band_rows > 0 and y % band_rows == 0
            sd = seeds[(y // band_rows - 1) * (X + 1):(y // band_rows) * (X + 1)] if band_first else None
            up = sd[1:] if band_first else out[y - 1]
            left = np.empty(X, dtype=np.uint32)
            left[1:] = up[:-1]
            if band_first:
                left[0] = sd[0]
            elif band_rows > 0 and y % band_rows == 1 and y > 1:
                left[0] = seeds[(y // band_rows - 1) * (X + 1) + X]      # last pixel of the seed row
            else:
                left[0] = out[y - 2, X - 1] if y >= 2 else 0
            v = np.where(kind == RUN_ABOVE, _add_bytes(up, val), v)
            v = np.where(kind == RUN_ABOVE_LEFT, left, v)
        else:
            v = np.where(kind == RUN_CONST, v, 0)
        out[y] = v
    return out.reshape(-1)


def expand_iframe_tiles(desc, X, Y):
    """What sp_iframe_tile_kernel computes: every tile (band x 256-column span) on its own, from its records,
    the seed row above its band and its column of left pixels — nothing another tile produced."""
    if desc["kind"] == KIND_FLAT:
        return np.full(X * Y, desc["flat_colour"], dtype=np.uint32)
    span, idx, seeds = desc["span_px"], desc["tile_idx"], desc["seeds"]
    recs = desc["runs"].reshape(-1)            # tile layout: 4-byte records (sp.h tile_record32) travelling in the run table's memory, padded to an even count
    left = desc["left"].reshape(-1, 2)         # per tile row: {pixel left of the span one row up, kind counts n_const | n_above << 16}
    rows_per = desc["band_rows"] if 0 < desc["band_rows"] < Y else Y
    nbands, nspans = (Y + rows_per - 1) // rows_per, (X + span - 1) // span
    assert idx.size == nbands * nspans * (rows_per + 1) and left.shape[0] == nbands * nspans * rows_per
    out = np.zeros((Y, X), dtype=np.uint32)
    covered = np.zeros(len(recs), dtype=bool)
    for b in range(nbands):
        for s in range(nspans):
            t = b * nspans + s
            xs, xe = s * span, min(X, (s + 1) * span)
            up = seeds[(b - 1) * (X + 1) + 1 + xs:(b - 1) * (X + 1) + 1 + xe].copy() if b > 0 else np.zeros(xe - xs, np.uint32)
            for r in range(min(rows_per, Y - b * rows_per)):
                y = b * rows_per + r
                e = int(idx[t * (rows_per + 1) + r])
                lo, hi = e & 0x7FFFFFFF, int(idx[t * (rows_per + 1) + r + 1]) & 0x7FFFFFFF
                if e >> 31:        # kRowRepeats: no records of its own, the words of the row above stay in force
                    assert r > 0 and lo == hi, (b, s, r)
                else:
                    rec = recs[lo:hi]
                    covered[lo:hi] = True
                    counts = int(left[t * rows_per + r, 1])
                    nc, na = counts & 0xFFFF, counts >> 16
                    assert nc + na <= len(rec), (b, s, r)
                    # a record's kind is its place among the row's records: constants, then "above", then "above-left"
                    kinds_sorted = np.concatenate([np.full(nc, RUN_CONST), np.full(na, RUN_ABOVE), np.full(len(rec) - nc - na, TILE_ABOVE_LEFT)]).astype(np.uint32)
                    order = np.argsort(rec >> 24, kind="stable")
                    cols = (rec >> 24).astype(np.int64)[order]
                    vals = (rec & 0xFFFFFF)[order]
                    kinds = kinds_sorted[order]
                    for part in (slice(0, nc), slice(nc, nc + na), slice(nc + na, len(rec))):   # inside a kind the columns ascend
                        assert np.all(np.diff((rec[part] >> 24).astype(np.int64)) > 0), (b, s, r)
                assert len(rec) and cols[0] == 0 and np.all(np.diff(cols) > 0) and cols[-1] < xe - xs, (b, s, r)
                k = np.searchsorted(cols, np.arange(xe - xs), side="right") - 1
                kind, val = kinds[k], vals[k]
                lft = np.empty(xe - xs, np.uint32)
                lft[1:] = up[:-1]
                lft[0] = left[t * rows_per + r, 0]
                # every kind is "start value + record value, byte by byte": the pixel above, the pixel above-left (value 0), or 0xFFFFFF for a
                # constant, whose record carries the colour plus one in every byte (sp.h: tile_record32)
                start = np.where(kind == RUN_ABOVE, up, np.where(kind == TILE_ABOVE_LEFT, lft, np.uint32(0xFFFFFF)))
                assert np.all(val[kind == TILE_ABOVE_LEFT] == 0)
                v = _add_bytes(start, val).astype(np.uint32)
                out[y, xs:xe] = v
                up = v
    assert covered[:int(idx[-1]) & 0x7FFFFFFF].all()
    return out.reshape(-1)


def expand_pframe(desc, prev, X, Y):
    """What sp_pframe_kernel computes from the block table + payload."""
    nbx, nby = (X + 15) // 16, (Y + 15) // 16
    prev2 = prev.reshape(Y, X)
    out = np.zeros((Y, X), dtype=np.uint32)
    blocks, payload = desc["blocks"], desc["payload"]
    flat_prev = prev.reshape(-1)
    for by in range(nby):
        for bx in range(nbx):
            b = blocks[by * nbx + bx]
            x16, y16 = bx * 16, by * 16
            xe, ye = min(x16 + 16, X), min(y16 + 16, Y)
            out[y16:ye, x16:xe] = prev2[y16:ye, x16:xe]
            flags = int(b[0])
            if not flags:
                continue
            x1, y1, x2, y2 = int(b[1]), int(b[2]), int(b[3]), int(b[4])
            mx, my = np.frombuffer(b[8:12].tobytes(), dtype=np.int16)
            off = int(np.frombuffer(b[12:16].tobytes(), dtype=np.uint32)[0])
            w = x2 - x1
            for ly in range(y1, y2):
                for lx in range(x1, x2):
                    x, y = x16 + lx, y16 + ly
                    if x >= X or y >= Y:
                        continue
                    if flags & PB_MOTION:
                        j = (y + int(my)) * X + x + int(mx)
                        out[y, x] = flat_prev[j] if 0 <= j < X * Y else 0
                    else:
                        out[y, x] = payload[off + (ly - y1) * w + (lx - x1)]
    return out.reshape(-1)
