// Second opinion on the JavaScript semantics the C++ ScreenPressor oracle emulates by hand, for the
// version-2 (range coder) streams (SURVEY.md §8c item 2).  The reference's range coder lives on JS
// numbers: `code` and `range` are doubles, Std.int() is `x | 0` (ToInt32, wraps), reads past the end of
// the stream are `undefined` and turn `code` into NaN for good, a NaN comparison ends the
// renormalisation loop, typed-array stores wrap or are dropped when out of range, a table index past
// the tables is `undefined` and raises a TypeError on use.  This file decodes with PLAIN typed-array
// code and no special cases, so whatever node does natively is what the reference's engine does; the
// test (tests/test_js_semantics.py) holds the C++ oracle — which spells those cases out — against it on
// valid, truncated, bit-flipped and random streams.
// Written from SURVEY.md §8a / Appendix B (stream layout, model steps, predictor rules), in this
// project's own structure; it is test infrastructure only.
//
//   node sp_js_semantics.js < cases.json > results.json
'use strict';

const RENORM_BELOW = 0x1000000, HALVE_ABOVE = 0x10000;
const ROW = 273;                       // colour table row: 16 group sums, total, 256 counts
const MV_RANGE = 256;

class HangError extends Error {}

// ---- adaptive tables ----------------------------------------------------------------------------
function flatTable(n) { const t = new Uint32Array(n + 1); return t; }
function resetFlat(t, n) { for (let i = 0; i < n; ++i) t[i] = 1; t[n] = n; }

class RangeReader {
  open(bytes, at) {
    this.bytes = bytes;
    this.range = 0xFFFF * 65536 + 0xFFFF;
    let c = 0;
    for (let k = 1; k <= 4; ++k) c = c * 256 + bytes[at + k];   // the byte at `at` is skipped
    this.code = c;
    this.at = at + 5;
  }
  slot(total) {
    this.range = (this.range / total) | 0;
    return (this.code / this.range) | 0;
  }
  take(cum, width) {
    this.code -= cum * this.range;
    this.range = this.range * width;
    let spins = 0;
    while (this.range < RENORM_BELOW) {
      // a healthy coder needs at most 3 rounds; range == 0 (corrupt stream) never leaves this loop in the
      // reference either
      if (++spins > 64) throw new HangError();
      this.code = this.code * 256 + this.bytes[this.at++];
      this.range *= 256;
    }
  }
  // plain table: n counts then their total
  symbol(t, n, step) {
    let total = t[n];
    const v = this.slot(total);
    let s = 0, below = 0, w = 0;
    for (; s < n; ++s) {
      w = t[s];
      if (v >= below + w) below += w; else break;
    }
    this.take(below, w);
    t[s] = w + step;
    total += step;
    if (total > HALVE_ABOVE) {
      total = 0;
      for (let i = 0; i < n; ++i) { const h = (t[i] >> 1) + 1; t[i] = h; total += h; }
    }
    t[n] = total;
    return s;
  }
  // colour row at word offset `o`: two-level search, group sums kept alongside the counts
  colour(t, o, step) {
    let total = t[o + 16];
    const v = this.slot(total);
    let g = 0, below = 0, gw = 0;
    for (; g < 16; ++g) {
      gw = t[o + g];
      if (v >= below + gw) below += gw; else break;
    }
    let s = g * 16, w = 0;
    for (; s < 256; ++s) {
      w = t[o + 17 + s];
      if (v >= below + w) below += w; else break;
    }
    this.take(below, w);
    t[o + 17 + s] = w + step;
    t[o + g] = gw + step;
    total += step;
    if (total > HALVE_ABOVE) {
      total = 0;
      for (let i = o + 17; i < o + 17 + 256; ++i) { const h = (t[i] >> 1) + 1; t[i] = h; total += h; }
      for (let k = 0; k < 16; ++k) {
        let sum = 0;
        for (let j = 0; j < 16; ++j) sum += t[o + 17 + k * 16 + j];
        t[o + k] = sum;
      }
    }
    t[o + 16] = total;
    return s;
  }
}

class RcModels {
  constructor() {
    this.rd = new RangeReader();
    this.colours = new Uint32Array(3 * 4096 * ROW);
    this.kinds = []; this.lengths = [];
    for (let i = 0; i < 6; ++i) { this.kinds[i] = flatTable(6); this.lengths[i] = flatTable(256); }
    this.span = flatTable(256); this.blockRuns = flatTable(256); this.blockKinds = flatTable(5);
    this.rect = [flatTable(16), flatTable(16), flatTable(16), flatTable(16)];
    this.mv = [flatTable(2 * MV_RANGE), flatTable(2 * MV_RANGE)];
  }
  firstUse() { for (let r = 0; r < 3 * 4096; ++r) this.colours[r * ROW + 16] = 0; }
  keyFrameReset() {
    const c = this.colours;
    for (let r = 0; r < 3 * 4096; ++r) {
      const o = r * ROW;
      if (c[o + 16] !== 256) {
        for (let i = 0; i < 256; ++i) c[o + 17 + i] = 1;
        for (let i = 0; i < 16; ++i) c[o + i] = 16;
        c[o + 16] = 256;
      }
    }
    for (let i = 0; i < 6; ++i) { resetFlat(this.lengths[i], 256); resetFlat(this.kinds[i], 6); }
    resetFlat(this.span, 256); resetFlat(this.blockRuns, 256); resetFlat(this.blockKinds, 5);
    for (let i = 0; i < 4; ++i) resetFlat(this.rect[i], 16);
    resetFlat(this.mv[0], 2 * MV_RANGE); resetFlat(this.mv[1], 2 * MV_RANGE);
  }
  open(bytes, at) { this.rd.open(bytes, at); }
  colour(ctx) { return this.rd.colour(this.colours, ctx * ROW, 400); }
  length(kind) { return this.rd.symbol(this.lengths[kind], 256, 400); }
  kind(prev) { return this.rd.symbol(this.kinds[prev], 6, 1000); }
  spanByte() { return this.rd.symbol(this.span, 256, 1); }
  blockKind() { return this.rd.symbol(this.blockKinds, 5, 10); }
  blockRun() { return this.rd.symbol(this.blockRuns, 256, 20); }
  rectEdge(i) { return this.rd.symbol(this.rect[i], 16, 100); }
  mvx() { return this.rd.symbol(this.mv[0], 2 * MV_RANGE, 100); }
  mvy() { return this.rd.symbol(this.mv[1], 2 * MV_RANGE, 100); }
}

// ---- the codec ----------------------------------------------------------------------------------
class Codec {
  constructor(X, Y, bpp) {
    this.X = X; this.Y = Y; this.bpp = bpp;
    this.ctxShift = bpp === 16 ? 0 : 2;
    this.cols = ((X + 15) / 16) | 0; this.rows = ((Y + 15) / 16) | 0;
    this.kindsOfBlocks = new Int32Array(this.cols * this.rows);
    this.prev = null; this.models = null; this.haveKey = false; this.lastFlat = null;
    this.hi = 0; this.lo = 0;   // colour context: previous component (hi) and the one before (lo, pre-shifted)
    this.budget = 0;
  }
  preinit(lines) { this.quietBlocks = this.cols * (((lines + 15) / 16) | 0); }
  tick() { if (--this.budget < 0) throw new HangError(); }
  forgetFrame() {
    this.prev = null;
    if (this.lastFlat !== null) return;
    this.models.keyFrameReset();          // TypeError when no coded key frame came first
  }
  literal() {
    const m = this.models;
    const c0 = m.colour(this.hi + this.lo);
    this.lo = (this.hi << 6) & 0xFC0; this.hi = c0 >> this.ctxShift;
    const c1 = m.colour(4096 + this.hi + this.lo);
    this.lo = (this.hi << 6) & 0xFC0; this.hi = c1 >> this.ctxShift;
    const c2 = m.colour(8192 + this.hi + this.lo);
    this.lo = (this.hi << 6) & 0xFC0; this.hi = c2 >> this.ctxShift;
    return (c2 << 16) + (c1 << 8) + c0;
  }
  runContext(px) {
    if (this.bpp === 16) { this.lo = (px & 0xFF00) >> 2; this.hi = px >> 16; }   // range-coder streams only
    else { this.lo = (px & 0xFC00) >> 4; this.hi = px >> 18; }
  }
  keyFrame(src, dst) {
    const X = this.X, end = X * this.Y;
    const head = src[0], version = (head >> 4) + 1;
    if ((head & 15) === 1) {
      this.forgetFrame();
      let px;
      if (this.bpp === 16) {
        const v = src[0] + src[1] * 256;
        px = ((((v >> 10) & 31) << 3) << 16) + ((((v >> 5) & 31) << 3) << 8) + ((v & 31) << 3);
      } else px = (src[3] << 16) + (src[2] << 8) + src[1];
      for (let i = 0; i < end; ++i) dst[i] = px;
      this.prev = dst; this.lastFlat = px; this.haveKey = true;
      return 0;
    }
    this.lastFlat = null;
    if ((head & 15) !== 2) return 2;
    if (this.models === null) {
      if (version !== 2) return 2;        // this file covers the range-coder streams only
      this.models = new RcModels();
      this.models.firstUse();
    }
    this.forgetFrame();
    const m = this.models;
    m.open(src, 1);
    this.hi = this.lo = 0;
    let at = 0, px = 0, covered = 0;
    while (covered < X + 1) {             // opening literal runs: a full row and one pixel more
      this.tick();
      px = this.literal();
      let n = m.length(0);
      covered += n;
      while (n-- > 0) dst[at++] = px;
    }
    let left = at - 1;
    const bytes = new Uint8Array(dst.buffer, dst.byteOffset, dst.length * 4);
    let kind = 0;
    while (at < end) {
      this.tick();
      kind = m.kind(kind);
      if (kind === 0) px = this.literal();
      let n = m.length(kind);
      if (kind === 0) { while (n-- > 0) dst[at++] = px; left = at - 1; }
      else if (kind === 1) { while (n-- > 0) { dst[at] = dst[left]; left = at; at++; } px = dst[left]; }
      else if (kind === 2) { while (n-- > 0) { px = dst[at - X]; dst[at] = px; at++; } left = at - 1; }
      else if (kind === 4) {
        while (n-- > 0) {
          const u = (at - X - 1) * 4;
          const c0 = bytes[left * 4] + bytes[u + 4] - bytes[u];
          const c1 = bytes[left * 4 + 1] + bytes[u + 5] - bytes[u + 1];
          const c2 = bytes[left * 4 + 2] + bytes[u + 6] - bytes[u + 2];
          px = ((c2 & 255) << 16) + ((c1 & 255) << 8) + (c0 & 255);
          dst[at] = px; left = at; at++;
        }
      } else if (kind === 5) { while (n-- > 0) { px = dst[at - X - 1]; dst[at] = px; at++; } left = at - 1; }
      this.runContext(px);
    }
    this.prev = dst; this.haveKey = true;
    return 0;
  }
  interFrame(src, dst) {
    this.lastFlat = null;
    if (src.length === 0 || !this.haveKey) return { frame: this.prev, signif: false };
    if (src[0] === 0) return { frame: this.prev, signif: false };
    const m = this.models, X = this.X, Y = this.Y, kinds = this.kindsOfBlocks;
    m.open(src, 1);
    let low = m.spanByte();
    const first = (m.spanByte() << 8) + low;
    low = m.spanByte();
    const last = (m.spanByte() << 8) + low;
    for (let i = 0; i < kinds.length; ++i) kinds[i] = 0;
    for (let b = first; b <= last;) {
      this.tick();
      const k = m.blockKind(), n = m.blockRun();
      for (let i = 0; i < n; ++i) kinds[b++] = k;
    }
    let signif = false;
    for (let i = this.quietBlocks; i < kinds.length; ++i) if (kinds[i] > 0) { signif = true; break; }
    const bytes = new Uint8Array(dst.buffer, dst.byteOffset, dst.length * 4);
    const prev = this.prev;
    this.hi = this.lo = 0;
    let px = 0, mvx = 0, mvy = 0;
    for (let r = 0; r < this.rows; ++r)
      for (let c = 0; c < this.cols; ++c) {
        let x1 = c * 16, y1 = r * 16, x2 = Math.min(x1 + 16, X), y2 = Math.min(y1 + 16, Y);
        const what = kinds[r * this.cols + c];
        if (what <= 0) {
          for (let y = y1; y < y2; ++y) for (let x = x1; x < x2; ++x) dst[y * X + x] = prev[y * X + x];
          continue;
        }
        if ((what - 1) & 1) {             // keep the block, then work inside a rectangle of it
          for (let y = y1; y < y2; ++y) for (let x = x1; x < x2; ++x) dst[y * X + x] = prev[y * X + x];
          const ox = c * 16, oy = r * 16;
          x1 = m.rectEdge(0) + ox; y1 = m.rectEdge(1) + oy; x2 = m.rectEdge(2) + ox + 1; y2 = m.rectEdge(3) + oy + 1;
        }
        if ((what - 1) & 2) {             // moved from elsewhere in the previous frame (range coder: no "same as last")
          mvx = m.mvx() - MV_RANGE; mvy = m.mvy() - MV_RANGE;
          for (let y = y1; y < y2; ++y)
            for (let x = 0; x < x2 - x1; ++x) dst[y * X + x1 + x] = prev[(y + mvy) * X + x1 + mvx + x];
          continue;
        }
        let x = x1, y = y1, kind = 0;
        while (y < y2) {
          this.tick();
          kind = m.kind(kind);
          if (kind === 0) px = this.literal();
          const n = m.length(kind);
          let at = y * X + x;
          for (let k = 0; k < n; ++k) {
            if (kind === 1) px = dst[at - 1];
            else if (kind === 2) px = dst[at - X];
            else if (kind === 3) px = prev[at];
            else if (kind === 4) {
              const u = (at - X - 1) * 4, l = (at - 1) * 4;
              const c0 = bytes[l] + bytes[u + 4] - bytes[u];
              const c1 = bytes[l + 1] + bytes[u + 5] - bytes[u + 1];
              const c2 = bytes[l + 2] + bytes[u + 6] - bytes[u + 2];
              px = ((c2 & 255) << 16) + ((c1 & 255) << 8) + (c0 & 255);
            } else if (kind === 5) px = dst[at - X - 1];
            dst[at] = px;
            if (++x >= x2) { x = x1; ++y; at = y * X + x; } else ++at;
          }
          this.runContext(px);
        }
      }
    this.prev = dst;
    return { frame: dst, signif };
  }
}

// ---- driver ---------------------------------------------------------------------------------------
function run(cs) {
  const n = cs.w * cs.h;
  const bufs = [0, 1, 2].map(() => new Int32Array(n).fill(cs.prefill));
  const codec = new Codec(cs.w, cs.h, cs.bpp);
  codec.preinit(cs.lines);
  const out = [];
  for (const f of cs.frames) {
    const src = Uint8Array.from(f.bytes);
    const dst = bufs.find(b => b !== codec.prev);
    const rec = { raised: false, hang: false };
    codec.budget = 200 * n + 100000;
    try {
      if (f.key) rec.state = codec.keyFrame(src, dst);
      else {
        const r = codec.interFrame(src, dst);
        rec.same = r.frame !== dst;
        rec.signif = r.signif;
      }
    } catch (e) {
      if (e instanceof HangError) rec.hang = true;
      else if (e instanceof TypeError) rec.raised = true;
      else throw e;
    }
    rec.dst = Array.from(dst);
    rec.which = bufs.indexOf(codec.prev);
    out.push(rec);
    if (rec.hang) break;                  // the reference would still be spinning: nothing after it is defined
  }
  return out;
}

let input = '';
process.stdin.on('data', d => { input += d; });
process.stdin.on('end', () => { process.stdout.write(JSON.stringify(JSON.parse(input).map(run))); });
