// Second opinion on the JavaScript semantics the C++ oracle emulates by hand (SURVEY.md §8c item 2).
// The reference runs on JS typed arrays: reads past the end give `undefined`, arithmetic on it gives
// NaN, bit operators turn NaN into 0, comparisons with NaN are false, typed-array stores wrap, a NaN
// loop bound runs zero times.  This file decodes MS Video 1 frames with PLAIN typed-array code and no
// special cases, so whatever node does natively is what the reference's engine does; the test
// compares the C++ oracle (which spells the special cases out) against it on truncated / garbage
// streams.  Written from the bit layout in SURVEY.md Appendix A, not from the reference's text.
//
//   node msv1_js_semantics.js < cases.json > results.json
'use strict';

function rgb15(c) { return ((c & 0x1F) << 3) + ((c & 0x3E0) << 6) + ((c & 0x7C00) << 9); }
function le16(s, i) { return s[i] + s[i + 1] * 256; }

class Decoder {
  constructor(bits, X, Y, paletteBytes) {
    this.bits = bits; this.X = X; this.Y = Y;
    this.rows = Y >> 2; this.cols = X >> 2;
    this.rowCoded = [];
    this.rowCoded[this.rows - 1] = false;
    this.prev = null;
    this.shortLimit = ((this.rows * this.cols / 1023) | 0) * 2 + 10;
    this.pal = new Int32Array(bits === 8 ? 256 : 8);
    this.paletteBytes = paletteBytes;
    this.insignLines = undefined;       // only the 16-bit Preinit sets it
  }
  preinit(lines) {
    this.insignBlocks = (lines + 3) >> 2;
    if (this.bits === 16) { this.insignLines = lines; return; }
    const p = this.paletteBytes;
    for (let i = 0, pos = 0; i < 256 && p.length - pos >= 4; ++i, pos += 4)
      this.pal[i] = p[pos] | (p[pos + 1] << 8) | (p[pos + 2] << 16) | (p[pos + 3] << 24);
  }
  onlySkips(s) {
    let n = 0;
    for (let i = 0; i < s.length; i += 2) {
      const a = s[i], b = s[i + 1];
      if ((b & 0xFC) !== 0x84) return false;
      n += ((b - 0x84) << 8) + a;
      if (n >= this.rows * this.cols) return true;
    }
    return true;
  }
  paint(dst, at, colours, flags, eight) {
    for (let y = 0; y < 4; ++y)
      for (let x = 0; x < 4; ++x) {
        const q = eight ? ((y & 2) << 1) + (x & 2) : 0;
        dst[at + y * this.X + x] = colours[q + (flags & 1)];
        flags >>= 1;
      }
  }
  copy(dst, at) {   // throws a TypeError when there is no previous frame, as in the reference
    for (let y = 0; y < 4; ++y)
      for (let x = 0; x < 4; ++x) dst[at + y * this.X + x] = this.prev[at + y * this.X + x];
  }
  decodeP(s, dst) {
    if (this.bits === 16 && (s.length === 0 || (s.length < this.shortLimit && this.onlySkips(s))))
      return { same: true, signif: false };
    let si = 0, skip = 0, changes = false;
    const c = this.bits === 16 ? this.pal : new Int32Array(8);
    body:
    for (let by = 0; by < this.rows; ++by) {
      this.rowCoded[by] = false;
      for (let bx = 0; bx < this.cols; ++bx) {
        const at = by * this.X * 4 + bx * 4;
        if (skip !== 0) { skip--; this.copy(dst, at); continue; }
        const a = s[si], b = s[si + 1];
        if (this.bits === 8 && a + b === 0) break body;
        si += 2;
        if ((b & 0xFC) === 0x84) { skip = ((b - 0x84) << 8) + a - 1; this.copy(dst, at); continue; }
        if (this.bits === 16) {
          if (b < 0x80) {
            const flags = ((b << 8) + a) ^ 0xFFFF, c0 = le16(s, si);
            c[0] = rgb15(c0); c[1] = rgb15(le16(s, si + 2));
            si += 4;
            if ((c0 & 0x8000) !== 0) {
              for (let k = 0; k < 6; ++k) c[2 + k] = rgb15(le16(s, si + 2 * k));
              si += 12;
              this.paint(dst, at, c, flags, true);
            } else this.paint(dst, at, c, flags, false);
          } else { c[0] = c[1] = rgb15((b << 8) + a); this.paint(dst, at, c, 0, false); }
        } else {
          if (b < 0x80) {
            c[1] = this.pal[s[si]]; c[0] = this.pal[s[si + 1]]; si += 2;
            this.paint(dst, at, c, (b << 8) + a, false);
          } else if (b >= 0x90) {
            for (let k = 0; k < 8; ++k) c[k] = this.pal[s[si + k]];
            si += 8;
            this.paint(dst, at, c, ((b << 8) + a) ^ 0xFFFF, true);
          } else { c[0] = c[1] = this.pal[a]; this.paint(dst, at, c, 0, false); }
        }
        changes = true;
        this.rowCoded[by] = true;
      }
    }
    let signif = false;
    if (changes)
      for (let i = this.insignBlocks; i < this.rows; ++i) if (this.rowCoded[i]) { signif = true; break; }
    if (signif && this.prev !== null) {
      signif = false;
      for (let i = this.insignLines * this.X; i < this.Y * this.X; ++i)
        if (dst[i] !== this.prev[i]) { signif = true; break; }
    }
    const same = !changes;
    if (changes) this.prev = dst;
    return { same, signif };
  }
}

// A case with `files` instead of `frames` reads each frame's bytes from a file and answers with the 64-bit truncated
// SHA-256 of the previous frame after the call (little-endian int32 pixels) instead of the pixels: full-size frames
// (tests/test_js_semantics.py: frame 0 of every 1920x1080 MSVideo1 bench workload against tests/golden/bench_digests.json).
const fs = require('fs');
const cases = JSON.parse(fs.readFileSync(0, 'utf8'));
const out = cases.map(cs => {
  const d = new Decoder(cs.bits, cs.w, cs.h, Uint8Array.from(cs.palette || []));
  d.preinit(cs.lines);
  const bufs = [0, 1, 2].map(() => new Int32Array(cs.w * cs.h).fill(cs.prefill | 0));
  if (cs.files) {
    return cs.files.map(name => {
      const dst = bufs.find(b => b !== d.prev);
      let r;
      try { r = d.decodeP(new Uint8Array(fs.readFileSync(name)), dst); } catch (e) { return { raised: true }; }
      const pic = d.prev;
      const digest = require('crypto').createHash('sha256').update(Buffer.from(pic.buffer, pic.byteOffset, pic.byteLength)).digest('hex').slice(0, 16);
      return { raised: false, same: r.same, signif: r.signif, digest };
    });
  }
  return cs.frames.map(f => {
    const dst = bufs.find(b => b !== d.prev);
    let r;
    try { r = d.decodeP(Uint8Array.from(f), dst); } catch (e) { return { raised: true, dst: Array.from(dst) }; }
    return { raised: false, same: r.same, signif: r.signif, dst: Array.from(dst), which: bufs.indexOf(d.prev) };
  });
});
process.stdout.write(JSON.stringify(out));
