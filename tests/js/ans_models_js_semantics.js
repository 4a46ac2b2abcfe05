// The colour-context models of ScreenPressor versions 3 and 4 (the "ladder": symbol lists -> sparse sorted models ->
// 40-entry table -> full 256-symbol model) under a real JS engine — the fourth engine-semantics check (SURVEY.md §8c item
// 2, Appendix C.2).  Every array the format keeps in a typed array is a typed array here (Uint8Array symbols, Uint16Array
// frequencies / counts / cumulative sums, Uint8Array start table), so wrap-on-store, `undefined` stored as 0 and never
// found again, and symbols beyond 255 coming out of a sparse model are decided by node, not by hand.  The C++ oracle
// (oracle/sp_entropy_oracle.cpp: Context, Cx1..Cx7, FixedSizeRansCtx, EntroANS.decodeClr) must agree symbol for symbol.
// Written in this project's own structure (it follows jsplayer_amd/csrc/sp_models.cpp: one record per context with a
// stage tag), test infrastructure only.
//
//   node ans_models_js_semantics.js < cases.json > results.json
//   case = {f0: 32|64, bytes: [...], pos: int, ctxs: [ctx, ...]}  ->  {syms: [...], pos: int, census: [entries per stage]}   (-1 = undefined)
'use strict';

const SCALE = 4096, SPARSE_STEP = 50, TABLE_STEP = 25;
const EMPTY = 0, LIST14 = 1, LIST64 = 2, LIST256 = 3, SPARSE4 = 4, SPARSE16 = 5, TABLE40 = 6, FULL = 7;

function shiftFor(tot) { let s = 0; while (tot <= SCALE / 2) { tot <<= 1; ++s; } return s; }
function sortBytes(a, n) { for (let i = 1; i < n; ++i) for (let j = i; j > 0 && a[j - 1] > a[j]; --j) { const t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; } }

class Full {                       // fixed alphabet with deferred adaptation
  constructor(n) { this.n = n; this.fc = new Uint16Array(2 * n); this.cnt = new Uint16Array(n); this.start = new Uint8Array(32); this.sum = 0; }
  mark(cf, fr, sym) { const k0 = (cf + 127) >> 7, k1 = ((cf + fr - 1) >> 7) + 1; for (let k = k0; k < k1; ++k) this.start[k] = sym; }
  take(slot) {
    let j = this.start[slot >> 7];
    while (j < this.n - 1 && this.fc[2 * (j + 1) + 1] <= slot) ++j;
    const iv = [j, this.fc[2 * j + 1], this.fc[2 * j]];
    this.cnt[j] += 16; this.sum += 16;
    if (this.sum + 16 > SCALE) {
      this.sum = 0; let cf = 0;
      for (let k = 0; k < this.n; ++k) {
        const fr = this.cnt[k];
        this.fc[2 * k] = fr; this.fc[2 * k + 1] = cf;
        this.mark(cf, fr, k);
        cf += fr;
        this.cnt[k] -= fr >> 1;
        this.sum += this.cnt[k];
      }
    }
    return iv;
  }
}

class Models {
  constructor(f0) { this.f0 = f0; this.census = [0, 0, 0, 0, 0, 0, 0, 0]; this.cx = new Map(); this.tot = 0; this.c256 = new Uint16Array(256); this.f512 = new Uint16Array(512); }
  get(ctx) { let s = this.cx.get(ctx); if (!s) { s = {stage: EMPTY}; this.cx.set(ctx, s); } return s; }
  coded(ctx) { return this.get(ctx).stage >= SPARSE4; }
  enter(s, st) { s.stage = st; ++this.census[st]; }   // how often a context entered each stage

  // ---- sparse
  sparseTotal(s) { let t = 256 - s.n; for (let i = 0; i < s.n; ++i) t += s.freq[i]; return t; }
  sparseHalve(s) { let sum = 256 - s.n; for (let i = 0; i < s.n; ++i) { s.freq[i] -= s.freq[i] >> 1; sum += s.freq[i]; } this.tot = sum; }
  sparseInsert(s, pos, c) {
    if (s.n === s.cap) return false;
    for (let i = s.n - 1; i >= pos; --i) { s.sym[i + 1] = s.sym[i]; s.freq[i + 1] = s.freq[i]; }
    s.sym[pos] = c; s.freq[pos] = SPARSE_STEP; ++s.n;
    if (s.maxpos >= pos) ++s.maxpos;
    this.tot += SPARSE_STEP;
    if (this.tot + SPARSE_STEP > SCALE) this.sparseHalve(s);
    return true;
  }
  sparseTake(s, slot, tot0) {      // -> [ok, sym, cum, freq]
    this.tot = tot0;
    const shift = shiftFor(tot0);
    slot >>= shift;
    const bonus = (SCALE - (tot0 << shift)) >> shift;
    const keep = s.freq[s.maxpos];
    s.freq[s.maxpos] += bonus;
    let cum = 0, last = 0;
    for (let pos = 0; pos < s.n; ++pos) {
      const sy = s.sym[pos], start = cum + sy - last;
      if (slot < start) {
        const c = slot - cum + last;
        s.freq[s.maxpos] = keep;
        return [this.sparseInsert(s, pos, c), c, slot << shift, 1 << shift];
      }
      const fr = s.freq[pos];
      if (start + fr > slot) {
        s.freq[s.maxpos] = keep;
        s.freq[pos] += SPARSE_STEP; this.tot += SPARSE_STEP;
        if (pos !== s.maxpos && s.freq[pos] > s.freq[s.maxpos]) s.maxpos = pos;
        if (this.tot + SPARSE_STEP > SCALE) this.sparseHalve(s);
        return [true, sy, start << shift, fr << shift];
      }
      cum += sy - last + fr;
      last = sy + 1;
    }
    s.freq[s.maxpos] = keep;
    const c = last + slot - cum;
    return [this.sparseInsert(s, s.n, c), c, slot << shift, 1 << shift];
  }
  sparseFromList(s, cap, c) {
    const n = s.n, list = s.list;
    sortBytes(list, n);
    s.cap = cap; s.maxpos = 0; s.sym = new Uint8Array(16); s.freq = new Uint16Array(16);
    for (let i = 0; i < n; ++i) { s.sym[i] = list[i]; if (s.sym[i] === c) { s.freq[i] = 2 * SPARSE_STEP; s.maxpos = i; } else s.freq[i] = SPARSE_STEP; }
  }
  sparse16FromSparse4(s, c) {
    const osym = s.sym, ofreq = s.freq, on = s.n;
    s.cap = 16; s.sym = new Uint8Array(16); s.freq = new Uint16Array(16); s.maxpos = 0;
    let i = 0, tot = 0;
    while (i < on && osym[i] < c) { s.sym[i] = osym[i]; tot += s.freq[i] = ofreq[i]; ++i; }
    let j = i;
    s.sym[j] = c; tot += s.freq[j] = SPARSE_STEP; ++j;
    while (i < on) { s.sym[j] = osym[i]; tot += s.freq[j] = ofreq[i]; ++i; ++j; }
    s.n = on + 1;
    if (tot > SCALE) this.sparseHalve(s);
    s.cachedTot = this.sparseTotal(s);
  }

  // ---- 40-entry table: tsym / tfreq / tcum / tcnt, the running sum in its own Uint16 slot
  tableNew(cap) { return {tcap: cap, td: 0, fshift: 0, tsym: new Uint8Array(64), tfreq: new Uint16Array(64), tcum: new Uint16Array(64), tcnt: new Uint16Array(64), tsum: new Uint16Array(1)}; }
  tableSwap(t, a, b) { for (const k of ['tsym', 'tfreq', 'tcum', 'tcnt']) { const x = t[k][a]; t[k][a] = t[k][b]; t[k][b] = x; } }
  tableCalcSum(t) { const sh = t.fshift > 0 ? t.fshift - 1 : 0; let sum = (256 - t.td) << sh; for (let i = 0; i < t.tcap; ++i) sum += t.tcnt[i]; t.tsum[0] = sum; }
  tableRebuild(t) {
    const sh = t.fshift > 0 ? t.fshift - 1 : 0;
    for (let i = 0; i < 256; ++i) this.c256[i] = 1 << sh;
    for (let i = 0; i < t.td; ++i) this.c256[t.tsym[i]] = t.tcnt[i];
    let cum = 0;
    for (let i = 0; i < 256; ++i) { this.f512[2 * i] = this.c256[i]; this.f512[2 * i + 1] = cum; cum += this.c256[i]; }
    if (t.fshift > 0) --t.fshift;
    const sh2 = t.fshift > 0 ? t.fshift - 1 : 0;
    let sum = (256 - t.td) << sh2;
    for (let i = 0; i < t.td; ++i) { t.tcnt[i] -= t.tcnt[i] >> 1; sum += t.tcnt[i]; t.tfreq[i] = this.f512[2 * t.tsym[i]]; t.tcum[i] = this.f512[2 * t.tsym[i] + 1]; }
    t.tsum[0] = sum;
  }
  tableBump(t, pos) {
    const step = TABLE_STEP << t.fshift;
    t.tcnt[pos] += step; t.tsum[0] += step;
    if (pos > 0 && t.tcnt[pos] > t.tcnt[pos - 1]) this.tableSwap(t, pos, pos - 1);
    if (t.tsum[0] + step > SCALE) this.tableRebuild(t);
  }
  tableAdd(t, c, freq, cum) { if (t.td >= 40 || t.td >= t.tcap) return -1; t.tsym[t.td] = c; t.tfreq[t.td] = freq; t.tcum[t.td] = cum; t.tcnt[t.td] = freq - (freq >> 1); return t.td++; }
  tableUnseenCum(t, c) {
    let lower = -1, lfreq = 0, lcum = 0;
    for (let i = 0; i < t.td; ++i) if (t.tsym[i] > lower && t.tsym[i] < c) { lower = t.tsym[i]; lfreq = t.tfreq[i]; lcum = t.tcum[i]; }
    return lfreq > 0 ? lcum + lfreq + ((c - lower - 1) << t.fshift) : c << t.fshift;
  }
  tableFromSparse16(s, c) {
    const t = this.tableNew(32), oldd = s.n, shift = shiftFor(this.sparseTotal(s));
    let cum = 0, last = 0;
    for (let pos = 0; pos < oldd; ++pos) {
      const sy = s.sym[pos];
      cum += sy - last;
      const fr = s.freq[pos] << shift;
      t.tsym[pos] = sy; t.tfreq[pos] = fr; t.tcum[pos] = cum << shift; t.tcnt[pos] = fr - (fr >> 1);
      cum += s.freq[pos];
      last = sy + 1;
    }
    t.td = oldd; t.fshift = shift;
    const f = 1 << t.fshift, cf = c > 0 ? this.tableUnseenCum(t, c) : 0;
    t.tsym[oldd] = c; t.tfreq[oldd] = f; t.tcum[oldd] = cf; t.tcnt[oldd] = f - (f >> 1);
    t.td = oldd + 1;
    const step = TABLE_STEP << t.fshift;
    t.tcnt[oldd] += step; t.tsum[0] += step;
    if (t.tsum[0] + step > SCALE) this.tableRebuild(t);
    this.tableCalcSum(t);
    for (let i = 0; i < t.td - 1; ++i) for (let j = i + 1; j < t.td; ++j) if (t.tfreq[j] > t.tfreq[i]) this.tableSwap(t, i, j);
    return t;
  }
  tableFromList(s, c) {
    const oldd = s.n, t = this.tableNew(oldd <= 32 ? 32 : 64), f0 = this.f0, shift = shiftFor(256 - oldd + oldd * f0 + f0);
    sortBytes(s.list, oldd);
    let cum = 0, last = 0, at = 0;
    for (let pos = 0; pos < oldd; ++pos) {
      const sy = s.list[pos];
      cum += sy - last;
      let cfr = f0;
      if (sy === c) { at = pos; cfr = 2 * f0; }
      const fr = cfr << shift;
      t.tsym[pos] = sy; t.tfreq[pos] = fr; t.tcum[pos] = cum << shift; t.tcnt[pos] = fr - (fr >> 1);
      cum += cfr;
      last = sy + 1;
    }
    t.td = oldd; t.fshift = shift;
    this.tableCalcSum(t);
    if (at > 0) this.tableSwap(t, 0, at);
    return t;
  }
  tableTake(t, slot) {             // -> [ok, sym, cum, freq]
    let lfreq = 0, lcum = 0, lower = 0;
    for (let i = 0; i < t.td; ++i) {
      const cf = t.tcum[i];
      if (cf <= slot) {
        const fr = t.tfreq[i];
        if (cf + fr > slot) { const iv = [true, t.tsym[i], cf, fr]; this.tableBump(t, i); return iv; }
        if (cf >= lcum) { lfreq = fr; lcum = cf; lower = t.tsym[i]; }
      }
    }
    const f = 1 << t.fshift;
    let c, cf;
    if (lfreq > 0) { const x = (slot - (lcum + lfreq)) >> t.fshift; c = x + lower + 1; cf = lcum + lfreq + (x << t.fshift); }
    else { c = slot >> t.fshift; cf = c << t.fshift; }
    let p = this.tableAdd(t, c, f, cf);
    if (p < 0) {
      if (t.tcap === 64) return [false, c, cf, f];
      t.tcap = 64;
      p = this.tableAdd(t, c, f, cf);
    }
    this.tableBump(t, p);
    return [true, c, cf, f];
  }
  fullFromList(s, c) {
    const m = new Full(256), d = s.n;
    for (let i = 0; i < 256; ++i) { m.fc[2 * i] = 1; m.cnt[i] = 1; }
    const f0 = ((SCALE - (256 - d)) / (d + 1)) | 0, c0 = f0 - (f0 >> 1);
    for (let i = 0; i < d; ++i) { m.fc[2 * s.list[i]] = f0; m.cnt[s.list[i]] = c0; }
    m.fc[2 * c] += f0; m.cnt[c] += 16;
    let sum = 0, cf = 0;
    for (let i = 0; i < 256; ++i) { sum += m.cnt[i]; m.fc[2 * i + 1] = cf; m.mark(cf, m.fc[2 * i], i); cf += m.fc[2 * i]; }
    m.sum = sum;
    return m;
  }
  fullFromTable(t) {
    const m = new Full(256);
    m.sum = t.tsum[0];
    for (let i = 0; i < t.tcap; ++i) if (t.tcnt[i] > 0) { m.fc[2 * t.tsym[i]] = t.tfreq[i]; m.fc[2 * t.tsym[i] + 1] = t.tcum[i]; m.cnt[t.tsym[i]] = t.tcnt[i]; }
    const f = 1 << t.fshift, cu = f - (f >> 1);
    let cf = 0;
    for (let i = 0; i < 256; ++i) {
      let fr;
      if (m.fc[2 * i] > 0) fr = m.fc[2 * i]; else { m.fc[2 * i] = f; m.fc[2 * i + 1] = cf; m.cnt[i] = cu; fr = f; }
      m.mark(cf, fr, i);
      cf += fr;
    }
    return m;
  }

  // ---- the state machine
  learn(ctx, c) {                  // c may be undefined: stored as 0 by the typed array, equal to nothing
    const s = this.get(ctx);
    const find = () => { for (let i = 0; i < s.n; ++i) if (s.list[i] === c) return true; return false; };
    switch (s.stage) {
      case EMPTY: s.list = new Uint8Array(256); s.n = 1; s.list[0] = c; this.enter(s, LIST14); break;
      case LIST14:
        if (find()) {
          if (s.n <= 4) { this.sparseFromList(s, 4, c); this.enter(s, SPARSE4); }
          else { this.sparseFromList(s, 16, c); s.cachedTot = this.sparseTotal(s); this.enter(s, SPARSE16); }
        } else { s.list[s.n++] = c; if (s.n > 14) this.enter(s, LIST64); }
        break;
      case LIST64:
        if (find()) { s.table = this.tableFromList(s, c); this.enter(s, TABLE40); }
        else { s.list[s.n++] = c; if (s.n > 64) this.enter(s, LIST256); }
        break;
      case LIST256:
        if (find()) { s.full = this.fullFromList(s, c); this.enter(s, FULL); }
        else if (s.n < 256) s.list[s.n++] = c;
        break;
      default: break;
    }
  }
  take(ctx, slot) {                // -> [sym, cum, freq]
    const s = this.get(ctx);
    switch (s.stage) {
      case SPARSE4: {
        const r = this.sparseTake(s, slot, s.freq[0] + s.freq[1] + s.freq[2] + s.freq[3] + 256 - s.n);
        if (!r[0]) { this.sparse16FromSparse4(s, r[1]); this.enter(s, SPARSE16); }
        return [r[1], r[2], r[3]];
      }
      case SPARSE16: {
        const r = this.sparseTake(s, slot, s.cachedTot);
        s.cachedTot = this.tot;
        if (!r[0]) { s.table = this.tableFromSparse16(s, r[1]); this.enter(s, TABLE40); }
        return [r[1], r[2], r[3]];
      }
      case TABLE40: {
        const r = this.tableTake(s.table, slot);
        if (!r[0]) { s.full = this.fullFromTable(s.table); this.enter(s, FULL); }
        return [r[1], r[2], r[3]];
      }
      default: return s.full.take(slot);
    }
  }
}

class Coder {                      // rANS: 32-bit state in a JS number, 12-bit slots, byte renormalisation below 2^23
  constructor(bytes, at) { this.bytes = bytes; this.seed(at); }
  seed(at) { const b = this.bytes; let s = b[at]; s |= b[at + 1] << 8; s |= b[at + 2] << 16; s |= b[at + 3] << 24; this.state = s; this.at = at + 4; }
  slot() { return this.state & 4095; }
  consume(start, width) {
    let s = this.state;
    s = width * (s >> 12) + (s & 4095) - start;
    let spins = 0;
    while (s < 8388608) { if (++spins > 64) return false; s = (s << 8) | this.bytes[this.at++]; }
    this.state = s;
    return true;
  }
}

function run(cs) {
  const coder = new Coder(Uint8Array.from(cs.bytes), cs.pos), models = new Models(cs.f0), syms = [];
  let ndec = 0;
  for (const ctx of cs.ctxs) {
    let c;
    if (models.coded(ctx)) {
      const iv = models.take(ctx, coder.slot());
      if (!coder.consume(iv[1], iv[2])) break;     // the reference would still be spinning
      c = iv[0];
    } else {
      c = coder.bytes[coder.at++];
      models.learn(ctx, c);
    }
    if (++ndec === 131072) { coder.seed(coder.at); ndec = 0; }
    syms.push(c === undefined ? -1 : c);
  }
  return {syms: syms, pos: coder.at, census: models.census};
}

let input = '';
process.stdin.on('data', d => { input += d; });
process.stdin.on('end', () => { process.stdout.write(JSON.stringify(JSON.parse(input).map(run))); });
