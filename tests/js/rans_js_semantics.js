// The rANS state machine of ScreenPressor versions 3 and 4 on plain JS numbers and typed arrays — the third of the
// engine-semantics checks (SURVEY.md §8c item 2).  Its state lives in JS `Int`s that the bit operators wrap to 32 bits
// (a state seeded from four bytes can be negative), its renormalisation reads past the end of the stream as
// `undefined`, and nothing guards the probabilities it is given.  The C++ oracle spells those cases out by hand
// (oracle/sp_entropy_oracle.cpp, struct Rans); this file lets node decide what they are.  Written from SURVEY.md §8a
// (Rans: LE32 seed, 12-bit slots, byte renormalisation below 2^23), test infrastructure only.
//
//   node rans_js_semantics.js < cases.json > results.json
//   case = {bytes:[...], pos:int, ops:[[start, freq], ...]}   freq -1: take a raw byte, -2: reseed from the stream
'use strict';

class Coder {
  constructor(bytes, at) { this.bytes = bytes; this.seed(at); }
  seed(at) {
    const b = this.bytes;
    let s = b[at];
    s |= b[at + 1] << 8;
    s |= b[at + 2] << 16;
    s |= b[at + 3] << 24;
    this.state = s;
    this.at = at + 4;
  }
  consume(start, width) {
    let s = this.state;
    s = width * (s >> 12) + (s & 4095) - start;
    let spins = 0;
    while (s < 8388608) {
      if (++spins > 64) return false;          // the reference would still be spinning
      s = (s << 8) | this.bytes[this.at++];
    }
    this.state = s;
    return true;
  }
}

function run(cs) {
  const c = new Coder(Uint8Array.from(cs.bytes), cs.pos);
  const out = [];
  for (const [start, freq] of cs.ops) {
    let v;
    if (freq === -1) { v = c.bytes[c.at++]; if (v === undefined) v = -1; }
    else if (freq === -2) { c.seed(c.at); v = c.state; }
    else { if (!c.consume(start, freq)) break; v = c.state; }
    // the state may be held as a number beyond 2^31 (only with probabilities no model hands out); every later use of
    // it goes through >> or &, i.e. through ToInt32, so it is reported the way it acts
    out.push([v | 0, c.at]);
  }
  return out;
}

let input = '';
process.stdin.on('data', d => { input += d; });
process.stdin.on('end', () => { process.stdout.write(JSON.stringify(JSON.parse(input).map(run))); });
