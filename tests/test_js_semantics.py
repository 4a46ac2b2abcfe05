"""The C++ oracle spells out JavaScript's out-of-range / NaN / typed-array behaviour by hand; here a
plain typed-array decoder runs under a real JS engine (node) on the same truncated and garbage
streams and must agree with it — frames, significant_changes, buffer identity, raised-or-not."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from jsplayer_amd import streamgen as sg
from oracle_binding import OracleAbort, OracleMSVideo1

HERE = os.path.dirname(os.path.abspath(__file__))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def build_cases():
    rng = np.random.default_rng(77)
    cases = []
    for bits in (16, 8):
        for (w, h) in [(16, 8), (12, 12), (37, 23), (64, 48)]:
            frames, keys, pal = sg.msv1_clip(4000 + bits + w, w, h, 4, bits=bits, p_mix=sg.msv1_p_mix(0.5, 5.0))
            seq = [frames[0]]
            for k in range(40):
                base = bytearray(frames[1 + k % 3])
                kind = k % 5
                if kind == 1 and base:
                    base = base[: int(rng.integers(0, len(base)))]
                elif kind == 2 and base:
                    base[int(rng.integers(0, len(base)))] = int(rng.integers(0, 256))
                elif kind == 3:
                    base = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 60)), dtype=np.uint8).tobytes())
                elif kind == 4:
                    base = base + b"\x07"      # odd length: AVI pad byte
                seq.append(bytes(base))
            cases.append(dict(bits=bits, w=w, h=h, lines=int(rng.integers(0, 9)), prefill=0x00A5A5A5,
                              palette=list(pal) if pal else [], frames=[list(f) for f in seq]))
        # a skip code before any frame exists: the reference raises a TypeError
        cases.append(dict(bits=bits, w=16, h=8, lines=0, prefill=7, palette=list(sg.random_palette(sg.SplitMix64(5))) if bits == 8 else [],
                          frames=[[0x00, 0xFC, 0x01, 0x84] + [0] * 8, [0x00, 0xFC] * 8]))
    return cases


def test_oracle_agrees_with_a_real_js_engine():
    cases = build_cases()
    res = subprocess.run([NODE, os.path.join(HERE, "js", "msv1_js_semantics.js")], input=json.dumps(cases).encode(),
                         stdout=subprocess.PIPE, check=True)
    js = json.loads(res.stdout)
    checked = 0
    for cs, jres in zip(cases, js):
        orc = OracleMSVideo1(cs["bits"], cs["w"], cs["h"], bytes(cs["palette"]) or None)
        orc.Preinit(cs["lines"])
        bufs = [np.full(cs["w"] * cs["h"], cs["prefill"], dtype=np.int32) for _ in range(3)]
        for f, jr in zip(cs["frames"], jres):
            dst = next(b for b in bufs if b is not orc.PreviousFrame())
            try:
                data, sig = orc.DecompressP(bytes(f), dst)
                raised = False
            except OracleAbort:
                raised = True
            assert raised == jr["raised"]
            assert dst.tolist() == jr["dst"], (cs["bits"], cs["w"], cs["h"], len(f))
            if not raised:
                assert sig == jr["signif"]
                assert (data is not dst) == jr["same"] or (data is dst and not jr["same"])
                which = next((i for i, b in enumerate(bufs) if b is orc.PreviousFrame()), -1)
                assert which == jr["which"]
            checked += 1
    assert checked > 300


def test_js_engine_decodes_the_full_size_bench_frames_to_the_recorded_digests(tmp_path):
    """The golden digests of the 1920x1080 bench workloads come from the C++ oracle; here the first frames of every MSVideo1
    workload go through the plain typed-array decoder under node and must hash to the same values: the oracle's second
    opinion at the size that is timed (it narrows what is unchecked; it pins nothing to the reference)."""
    from jsplayer_amd import workloads as wl
    gold_doc = json.load(open(wl.GOLDEN))["digests"]
    cases, wanted = [], []
    for name, spec in wl.WORKLOADS.items():
        if spec["codec"] != "msv1":
            continue
        nfr = 3 if "inter" in spec else 1
        clip = wl.build_clips(name, 0, frames=nfr)[0]
        files = []
        for i, f in enumerate(clip.frames):
            path = tmp_path / f"{name}_{i}.bin"
            path.write_bytes(f)
            files.append(str(path))
        cases.append(dict(bits=spec["bits"], w=wl.W, h=wl.H, lines=36, prefill=0, palette=list(clip.palette) if clip.palette else [], files=files))
        wanted.append((name, gold_doc[f"{name}/rank0"][0][:nfr]))
    res = subprocess.run([NODE, os.path.join(HERE, "js", "msv1_js_semantics.js")], input=json.dumps(cases).encode(),
                         stdout=subprocess.PIPE, check=True)
    js = json.loads(res.stdout)
    checked = 0
    for (name, gold), frames in zip(wanted, js):
        for i, (g, r) in enumerate(zip(gold, frames)):
            assert not r["raised"], (name, i)
            assert g != "-" and r["digest"] == g, (name, i)
            checked += 1
    assert checked >= 8


# ---- ScreenPressor, range-coder (version 2) streams ---------------------------------------------------
def build_sp_cases():
    from jsplayer_amd import streamgen as sg2
    rng = np.random.default_rng(78)
    cases = []
    for (w, h, bpp) in [(64, 48, 24), (37, 23, 24), (64, 48, 16), (100, 52, 24)]:
        chunks, keys, _ = sg2.sp_clip(4100 + w + bpp, w, h, 7, bpp=bpp, version=2, key_every=4, flat_at=(5,),
                                      unchanged_at=(2,))
        clip = [dict(key=bool(k), bytes=list(c)) for c, k in zip(chunks, keys)]
        cases.append(dict(w=w, h=h, bpp=bpp, lines=4, prefill=0x00A5A5A5, frames=clip))        # the valid clip
        for k in range(45):
            victim = 1 + k % 6                       # never frame 0: the models must exist (this JS file is v2 only)
            b = bytearray(chunks[victim])
            kind = k % 5
            if kind == 0 and len(b) > 1:
                b = b[: int(rng.integers(1, len(b)))]                                     # truncated
            elif kind == 1 and len(b) > 1:
                b[int(rng.integers(1, len(b)))] ^= int(rng.integers(1, 256))              # one byte flipped
            elif kind == 2:
                b = bytearray([b[0] if b else 0x12]) + bytearray(rng.integers(0, 256, size=int(rng.integers(0, 200)), dtype=np.uint8).tobytes())
            elif kind == 3:
                b = bytearray(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8).tobytes())
                b[0] = int(rng.choice([0x12, 0x11, 0x13, 0x10, 0x02, 0x01]))             # heads of this version only
            else:
                b = bytearray()
            seq = clip[:victim] + [dict(key=bool(keys[victim]), bytes=list(b))] + clip[victim + 1:] + clip[4:]
            cases.append(dict(w=w, h=h, bpp=bpp, lines=int(rng.integers(0, 40)), prefill=int(rng.integers(0, 1 << 24)), frames=seq))
    # a flat key frame before any coded one: the reference dereferences a null coder
    cases.append(dict(w=16, h=16, bpp=24, lines=0, prefill=5, frames=[dict(key=True, bytes=[0x11, 1, 2, 3]), dict(key=False, bytes=[1, 2, 3])]))
    return cases


def test_screenpressor_oracle_agrees_with_a_real_js_engine():
    from oracle_binding import OracleScreenPressor
    cases = build_sp_cases()
    res = subprocess.run([NODE, os.path.join(HERE, "js", "sp_js_semantics.js")], input=json.dumps(cases).encode(),
                         stdout=subprocess.PIPE, check=True, timeout=300)
    js = json.loads(res.stdout)
    checked = aborted = hung = 0
    for ci, (cs, jres) in enumerate(zip(cases, js)):
        orc = OracleScreenPressor(cs["w"], cs["h"], cs["bpp"])
        orc.Preinit(cs["lines"])
        bufs = [np.full(cs["w"] * cs["h"], cs["prefill"], dtype=np.int32) for _ in range(3)]
        for fi, (f, jr) in enumerate(zip(cs["frames"], jres)):
            dst = next(b for b in bufs if b is not orc.PreviousFrame())
            where = (ci, fi, cs["w"], cs["h"], cs["bpp"], len(f["bytes"]))
            if f["key"]:
                rc = orc.DecompressI(bytes(f["bytes"]), dst)
                assert (rc == 3) == (jr["raised"] or jr["hang"]), where
                if rc != 3:
                    assert rc == jr["state"], where
            else:
                try:
                    data, sig = orc.DecompressP(bytes(f["bytes"]), dst)
                    assert not (jr["raised"] or jr["hang"]), where
                    assert sig == jr["signif"], where
                    assert (data is not dst) == jr["same"], where
                except OracleAbort:
                    assert jr["raised"] or jr["hang"], where
            if jr["hang"]:
                hung += 1
                break                                 # the reference would still be spinning
            aborted += 1 if jr["raised"] else 0
            assert dst.tolist() == jr["dst"], where
            which = next((i for i, b in enumerate(bufs) if b is orc.PreviousFrame()), -1)
            assert which == jr["which"], where
            checked += 1
    assert checked > 1500 and aborted > 0 and hung > 0


# ---- ScreenPressor versions 3 / 4: the rANS state machine alone ------------------------------------------------
def test_rans_state_machine_agrees_with_a_real_js_engine():
    """oracle/sp_entropy_oracle.cpp's Rans (int32 wrap of the seed and of `(x << 8) | byte`, reads past the end,
    unguarded probabilities, the renormalisation that never ends) against the same machine on plain JS numbers under
    node: states and stream positions after every operation, and where a sequence hangs."""
    import ctypes as C
    import oracle_binding
    L = oracle_binding.lib()
    L.orc_rans_trace.restype = C.c_int
    L.orc_rans_trace.argtypes = [C.c_char_p, C.c_size_t, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(79)
    cases = []
    for k in range(400):
        nbytes = int(rng.integers(0, 60))
        data = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        if k % 3 == 0 and nbytes >= 4:
            data[3] |= 0x80                                   # a negative seed
        ops = []
        for _ in range(int(rng.integers(1, 40))):
            r = rng.random()
            if r < 0.1:
                ops.append([0, -1])
            elif r < 0.15:
                ops.append([0, -2])
            elif r < 0.6:                                      # what a model would hand over
                freq = int(rng.integers(1, 4097))
                ops.append([int(rng.integers(0, 4097 - freq)), freq])
            else:                                              # anything
                ops.append([int(rng.integers(-5000, 9000)), int(rng.integers(0, 5000))])
        cases.append(dict(bytes=data.tolist(), pos=int(rng.integers(0, max(1, nbytes + 3))), ops=ops))
    res = subprocess.run([NODE, os.path.join(HERE, "js", "rans_js_semantics.js")], input=json.dumps(cases).encode(),
                         stdout=subprocess.PIPE, check=True, timeout=120)
    js = json.loads(res.stdout)
    hangs = steps = 0
    for cs, jr in zip(cases, js):
        ops = np.array(cs["ops"], dtype=np.int32).reshape(-1)
        out = np.zeros(2 * len(cs["ops"]), dtype=np.int64)
        src = bytes(cs["bytes"])
        done = L.orc_rans_trace(src, len(src), cs["pos"], ops.ctypes.data, len(cs["ops"]), out.ctypes.data)
        assert done == len(jr), (cs["pos"], len(cs["bytes"]), done, len(jr))
        assert out[:2 * done].reshape(-1, 2).tolist() == jr
        hangs += done < len(cs["ops"])
        steps += done
    assert steps > 4000 and hangs > 0


# ---- ScreenPressor versions 3 / 4: the colour-context model ladder -------------------------------------------------
def test_colour_context_ladder_agrees_with_a_real_js_engine():
    """oracle/sp_entropy_oracle.cpp's Context / Cx1..Cx7 / FixedSizeRansCtx behind EntroANS.decodeClr against the same
    models kept in JS typed arrays under node (tests/js/ans_models_js_semantics.js): arbitrary byte streams — small and
    large raw alphabets, so that every stage of the ladder is entered — decoded through a handful of contexts, symbol
    for symbol and position for position, both f0 settings (64: version 3, whose table can outgrow the code space)."""
    import ctypes as C
    import oracle_binding
    L = oracle_binding.lib()
    L.orc_ans_clr_trace.restype = C.c_int
    L.orc_ans_clr_trace.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(83)
    cases = []
    for k in range(160):
        alphabet = rng.choice(256, size=int(rng.choice([2, 5, 12, 30, 70, 256])), replace=False)
        nbytes = int(rng.integers(40, 9000))
        data = alphabet[rng.integers(0, alphabet.size, size=nbytes)].astype(np.uint8)
        if k % 4 == 0:                                      # bursts of anything in between
            at = rng.integers(0, nbytes, size=nbytes // 6)
            data[at] = rng.integers(0, 256, size=at.size, dtype=np.uint8)
        nctx = int(rng.choice([1, 2, 3, 6, 12]))
        ids = rng.choice(3 * 4096, size=nctx, replace=False)
        ctxs = ids[rng.integers(0, nctx, size=int(rng.integers(50, 7000)))]
        cases.append(dict(f0=int(rng.choice([32, 64])), bytes=data.tolist(), pos=1, ctxs=ctxs.tolist()))
    for k in range(12):   # 65+ distinct raw bytes before the first repeat: the 256-entry list, then the full model built from it
        perm = rng.permutation(256).astype(np.uint8)
        data = np.concatenate([rng.integers(0, 256, 5, dtype=np.uint8), perm[:int(rng.integers(66, 257))], perm[::-1],
                               rng.integers(0, 256, 3000, dtype=np.uint8)])
        cases.append(dict(f0=32 if k % 2 else 64, bytes=data.tolist(), pos=1, ctxs=[int(rng.integers(0, 3 * 4096))] * int(rng.integers(300, 3000))))
    res = subprocess.run([NODE, os.path.join(HERE, "js", "ans_models_js_semantics.js")], input=json.dumps(cases).encode(),
                         stdout=subprocess.PIPE, check=True, timeout=300)
    js = json.loads(res.stdout)
    total = past_end = hangs = big = 0
    census = np.zeros(8, dtype=np.int64)
    for cs, jr in zip(cases, js):
        ctxs = np.array(cs["ctxs"], dtype=np.int32)
        out = np.zeros(ctxs.size, dtype=np.int32)
        pos = C.c_int64(0)
        src = bytes(cs["bytes"])
        done = L.orc_ans_clr_trace(cs["f0"], src, len(src), cs["pos"], ctxs.ctypes.data, ctxs.size, out.ctypes.data, C.byref(pos))
        assert done == len(jr["syms"]), (done, len(jr["syms"]))
        assert out[:done].tolist() == jr["syms"]
        assert pos.value == jr["pos"]
        total += done
        past_end += pos.value > len(src)
        hangs += done < ctxs.size
        big += int((out[:done] > 255).sum())
        census += np.array(jr["census"])
    assert total > 100000 and past_end > 0
    assert (census[1:] > 0).all(), census     # every stage was entered: lists 14 / 64 / 256, sparse 4 / 16, table, full
