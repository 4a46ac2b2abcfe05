# Driver of tools/sanitize_cpu.sh (CPU only): oracle, host stage and encoder built with -fsanitize=address,undefined,
# driven with valid, truncated and garbage ScreenPressor / MSVideo1 streams.
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ['JSP_SANITIZE_DIR']
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_binding, hoststage_binding
from jsplayer_amd import streamgen as sg
oracle_binding.ORACLE_PATH = os.path.join(OUT, 'liboracle.so')
hoststage_binding._PATH = os.path.join(OUT, 'libhoststage.so')
hoststage_binding.subprocess.check_call = lambda *a, **k: 0
sg._GEN_PATH = os.path.join(OUT, 'libjspgen.so')
from oracle_binding import OracleMSVideo1, OracleScreenPressor, OracleAbort
import hoststage_binding as hs
rng = np.random.default_rng(9)
_L = hs.lib()
_L.hs_msv1_parse.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
def hs_msv1(bits, w, h, src, have_prev):
    desc = np.zeros(max((w >> 2) * (h >> 2), 1), np.uint32); bc = np.zeros(max(h >> 2, 1), np.uint8); out = np.zeros(8, np.uint64)
    _L.hs_msv1_parse(bits, w, h, src, len(src), int(have_prev), 4, desc.ctypes.data, bc.ctypes.data, out.ctypes.data)
n = 0
for version in (2, 3, 4):
    for (w, h) in [(64, 48), (100, 52), (37, 23), (320, 240)]:
        chunks, keys, frames = sg.sp_clip(8000 + version, w, h, 6, version=version, flat_at=(3,), unchanged_at=(2,))
        host = hs.HostStage(w, h, 24); host.preinit(36)
        host.set_iframe_layout([0, 5, 24][version - 2], [0, 256, 512][version - 2])   # row-major / tile layouts
        o = OracleScreenPressor(w, h, 24); o.Preinit(36)
        bufs = [np.zeros(w * h, np.int32) for _ in range(3)]
        seq = list(zip(chunks, keys))
        # mutations: truncated, flipped, garbage, wrong order
        for k in range(60):
            c, key = seq[int(rng.integers(0, len(seq)))]
            b = bytearray(c)
            kind = k % 4
            if kind == 0 and b: b = b[: int(rng.integers(0, len(b)))]
            elif kind == 1 and b: b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
            elif kind == 2: b = bytearray([c[0] if c else 0x32]) + bytearray(rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8).tobytes())
            seq.append((bytes(b), key))
        for c, key in seq:
            dst = next(x for x in bufs if x is not o.PreviousFrame())
            try:
                (o.DecompressI if key else o.DecompressP)(c, dst)
            except OracleAbort:
                pass
            d = host.decode(key, c)
            if d["kind"] == hs.KIND_INTER:
                host.literalise_motion(d)
            n += 1
for bits in (16, 8):
    for (w, h) in [(16, 8), (37, 23), (64, 48)]:
        frames, keys, pal = sg.msv1_clip(8100, w, h, 4, bits=bits, p_mix=sg.msv1_p_mix(0.5, 5.0))
        o = OracleMSVideo1(bits, w, h, pal); o.Preinit(4)
        bufs = [np.zeros(w * h, np.int32) for _ in range(3)]
        for k in range(200):
            b = bytearray(frames[k % 4])
            if k % 3 == 1 and b: b = b[: int(rng.integers(0, len(b)))]
            if k % 3 == 2: b = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 80)), dtype=np.uint8).tobytes())
            dst = next(x for x in bufs if x is not o.PreviousFrame())
            hs_msv1(bits, w, h, bytes(b), o.PreviousFrame() is not None)
            try: o.DecompressP(bytes(b), dst)
            except OracleAbort: pass
            n += 1
print("sanitizer run finished,", n, "frames")
