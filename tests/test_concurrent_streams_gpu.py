"""Several independent streams decoded concurrently from several host threads on one GPU (the
reference is single-threaded with static scratch state; the contract here is one thread per codec
instance, instances independent — SURVEY.md §8b "Threading").  Every stream must come out bit-exact."""
import threading

import numpy as np
import pytest

from jsplayer_amd import MSVideo1_16bit, ScreenPressor
from jsplayer_amd import streamgen as sg
from oracle_binding import OracleMSVideo1, OracleScreenPressor

pytestmark = pytest.mark.gpu


def _expected(kind, w, h, chunks, keys):
    orc = OracleMSVideo1(16, w, h) if kind == "msv1" else OracleScreenPressor(w, h, 24)
    orc.Preinit(36)
    bufs = [np.zeros(w * h, np.int32) for _ in range(3)]
    out = []
    for c, k in zip(chunks, keys):
        dst = next(b for b in bufs if b is not orc.PreviousFrame())
        (orc.DecompressI if k else orc.DecompressP)(c, dst)
        out.append(orc.PreviousFrame().copy())
    return out


def test_eight_streams_from_eight_threads():
    import torch
    w, h, n = 320, 240, 12
    jobs = []
    for i in range(8):
        if i % 2 == 0:
            chunks, keys, _ = sg.msv1_clip(5000 + i, w, h, n, p_mix=sg.msv1_p_mix(0.6, 20.0))
            jobs.append(("msv1", chunks, keys))
        else:
            chunks, keys, _ = sg.sp_clip(5000 + i, w, h, n, version=2 + (i % 3))
            jobs.append(("sp", chunks, keys))
    expected = [_expected(k, w, h, c, ks) for k, c, ks in jobs]
    results, errors = [None] * len(jobs), []

    def run(idx):
        try:
            kind, chunks, keys = jobs[idx]
            codec = MSVideo1_16bit(w, h) if kind == "msv1" else ScreenPressor(w, h, 24)
            codec.Preinit(36)
            bufs = [torch.zeros(w * h, dtype=torch.int32, device="cuda") for _ in range(3)]
            got = []
            for c, k in zip(chunks, keys):
                dst = next(b for b in bufs if b is not codec.PreviousFrame())
                if k:
                    assert codec.DecompressI(c, dst) == 0
                else:
                    codec.DecompressP(c, dst)
                got.append(codec.PreviousFrame().cpu().numpy().copy())
            codec.StopAndClean()
            results[idx] = got
        except Exception as e:  # surfaced in the main thread
            errors.append((idx, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for i, (exp, got) in enumerate(zip(expected, results)):
        assert got is not None
        for f, (a, b) in enumerate(zip(exp, got)):
            assert np.array_equal(a, b), f"stream {i} frame {f}"
