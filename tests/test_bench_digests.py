"""tests/golden/bench_digests.json must describe what jsplayer_amd.workloads generates today and what the oracle
decodes today: the first frames of every recorded workload are regenerated and re-decoded here (CPU only)."""
import json

import numpy as np
import pytest

from jsplayer_amd import streamgen as sg
from jsplayer_amd import workloads as wl
from oracle_binding import OracleMSVideo1, OracleScreenPressor

DOC = json.load(open(wl.GOLDEN))


def test_default_workload_is_recorded_for_all_eight_ranks():
    for r in range(8):
        d = DOC["digests"][f"{wl.DEFAULT}/rank{r}"]
        assert len(d) == 1 and len(d[0]) == wl.WORKLOADS[wl.DEFAULT]["frames"]
    firsts = {DOC["digests"][f"{wl.DEFAULT}/rank{r}"][0][0] for r in range(8)}
    assert len(firsts) == 8, "ranks decode distinct streams"


@pytest.mark.parametrize("key", sorted(k for k in DOC["digests"] if k.endswith("/rank0")))
def test_first_frames_still_decode_to_the_recorded_digests(key):
    name = key.split("/")[0]
    spec = wl.WORKLOADS[name]
    gold = DOC["digests"][key]
    assert len(gold) == spec.get("clips", 1) and all(len(g) == spec["frames"] for g in gold)
    clip = wl.build_clips(name, 0, frames=2)[0]     # per-frame seeds / a sequential generator: a prefix of the full clip
    orc = OracleScreenPressor(wl.W, wl.H, 24) if spec["codec"] == "sp" else OracleMSVideo1(spec["bits"], wl.W, wl.H, clip.palette)
    orc.Preinit(36)
    bufs = [np.zeros(wl.W * wl.H, np.int32) for _ in range(2)]
    for i, (src, k) in enumerate(zip(clip.frames, clip.keys)):
        dst = bufs[0] if orc.PreviousFrame() is bufs[1] else bufs[1]
        if k:
            assert orc.DecompressI(src, dst) == 0
        else:
            orc.DecompressP(src, dst)
        assert wl.digest(orc.PreviousFrame()) == gold[0][i], (key, i)
