"""tests/golden/bench_digests.json must describe what jsplayer_amd.workloads generates today and what the oracle
decodes today: the first frames of every recorded workload are regenerated and re-decoded here (CPU only)."""
import json

import numpy as np
import pytest

from jsplayer_amd import streamgen as sg
from jsplayer_amd import workloads as wl
from oracle_binding import OracleMSVideo1, OracleScreenPressor

DOC = json.load(open(wl.GOLDEN))


@pytest.mark.parametrize("name", [wl.DEFAULT, "screenpressor_v4_1080p_pclip300"])
def test_eight_stream_workloads_are_recorded_for_all_eight_ranks(name):
    """the 8-stream configuration is 8 copies (seeds +0..+7) of the MSVideo1 key-frame workload and
    of the ScreenPressor inter-frame clip."""
    spec = wl.WORKLOADS[name]
    for r in range(8):
        d = DOC["digests"][f"{name}/rank{r}"]
        assert len(d) == spec.get("clips", 1) and all(len(c) == spec["frames"] for c in d)
    firsts = {DOC["digests"][f"{name}/rank{r}"][c][0] for r in range(8) for c in range(spec.get("clips", 1))}
    assert len(firsts) == 8 * spec.get("clips", 1), "ranks and clips decode distinct streams"


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
