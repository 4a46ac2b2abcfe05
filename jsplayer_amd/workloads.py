"""The synthetic 1920x1080 workloads of BASELINE.json's configs, shared by bench.py, the
full-size parity tests (tests/test_bench_workloads_gpu.py) and the script that writes their golden digests
(tests/golden/make_bench_digests.py), so that what is timed is exactly what is checked.

A workload is one or more independent CLIPS (each with a codec instance of its own: its entropy models and its
previous-frame chain).  A "step" of bench.py decodes every clip of the workload once.  Clips are sized so that a
step's inputs exceed the 256 MiB Infinity Cache and keep the GPU busy for about a millisecond: nothing a step reads
is left in a cache by the step before it.

Frames of one clip come from per-frame seeds (splitmix64, SEED_BASE + config + (rank, clip, frame)), so clips
can be generated on several host threads and any single frame can be regenerated alone.
"""
from __future__ import annotations

import hashlib
import os
import sys
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import streamgen as sg

W, H = 1920, 1080

WORKLOADS = {
    # BASELINE.json configs[1]: MSVideo1 1920x1080 key frames, mix M1, decoded from the RAW STREAM BYTES resident
    # in HBM (on-GPU parse + block reconstruction every step).  3 clips of 512 distinct frames: 1.6 GB of stream and 12.7 GB of
    # frames per step, three launches — a step is ~2.4 ms of GPU time, so that 20 timed steps are ~50 ms.
    "msvideo1_16_1080p_keyframes_m1": dict(codec="msv1", bits=16, frames=512, clips=3, mix="m1", config_index=2, parse="gpu"),
    # the same frames with the descriptor table built by the sequential host parser at staging (round-1 headline)
    "msvideo1_16_1080p_keyframes_m1_hostdesc": dict(codec="msv1", bits=16, frames=512, mix="m1", config_index=2, parse="host"),
    "msvideo1_16_1080p_keyframes_solid": dict(codec="msv1", bits=16, frames=512, mix="solid", config_index=2, parse="gpu"),
    "msvideo1_16_1080p_keyframes_eight": dict(codec="msv1", bits=16, frames=256, mix="eight", config_index=2, parse="gpu"),
    "msvideo1_8_1080p_keyframes_m1": dict(codec="msv1", bits=8, frames=512, mix="m1", config_index=2, parse="gpu"),
    # inter frames: 70 % of the blocks skipped (geometric skip runs, mean 40), one temporal launch
    "msvideo1_16_1080p_inter70": dict(codec="msv1", bits=16, frames=512, mix="m1", config_index=2, inter=0.70, parse="gpu"),
    # BASELINE.json configs[2]: ScreenPressor 1080p key frames (host rANS -> GPU tile reconstruction), 256 distinct
    "screenpressor_v4_1080p_iframes": dict(codec="sp", version=4, frames=256, config_index=3, mode="intra"),
    "screenpressor_v2_1080p_iframes": dict(codec="sp", version=2, frames=256, config_index=3, mode="intra"),
    # BASELINE.json configs[3]: ScreenPressor 1080p 300-frame clips, inter-frame kernel; two clips per step
    # (each clip's key frame is decoded up front, the step is the 2 x 299 inter frames)
    "screenpressor_v4_1080p_pclip300": dict(codec="sp", version=4, frames=300, clips=2, config_index=4, mode="inter"),
}
DEFAULT = "msvideo1_16_1080p_keyframes_m1"


@dataclass
class Clip:
    frames: List[bytes]
    keys: List[bool]
    palette: Optional[bytes]


def _threads() -> int:
    return max(1, min(16, os.cpu_count() or 1))


def _seed(config_index: int, rank: int, clip: int, frame: int = 0) -> int:
    # seeds +0..+7 for the 8-stream configuration, clips and frames in higher bits
    return config_index + 1000 * rank + 100 * clip + (frame << 20)


def _msv1_clip(spec, rank, clip) -> Clip:
    mix = {"m1": sg.MIX_M1, "solid": sg.MIX_ALL_SOLID, "eight": sg.MIX_ALL_EIGHT}[spec["mix"]]
    p_mix = sg.msv1_p_mix(spec["inter"], 40.0) if "inter" in spec else None
    bits, n = spec["bits"], spec["frames"]
    pal = sg.random_palette(sg.SplitMix64(sg.SEED_BASE + _seed(spec["config_index"], rank, clip) + 7)) if bits == 8 else None
    gen = sg.msv1_frame_16 if bits == 16 else sg.msv1_frame_8

    def one(i):
        rng = sg.SplitMix64(sg.SEED_BASE + _seed(spec["config_index"], rank, clip, i))
        return gen(rng, W, H, mix if (i == 0 or p_mix is None) else p_mix)

    with ThreadPoolExecutor(_threads()) as ex:
        frames = list(ex.map(one, range(n)))
    return Clip(frames, [i == 0 or p_mix is None for i in range(n)], pal)


def _sp_clip(spec, rank, clip) -> Clip:
    n, version = spec["frames"], spec["version"]
    if spec["mode"] == "intra":   # every frame a key frame of its own synthetic desktop: frames are independent
        def one(i):
            c, _, _ = sg.sp_clip(_seed(spec["config_index"], rank, clip, i), W, H, 1, version=version)
            return c[0]
        with ThreadPoolExecutor(_threads()) as ex:
            frames = list(ex.map(one, range(n)))
        return Clip(frames, [True] * n, None)
    chunks, keys, _ = sg.sp_clip(_seed(spec["config_index"], rank, clip), W, H, n, version=version)
    return Clip(chunks, keys, None)


def build_clips(name: str, rank: int = 0, frames: Optional[int] = None) -> List[Clip]:
    """The clips of workload `name` for rank `rank` (`frames` overrides the clip length: experiments only)."""
    spec = dict(WORKLOADS[name])
    if frames:
        spec["frames"] = int(frames)
    make = _msv1_clip if spec["codec"] == "msv1" else _sp_clip
    return [make(spec, rank, c) for c in range(spec.get("clips", 1))]


def make_codec(name: str, palette: Optional[bytes] = None, device: int = 0, options: Optional[dict] = None):
    """A product codec instance configured as bench.py runs workload `name` (`options`: further set_option pairs, tests)."""
    from . import MSVideo1_16bit, MSVideo1_8bit, ScreenPressor
    spec = WORKLOADS[name]
    if spec["codec"] == "sp":
        codec = ScreenPressor(W, H, 24, device=device)
    else:
        codec = MSVideo1_16bit(W, H, device=device) if spec["bits"] == 16 else MSVideo1_8bit(W, H, palette, device=device)
    codec.Preinit(36)
    if spec.get("parse"):
        codec.set_option("msv1_parse", spec["parse"])
    for key, value in (options or {}).items():
        codec.set_option(key, value)
    return codec


def digest(frame) -> str:
    """64-bit truncated SHA-256 of one RGB32 frame (numpy int32/uint32 array or bytes)."""
    data = frame if isinstance(frame, (bytes, bytearray, memoryview)) else np.ascontiguousarray(frame).tobytes()
    return hashlib.sha256(data).hexdigest()[:16]


_KERNEL_SOURCES = (      # kernel name prefix -> the files under csrc/ its code comes from
    ("msv1_fused", ("msv1_parse_kernels.hip", "msv1_lanes.h", "msv1_decode.h", "msv1_fused_hooks.h", "msv1.h")),
    ("msv1_parse", ("msv1_parse_kernels.hip", "msv1_lanes.h", "msv1.h")),
    ("msv1_blocks", ("msv1_kernels.hip", "msv1_decode.h", "msv1.h")),
    ("sp_", ("sp_kernels.hip", "sp.h")),
)


def kernel_source_digest(kernels: str) -> str:
    """A digest of the source files behind the kernels a workload names ("a + b | c"): what a recorded measurement of those kernels
    (profiles/traffic_by_workload.json) is tied to — bench.py quotes a PMC traffic figure only while the kernels' sources are the ones it was
    taken from.  (Content, not commits: the GPU box has no .git.)"""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    files = set()
    for part in kernels.replace("|", "+").split("+"):
        name = part.strip()
        for prefix, srcs in _KERNEL_SOURCES:
            if name.startswith(prefix):
                files.update(srcs)
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(f.encode())
        with open(os.path.join(here, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_digests.json")


def golden_digests(name: str, rank: int):
    """Per-clip lists of the oracle's frame digests for (workload, rank) from tests/golden/bench_digests.json
    (written by tests/golden/make_bench_digests.py from the CPU oracle), or None when not recorded."""
    import json
    if not os.path.exists(GOLDEN):
        return None
    with open(GOLDEN) as f:
        return json.load(f)["digests"].get(f"{name}/rank{rank}")


class StagedWorkload:
    """Workload `name` staged on the current device the way bench.py times it: per clip a codec instance, the
    destination frame buffers and a staged batch (inter-mode clips: the key frame is decoded up front into
    `firsts[i]`, the batch is the inter frames)."""

    def __init__(self, name: str, clips: List[Clip], device: int = 0, hip_stream: Optional[int] = None, options: Optional[dict] = None):
        import torch
        spec = WORKLOADS[name]
        self.name, self.clips, self.inter = name, clips, spec.get("mode") == "inter"
        self.codecs, self.staged, self.dsts, self.firsts = [], [], [], []
        self.pools = []
        for clip in clips:
            codec = make_codec(name, clip.palette, device=device, options=options)
            if hip_stream:
                codec.set_stream(hip_stream)
            frames, keys, first = clip.frames, clip.keys, None
            if self.inter:
                first = torch.empty(W * H, dtype=torch.int32, device=f"cuda:{device}")
                assert codec.DecompressI(frames[0], first) == 0
                frames, keys = frames[1:], keys[1:]
            # The frames a clip is decoded into come from the product's frame pool (jsp_pool_create / FramePool, the counterpart of the
            # Manager's buffer pool, Manager.hx:114-118), which PLACES a pool of this size: where the frames lie in physical memory moves
            # the store-bound kernels by a quarter from one set of allocations to the next on the same GPU, and the pool measures up to
            # sixteen candidates with the kernels' store shape and keeps a fast one (in one process, same clips: one torch
            # tensor per frame 0.63 - 0.80 of 8 TB/s on M1 by session, the pool 0.76 - 0.78 in every one).  Lab knobs:
            # JSP_BENCH_FRAME_POOL=torch (one torch tensor per frame: what rounds 1-3 timed), =1 (one torch allocation, back to back).
            how = os.environ.get("JSP_BENCH_FRAME_POOL", "probed")
            if how == "probed":
                from .codec import FramePool
                fp = FramePool(W, H, len(frames), device=device)
                self.pools.append(fp)
                dsts = list(fp.frames)
            elif how == "torch":
                dsts = [torch.empty(W * H, dtype=torch.int32, device=f"cuda:{device}") for _ in frames]
            else:
                pool = torch.empty(len(frames) * W * H, dtype=torch.int32, device=f"cuda:{device}")
                dsts = [pool[i * W * H:(i + 1) * W * H] for i in range(len(frames))]
            self.staged.append(codec.stage_batch(frames, dsts, is_key=keys))   # host stage + H2D
            self.codecs.append(codec)
            self.dsts.append(dsts)
            self.firsts.append(first)
        self.infos = [s.info() for s in self.staged]
        self.frames_per_step = sum(i["frames"] for i in self.infos)

    def step(self) -> None:
        """One pass of the hot path over every clip (asynchronous on the codecs' stream)."""
        for s in self.staged:
            s.decode()

    def sync(self) -> None:
        for c in self.codecs:
            c.sync()

    def scrub(self) -> None:
        """Every destination frame of the batches overwritten with 0xEE bytes (ordered before the next step() on the codecs' stream):
        what a later step() leaves is then what THAT step wrote, not what an earlier one left behind.  (An inter-mode clip's key
        frame in `firsts`, which a step only reads, stays.)"""
        import torch
        for c in self.codecs:
            c.sync()
        for dsts in self.dsts:
            for d in dsts:
                d.fill_(-286331154)            # 0xEEEEEEEE
        torch.cuda.synchronize()

    def lookback_fallbacks(self) -> int:
        """Staged MSVideo1 batches that were re-run through the descriptor kernels because a tile of the fused kernel gave up waiting
        (jsp_counter): 0 when the launches named by kernels() are what ran.  ScreenPressor has no such path."""
        from .codec import CodecError
        total = 0
        for c in self.codecs:
            try:
                total += c.counter("lookback_fallbacks")
            except CodecError:
                pass
        return total

    def kernels(self) -> str:
        return " | ".join(sorted({s.kernels() for s in self.staged}))

    def mismatches(self, gold) -> List[tuple]:
        """(clip, frame) of every frame in HBM whose digest differs from `gold` (golden_digests()); frames the
        oracle did not adopt ("-") must not have been adopted here either.  Call after step() + sync()."""
        bad = []
        with ThreadPoolExecutor(8) as ex:
            for ci, (dsts, g, st) in enumerate(zip(self.dsts, gold, self.staged)):
                status, adopted, _ = st.results()
                if any(status):
                    bad.append((ci, status.index(next(v for v in status if v))))
                if self.inter:
                    dsts, adopted = [self.firsts[ci]] + dsts, [1] + adopted
                if len(g) != len(dsts):
                    raise ValueError("golden digests do not describe this clip")
                for lo in range(0, len(dsts), 64):      # 64 frames (0.5 GB) in host memory at a time
                    host = [d.cpu().numpy() for d in dsts[lo:lo + 64]]
                    for k, dg in enumerate(ex.map(digest, host)):
                        want = g[lo + k]
                        if (want == "-") != (not adopted[lo + k]) or (want != "-" and want != dg):
                            bad.append((ci, lo + k))
        return bad

    def close(self) -> None:
        for s in self.staged:
            s.close()
        for c in self.codecs:
            c.StopAndClean()
        self.staged, self.codecs, self.dsts, self.firsts = [], [], [], []
        for fp in self.pools:
            fp.close()
        self.pools = []
