"""lab: the tree's library against the snapshot under .lab_prev/ in ONE process, decoding the SAME staged clips into the SAME destination frames,
alternately (every process gets its own luck with where its pool lies — runs of bench.py in two processes differ by more than most kernel
changes).  Both libraries are loaded side by side: the package under .lab_prev is imported as `jsplayer_amd_prev` (its modules import each
other relatively).  Digests of the frames each side leaves are compared with the golden ones once (after a 0xEE scrub).

    .lab_prev:  git archive <commit> jsplayer_amd include bench.py tests oracle | tar -x -C .lab_prev; make -C .lab_prev/jsplayer_amd/csrc
    usage:      [LAB_SIDES="tag ..."] python tools/lab/two_libs.py <workload> [rounds=4] [steps=20]      (tags: tools/lab/mk_variant.sh)
"""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import jsplayer_amd as new_pkg  # noqa: E402
from jsplayer_amd import workloads as wl  # noqa: E402
from jsplayer_amd.codec import FramePool  # noqa: E402


def load_side(tag):
    """The package under .lab_<tag>/ imported as jsplayer_amd_<tag> (with its own libjsplayer_amd.so); returns its workloads module."""
    d = os.path.join(ROOT, ".lab_" + tag, "jsplayer_amd")
    name = "jsplayer_amd_" + tag
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return importlib.import_module(name + ".workloads")


def load_prev():
    return load_side("prev")


def main():
    name = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    sides = {"prev": load_prev(), "new": wl}
    for tag in os.environ.get("LAB_SIDES", "").split():       # further variants: LAB_SIDES="nomask oldstore" -> .lab_nomask/, .lab_oldstore/
        # "prev#2" / "new#2": the SAME library staged a second time (codecs, tables and all of its own) — what two stagings of one kernel differ by
        base = tag.split("#")[0]
        sides[tag] = sides[base] if base in sides else load_side(base)
    spec = wl.WORKLOADS[name]
    inter = spec.get("mode") == "inter"
    clips = wl.build_clips(name, 0)
    gold = wl.golden_digests(name, 0)
    W, H = wl.W, wl.H
    pools, dsts_all, firsts = [], [], []
    for clip in clips:
        n = len(clip.frames) - (1 if inter else 0)
        fp = FramePool(W, H, n, device=0)
        pools.append(fp)
        dsts_all.append(list(fp.frames))
        firsts.append(torch.empty(W * H, dtype=torch.int32, device="cuda:0") if inter else None)
    print(f"{name}: pools {[round(p.store_rate) for p in pools]} GB/s after {[p.attempts for p in pools]} candidates", flush=True)
    staged = {}
    for side, mod in sides.items():
        items = []
        for clip, dsts, first in zip(clips, dsts_all, firsts):
            codec = mod.make_codec(name, clip.palette, device=0)
            for kv in os.environ.get("LAB_OPT_" + side.replace("#", "_"), "").split(","):      # LAB_OPT_new_2="msv1_parse_ahead=off": options of that side's codecs
                if "=" in kv:
                    codec.set_option(*kv.split("=", 1))
            frames, keys = clip.frames, clip.keys
            if inter:
                assert codec.DecompressI(frames[0], first) == 0
                frames, keys = frames[1:], keys[1:]
            items.append((codec, codec.stage_batch(frames, dsts, is_key=keys)))
        staged[side] = items

    def run(side, n):
        for _ in range(n):
            for _, st in staged[side]:
                st.decode()
        for codec, _ in staged[side]:
            codec.sync()

    def verify(side):
        for dsts in dsts_all:
            for d in dsts:
                d.fill_(-286331154)
        torch.cuda.synchronize()
        run(side, 1)
        bad = 0
        for ci, dsts in enumerate(dsts_all):
            g = gold[ci][1:] if inter else gold[ci]
            for k in range(0, len(dsts), 37):            # a sample: every 37th frame
                if g[k] != "-" and wl.digest(dsts[k].cpu().numpy()) != g[k]:
                    bad += 1
        return bad

    for side in sides:
        print(f"{side}: kernels {staged[side][0][1].kernels()}; sampled digests wrong: {verify(side)}", flush=True)
    info = staged["new"][0][1].info()
    moved = sum(min(st.info()["algorithmic_bytes"], st.info().get("moved_bytes") or st.info()["algorithmic_bytes"]) for _, st in staged["new"])
    for side in sides:
        run(side, 3)
    best = {s: 1e9 for s in sides}
    order = list(sides)
    for r in range(rounds):
        order = order[1:] + order[:1]                          # (every side gets every place in the round)
        for side in order:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(side, steps)
            dt = (time.perf_counter() - t0) / steps
            best[side] = min(best[side], dt)
            print(f"round {r} {side:8s}: {dt * 1e3:.4f} ms per step  {moved / dt / 8e12:.4f} of 8 TB/s", flush=True)
    print("best: " + ", ".join(f"{s} {best[s] * 1e3:.4f} ms ({best[s] / best['prev']:.4f} of prev)" for s in sides), flush=True)
    del info


if __name__ == "__main__":
    main()
