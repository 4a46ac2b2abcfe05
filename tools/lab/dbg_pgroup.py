import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from jsplayer_amd import workloads as wl
from oracle_binding import OracleScreenPressor
name = "screenpressor_v4_1080p_pclip300"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
clip = wl.build_clips(name, 0, frames=n)[0]
W, H = wl.W, wl.H
codec = wl.make_codec(name)
first = torch.empty(W * H, dtype=torch.int32, device="cuda")
assert codec.DecompressI(clip.frames[0], first) == 0
dsts = [torch.empty(W * H, dtype=torch.int32, device="cuda") for _ in clip.frames[1:]]
st = codec.stage_batch(clip.frames[1:], dsts, is_key=clip.keys[1:])
print(st.kernels(), st.info()["kernel_launches"])
st.decode(); codec.sync()
orc = OracleScreenPressor(W, H, 24); orc.Preinit(36)
bufs = [np.zeros(W * H, np.int32) for _ in range(3)]
orc.DecompressI(clip.frames[0], bufs[0])
for i, src in enumerate(clip.frames[1:]):
    dst = next(b for b in bufs if b is not orc.PreviousFrame())
    orc.DecompressP(src, dst)
    ref = orc.PreviousFrame()
    got = dsts[i].cpu().numpy()
    bad = np.nonzero(ref != got)[0]
    if len(bad):
        ys, xs = bad // W, bad % W
        print("inter frame", i, "first bad px", (xs[0], ys[0]), "count", len(bad), "x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
        blk = set(zip((xs // 16).tolist(), (ys // 16).tolist()))
        print("  blocks (bx,by):", sorted(blk)[:20], len(blk))
        print("  rows within block:", sorted(set((ys % 16).tolist())), "cols within block:", sorted(set((xs % 16).tolist())))
        break
else:
    print("all", len(dsts), "inter frames match")
