import sys, time
sys.path.insert(0, '.')
import torch
from jsplayer_amd import FramePool
free, total = torch.cuda.mem_get_info()
print("free GB", free / 2**30)
n = int(free * 0.30 / (1920 * 1080 * 4))          # a pool that fits three times, not four
t = time.time()
p = FramePool(1920, 1080, n)
print("frames", n, "attempts", p.attempts, "rate", round(p.store_rate), "s", round(time.time() - t, 1))
assert 1 <= p.attempts <= 4
assert int(p.frames[-1][-1].item()) == 0
p.close()
free2, _ = torch.cuda.mem_get_info()
print("free after close GB", free2 / 2**30)
assert free2 > free * 0.95
print("ok")
