"""BASELINE.json configs[4] rehearsed on the one GPU a test box has: `bench.py --gpus 2 --ranks-share-device` starts two ranks of itself
(torch.distributed.run as a child process, exactly as `--gpus 2` does on a 2-GPU node) which both decode on device 0 — every rank
its own streams (the seeds and oracle digests of ITS rank), the full default line per rank (headline leg, the `also` legs, the
end-to-end legs), the counter / verdict / per-rank collectives over gloo on host tensors.  What this cannot show is RCCL between two
devices; everything else of the N-GPU path has then executed before the driver's multi-GPU run (the caller side of Manager.hx:97-142
per rank)."""
import json
import os
import subprocess
import sys

import pytest

from jsplayer_amd import workloads as wl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_of_the_default_line_on_one_device():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["JSP_BENCH_E2E_SECONDS"] = "0.5"          # (the end-to-end legs run on every rank; half a second each is enough to have run them)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-share-device", "--steps", "3", "--warmup", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    doc = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert doc["ranks_share_device"] is True and doc["ranks"] == 2 and doc["n_gpus"] == 1
    spec = wl.WORKLOADS[wl.DEFAULT]
    per_rank = spec["frames"] * spec["clips"] * 3
    assert doc["verified"] is True and doc["lookback_fallbacks"] == 0
    assert doc["per_rank_frames"] == [per_rank, per_rank] and doc["total_frames"] == 2 * per_rank
    assert doc["value"] > 0 and doc["e2e"]["streams"] == 2 and doc["e2e"]["value"] > 0
    assert doc["e2e"]["async_reruns"] == 0 and doc["e2e"]["all_threads"]["async_reruns"] == 0
    assert [a["workload"] for a in doc["also"]] == ["screenpressor_v4_1080p_iframes", "screenpressor_v4_1080p_pclip300"]
    for a in doc["also"]:
        s = wl.WORKLOADS[a["workload"]]
        n = (s["frames"] - (1 if s.get("mode") == "inter" else 0)) * s.get("clips", 1) * 3
        assert a["verified"] is True and a["per_rank_frames"] == [n, n] and a["total_frames"] == 2 * n, a["workload"]
        assert a["value"] > 0 and a["e2e"]["all_threads"]["value"] > 0
