"""Multi-rank path on CPU: world_size-2 `gloo` process group, the same sharding + counter
reduction code bench.py runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jsplayer_amd.sharding import assign_streams, gather_per_rank, reduce_counters


def test_assignment_is_a_partition():
    for world in (1, 2, 3, 8):
        got = sorted(i for r in range(world) for i in assign_streams(11, world, r))
        assert got == list(range(11))
    assert assign_streams(8, 8, 3) == [3]
    with pytest.raises(ValueError):
        assign_streams(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank "decodes" its own streams: stream i has (i + 1) frames of 1920x1080
        mine = assign_streams(5, world, rank)
        frames = sum(i + 1 for i in mine)
        pixels = frames * 1920 * 1080
        elapsed = 0.5 + rank  # rank 1 is the slow one
        dist.barrier()
        tf, tp, te = reduce_counters(frames, pixels, elapsed)
        out.put((rank, mine, tf, tp, te, gather_per_rank(frames)))
    finally:
        dist.destroy_process_group()


def test_two_rank_counter_reduce_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] == [0, 2, 4] and results[1][1] == [1, 3]
    for _, _, tf, tp, te, per_rank in results:
        assert tf == 15 and tp == 15 * 1920 * 1080 and te == 1.5
        assert per_rank == [9, 6]


def test_single_process_passthrough():
    assert reduce_counters(3, 30, 0.25) == (3, 30, 0.25)
    assert gather_per_rank(7) == [7]


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it starts torch.distributed.run as a child process (before anything
    touches a GPU) and the ranks meet in the counter collectives; --dry-run: gloo, nothing decoded (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert res.returncode == 0, res.stderr.decode()[-800:]
    line = [l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1]
    doc = json.loads(line)
    assert doc["n_gpus"] == 2 and doc["dry_run"] is True and doc["value"] is None
    assert len(doc["per_rank_frames"]) == 2 and doc["per_rank_frames"][0] == doc["per_rank_frames"][1] > 0
    assert doc["total_frames"] == sum(doc["per_rank_frames"])


def test_native_stream_assignment_and_counter_reduce_without_a_gpu():
    """The native (one process, several devices) form of the same sharding, jsp_assign_stream / jsp_reduce_counters of the C ABI:
    stream i -> devices[i mod G]; the counters' sums are the host's when no device (hence no RCCL communicator) is there."""
    import ctypes as C
    from jsplayer_amd import _native as N
    lib = N.lib()
    devs = (C.c_int * 3)(4, 0, 2)
    assert [lib.jsp_assign_stream(i, devs, 3) for i in range(7)] == [4, 0, 2, 4, 0, 2, 4]
    for rank in range(3):      # the same rule as the one-process-per-GPU form (sharding.assign_streams)
        assert [i for i in range(7) if lib.jsp_assign_stream(i, devs, 3) == devs[rank]] == assign_streams(7, 3, rank)
    assert lib.jsp_assign_stream(-1, devs, 3) == -1 and lib.jsp_assign_stream(0, None, 3) == -1 and lib.jsp_assign_stream(0, devs, 0) == -1
    per = (C.c_uint64 * 6)(10, 1000, 20, 2000, 12, 1200)
    total, via = (C.c_uint64 * 2)(), C.c_int(7)
    assert lib.jsp_reduce_counters(devs, 3, per, total, C.byref(via)) == 0
    assert list(total) == [42, 4200]
    import torch
    if not torch.cuda.is_available():
        assert via.value == 0 and lib.jsp_device_count() == 0
    assert lib.jsp_reduce_counters(None, 3, per, total, None) != 0
