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
