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


def _dry_run(extra_env=None, extra_args=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", *extra_args],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert res.returncode == 0, res.stderr.decode()[-800:]
    return json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it starts torch.distributed.run as a child process (before anything
    touches a GPU) and the ranks meet in the counter collectives; --dry-run: nothing decoded (no GPU here).  Without a GPU the RCCL
    attempt of an N-rank run cannot succeed, so this IS the rehearsal of the fall-back the first real N-GPU run may need: every rank
    fails to set up "nccl", the ranks agree on that through the rendezvous store (not through the group under test), the same
    processes go on over gloo with host tensors, and the line says what carried its collectives."""
    doc = _dry_run()
    assert doc["n_gpus"] == 2 and doc["dry_run"] is True and doc["value"] is None
    assert len(doc["per_rank_frames"]) == 2 and doc["per_rank_frames"][0] == doc["per_rank_frames"][1] > 0
    assert doc["total_frames"] == sum(doc["per_rank_frames"])
    import torch
    if not torch.cuda.is_available():
        assert doc["collectives"].startswith("gloo (rccl failed on 2 of 2 ranks; rank 0: "), doc["collectives"]


def test_forced_rccl_failure_takes_every_rank_to_gloo_and_the_line_says_so():
    """JSP_BENCH_FORCE_RCCL_FAILURE=all: the RCCL attempt raises on every rank; =1: on rank 1 only — rank 0 learns of it from the store and
    falls back with it.  Counters and per-rank frames are what the undisturbed run prints."""
    plain = _dry_run()
    forced = _dry_run({"JSP_BENCH_FORCE_RCCL_FAILURE": "all"})
    assert forced["collectives"] == "gloo (rccl failed on 2 of 2 ranks; rank 0: RuntimeError: forced by JSP_BENCH_FORCE_RCCL_FAILURE)"
    assert forced["total_frames"] == plain["total_frames"] and forced["per_rank_frames"] == plain["per_rank_frames"]
    one = _dry_run({"JSP_BENCH_FORCE_RCCL_FAILURE": "1"})
    assert one["collectives"].startswith("gloo (rccl failed on ") and one["per_rank_frames"] == plain["per_rank_frames"]
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert one["collectives"].startswith("gloo (rccl failed on 1 of 2 ranks; rank 1: RuntimeError: forced")


def test_ranks_that_share_a_device_do_not_try_rccl():
    assert _dry_run(extra_args=("--ranks-share-device",))["collectives"] == "gloo (ranks share a device)"


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


def test_counter_reduce_over_eight_devices_falls_back_to_the_host_sum_when_rccl_cannot_be_loaded():
    """jsp_reduce_counters with eight distinct device ordinals and a librccl that cannot be loaded (JSP_RCCL_LIB names nothing; the device count is
    assumed, this box has no such devices): the sums are the host's, via_rccl = 0, jsp_shard_last_error says why — the native caller's line
    (examples/jsp_play --devices) is printed either way."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes as C, sys; sys.path.insert(0, %r)\n"
        "from jsplayer_amd import _native as N\n"
        "lib = N.lib(); lib.jsp_shard_last_error.restype = C.c_char_p\n"
        "devs = (C.c_int * 8)(*range(8)); per = (C.c_uint64 * 16)(*[v for i in range(8) for v in (100 + i, 1000 * (i + 1))])\n"
        "total, via = (C.c_uint64 * 2)(), C.c_int(7)\n"
        "rc = lib.jsp_reduce_counters(devs, 8, per, total, C.byref(via))\n"
        "print(rc, list(total), via.value, lib.jsp_shard_last_error().decode())\n" % root)
    env = dict(os.environ, JSP_RCCL_LIB="/nonexistent/librccl.so", JSP_SHARD_ASSUME_DEVICES="8")
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert res.returncode == 0, res.stderr.decode()[-800:]
    rc, rest = res.stdout.decode().strip().split(" ", 1)
    assert rc == "0" and rest.startswith("[828, 36000] 0 librccl not loadable (tried /nonexistent/librccl.so)"), rest
    # an ordinal the box does not have: the same host sum, and the reason names it
    env = dict(os.environ, JSP_SHARD_ASSUME_DEVICES="4")
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert res.returncode == 0 and "[828, 36000] 0 device ordinal 4 is not one of the 4 visible" in res.stdout.decode(), res.stdout.decode()
