"""Stream-per-GPU sharding: the decode path partitions by independent AVI stream —
one codec instance, its previous-frame chain and its entropy models per stream — so ranks never
exchange frame data.  The only collective is the reduction of the job counters.

One process per GPU under `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def assign_streams(n_streams: int, world_size: int, rank: int) -> List[int]:
    """Stream i -> rank i mod world_size."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside the world")
    return [i for i in range(n_streams) if i % world_size == rank]


def reduce_counters(frames: int, pixels: int, elapsed_s: float, device=None) -> Tuple[int, int, float]:
    """Whole-job (frames, pixels) = sum over ranks; elapsed = max over ranks.  With no process group
    (single process) the inputs are returned unchanged."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return int(frames), int(pixels), float(elapsed_s)
    counts = torch.tensor([int(frames), int(pixels)], dtype=torch.int64, device=device)
    tmax = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return int(counts[0]), int(counts[1]), float(tmax[0])


def gather_per_rank(value: int, device=None) -> List[int]:
    """Every rank's `value` (frames decoded), in rank order, on every rank: what shows that the collective saw
    all N ranks.  Single process: [value]."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [int(value)]
    mine = torch.tensor([int(value)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(t[0]) for t in out]
