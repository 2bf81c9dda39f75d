"""Multi-GPU = independent replicas.  The hot path does not shard (one stream is sequential over blocks and denoise
steps, B = 1), so N GPUs run N prompt streams with NO collective on the data path, exactly like the
reference's inference.py: prompts sharded by rank (DistributedSampler(shuffle=False, drop_last=True), inference.py:146),
seed = seed + rank (inference.py:49).  The only collectives are the timing barrier and a MAX/SUM of scalars."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def shard_prompts(prompts: Sequence, rank: int, world: int) -> List:
    """Indices rank, rank+world, ... of the first (len // world) * world prompts."""
    per = len(prompts) // world
    return [prompts[i] for i in range(rank, per * world, world)]


def replica_seed(seed: int, rank: int) -> int:
    return seed + rank


def aggregate_throughput(frames_local: float, elapsed_local: float, device="cpu") -> Tuple[float, float]:
    """(frames of all ranks, max elapsed over ranks).  Works without a process group (single replica)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(frames_local), float(elapsed_local)
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    f = torch.tensor([frames_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(f.item()), float(t.item())
