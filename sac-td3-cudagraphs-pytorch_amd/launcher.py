"""Seed-parallel runs: the reference's only "multi-GPU" mechanism is spawner.py emitting N independent
`python main.py train --seed k` jobs, one GPU each (spawner.py:147-178,210-211,291).  There is no exchange step on
the path, so the MI355X counterpart is replicas only: one process per GPU (torch.distributed.run), each owning an
independent learner for its share of the seeds; torch.distributed is used for the start barrier and for combining
the timings, never on the data path.
"""
from __future__ import annotations

import os
from typing import List, Tuple


def rank_info() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment (1 process: 0, 1, 0)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_seeds(num_seeds: int, world_size: int, rank: int, first_seed: int = 0) -> List[int]:
    """Seeds first_seed .. first_seed+num_seeds-1 dealt round-robin over the ranks (spawner.py:166-178 numbers
    its jobs the same way: one job per (env, seed))."""
    assert 0 <= rank < world_size
    return [first_seed + s for s in range(num_seeds) if s % world_size == rank]


def init_process_group(backend: str = "nccl"):
    """Join the job's process group (nccl == RCCL on ROCm; gloo for CPU rehearsals).  Returns the module or None."""
    rank, world, local = rank_info()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return dist


def aggregate(dist, local_units: float, local_seconds: float, device=None) -> Tuple[float, float]:
    """Whole-job figures: units summed over ranks, time = the slowest rank's.  Returns (total_units, seconds)."""
    if dist is None:
        return float(local_units), float(local_seconds)
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([local_units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())
