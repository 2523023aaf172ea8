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


# The reference's sweep unit (spawner.py:21-38,147-178): an "environment bundle" names a list of tasks and a run is one job
# per (task, seed), tasks outermost.  Only the enumeration and its mapping onto GPUs are restated here -- the cluster side
# of spawner.py (SLURM / tmux scripts, wandb) is out of scope.
ENV_BUNDLES = {
    "debug": ["Hopper-v4"],
    "low": ["Hopper-v4", "Pusher-v4"],
    "medium": ["HalfCheetah-v4", "Walker2d-v4", "Ant-v4"],
    "high": ["Humanoid-v4", "HumanoidStandup-v4"],
}


def sweep_jobs(env_bundle: str, num_seeds: int) -> List[Tuple[str, int]]:
    """[(env_id, seed)] in the reference's job order: for env in bundle: for seed in range(num_seeds)."""
    if env_bundle not in ENV_BUNDLES:
        raise ValueError(f"unknown env bundle {env_bundle!r} (one of {sorted(ENV_BUNDLES)})")
    if num_seeds < 1:
        raise ValueError("num_seeds must be positive")
    return [(env, seed) for env in ENV_BUNDLES[env_bundle] for seed in range(num_seeds)]


def shard_jobs(jobs: List[Tuple[str, int]], world_size: int, rank: int) -> List[Tuple[str, int]]:
    """This rank's share of a sweep: job k runs on GPU k mod world_size (one learner per GPU at a time, in order)."""
    assert 0 <= rank < world_size
    return [j for k, j in enumerate(jobs) if k % world_size == rank]


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


class SharedReplay:
    """Optional shared-replay variant (BASELINE.json north_star; not in the reference): every rank appends the new
    transitions of ALL ranks, so each GPU's ring holds every seed's data.  The exchange is one all-gather per key
    and env step (num_envs x ~100 B per rank: latency-bound; RCCL over xGMI when the backend is nccl) -- the only
    collective on the data path, and only in this variant.  Wraps any buffer with the reference's `extend(td)` /
    `sample(B)` / `len()`; rows are appended in rank order, so all rings stay identical."""

    KEYS = ("observations", "next_observations", "actions", "rewards", "terminations", "dones")

    def __init__(self, dist, rb, device=None):
        self.dist, self.rb, self.device = dist, rb, device

    def extend(self, td) -> None:
        import numpy as np
        import torch
        if self.dist is None:
            return self.rb.extend(td)
        world = self.dist.get_world_size()
        out = {}
        for k in self.KEYS:
            if k not in td:
                continue
            x = td[k]
            x = x.detach() if hasattr(x, "detach") else torch.as_tensor(np.asarray(x))
            x = x.to(torch.float32).reshape(x.shape[0], -1).contiguous()
            if self.device is not None:
                x = x.to(self.device)
            buf = [torch.empty_like(x) for _ in range(world)]
            self.dist.all_gather(buf, x)
            out[k] = torch.cat(buf, 0).cpu().numpy()
        for k in ("terminations", "dones"):
            if k in out:
                out[k] = out[k] != 0
        self.rb.extend(out)

    def sample(self, batch_size):
        return self.rb.sample(batch_size)

    def __len__(self):
        return len(self.rb)
