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
# per (task, seed), tasks outermost.  The cluster side of spawner.py (SLURM / tmux scripts, wandb) is out of scope; `main()`
# below runs the sweep on this node, one learner process per GPU.
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
    transitions of ALL ranks, so each GPU's ring holds every seed's data, in rank order (all rings stay identical).  The
    exchange is the only collective on the data path, and only in this variant; it is latency-bound (num_envs x ~100 B per
    rank and env step), so `every=K` batches K env steps into one exchange.

    Two paths, chosen by what `rb` is:
      * device path (`rb` is this package's ReplayBuffer bound to an engine, `device` a cuda device): the rank's rows are packed
        into the ring's record layout, ONE all_gather_into_tensor (RCCL over xGMI when the backend is nccl) fills a device slab
        [world x rows, record] and the engine appends the slab with one kernel (`sactd3_rb_extend_device`): no device -> host
        copy, no per-key collectives;
      * host path (any buffer with the reference's `extend(td)` / `sample(B)` / `len()`; gloo rehearsals): one all_gather per
        key, rows appended through `rb.extend`."""

    KEYS = ("observations", "next_observations", "actions", "rewards", "terminations", "dones")

    def __init__(self, dist, rb, device=None, every: int = 1, force_device_path: bool = False):
        self.dist, self.rb, self.device, self.every = dist, rb, device, max(int(every), 1)
        self._pending: list = []
        eng = getattr(rb, "_engine", None)
        self._engine = eng if (eng is not None and device is not None and str(device).startswith("cuda")
                               and (force_device_path or dist is not None)) else None
        self._slab = None                         # kept alive until the engine has consumed it

    # -- device path
    def _exchange_device(self, td_list) -> None:
        import numpy as np
        import torch
        eng = self._engine
        rec = np.concatenate([eng.pack_records(_host(td["observations"]), _host(td["actions"]), _host(td["rewards"]),
                                               _host(td["next_observations"]), _host(td["dones"] if "dones" in td else td["terminations"]))
                              for td in td_list])
        world = self.dist.get_world_size() if self.dist is not None else 1
        eng.sync()                                # the previous slab has been read by now: its memory may be reused
        mine = torch.from_numpy(rec).to(self.device, non_blocking=False)
        if self.dist is not None and world > 1:
            slab = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=torch.float32, device=self.device)
            self.dist.all_gather_into_tensor(slab, mine)
        else:
            slab = mine
        torch.cuda.current_stream(slab.device).synchronize()      # the collective ran on torch's stream, the ingest runs on the engine's
        self._slab = slab
        eng.rb_extend_device(slab.data_ptr(), slab.shape[0])

    # -- host path
    def _exchange_host(self, td) -> None:
        import numpy as np
        import torch
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

    def extend(self, td) -> None:
        if self._engine is not None:
            self._pending.append(td)
            if len(self._pending) >= self.every:
                pend, self._pending = self._pending, []
                self._exchange_device(pend)
            return
        if self.dist is None:
            return self.rb.extend(td)
        self._exchange_host(td)

    def flush(self) -> None:
        if self._engine is not None and self._pending:
            pend, self._pending = self._pending, []
            self._exchange_device(pend)

    def sample(self, batch_size):
        return self.rb.sample(batch_size)

    def __len__(self):
        return len(self.rb)


def _host(x):
    import numpy as np
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


# ------------------------------------------------------------------------------------------------ the sweep runner
# spawner.py:147-178,240-349 turns (env bundle x seeds) into one `python main.py train --env_id .. --seed ..` job per pair and
# hands each to Slurm with --gres=gpu:1 (or to a tmux window).  Here the node IS the scheduler: the parent process (which
# never touches a GPU) starts one worker process per GPU, worker r runs jobs r, r + N, r + 2N, ... one after the other with
# `loop.train`, each job with its own engine, replay ring, seed and output directory, and prints one JSON line per job;
# the parent relays them and ends with a summary line.  No collective, no shared state: replicas only.
ENV_DIMS = {   # (ob_dim, ac_dim, action bound) of the Gymnasium MuJoCo -v4 tasks named by the bundles (public task specs)
    "Hopper-v4": (11, 3, 1.0), "Pusher-v4": (23, 7, 2.0), "HalfCheetah-v4": (17, 6, 1.0), "Walker2d-v4": (17, 6, 1.0),
    "Ant-v4": (27, 8, 1.0), "Humanoid-v4": (376, 17, 0.4), "HumanoidStandup-v4": (376, 17, 0.4),
}
DEFAULTS = {   # tasks/defaults/sac.yml:1-48 (td3.yml differs in the keys listed under "td3")
    "sac": dict(num_envs=4, action_repeat=1, measure_burnin=3, num_timesteps=10_000_000, learning_starts=5000, eval_steps=10,
                eval_every=10_000, layer_norm=True, actor_lr=3e-4, qnets_lr=1e-3, clip_norm=0.0, segment_len=1, batch_size=256,
                gamma=0.99, rb_capacity=1_000_000, polyak=0.005, prefer_td3_over_sac=False, bcq_style_targ_mix=False,
                actor_update_delay=2, crit_targ_update_freq=1, alpha_init=0.2, autotune=True, log_alpha_lr=1e-3),
    "td3": dict(prefer_td3_over_sac=True, bcq_style_targ_mix=True, qnets_lr=3e-4, actor_noise_std=0.1, targ_actor_smoothing=True,
                td3_std=0.2, td3_c=0.5),
}


def job_config(algo: str, env_id: str, seed: int, **over):
    from types import SimpleNamespace
    cfg = dict(DEFAULTS["sac"])
    if algo == "td3":
        cfg.update(DEFAULTS["td3"])
    cfg.update(env_id=env_id, seed=seed, cudagraphs=False, compile=False)
    cfg.update({k: v for k, v in over.items() if v is not None})
    return SimpleNamespace(**cfg)


def run_job(algo: str, env_id: str, seed: int, device_index: int, out_dir, env_factory=None, **over):
    """One (env, seed) run = what one `main.py train` process of the reference does (main.py:126-195), on one GPU."""
    import json
    import time
    from pathlib import Path
    import numpy as np
    import torch
    from . import loop
    from .agent import Agent, ReplayBuffer
    cfg = job_config(algo, env_id, seed, **over)
    o, a, bound = ENV_DIMS[env_id]
    if env_factory is None:   # no simulator in this image: the synthetic vector env has the task's shapes and the gymnasium protocol
        make = lambda n: loop.SyntheticVecEnv(o, a, n, horizon=200, term_at=6.0, bound=bound)
    else:
        make = lambda n: env_factory(env_id, n, seed)
    env, eval_env = make(cfg.num_envs), make(1)
    torch.manual_seed(seed)                                           # main.py:145-146
    rb = ReplayBuffer(cfg.rb_capacity)
    agent = Agent({"ob_shape": (o,), "ac_shape": (a,)}, np.full(a, -bound, np.float32), np.full(a, bound, np.float32),
                  torch.device("cuda", device_index), cfg, rb, seed=seed)
    run_dir = Path(out_dir) / f"{env_id}__{algo}__seed{seed:02d}"
    tab = loop.Tabular(run_dir)
    ev = loop.Evaluator(cfg, eval_env, agent, tabular=tab, ckpt_dir=run_dir)
    t0 = time.time()
    metrics = loop.train(cfg, env, agent, fused=True, evaluator=ev)
    agent.engine.sync()
    dt = time.time() - t0
    tab.close()
    out = {"env_id": env_id, "algo": algo, "seed": seed, "gpu": device_index, "timesteps": int(agent.timesteps_so_far),
           "gradient_steps": int(agent.qnet_updates_so_far), "actor_updates": int(agent.actor_updates_so_far), "seconds": dt,
           "gradient_steps_per_s": agent.qnet_updates_so_far / max(dt, 1e-9), "best_eval_return": float(agent.best_eval_ep_ret),
           "final_metrics": metrics, "dir": str(run_dir)}
    (run_dir / "summary.json").write_text(json.dumps(out))
    agent.engine.close()
    return out


def _worker_main(args) -> int:
    import importlib
    import json
    jobs = shard_jobs(sweep_jobs(args.env_bundle, args.num_seeds), args.world, args.rank)
    factory = None
    if args.env_factory:
        mod, fn = args.env_factory.split(":")
        factory = getattr(importlib.import_module(mod), fn)
    for env_id, seed in jobs:
        if args.dry_run:
            out = {"env_id": env_id, "algo": args.algo, "seed": seed, "gpu": args.rank, "dry_run": True}
        else:
            out = run_job(args.algo, env_id, seed, args.device if args.device >= 0 else args.rank, args.out, factory,
                          num_timesteps=args.num_timesteps, learning_starts=args.learning_starts, eval_every=args.eval_every,
                          eval_steps=args.eval_steps, batch_size=args.batch_size, rb_capacity=args.rb_capacity)
        print("JOB " + json.dumps(out), flush=True)
    return 0


def main(argv=None) -> int:
    import argparse
    import json
    import subprocess
    import sys
    import time
    ap = argparse.ArgumentParser(prog="python -m sac_td3_cudagraphs_pytorch_amd.launcher",
                                 description="(env bundle x seeds) sweep, one learner process per GPU (spawner.py semantics, no Slurm / tmux)")
    ap.add_argument("--env_bundle", default="debug", choices=sorted(ENV_BUNDLES))
    ap.add_argument("--num_seeds", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--algo", default="sac", choices=["sac", "td3"])
    ap.add_argument("--out", default="runs")
    ap.add_argument("--num_timesteps", type=int, default=None)
    ap.add_argument("--learning_starts", type=int, default=None)
    ap.add_argument("--eval_every", type=int, default=None)
    ap.add_argument("--eval_steps", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--rb_capacity", type=int, default=None)
    ap.add_argument("--env_factory", default=None, help="module:function(env_id, num_envs, seed) -> gymnasium-protocol vector env (default: synthetic)")
    ap.add_argument("--device", type=int, default=-1, help="run every worker on this device ordinal (one-GPU rehearsals of N workers)")
    ap.add_argument("--dry-run", action="store_true", help="enumerate and shard the jobs, start the workers, run nothing on a GPU")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--world", type=int, default=1, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if args.worker:
        return _worker_main(args)
    # parent: touches no GPU (no torch import, no HIP call); one fresh child per GPU
    jobs = sweep_jobs(args.env_bundle, args.num_seeds)
    world = max(1, min(args.gpus, len(jobs)))
    base = [sys.executable, "-m", "sac_td3_cudagraphs_pytorch_amd.launcher", "--worker", "--world", str(world)] + \
           [x for x in (argv if argv is not None else sys.argv[1:]) if x != "--worker"]
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    procs = [subprocess.Popen(base + ["--rank", str(r)], env=env, stdout=subprocess.PIPE, text=True) for r in range(world)]
    done, rc = [], 0
    for p in procs:
        out, _ = p.communicate()
        rc = rc or p.returncode
        for ln in out.splitlines():
            if ln.startswith("JOB "):
                done.append(json.loads(ln[4:]))
                print(ln, flush=True)
    dt = time.time() - t0
    steps = sum(j.get("gradient_steps", 0) for j in done)
    print("SWEEP " + json.dumps({"env_bundle": args.env_bundle, "num_seeds": args.num_seeds, "gpus": world, "jobs": len(done),
                                 "expected_jobs": len(jobs), "seconds": dt, "gradient_steps": steps,
                                 "aggregate_gradient_steps_per_s": steps / max(dt, 1e-9)}), flush=True)
    return rc if rc else (0 if len(done) == len(jobs) else 1)


if __name__ == "__main__":
    raise SystemExit(main())
