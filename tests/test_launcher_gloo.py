"""CPU, world_size 2 over gloo: the seed-parallel launcher's sharding, barrier and timing aggregation (the N > 1
path of bench.py).  Replicas only: no tensor of the learners ever crosses ranks."""
import json
import os
import pytest
import socket
import subprocess
import sys

import torch
import torch.multiprocessing as mp

from sac_td3_cudagraphs_pytorch_amd import launcher


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = launcher.init_process_group("gloo")
    seeds = launcher.shard_seeds(8, world, rank)
    dist.barrier()
    total, secs = launcher.aggregate(dist, local_units=1000.0 * len(seeds), local_seconds=1.0 + rank)
    q.put((rank, seeds, total, secs))
    dist.destroy_process_group()


def test_two_rank_seed_sharding_and_aggregation():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in procs)
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert out[0][1] == [0, 2, 4, 6] and out[1][1] == [1, 3, 5, 7]          # disjoint, complete
    for _, _, total, secs in out:
        assert total == 8000.0 and secs == 2.0                              # sum of units, max of times


def test_single_process_is_the_identity():
    assert launcher.shard_seeds(3, 1, 0, first_seed=5) == [5, 6, 7]
    assert launcher.aggregate(None, 10.0, 2.0) == (10.0, 2.0)


class _ListRB:
    def __init__(self):
        self.rows = []

    def extend(self, td):
        import numpy as np
        n = len(td["observations"])
        for i in range(n):
            self.rows.append({k: np.array(v[i]) for k, v in td.items()})

    def __len__(self):
        return len(self.rows)


def _shared_worker(rank, world, port, q):
    import numpy as np
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = launcher.init_process_group("gloo")
    rb = launcher.SharedReplay(dist, _ListRB())
    for step in range(3):   # each rank contributes 2 rows per step, tagged with its rank and the step
        base = 100.0 * rank + 10.0 * step
        rb.extend({"observations": np.full((2, 3), base, np.float32), "next_observations": np.full((2, 3), base + 1, np.float32),
                   "actions": np.full((2, 1), base + 2, np.float32), "rewards": np.full((2, 1), base + 3, np.float32),
                   "terminations": np.array([[rank == 1], [False]]), "dones": np.array([[rank == 1], [False]])})
    q.put((rank, len(rb), [float(r["observations"][0]) for r in rb.rb.rows], [bool(r["dones"].any()) for r in rb.rb.rows]))
    dist.destroy_process_group()


def test_shared_replay_variant_keeps_all_rings_identical():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_shared_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=120) for _ in procs)
    [p.join(30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, n0, obs0, d0), (_, n1, obs1, d1) = out
    assert n0 == n1 == 12 and obs0 == obs1 and d0 == d1                 # every ring holds every rank's rows, same order
    assert obs0[:4] == [0.0, 0.0, 100.0, 100.0] and obs0[4:8] == [10.0, 10.0, 110.0, 110.0]
    assert d0[:4] == [False, False, True, False]                        # rank 1's termination flag travelled with its row


def test_sweep_enumeration_and_gpu_assignment():
    """spawner.py:147-178: one job per (env, seed), envs outermost; here job k goes to GPU k mod N."""
    jobs = launcher.sweep_jobs("medium", 3)
    assert jobs[:4] == [("HalfCheetah-v4", 0), ("HalfCheetah-v4", 1), ("HalfCheetah-v4", 2), ("Walker2d-v4", 0)] and len(jobs) == 9
    shards = [launcher.shard_jobs(jobs, 8, r) for r in range(8)]
    assert sorted(j for s in shards for j in s) == sorted(jobs) and [len(s) for s in shards] == [2, 1, 1, 1, 1, 1, 1, 1]
    assert launcher.shard_jobs(launcher.sweep_jobs("debug", 8), 8, 5) == [("Hopper-v4", 5)]      # the 8-seed node of BASELINE.json
    with pytest.raises(ValueError):
        launcher.sweep_jobs("nope", 1)


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` WITHOUT a torch.distributed.run environment (the form the driver uses for SCALE runs)
    must come back as ONE line with n_gpus == 2: the parent starts the ranks as fresh children before anything touches a
    GPU (the reference's counterpart: one OS process per seed, spawner.py:291,313-349).  Here without a GPU: --dry-run-ranks
    replaces the learner by a no-op, everything else (launch, gloo rendezvous, barrier, max-over-ranks, JSON) is real."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "2", "--dry-run-ranks"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 2 and d["dry_run"] is True and d["value"] is None
    assert d["config"]["parallelism"].startswith("2 independent seeds")
    assert lines[0] == out.stdout.splitlines()[-1] and len(lines[0]) < 1800      # the LAST stdout line, short enough for the driver's tail
    # the N = 1 line (in-process, no rendezvous) and the N = 2 line carry the same keys, so a SCALE run parses like a BENCH run
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "7", "--warmup", "2", "--dry-run-ranks"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.splitlines()[-1])
    contract = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"}
    assert contract <= set(d1) and set(d1) == set(d) and d1["n_gpus"] == 1 and set(d1["config"]) == set(d["config"])


def test_sweep_launcher_starts_one_worker_per_gpu_and_shards_the_jobs(tmp_path):
    """python -m sac_td3_cudagraphs_pytorch_amd.launcher: (env bundle x seeds) -> one worker process per GPU, job k on worker
    k mod N (spawner.py:147-178,291,313-349 semantics on one node).  --dry-run: the workers are real processes, nothing runs
    on a GPU."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "sac_td3_cudagraphs_pytorch_amd.launcher", "--env_bundle", "low", "--num_seeds", "3",
                          "--gpus", "4", "--dry-run", "--out", str(tmp_path)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    jobs = [json.loads(ln[4:]) for ln in out.stdout.splitlines() if ln.startswith("JOB ")]
    sweep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("SWEEP ")][0][6:])
    assert sweep["jobs"] == sweep["expected_jobs"] == 6 and sweep["gpus"] == 4
    want = launcher.sweep_jobs("low", 3)
    assert sorted((j["env_id"], j["seed"]) for j in jobs) == sorted(want)
    for j in jobs:
        assert j["gpu"] == want.index((j["env_id"], j["seed"])) % 4
