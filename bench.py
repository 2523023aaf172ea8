#!/usr/bin/env python3
"""Headline benchmark: gradient-steps/s of the SAC hot path (BASELINE.json configs[1]: SAC Hopper-v4, batch 256,
hipGraph-captured update step on 1x MI355X), seeds sharded one per GPU for --gpus N (replicas only, no collective).

One "step" = one loop iteration of the reference's orchestrator.py:337-352 on synthetic transitions already
resident in HBM: replay index draw + gather, critic update, (every 3rd iteration) 2x actor+alpha update on the
same batch, Polyak.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hopper_sac|halfcheetah_td3|humanoid_sac]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # SURVEY.md section 8d / BASELINE.md section 3
    "hopper_sac": dict(env="Hopper-v4", o=11, a=3, bound=1.0, td3=False, batch=256, rows=100_000, capacity=1_000_000),
    "halfcheetah_td3": dict(env="HalfCheetah-v4", o=17, a=6, bound=1.0, td3=True, batch=256, rows=100_000, capacity=1_000_000),
    "humanoid_sac": dict(env="Humanoid-v4", o=376, a=17, bound=0.4, td3=False, batch=1024, rows=1_000_000, capacity=1_000_000),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def gather_algo_bytes(o, a, batch):
    """2*B*T + 4*B, T = 4*(2o+a+1)+1 (s, a, r, s', one done byte; read once + write once; int32 index)."""
    return 2 * batch * (4 * (2 * o + a + 1) + 1) + 4 * batch


def make_engine(w, seed, device_id):
    import torch
    import sac_td3_cudagraphs_pytorch_amd as pkg
    from sac_td3_cudagraphs_pytorch_amd import schema
    cfg = pkg.Config(ob_dim=w["o"], ac_dim=w["a"], batch_size=w["batch"], rb_capacity=w["capacity"], max_envs=4,
                     prefer_td3_over_sac=w["td3"], bcq_style_targ_mix=w["td3"], qnets_lr=3e-4 if w["td3"] else 1e-3,
                     seed=seed, device_id=device_id, use_graphs=os.environ.get("SACTD3_BENCH_GRAPHS", "1") != "0")   # (0: same launches, no hipGraph)
    eng = pkg.Engine(cfg, [-w["bound"]] * w["a"], [w["bound"]] * w["a"])
    torch.manual_seed(seed)  # reference init (agents/nets.py:34-49) under the run's seed
    actor, critics = schema.reference_initial_params(w["o"], w["a"], w["td3"], True)
    for which, flat in ((0, actor), (2, actor), (1, critics), (3, critics)):
        eng.set_params(which, flat)
    eng.rb_fill_synthetic(w["rows"], seed=0)  # data seed 0
    eng.sync()
    return eng


def oracle_rate(w, device, seconds, threads=None, min_iters=0):
    """iterations/s of the plain-torch restatement (oracle) on `device`: the reference's CPU / eager-ROCm path."""
    import torch
    from oracle.sac_td3_ref import Hps, RefAgent
    if threads:
        torch.set_num_threads(threads)
    o, a, B = w["o"], w["a"], w["batch"]
    hps = (Hps.td3 if w["td3"] else Hps.sac)(batch_size=B)
    torch.manual_seed(0)
    ag = RefAgent(o, a, [-w["bound"]] * a, [w["bound"]] * a, hps, device=device)
    ag.keep_trace = False                     # the parity tests' bookkeeping (gradient clones) is not part of the step
    n = min(w["rows"], 100_000)
    g = torch.Generator().manual_seed(0)
    data = [torch.randn(n, o, generator=g), (torch.rand(n, a, generator=g) * 2 - 1) * w["bound"], torch.randn(n, generator=g),
            torch.randn(n, o, generator=g), torch.rand(n, generator=g) < 0.01]
    data = [d.to(device) for d in data]
    sync = (lambda: torch.cuda.synchronize()) if str(device).startswith("cuda") else (lambda: None)

    def one(i):
        idx = torch.randint(0, n, (B,), device=device)
        b = ag.to_batch(*[d[idx] for d in data])
        ag.iteration(b, i)

    for i in range(6):
        one(i)
    sync()
    t0, i = time.perf_counter(), 0
    while True:
        one(i)
        i += 1
        if i % 3 == 0:
            sync()
            if time.perf_counter() - t0 >= seconds and i >= min_iters:
                break
    sync()
    return i / (time.perf_counter() - t0), i


def oracle_graph_rate(w, seconds):
    """BASELINE.md baseline (iii): the same restatement on cuda:0 with update_qnets / update_actor captured as
    torch.cuda.CUDAGraph and replayed -- the closest stand-in for the reference's CudaGraphModule mode (orchestrator.py:
    313-315; sampling and the target update stay eager between the two graphs, as there).  Returns (iterations/s, n)."""
    import torch
    from oracle.sac_td3_ref import Hps, RefAgent
    o, a, B = w["o"], w["a"], w["batch"]
    hps = (Hps.td3 if w["td3"] else Hps.sac)(batch_size=B)
    torch.manual_seed(0)
    ag = RefAgent(o, a, [-w["bound"]] * a, [w["bound"]] * a, hps, device="cuda")
    ag.keep_trace = False
    for opt in (ag.q_optimizer, ag.actor_optimizer, getattr(ag, "alpha_optimizer", None)):
        if opt is not None:
            for grp in opt.param_groups:
                grp["capturable"] = True
    n = min(w["rows"], 100_000)
    g = torch.Generator().manual_seed(0)
    data = [torch.randn(n, o, generator=g), (torch.rand(n, a, generator=g) * 2 - 1) * w["bound"], torch.randn(n, generator=g),
            torch.randn(n, o, generator=g), torch.rand(n, generator=g) < 0.01]
    data = [d.cuda() for d in data]
    static = ag.to_batch(*[d[:B].clone() for d in data])       # the graphs read these tensors; each sample is copied into them

    def load():
        idx = torch.randint(0, n, (B,), device="cuda")
        for dst, src in zip((static.observations, static.actions, static.rewards, static.next_observations, static.dones), data):
            dst.copy_(src[idx].to(dst.dtype))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                               # warm-up outside capture (allocations, Adam state)
        for _ in range(3):
            load(); ag.update_qnets(static); ag.update_actor(static); ag.update_targ_nets()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gq, ga = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gq):
        ag.update_qnets(static)
    with torch.cuda.graph(ga):
        ag.update_actor(static)

    def one(i):
        load()
        gq.replay()
        ag.qnet_updates_so_far += 1
        if i % (hps.actor_update_delay + 1) == 0:
            for _ in range(hps.actor_update_delay):
                ga.replay()
        ag.update_targ_nets()

    for i in range(6):
        one(i)
    torch.cuda.synchronize()
    t0, i = time.perf_counter(), 0
    while True:
        one(i)
        i += 1
        if i % 3 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
    return i / (time.perf_counter() - t0), i


def parity_deltas(w, device_id, iters=6):
    """SURVEY 8(d) 'parity deltas': `iters` iterations of the workload (same batches, same injected noise, reference
    schedule) through the engine's call-by-call API and through the oracle; max abs difference of the reported scalars
    and of the parameters afterwards.  Part of the baseline leg: the oracle is the checker, never the thing measured."""
    import numpy as np
    import torch
    import sac_td3_cudagraphs_pytorch_amd as pkg
    from sac_td3_cudagraphs_pytorch_amd import _lib, schema
    from oracle.sac_td3_ref import Hps, RefAgent
    o, a, B, td3 = w["o"], w["a"], w["batch"], w["td3"]
    hps = (Hps.td3 if td3 else Hps.sac)(batch_size=B)
    torch.manual_seed(0)
    lo, hi = [-w["bound"]] * a, [w["bound"]] * a
    ref = RefAgent(o, a, lo, hi, hps)
    eng = pkg.Engine(pkg.Config.from_hps(hps, o, a, rb_capacity=1024, seed=0, device_id=device_id), lo, hi)
    nh = a if td3 else 2 * a
    flat_a = lambda m: schema.dict_to_flat({k: v for k, v in m.state_dict().items() if k.startswith(("fc_stack", "head"))}, o, nh, True)
    flat_c = lambda ms: np.concatenate([schema.dict_to_flat(q.state_dict(), o + a, 1, True) for q in ms])
    eng.set_params(_lib.ACTOR, flat_a(ref.actor)); eng.set_params(_lib.ACTOR_TARGET, flat_a(ref.actor_target))
    eng.set_params(_lib.CRITICS, flat_c(ref.qnets)); eng.set_params(_lib.CRITICS_TARGET, flat_c(ref.qnets_target))
    g = torch.Generator().manual_seed(1)
    worst = {}
    for i in range(iters):
        obs, nobs = torch.randn(B, o, generator=g), torch.randn(B, o, generator=g)
        act = (torch.rand(B, a, generator=g) * 2 - 1) * w["bound"]
        rew, done = torch.randn(B, generator=g), torch.rand(B, generator=g) < 0.01
        noise = {"critic": torch.randn(B, a, generator=g), "actor": [torch.randn(B, a, generator=g) for _ in range(2)],
                 "alpha": [torch.randn(B, a, generator=g) for _ in range(2)]}
        want = ref.iteration(ref.to_batch(obs, act, rew, nobs, done), i, noise)
        eng.load_batch(obs, act, rew, nobs, done)
        eng.set_noise(_lib.SITE_CRITIC, noise["critic"])
        eng.update_qnets()
        if i % (hps.actor_update_delay + 1) == 0:
            for j in range(hps.actor_update_delay):
                eng.set_noise(_lib.SITE_ACTOR0, noise["actor"][j]); eng.set_noise(_lib.SITE_ALPHA0, noise["alpha"][j])
                eng.update_actor()
        eng.update_targ_nets(i + 1)
        got = eng.read_metrics()
        for k, v in want.items():
            worst[k] = max(worst.get(k, 0.0), abs(got[k] - float(v)))
    worst["params/critics"] = float(np.abs(eng.get_params(_lib.CRITICS) - flat_c(ref.qnets)).max())
    worst["params/actor"] = float(np.abs(eng.get_params(_lib.ACTOR) - flat_a(ref.actor)).max())
    eng.close()
    return {"iterations": iters, "max_abs_delta": worst,
            "note": "fp32 vs the plain-PyTorch oracle on the same batches and noise; parameters after the Adam steps can differ by "
                    "up to 2 lr per step where a gradient is near zero (sign-like first steps), see tests/helpers.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="hopper_sac", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-baselines", action="store_true")
    ap.add_argument("--eager-profile", type=int, default=0, metavar="ITERS",
                    help="run ONLY the eager PyTorch-ROCm restatement for ITERS iterations (for rocprofv3 launch counts)")
    ap.add_argument("--gather-profile", type=int, default=0, metavar="BATCH",
                    help="run ONLY the replay gather at this batch size 20 times (for rocprofv3 --pmc traffic counters)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]
    if args.eager_profile:
        v, n = oracle_rate(w, "cuda", 0.0, min_iters=args.eager_profile)
        print(json.dumps({"eager_rocm_iterations": n, "rate": v}))
        return
    if args.gather_profile:
        eng = make_engine(w, 0, 0)
        us, by = eng.time_gather_sweep(args.gather_profile, 20)
        print(json.dumps({"batch": args.gather_profile, "us": us, "algo_bytes": by, "GB/s": by / us * 1e-3}))
        return

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch
    # rehearsal knobs (one-GPU boxes): SACTD3_BENCH_BACKEND=gloo and SACTD3_BENCH_DEVICE=0 let N ranks share one card
    backend = os.environ.get("SACTD3_BENCH_BACKEND", "nccl")
    if "SACTD3_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SACTD3_BENCH_DEVICE"])
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    eng = make_engine(w, seed=rank, device_id=local)  # seed = GPU index, one independent learner per GPU
    delay = 2
    it = 0
    for _ in range(args.warmup):
        eng.step(it % (delay + 1) == 0)
        it += 1
    eng.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step(it % (delay + 1) == 0)
        it += 1
    eng.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # slowest rank defines the job's time
        dt = float(t.item())
        dist.barrier()
    metrics = eng.read_metrics()

    if rank == 0:
        out = {
            "metric": "gradient-steps/sec (SAC, batch=256) at 1 GPU + 8-seed node; replay HBM GB/s",
            "value": world * args.steps / dt, "unit": "gradient-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'TD3' if w['td3'] else 'SAC'} {w['env']} batch={w['batch']}, 2x256 MLP + LayerNorm, "
                                   f"{w['rows']} rows resident in a {w['capacity']}-row HBM replay ring, "
                                   "one hipGraph launch per iteration (1 critic update, 2 actor+alpha updates every 3rd, Polyak)",
                       "parallelism": f"{world} independent seeds, one per GPU, no collective"},
            # one iteration = 1 critic update (+ Polyak) and, every 3rd iteration, 2 actor(+alpha) updates
            "critic_updates_per_s": world * args.steps / dt, "actor_updates_per_s": world * args.steps / dt * 2.0 / 3.0,
            "kernels_per_iteration": {"critic_only": eng.graph_kernel_count(2), "critic_plus_2_actor": eng.graph_kernel_count(3)},
            "final_metrics": metrics,
        }
        # roofline of the replay gather (the path's HBM-bound kernel; north_star's "replay HBM GB/s"):
        # algorithmic bytes per launch / average launch duration from HIP events on the engine's stream
        us = eng.time_kernel("gather", 2000)
        ab = gather_algo_bytes(w["o"], w["a"], w["batch"])
        out["roofline"] = {"kernel": "k_gather", "bound": "hbm", "achieved": ab / us * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ab / us * 1e-3 / HBM_PEAK_GBS, "traffic": None, "algo_bytes_per_launch": ab, "avg_launch_us": us,
                           "note": "B=256 is launch/latency-bound by construction; see gather_batch_sweep for the bandwidth end "
                                   "and profiles/r01_gather_humanoid_b65536_pmc.csv for the PMC traffic (1.01x algorithmic)"}
        # the kernel that takes the most time: hidden layers of the 4 critics (fp32 MFMA); FLOPs = 2*MAC of both layers
        us_t = eng.time_kernel("trunk_critics", 500)
        fl = 4 * 2.0 * w["batch"] * 256 * ((w["o"] + w["a"]) + 256)
        out["roofline_mfma"] = {"kernel": "k_nt (4-net critic trunk)", "bound": "mfma", "achieved": fl / us_t * 1e-6, "peak": 157.3,
                                "unit": "TFLOP/s", "frac": fl / us_t * 1e-6 / 157.3, "flops_per_launch": fl, "avg_launch_us": us_t}
        # acting side (agents/agent.py:172-181): one predict round trip = H2D of the observations, 2 kernels, D2H + sync
        import numpy as np
        ob = np.zeros((4, w["o"]), np.float32)
        for _ in range(20):
            eng.predict(ob, True)
        tp = time.perf_counter()
        for _ in range(300):
            eng.predict(ob, True)
        out["predict_round_trip_us"] = (time.perf_counter() - tp) / 300 * 1e6
        # rb.extend of one env step (num_envs = 4 rows, orchestrator.py:100-113): host pack + async H2D + length publish
        rows = [np.zeros((4, w["o"]), np.float32), np.zeros((4, w["a"]), np.float32), np.zeros(4, np.float32),
                np.zeros((4, w["o"]), np.float32), np.zeros(4, bool)]
        for _ in range(700):   # (the HIP runtime grows its signal pools during the first few hundred async copies)
            eng.rb_extend(*rows)
        eng.sync()
        tp = time.perf_counter()
        for _ in range(300):
            eng.rb_extend(*rows)
        eng.sync()
        out["rb_extend_call_us"] = (time.perf_counter() - tp) / 300 * 1e6
        # SURVEY 8(d)'s optional second figure: the loop as orchestrator.py:325-352 runs it, minus the simulator -- per
        # iteration one synchronous predict for the 4 envs, one rb.extend of their 4 transitions, one fused update
        tp = time.perf_counter()
        for i in range(600):
            eng.predict(ob, True)
            eng.rb_extend(*rows)
            eng.step(i % 3 == 0)
        eng.sync()
        out["loop_with_acting_per_s"] = 600 / (time.perf_counter() - tp)
        if world == 1 and not args.no_baselines:
            # Seed sweeps are the reference's unit of work (spawner.py: one job per seed) and one learner leaves most of an
            # MI355X idle: S independent engines (own stream, own graphs, seeds 0..S-1) in this process, one host thread each,
            # aggregate gradient-steps/s.  An extra figure -- `value` above stays the one-learner-per-GPU number.
            multi = {}
            for S in (2, 4, 8):
                engs = [eng] + [make_engine(w, seed=100 + k, device_id=local) for k in range(1, S)]
                for i in range(150):
                    for e2 in engs:
                        e2.step(i % 3 == 0)
                for e2 in engs:
                    e2.sync()
                import threading

                def drive(e2):                              # one host thread per learner (ctypes drops the GIL in the call)
                    for i in range(600):
                        e2.step(i % 3 == 0)
                    e2.sync()
                th = [threading.Thread(target=drive, args=(e2,)) for e2 in engs]
                tp = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                multi[str(S)] = S * 600 / (time.perf_counter() - tp)
                for e2 in engs[1:]:
                    e2.close()
            out["learners_per_gpu_aggregate_steps_per_s"] = multi
            # large-batch asymptote of the same kernel (B=256 is launch-bound by construction)
            sweep = {}
            for bs in (256, 4096, 65536):
                us_b, by = eng.time_gather_sweep(bs, 200 if bs < 65536 else 50)
                sweep[str(bs)] = {"us": us_b, "GB/s": by / us_b * 1e-3}
            out["gather_batch_sweep"] = sweep
            eng.close()
            # north_star's "replay-gather HBM GB/s as fraction of 8 TB/s": the same kernel on the Humanoid-v4 record
            # (3136 B/row) out of a full 1M-row (3.1 GB, far beyond the 256 MB Infinity Cache) ring, 65 536 rows per launch
            wh = WORKLOADS["humanoid_sac"]
            eh = make_engine(wh, 0, local)
            us_h, by_h = eh.time_gather_sweep(65536, 50)
            out["replay_gather_hbm"] = {"record": "Humanoid-v4 (o=376, a=17), 1M-row ring", "rows_per_launch": 65536, "us": us_h,
                                        "algo_bytes": by_h, "achieved": by_h / us_h * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": by_h / us_h * 1e-3 / HBM_PEAK_GBS,
                                        "traffic": 409.8e6, "traffic_unit": "bytes per launch",
                                        "traffic_note": "PMC: FETCH_SIZE 103 539 KB x2 (gfx950 correction) + WRITE_SIZE 202 715 KB, separate --pmc passes, "
                                                        "profiles/r01_gather_humanoid_b65536_pmc.csv = 1.01x the algorithmic bytes"}
            eh.close()
            # the op sizes are tiny: torch's default of one thread per host core (128 here) is slower than a few threads,
            # so time 1 and 8 threads on a bounded sample each and report the faster one
            best = None
            for nt in (1, 8):
                v, n = oracle_rate(w, "cpu", args.cpu_seconds / 2, threads=nt)
                if best is None or v > best[0]:
                    best = (v, n, nt)
            out["cpu_baseline"] = {"value": best[0], "unit": "gradient-steps/s", "cores": best[2], "kind": "port",
                                   "sample": f"{best[1]} iterations of the same workload through oracle/sac_td3_ref.py "
                                             f"(plain PyTorch CPU eager, torch.set_num_threads({best[2]}); faster of 1 and 8 threads)"}
            out["parity"] = parity_deltas(w, local)
            v, n = oracle_rate(w, "cuda", 6.0)
            out["eager_rocm_baseline"] = {"value": v, "unit": "gradient-steps/s",
                                          "sample": f"{n} iterations, same restatement on cuda:0, eager PyTorch-ROCm, no graphs"}
            out["speedup_vs_eager_rocm"] = out["value"] / v
            try:
                vg, ng = oracle_graph_rate(w, 4.0)
                out["torch_cudagraph_rocm_baseline"] = {"value": vg, "unit": "gradient-steps/s",
                                                        "sample": f"{ng} iterations, same restatement, update_qnets / update_actor as "
                                                                  "torch.cuda.CUDAGraph replays (the reference's cudagraphs: true mode), sampling and target update eager"}
                out["speedup_vs_torch_cudagraph_rocm"] = out["value"] / vg
            except Exception as ex:   # capture support for an op can differ between torch / ROCm versions: report, do not fail the run
                out["torch_cudagraph_rocm_baseline"] = {"value": None, "error": repr(ex)[:300]}
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
