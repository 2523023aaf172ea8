#!/usr/bin/env python3
"""Headline benchmark: gradient-steps/s of the SAC hot path (BASELINE.json configs[1]: SAC Hopper-v4, batch 256,
hipGraph-captured update step on 1x MI355X), seeds sharded one per GPU for --gpus N (replicas only, no collective).

One "step" = one loop iteration of the reference's orchestrator.py:337-352 on synthetic transitions already
resident in HBM: replay index draw + gather, critic update, (every 3rd iteration) 2x actor+alpha update on the
same batch, Polyak.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hopper_sac|halfcheetah_td3|humanoid_sac]

`--gpus N` (N > 1) started WITHOUT a torch.distributed.run environment launches the N one-GPU ranks itself (the
parent touches no GPU: it only starts `python -m torch.distributed.run --nproc-per-node N ... bench.py ...` as a child,
the reference's counterpart being one OS process per seed, spawner.py:291,313-349); started by torch.distributed.run
(RANK / WORLD_SIZE set) it is one of those ranks.
"""
import argparse
import csv
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # SURVEY.md section 8d / BASELINE.md section 3
    "hopper_sac": dict(env="Hopper-v4", o=11, a=3, bound=1.0, td3=False, batch=256, rows=100_000, capacity=1_000_000),
    "halfcheetah_td3": dict(env="HalfCheetah-v4", o=17, a=6, bound=1.0, td3=True, batch=256, rows=100_000, capacity=1_000_000),
    "humanoid_sac": dict(env="Humanoid-v4", o=376, a=17, bound=0.4, td3=False, batch=1024, rows=1_000_000, capacity=1_000_000),
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32), the type the path computes in
DELAY = 2                 # actor_update_delay (sac.yml:44, td3.yml): actor updates every 3rd iteration


def describe(w):
    return (f"{'TD3' if w['td3'] else 'SAC'} {w['env']} batch={w['batch']}, 2x256 MLP + LayerNorm, {w['rows']} rows resident in a "
            f"{w['capacity']}-row HBM replay ring, per iteration 1 critic update + Polyak and every 3rd iteration 2 actor"
            f"{'' if w['td3'] else '+alpha'} updates; one hipGraph launch per 3-iteration schedule period")


def describe_short(w):
    return (f"{'TD3' if w['td3'] else 'SAC'} {w['env']} batch={w['batch']} 2x256 MLP+LN, {w['rows']} rows in a {w['capacity']}-row HBM ring, "
            f"1 critic update+Polyak per iteration, 2 actor{'' if w['td3'] else '+alpha'} updates every 3rd, hipGraph per 3 iterations")


LINE_MAX = 1800           # the driver parses the LAST stdout line out of an 8 KB tail: the result line stays well under it
DETAIL_FILE = "bench_detail.json"


def _sig(x, n=5):
    """floats to n significant digits (bytes on the result line), everything else untouched"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{n}g}")


def _pick(d, keys, n=5):
    return {k: _sig(d[k], n) for k in keys if d is not None and k in d}


def result_line(full, w):
    """The ONE line the driver parses (<= LINE_MAX bytes): the contract's keys, the roofline of the dominant kernel, the CPU
    baseline, and three numbers per secondary config.  Everything else (node tables, per-grid rooflines, sweeps, full
    configs) is in the side file `DETAIL_FILE` (written next to bench.py and, when it exists, into gpurun_out/)."""
    out = {k: _sig(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data") if k in full}
    out["config"] = {"workload": describe_short(w), "parallelism": full["config"]["parallelism"]}
    for k in ("dry_run", "note"):
        if k in full:
            out[k] = full[k]
    if "kernels_per_iteration" in full:
        out["kernel_nodes_per_iteration"] = _sig(full["kernels_per_iteration"].get("average_per_iteration"), 4)
    if "roofline" in full:
        out["roofline"] = _pick(full["roofline"], ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algo_flops_per_launch",
                                                   "algo_bytes_per_launch", "avg_launch_us", "share_of_node_time"), 4)
    if "cpu_baseline" in full:
        out["cpu_baseline"] = _pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"), 4)
    for k in ("speedup_vs_eager_rocm", "speedup_vs_torch_cudagraph_rocm"):
        if k in full:
            out[k] = _sig(full[k], 4)
    if "parity" in full:
        out["parity_max_rel"] = _sig(full["parity"].get("max_rel"), 3)   # |engine - oracle| / max(|oracle|, 1), worst reported scalar
    if "value_3000_steps" in full:
        out["value_3000_steps"] = _pick(full["value_3000_steps"], ("median", "min", "max", "repeats"), 5)
    if "replay_gather_hbm" in full:
        out["replay_gather_hbm"] = _pick(full["replay_gather_hbm"], ("rows_per_launch", "achieved", "unit", "frac", "traffic"), 4)
    if "configs" in full:
        out["configs"] = {name: {"value": _sig(c["value"]), "ms_per_step": _sig(c["ms_per_step"]), "roofline_frac": _sig(c["roofline"]["frac"], 4),
                                 "roofline_kernel": c["roofline"]["kernel"], "speedup_vs_eager_rocm": _sig(c.get("speedup_vs_eager_rocm"), 4)}
                          for name, c in full["configs"].items()}
    if "detail" in full:
        out["detail"] = full["detail"]
    line = json.dumps(out, separators=(",", ":"))
    for drop in ("replay_gather_hbm", "speedup_vs_torch_cudagraph_rocm", "kernel_nodes_per_iteration", "detail"):   # never expected: the
        if len(line) <= LINE_MAX:                                                                                  # line is ~1.5 KB
            break
        out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= LINE_MAX, len(line)
    return line


def write_detail(full):
    """the full measurement record -> bench_detail.json beside bench.py (and a copy under gpurun_out/, which gpurun merges back)"""
    paths = [os.path.join(ROOT, DETAIL_FILE)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", DETAIL_FILE))
    done = []
    for path in paths:
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            done.append(os.path.relpath(path, ROOT))
        except OSError:
            pass
    return done


def gather_algo_bytes(o, a, batch):
    """2*B*T + 4*B, T = 4*(2o+a+1)+1 (s, a, r, s', one done byte; read once + write once; int32 index)."""
    return 2 * batch * (4 * (2 * o + a + 1) + 1) + 4 * batch


def make_engine(w, seed, device_id, rows=None, capacity=None):
    import torch
    import sac_td3_cudagraphs_pytorch_amd as pkg
    from sac_td3_cudagraphs_pytorch_amd import schema
    cfg = pkg.Config(ob_dim=w["o"], ac_dim=w["a"], batch_size=w["batch"], rb_capacity=capacity or w["capacity"], max_envs=4,
                     prefer_td3_over_sac=w["td3"], bcq_style_targ_mix=w["td3"], qnets_lr=3e-4 if w["td3"] else 1e-3,
                     seed=seed, device_id=device_id, use_graphs=os.environ.get("SACTD3_BENCH_GRAPHS", "1") != "0")   # (0: same launches, no hipGraph)
    eng = pkg.Engine(cfg, [-w["bound"]] * w["a"], [w["bound"]] * w["a"])
    torch.manual_seed(seed)  # reference init (agents/nets.py:34-49) under the run's seed
    actor, critics = schema.reference_initial_params(w["o"], w["a"], w["td3"], True)
    for which, flat in ((0, actor), (2, actor), (1, critics), (3, critics)):
        eng.set_params(which, flat)
    if rows is None:
        eng.rb_fill_synthetic(w["rows"], seed=0)  # data seed 0
    eng.instantiate_graphs()   # capture now: a short --warmup must not leave a graph instantiation inside the timed steps
    eng.sync()
    return eng


def run_steps(eng, it, n):
    """n loop iterations from iteration number `it` on: whole periods of the actor schedule (1 iteration with the actor updates
    + DELAY critic-only ones) go out as ONE graph launch (sactd3_step_period), iterations outside a whole period one by one."""
    if hasattr(eng, "run_iterations") and os.environ.get("SACTD3_BENCH_PERIOD", "1") != "0":
        return eng.run_iterations(it, n)
    for _ in range(n):
        eng.step(it % (DELAY + 1) == 0)
        it += 1
    return it


def timed_windows(eng, it, steps=3000, repeats=5):
    """SURVEY.md 8(d)'s protocol: `repeats` windows of `steps` iterations each (the engine already warm), every window bracketed
    by a stream sync; returns (iteration number after, {"median", "min", "max", "repeats", "steps"} in gradient-steps/s)."""
    vals = []
    for _ in range(repeats):
        eng.sync()
        t0 = time.perf_counter()
        it = run_steps(eng, it, steps)
        eng.sync()
        vals.append(steps / (time.perf_counter() - t0))
    vals.sort()
    return it, {"median": vals[len(vals) // 2], "min": vals[0], "max": vals[-1], "repeats": repeats, "steps": steps}


# ------------------------------------------------------------------------------------------------ baselines (the oracle, timed)
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
    """SURVEY 8(d) 'parity deltas' of the TIMED path: `iters` fused iterations (sactd3_step, native Philox sampling and
    noise, graph replay) at the workload's own shapes; after each one the indices and normals the engine used are read
    back and the oracle's iteration is driven with exactly those (the reference's draw order, agents/agent.py:205,254,298).
    Max abs difference of the reported scalars and of the parameters afterwards.  The oracle is the checker here, never
    the thing measured."""
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
    n = 16384
    eng = pkg.Engine(pkg.Config.from_hps(hps, o, a, rb_capacity=n, seed=0, device_id=device_id), lo, hi)
    nh = a if td3 else 2 * a
    flat_a = lambda m: schema.dict_to_flat({k: v for k, v in m.state_dict().items() if k.startswith(("fc_stack", "head"))}, o, nh, True)
    flat_c = lambda ms: np.concatenate([schema.dict_to_flat(q.state_dict(), o + a, 1, True) for q in ms])
    eng.set_params(_lib.ACTOR, flat_a(ref.actor)); eng.set_params(_lib.ACTOR_TARGET, flat_a(ref.actor_target))
    eng.set_params(_lib.CRITICS, flat_c(ref.qnets)); eng.set_params(_lib.CRITICS_TARGET, flat_c(ref.qnets_target))
    g = torch.Generator().manual_seed(1)
    rows = [torch.randn(n, o, generator=g).numpy(), ((torch.rand(n, a, generator=g) * 2 - 1) * w["bound"]).numpy(),
            torch.randn(n, generator=g).numpy(), torch.randn(n, o, generator=g).numpy(), (torch.rand(n, generator=g) < 0.01).numpy()]
    for lo_ in range(0, n, 4096):
        eng.rb_extend(*[r[lo_:lo_ + 4096] for r in rows])
    worst, worst_rel = {}, {}
    for i in range(iters):
        do_actor = i % (DELAY + 1) == 0
        eng.step(do_actor)
        idx = eng.read_batch()["index"]
        noise = {"critic": torch.from_numpy(eng.read_noise(_lib.SITE_CRITIC))}
        if do_actor and not td3:
            noise["actor"] = [torch.from_numpy(eng.read_noise(s)) for s in (_lib.SITE_ACTOR0, _lib.SITE_ACTOR1)]
            noise["alpha"] = [torch.from_numpy(eng.read_noise(s)) for s in (_lib.SITE_ALPHA0, _lib.SITE_ALPHA1)]
        want = ref.iteration(ref.to_batch(*[r[idx] for r in rows]), i, noise)
        got = eng.read_metrics()
        for k, v in want.items():
            worst[k] = max(worst.get(k, 0.0), abs(got[k] - float(v)))
            worst_rel[k] = max(worst_rel.get(k, 0.0), abs(got[k] - float(v)) / max(abs(float(v)), 1.0))
    worst["params/critics"] = float(np.abs(eng.get_params(_lib.CRITICS) - flat_c(ref.qnets)).max())
    worst["params/actor"] = float(np.abs(eng.get_params(_lib.ACTOR) - flat_a(ref.actor)).max())
    eng.close()
    return {"path": "sactd3_step (fused iteration, graph replay, native RNG; oracle driven with the read-back indices and noise)",
            "iterations": iters, "max_abs_delta": worst, "max_rel_delta": worst_rel, "max_rel": max(worst_rel.values()),
            "rel_definition": "|engine - oracle| / max(|oracle|, 1) per reported scalar, max over the iterations (north_star: within 1e-5)",
            "note": "fp32 vs the plain-PyTorch oracle; parameters after the Adam steps can differ by up to 2 lr per step where a "
                    "gradient is near zero (sign-like first steps), see tests/helpers.py"}


# ------------------------------------------------------------------------------------------------ per-node budget and rooflines
def _norm_kernel(name):
    """rocprofv3's 'void k_nt<1, true, 2, 1, 2>(NtArgs)' -> 'k_nt<1,true,2,1,2>' (= the node registry's instance names)."""
    name = re.sub(r"^void\s+", "", name.strip())
    name = re.sub(r"\([^()]*\)\s*$", "", name)
    return name.replace(" ", "")


def pmc_traffic(workload):
    """{(kernel instance, threads): HBM-side bytes per launch} from the committed rocprofv3 --pmc summaries
    profiles/r*_pmc_<workload>.csv (written by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes; the
    gfx950 x2 correction of FETCH_SIZE is applied there, per MI355X_MICROARCH.md section HBM).  Newest round wins."""
    out, src = {}, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{workload}.csv"))):
        src = os.path.relpath(path, ROOT)
        with open(path) as f:
            for r in csv.DictReader(f):
                out[(r["kernel"], int(r["threads"]))] = dict(traffic=float(r["traffic_bytes"]), fetch=float(r["fetch_bytes"]),
                                                             write=float(r["write_bytes"]), calls=int(r["calls"]), source=src)
    return out


def node_budget(w, workload, device_id, ms_per_step, iters=200):
    """DESIGN.md section 5's budget as numbers: every kernel node of the period graph (and of the single-iteration graphs) timed alone
    (HIP events on the engine's stream, un-profiled, back-to-back launches of that node on a scratch engine), the sum over
    the reference's schedule (2 critic-only iterations + 1 with the actor updates per period) against the measured
    ms_per_step, and the roofline of every kernel instance from its algorithmic FLOPs / bytes."""
    eng = make_engine(w, seed=12345, device_id=device_id)
    run_steps(eng, 0, 6)
    eng.sync()
    g0, g1, gp = eng.time_nodes(0, iters), eng.time_nodes(1, iters), eng.time_nodes(2, iters)
    eng.close()
    s0, s1, sp = sum(n["us"] for n in g0), sum(n["us"] for n in g1), sum(n["us"] for n in gp)
    # the timed loop replays the PERIOD graph (one iteration with the actor updates + DELAY critic-only ones; for SAC the critic-only
    # iterations' sampling and next-action passes run ahead inside the first one's launches): its nodes are the budget
    per_iter = sp / (DELAY + 1)
    by_name, by_grid = roofline_groups([(1.0 / (DELAY + 1), gp)], pmc_traffic(workload), per_iter)
    return {"period_of_3_iterations": [[n["name"], round(n["us"], 3)] for n in gp], "sum_period_us": sp,
            "critic_only": [[n["name"], round(n["us"], 3)] for n in g0], "critic_plus_2_actor": [[n["name"], round(n["us"], 3)] for n in g1],
            "sum_critic_only_us": s0, "sum_critic_plus_2_actor_us": s1, "sum_per_iteration_us": per_iter,
            "measured_us_per_iteration": 1e3 * ms_per_step, "unaccounted_us_per_iteration": 1e3 * ms_per_step - per_iter,
            "note": "each node alone, back to back (includes its ~1.5 us launch boundary); unaccounted = what the dependent chain of "
                    "DIFFERENT kernels and the gap between graph replays add or save"}, by_name, by_grid


def roofline_groups(parts, traffic, per_iter):
    """parts = [(launches per iteration of each node, node table)] -> rooflines per kernel INSTANCE NAME (all grids of one instance
    summed: how rocprofv3 --stats groups them; frac is FLOP- / byte-weighted = sum of work / sum of time) and, for the side file,
    per (instance, grid).  Sorted by share of the iteration's node time: [0] is the dominant kernel."""
    def build(keyf):
        groups = {}
        for weight, nodes in parts:
            for n in nodes:
                d = groups.setdefault(keyf(n), dict(us=0.0, launches=0.0, flops=0.0, bytes=0.0, roles=[], grids=set()))
                d["us"] += weight * n["us"]; d["launches"] += weight
                d["flops"] += weight * n["flops"]; d["bytes"] += weight * n["bytes"]
                d["grids"].add(n["threads"])
                role = n["name"].split(":", 1)[1]
                if role not in d["roles"]:
                    d["roles"].append(role)
        out = []
        for key, d in sorted(groups.items(), key=lambda kv: -kv[1]["us"]):
            kern = key[0] if isinstance(key, tuple) else key
            us, L = d["us"] / d["launches"], d["launches"]
            t_mfma, t_hbm = d["flops"] / (MFMA_F32_PEAK_TF * 1e12), d["bytes"] / (HBM_PEAK_GBS * 1e9)
            if t_mfma >= t_hbm:
                ach, peak, unit, bound = d["flops"] / d["us"] * 1e-6, MFMA_F32_PEAK_TF, "TFLOP/s", "mfma"
            else:
                ach, peak, unit, bound = d["bytes"] / d["us"] * 1e-3, HBM_PEAK_GBS, "GB/s", "hbm"
            trs = [traffic[(kern, g)] for g in sorted(d["grids"]) if (kern, g) in traffic]
            calls = sum(t["calls"] for t in trs)
            tr = sum(t["traffic"] * t["calls"] for t in trs) / calls if calls else None
            out.append({"kernel": kern, "grids": sorted(d["grids"]), "roles": d["roles"], "bound": bound, "achieved": ach, "peak": peak,
                        "unit": unit, "frac": ach / peak, "traffic": tr, "traffic_source": trs[0]["source"] if trs else None,
                        "algo_flops_per_launch": d["flops"] / L, "algo_bytes_per_launch": d["bytes"] / L, "avg_launch_us": us,
                        "launches_per_iteration": L, "us_per_iteration": d["us"], "share_of_node_time": d["us"] / per_iter})
        return out
    return build(lambda n: n["name"].split(":")[0]), build(lambda n: (n["name"].split(":")[0], n["threads"]))


def kernel_counts(eng):
    """kernel nodes: of the 3-iteration period graph the timed loop replays, and per kind of iteration (their sum = the period)"""
    per = eng.graph_kernel_count(4)
    if not eng.graph_kernel_count(2):     # (after the timed region) instantiate the single-iteration graphs to count their nodes
        eng.step(False)
    if not eng.graph_kernel_count(3):
        eng.step(True)
    c0, c1 = eng.graph_kernel_count(2), eng.graph_kernel_count(3)
    return {"period_of_3_iterations": per, "critic_only": c0, "critic_plus_2_actor": c1, "average_per_iteration": per / (DELAY + 1) if per else None}


def gather_cold(eng, w, iters=300):
    """the replay gather at the workload's own batch size with a fresh index draw per launch (rows from HBM / MALL, not L2)."""
    us = eng.time_kernel("gather", iters)
    ab = gather_algo_bytes(w["o"], w["a"], w["batch"])
    return {"kernel": "k_gather", "batch": w["batch"], "us_incl_counter_tick_kernel": us, "algo_bytes": ab, "GB/s": ab / us * 1e-3,
            "frac_of_8TBs": ab / us * 1e-3 / HBM_PEAK_GBS}


def baselines(w, value, cpu_seconds, eager_seconds, graph_seconds):
    out = {}
    best = None
    for nt in (1, 8):   # torch's default of one thread per host core is slower than a few threads at these op sizes
        v, n = oracle_rate(w, "cpu", cpu_seconds / 2, threads=nt)
        if best is None or v > best[0]:
            best = (v, n, nt)
    out["cpu_baseline"] = {"value": best[0], "unit": "gradient-steps/s", "cores": best[2], "kind": "port",
                           "sample": f"{best[1]} iterations of this workload, oracle/sac_td3_ref.py, PyTorch CPU eager, "
                                     f"{best[2]} threads (faster of 1 and 8)"}
    v, n = oracle_rate(w, "cuda", eager_seconds)
    out["eager_rocm_baseline"] = {"value": v, "unit": "gradient-steps/s",
                                  "sample": f"{n} iterations, same restatement on cuda:0, eager PyTorch-ROCm, no graphs"}
    out["speedup_vs_eager_rocm"] = value / v
    try:
        vg, ng = oracle_graph_rate(w, graph_seconds)
        out["torch_cudagraph_rocm_baseline"] = {"value": vg, "unit": "gradient-steps/s",
                                                "sample": f"{ng} iterations, same restatement, update_qnets / update_actor as "
                                                          "torch.cuda.CUDAGraph replays (the reference's cudagraphs: true mode), sampling and target update eager"}
        out["speedup_vs_torch_cudagraph_rocm"] = value / vg
    except Exception as ex:   # capture support for an op can differ between torch / ROCm versions: report, do not fail the run
        out["torch_cudagraph_rocm_baseline"] = {"value": None, "error": repr(ex)[:300]}
    return out


def secondary_config(name, device_id, steps=3000, warmup=300):
    """BASELINE.json configs 3 / 4, measured like the headline: own engine, median of 5 windows of 3000 iterations, node
    budget, roofline, parity, baselines (the full objects go to the side file, three numbers each to the result line)."""
    w = WORKLOADS[name]
    eng = make_engine(w, seed=0, device_id=device_id)
    it = run_steps(eng, 0, warmup)
    it, win = timed_windows(eng, it, steps)
    out = {"workload": describe(w), "value": win["median"], "value_windows": win, "unit": "gradient-steps/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 / win["median"], "kernels_per_iteration": kernel_counts(eng),
           "final_metrics": eng.read_metrics(), "replay_gather_cold": gather_cold(eng, w)}
    eng.close()
    out["node_us"], rl, out["rooflines_per_grid"] = node_budget(w, name, device_id, out["ms_per_step"])
    out["roofline"], out["rooflines_top"] = rl[0], rl[1:5]
    out["parity"] = parity_deltas(w, device_id)
    out.update(baselines(w, out["value"], cpu_seconds=6.0, eager_seconds=3.0, graph_seconds=3.0))
    return out


# ------------------------------------------------------------------------------------------------ ranks
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Parent of an N-GPU run.  NOTHING here touches the GPU (no torch import, no HIP call): the ranks are fresh child
    processes started through torch.distributed.run, one per GPU, and this process only waits and passes on the exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


class _DryEngine:
    """--dry-run-ranks: no GPU work at all.  Checks launching, rendezvous, barrier, max-over-ranks and the JSON line on a
    box without a GPU (tests/test_launcher_gloo.py); the line it prints is marked dry_run and carries no measurement."""

    def step(self, do_actor):
        time.sleep(1e-4)

    def sync(self):
        pass

    def read_metrics(self):
        return {}

    def graph_kernel_count(self, which):
        return 0

    def run_iterations(self, i0, n):
        for _ in range(n):
            self.step(False)
        return i0 + n

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="hopper_sac", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-baselines", action="store_true", help="skip the baselines, the other configs and the extra figures")
    ap.add_argument("--timed-only", action="store_true", help="only the warm-up and the timed steps (rocprofv3 runs)")
    ap.add_argument("--dry-run-ranks", action="store_true", help="no GPU work: rehearse rank launching / rendezvous / aggregation on CPU")
    ap.add_argument("--eager-profile", type=int, default=0, metavar="ITERS",
                    help="run ONLY the eager PyTorch-ROCm restatement for ITERS iterations (for rocprofv3 launch counts)")
    ap.add_argument("--gather-profile", type=int, default=0, metavar="BATCH",
                    help="run ONLY the replay gather at this batch size 20 times (for rocprofv3 --pmc traffic counters)")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
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
    backend = os.environ.get("SACTD3_BENCH_BACKEND", "gloo" if args.dry_run_ranks else "nccl")
    if "SACTD3_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SACTD3_BENCH_DEVICE"])
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    gpu_sync = (lambda: None) if args.dry_run_ranks else torch.cuda.synchronize

    # seed = GPU index, one independent learner per GPU
    eng = _DryEngine() if args.dry_run_ranks else make_engine(w, seed=rank, device_id=local)
    # iteration numbers: the warm-up is numbered so that the timed window starts at a multiple of DELAY + 1 -- an iteration with the
    # actor updates, like iteration 0 of the reference loop (orchestrator.py:345-349) -- whatever --warmup is
    it = run_steps(eng, (-args.warmup) % (DELAY + 1), args.warmup)
    eng.sync()
    gpu_sync()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    it = run_steps(eng, it, args.steps)
    eng.sync()
    gpu_sync()
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
            "config": {"workload": describe(w), "parallelism": f"{world} independent seeds, one per GPU, no collective"},
            # one iteration = 1 critic update (+ Polyak) and, every 3rd iteration, 2 actor(+alpha) updates
            "critic_updates_per_s": world * args.steps / dt, "actor_updates_per_s": world * args.steps / dt * 2.0 / 3.0,
            "kernels_per_iteration": kernel_counts(eng),
            "final_metrics": metrics,
        }
        if args.dry_run_ranks:
            out.update(dry_run=True, value=None, note="no GPU work was done: rank launching / rendezvous / aggregation rehearsal only")
        elif not args.timed_only and world == 1:     # N > 1 lines carry the timed region only (roofline / cpu_baseline: N = 1)
            extras(out, eng, w, args, local, world)
            out["detail"] = DETAIL_FILE
            print("bench.py: full record (node tables, per-grid rooflines, sweeps, secondary configs) ->", ", ".join(write_detail(out)),
                  file=sys.stderr, flush=True)
        print(result_line(out, w), flush=True)
    if dist:
        dist.destroy_process_group()


def extras(out, eng, w, args, local, world):
    """Everything in the line beyond the timed steps (rank 0): rooflines, node budget, acting-side figures, the other
    BASELINE configs, baselines.  None of it is inside the timed region above."""
    import numpy as np
    # a longer window of the same loop (the driver's 20-step window is ~1.7 ms): agreement check for `value`
    it = run_steps(eng, 0, 300)
    it, out["value_3000_steps"] = timed_windows(eng, it, 3000, 5)
    out["replay_gather_cold"] = gather_cold(eng, w)
    # acting side (agents/agent.py:172-181): one predict round trip = H2D of the observations, 2 kernels, D2H + sync
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
    if args.no_baselines:
        eng.close()
        return
    # Seed sweeps are the reference's unit of work (spawner.py: one job per seed) and one learner leaves most of an
    # MI355X idle: S independent engines (own stream, own graphs, seeds 0..S-1) in this process, one host thread each,
    # aggregate gradient-steps/s.  An extra figure -- `value` above stays the one-learner-per-GPU number.
    import threading
    multi = {}
    for S in (2, 4, 8):
        engs = [eng] + [make_engine(w, seed=100 + k, device_id=local) for k in range(1, S)]
        for i in range(150):
            for e2 in engs:
                e2.step(i % 3 == 0)
        for e2 in engs:
            e2.sync()

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
    # large-batch asymptote of the gather (B=256 is launch-bound by construction)
    sweep = {}
    for bs in (256, 4096, 65536):
        us_b, by = eng.time_gather_sweep(bs, 200 if bs < 65536 else 50)
        sweep[str(bs)] = {"us": us_b, "GB/s": by / us_b * 1e-3}
    out["gather_batch_sweep"] = sweep
    eng.close()
    # roofline of the DOMINANT kernel of the timed graph (largest share of the iteration's node time), and the others
    out["node_us"], rl, out["rooflines_per_grid"] = node_budget(w, args.workload, local, out["ms_per_step"])
    out["roofline"], out["rooflines_top"] = rl[0], rl[1:5]
    # north_star's "replay-gather HBM GB/s as fraction of 8 TB/s": the same kernel on the Humanoid-v4 record
    # (3136 B/row) out of a full 1M-row (3.1 GB, far beyond the 256 MB Infinity Cache) ring, 65 536 rows per launch
    wh = WORKLOADS["humanoid_sac"]
    eh = make_engine(wh, 0, local)
    us_h, by_h = eh.time_gather_sweep(65536, 50)
    trs = [v for (k, _), v in sorted(pmc_traffic("gather_humanoid_b65536").items(), key=lambda kv: kv[0][1]) if k == "k_gather"]
    tr = trs[-1] if trs else None     # the 65 536-row launches (largest grid) of profiles/r*_pmc_gather_humanoid_b65536.csv
    out["replay_gather_hbm"] = {"record": "Humanoid-v4 (o=376, a=17), 1M-row ring", "rows_per_launch": 65536, "us": us_h,
                                "algo_bytes": by_h, "achieved": by_h / us_h * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": by_h / us_h * 1e-3 / HBM_PEAK_GBS, "bound": "hbm",
                                "traffic": tr["traffic"] if tr else None, "traffic_unit": "bytes per launch",
                                "traffic_source": tr["source"] if tr else None}
    eh.close()
    out["parity"] = parity_deltas(w, local)
    out.update(baselines(w, out["value"], cpu_seconds=args.cpu_seconds, eager_seconds=6.0, graph_seconds=4.0))
    # BASELINE.json configs 3 and 4, each measured the same way (own 3000 timed steps)
    out["configs"] = {name: secondary_config(name, local) for name in ("halfcheetah_td3", "humanoid_sac") if name != args.workload}


if __name__ == "__main__":
    main()
