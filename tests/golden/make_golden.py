#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

1. `nets_ref_*.npz`  -- outputs of the REFERENCE's own network classes (agents/nets.py, imported from
   /root/reference with an identity stand-in for the absent `beartype` decorator package, which the reference
   itself disables with `python -O`) on fixed seeds / inputs.  tests/test_golden.py rebuilds the oracle's nets
   under the same seeds and must reproduce these numbers bit for bit -> pins oracle/sac_td3_ref.py's network
   classes (init, forward, get_action, explore) to the reference.
2. `nets_bwd_*.npz`  -- BACKWARD of the reference's own network classes at the BASELINE batch sizes (Hopper / HalfCheetah B = 256,
   Humanoid B = 1024): autograd gradients, w.r.t. every parameter (and the action input), of (a) the plain functionals
   `get_action(ob)["log_prob"].sum() + ["sample"].sum()`, `Critic(ob, ac).sum()`, `Actor(ob).sum()` and (b) the two losses the
   update path differentiates, written with the reference's modules and the formulas of agents/agent.py:216-233 (twin MSE against
   a given target) and :272-281 (`(alpha * logp - min Q).mean()`, TD3: `(-Q1).mean()`).  tests/test_golden.py requires the
   oracle's classes to reproduce every stored number bit for bit on the CPU; tests/test_gpu_engine.py drives the HIP kernels
   with the same parameters / inputs / noise and compares per key.  Large matrices are stored as every 8th row + float64 row and
   column sums (inputs are regenerated from their seed; a float64 digest guards the regeneration).
3. `traj_*.npz`      -- a short injected-noise trajectory of the oracle agent (losses per iteration, parameter
   digests).  agents/agent.py cannot be imported here (tensordict / torchrl / omegaconf / wandb are absent) and the
   reference ships no golden vectors, so these pin the ORACLE against regressions and give the GPU tests a
   committed target; they are not reference outputs (parity unpinned at agent level, see oracle/sac_td3_ref.py).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {"hopper": (11, 3, 1.0), "halfcheetah": (17, 6, 1.0), "humanoid": (376, 17, 0.4)}


def reference_nets():
    bt = types.ModuleType("beartype")
    bt.beartype = lambda f: f
    sys.modules.setdefault("beartype", bt)
    sys.path.insert(0, "/root/reference")
    from agents import nets  # noqa: E402  (the reference's own file, imported in place, never copied)
    return nets


def make_nets_fixture(R):
    for env, (o, a, bound) in CASES.items():
        out = {}
        mn, mx = torch.full((a,), -bound), torch.full((a,), bound)
        for ln in (True, False):
            tag = "ln" if ln else "noln"
            g = torch.Generator().manual_seed(123)
            ob, ac = torch.randn(8, o, generator=g), (torch.rand(8, a, generator=g) * 2 - 1) * bound
            out["ob"], out["ac"] = ob.numpy(), ac.numpy()
            torch.manual_seed(11)
            net = R.TanhGaussActor((o,), (a,), (256, 256), mn, mx, layer_norm=ln, device="cpu")
            torch.manual_seed(12)
            act = net.get_action(ob)
            for k, v in act.items():
                out[f"sac_{tag}_{k}"] = v.detach().numpy()
            out[f"sac_{tag}_w1_digest"] = np.array([net.fc_stack.fc_block_1.fc.weight.double().sum().item(),
                                                    net.head.weight.double().abs().sum().item()])
            torch.manual_seed(13)
            q = R.Critic((o,), (a,), (256, 256), layer_norm=ln, device="cpu")
            out[f"q_{tag}"] = q(ob, ac).detach().numpy()
            torch.manual_seed(14)
            pi = R.Actor((o,), (a,), (256, 256), mn, mx, exploration_noise=0.1, layer_norm=ln, device="cpu")
            out[f"td3_{tag}_action"] = pi(ob).detach().numpy()
            torch.manual_seed(15)
            out[f"td3_{tag}_explore"] = pi.explore(ob)["action"].detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"nets_ref_{env}.npz"), **out)
        print("wrote nets_ref_%s.npz" % env)


BWD_B = {"hopper": 256, "halfcheetah": 256, "humanoid": 1024}
BWD_ALPHA = 0.2


def bwd_inputs(env):
    """the inputs of the backward fixtures, a function of the seed only (tests regenerate them; `digest` guards that)"""
    o, a, bound = CASES[env]
    B = BWD_B[env]
    g = torch.Generator().manual_seed(321)
    ob, ac = torch.randn(B, o, generator=g), (torch.rand(B, a, generator=g) * 2 - 1) * bound
    y = torch.randn(B, generator=g)
    torch.manual_seed(12)
    eps = torch.empty(B, a).normal_()          # what Normal.rsample() draws under torch.manual_seed(12) (checked below)
    digest = np.array([t.double().sum().item() for t in (ob, ac, y, eps)] + [t.double().abs().sum().item() for t in (ob, ac, y, eps)])
    return ob, ac, y, eps, digest


def pack_grad(out, name, t):
    """full tensor, except the gradients of the big weight matrices: every 8th row + float64 row / column sums"""
    t = t.detach()
    if t.ndim == 2 and t.numel() > 8192 and name.endswith(".weight"):
        out[name + "/rows8"] = t[::8].numpy()
        out[name + "/rowsum"] = t.double().sum(1).numpy()
        out[name + "/colsum"] = t.double().sum(0).numpy()
    else:
        out[name] = t.numpy()


def bwd_functionals(make_sac, make_q, make_td3, env, out=None):
    """Everything the backward fixtures hold, computed with the given network constructors (the reference's classes when the
    fixture is generated, the oracle's when tests/test_golden.py checks it): returns {name: tensor} before packing."""
    o, a, bound = CASES[env]
    ob, ac, y, eps, digest = bwd_inputs(env)
    res = {"digest": torch.from_numpy(digest)}
    torch.manual_seed(11); actor = make_sac()
    torch.manual_seed(13); q1 = make_q()
    torch.manual_seed(14); pi = make_td3()
    torch.manual_seed(16); q2 = make_q()
    names = lambda m: [k for k, _ in m.named_parameters()]
    # (a) plain functionals
    torch.manual_seed(12)
    act = actor.get_action(ob)
    for k, gr in zip(names(actor), torch.autograd.grad(act["log_prob"].sum() + act["sample"].sum(), list(actor.parameters()))):
        res[f"sum/actor/{k}"] = gr
    res["sum/actor/sample"], res["sum/actor/log_prob"] = act["sample"].detach(), act["log_prob"].detach()
    ac_r = ac.clone().requires_grad_(True)
    gq = torch.autograd.grad(q1(ob, ac_r).sum(), list(q1.parameters()) + [ac_r])
    for k, gr in zip(names(q1) + ["d_action"], gq):
        res[f"sum/critic/{k}"] = gr
    for k, gr in zip(names(pi), torch.autograd.grad(pi(ob).sum(), list(pi.parameters()))):
        res[f"sum/td3/{k}"] = gr
    # (b) the critic loss of agents/agent.py:216-233 against a given target y: sum over the twin of mse_loss(q.view(-1), y)
    qs = [q(ob, ac) for q in (q1, q2)]
    loss_q = sum(torch.nn.functional.mse_loss(qv.view(-1), y) for qv in qs)
    gq = torch.autograd.grad(loss_q, list(q1.parameters()) + list(q2.parameters()))
    for i, q in enumerate((q1, q2)):
        for k, gr in zip(names(q), gq[i * len(names(q)):(i + 1) * len(names(q))]):
            res[f"qloss/critic{i}/{k}"] = gr
    res["qloss/q"], res["qloss/loss"] = torch.stack([qv.detach().view(-1) for qv in qs]), loss_q.detach()
    # (b) the SAC actor loss of agents/agent.py:272-281: (alpha * logp - min_i Q_i(s, a_pi)).mean(), critics as constants
    torch.manual_seed(12)
    a_pi, logp, _ = actor.get_action(ob).values()
    a_pi.retain_grad()
    for q in (q1, q2):
        for p_ in q.parameters():
            p_.requires_grad_(False)
    q_pi = torch.stack([q1(ob, a_pi), q2(ob, a_pi)])
    loss_a = (BWD_ALPHA * logp - q_pi.min(0).values).mean()
    ga = torch.autograd.grad(loss_a, list(actor.parameters()) + [a_pi])
    for k, gr in zip(names(actor) + ["d_action"], ga):
        res[f"aloss/sac/{k}"] = gr
    res["aloss/sac/a_pi"], res["aloss/sac/logp"] = a_pi.detach(), logp.detach().view(-1)
    res["aloss/sac/q_pi"], res["aloss/sac/loss"] = q_pi.detach().squeeze(-1), loss_a.detach()
    # ... and TD3's: (-Q1(s, pi(s))).mean()
    a_t = pi(ob)
    a_t.retain_grad()
    loss_t = (-q1(ob, a_t)).mean()
    gt = torch.autograd.grad(loss_t, list(pi.parameters()) + [a_t])
    for k, gr in zip(names(pi) + ["d_action"], gt):
        res[f"aloss/td3/{k}"] = gr
    res["aloss/td3/a_pi"], res["aloss/td3/loss"] = a_t.detach(), loss_t.detach()
    # fp32 noise floor of the two composed losses' gradients: the same modules cast to float64, same inputs and noise; `noise/<key>` =
    # max |fp32 gradient - fp64 gradient|.  Where squashed actions saturate (1 - y^2 -> 0: 2.6 % of Humanoid's at these initial
    # parameters) the log-prob terms cancel badly and torch's own fp32 result sits 1e-4 .. 7e-4 from the float64 one; a second fp32
    # implementation is compared with a tolerance scaled by this measured floor instead of a guessed constant.
    import copy
    a64, p64, c64 = copy.deepcopy(actor).double(), copy.deepcopy(pi).double(), [copy.deepcopy(q).double() for q in (q1, q2)]
    for q in c64:
        for p_ in q.parameters():
            p_.requires_grad_(True)
    ob64, ac64, y64, eps64 = ob.double(), ac.double(), y.double(), eps.double()
    lq = sum(torch.nn.functional.mse_loss(q(ob64, ac64).view(-1), y64) for q in c64)
    g64 = torch.autograd.grad(lq, list(c64[0].parameters()) + list(c64[1].parameters()))
    for i in range(2):
        for k, gr in zip(names(q1), g64[i * len(names(q1)):(i + 1) * len(names(q1))]):
            res[f"noise/qloss/critic{i}/{k}"] = (res[f"qloss/critic{i}/{k}"].double() - gr).abs().max().float()
    mean64, std64 = a64(ob64)
    x64 = mean64 + eps64 * std64
    y_t = torch.tanh(x64)
    ap64 = y_t * a64.action_scale + a64.action_bias
    lp64 = (-((x64 - mean64) ** 2) / (2 * std64 ** 2) - std64.log() - 0.9189385332046727
            - torch.log(a64.action_scale * (1 - y_t.pow(2)) + 1e-6)).sum(1, keepdim=True)
    for q in c64:
        for p_ in q.parameters():
            p_.requires_grad_(False)
    la = (BWD_ALPHA * lp64 - torch.stack([c64[0](ob64, ap64), c64[1](ob64, ap64)]).min(0).values).mean()
    for k, gr in zip(names(actor), torch.autograd.grad(la, list(a64.parameters()))):
        res[f"noise/aloss/sac/{k}"] = (res[f"aloss/sac/{k}"].double() - gr).abs().max().float()
    at64 = p64(ob64)
    lt = (-c64[0](ob64, at64)).mean()
    for k, gr in zip(names(pi), torch.autograd.grad(lt, list(p64.parameters()))):
        res[f"noise/aloss/td3/{k}"] = (res[f"aloss/td3/{k}"].double() - gr).abs().max().float()
    # the noise: Normal.rsample() under seed 12 is mean + eps * std with eps = torch.empty(B, a).normal_() under the same seed
    mean, std = actor(ob)
    resample = torch.tanh(mean + eps * std) * actor.action_scale + actor.action_bias
    assert torch.equal(resample.detach(), res["aloss/sac/a_pi"]), "rsample() noise is not the seed-12 normal_() draw"
    res["eps"], res["y"] = eps, y
    return res


def make_bwd_fixture(R):
    for env, (o, a, bound) in CASES.items():
        mn, mx = torch.full((a,), -bound), torch.full((a,), bound)
        res = bwd_functionals(lambda: R.TanhGaussActor((o,), (a,), (256, 256), mn, mx, layer_norm=True, device="cpu"),
                              lambda: R.Critic((o,), (a,), (256, 256), layer_norm=True, device="cpu"),
                              lambda: R.Actor((o,), (a,), (256, 256), mn, mx, exploration_noise=0.1, layer_norm=True, device="cpu"), env)
        out = {}
        for k, v in res.items():
            pack_grad(out, k, v)
        path = os.path.join(HERE, f"nets_bwd_{env}.npz")
        np.savez_compressed(path, **out)
        print("wrote nets_bwd_%s.npz (%d arrays, %.0f KB)" % (env, len(out), os.path.getsize(path) / 1024))


def run_traj(algo, env, B=32, iters=6, seed=5):
    from oracle.sac_td3_ref import Hps, RefAgent
    o, a, bound = CASES[env]
    hps = (Hps.td3 if algo == "td3" else Hps.sac)(batch_size=B)
    torch.manual_seed(seed)
    ag = RefAgent(o, a, [-bound] * a, [bound] * a, hps)
    g = torch.Generator().manual_seed(seed + 1)
    n = 500
    data = dict(obs=torch.randn(n, o, generator=g), act=(torch.rand(n, a, generator=g) * 2 - 1) * bound,
                rew=torch.randn(n, generator=g), nobs=torch.randn(n, o, generator=g), done=torch.rand(n, generator=g) < 0.05)
    out = {"seed": np.array(seed), "B": np.array(B), **{k: v.numpy() for k, v in data.items()}}
    losses = []
    for i in range(iters):
        idx = torch.randint(0, n, (B,), generator=g)
        noise = {"critic": torch.randn(B, a, generator=g), "actor": [torch.randn(B, a, generator=g) for _ in range(2)],
                 "alpha": [torch.randn(B, a, generator=g) for _ in range(2)]}
        out[f"idx_{i}"] = idx.numpy()
        out[f"eps_critic_{i}"] = noise["critic"].numpy()
        for j in range(2):
            out[f"eps_actor_{i}_{j}"], out[f"eps_alpha_{i}_{j}"] = noise["actor"][j].numpy(), noise["alpha"][j].numpy()
        b = ag.to_batch(data["obs"][idx], data["act"][idx], data["rew"][idx], data["nobs"][idx], data["done"][idx])
        r = ag.iteration(b, i, noise)
        losses.append([float(r.get(k, float("nan"))) for k in ("loss/qf_loss", "loss/actor_loss", "loss/alpha_loss", "vitals/alpha")])
    out["losses"] = np.array(losses, np.float64)
    dig = lambda mods: np.array([[p.double().sum().item(), p.double().abs().sum().item()] for m in mods for p in m.parameters()])
    out["digest_actor"], out["digest_critics"], out["digest_targets"] = dig([ag.actor]), dig(ag.qnets), dig(ag.qnets_target)
    return out


def make_traj_fixture():
    for algo, env in (("sac", "hopper"), ("td3", "halfcheetah")):
        np.savez_compressed(os.path.join(HERE, f"traj_{algo}_{env}.npz"), **run_traj(algo, env))
        print("wrote traj_%s_%s.npz" % (algo, env))


if __name__ == "__main__":
    R = reference_nets()
    make_nets_fixture(R)
    make_bwd_fixture(R)
    make_traj_fixture()
