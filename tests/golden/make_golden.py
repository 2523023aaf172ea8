#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

1. `nets_ref_*.npz`  -- outputs of the REFERENCE's own network classes (agents/nets.py, imported from
   /root/reference with an identity stand-in for the absent `beartype` decorator package, which the reference
   itself disables with `python -O`) on fixed seeds / inputs.  tests/test_golden.py rebuilds the oracle's nets
   under the same seeds and must reproduce these numbers bit for bit -> pins oracle/sac_td3_ref.py's network
   classes (init, forward, get_action, explore) to the reference.
2. `traj_*.npz`      -- a short injected-noise trajectory of the oracle agent (losses per iteration, parameter
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
    make_nets_fixture(reference_nets())
    make_traj_fixture()
