"""Golden vectors (tests/golden/, produced by tests/golden/make_golden.py in the build container).

* nets_ref_*.npz hold outputs of the REFERENCE's own agents/nets.py classes: the oracle's network classes must
  reproduce them bit for bit from the same seeds (pins the oracle to the reference without the reference present).
* traj_*.npz hold an injected-noise trajectory of the oracle agent: the oracle must reproduce it (regression pin),
  and on a GPU the HIP engine must follow it within the parity tolerance.
"""
import os

import numpy as np
import pytest
import torch

from oracle.sac_td3_ref import DetPolicy, Hps, QNet, RefAgent, SquashedGaussPolicy
from tests import helpers

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"hopper": (11, 3, 1.0), "halfcheetah": (17, 6, 1.0), "humanoid": (376, 17, 0.4)}


@pytest.mark.parametrize("env", sorted(CASES))
def test_oracle_nets_reproduce_reference_outputs(env):
    o, a, bound = CASES[env]
    fx = np.load(os.path.join(HERE, "golden", f"nets_ref_{env}.npz"))
    ob, ac = torch.from_numpy(fx["ob"]), torch.from_numpy(fx["ac"])
    mn, mx = torch.full((a,), -bound), torch.full((a,), bound)
    for ln in (True, False):
        tag = "ln" if ln else "noln"
        torch.manual_seed(11)
        net = SquashedGaussPolicy(o, a, mn, mx, ln)
        torch.manual_seed(12)
        act = net.get_action(ob)
        for k, v in act.items():
            assert np.array_equal(v.detach().numpy(), fx[f"sac_{tag}_{k}"]), (env, tag, k)
        torch.manual_seed(13)
        assert np.array_equal(QNet(o, a, ln)(ob, ac).detach().numpy(), fx[f"q_{tag}"])
        torch.manual_seed(14)
        pi = DetPolicy(o, a, mn, mx, 0.1, ln)
        assert np.array_equal(pi(ob).detach().numpy(), fx[f"td3_{tag}_action"])
        torch.manual_seed(15)
        assert np.array_equal(pi.explore(ob).detach().numpy(), fx[f"td3_{tag}_explore"])


def _unpacked(fx, name, t):
    """(stored arrays, the same views of tensor t) for one fixture entry -- see make_golden.pack_grad"""
    if name in fx.files:
        return [(name, fx[name], t.numpy())]
    return [(name + "/rows8", fx[name + "/rows8"], t[::8].numpy()), (name + "/rowsum", fx[name + "/rowsum"], t.double().sum(1).numpy()),
            (name + "/colsum", fx[name + "/colsum"], t.double().sum(0).numpy())]


@pytest.mark.parametrize("env", sorted(CASES))
def test_oracle_nets_reproduce_reference_backward(env):
    """tests/golden/nets_bwd_*.npz hold autograd gradients through the REFERENCE's own TanhGaussActor / Critic / Actor
    (agents/nets.py:52-234) at the BASELINE batch sizes -- of log_prob.sum() + sample.sum(), Critic(ob, ac).sum(), Actor(ob).sum()
    w.r.t. every parameter and the action input, and of the critic / actor losses of agents/agent.py:216-233,272-281 built
    from those modules.  The oracle's classes, under the same seeds, must give the same numbers BIT FOR BIT: this pins the
    oracle's backward (not only its forward) to the reference for rows A3 / A5 / A6 of SURVEY.md section 8."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    o, a, bound = CASES[env]
    fx = np.load(os.path.join(HERE, "golden", f"nets_bwd_{env}.npz"))
    mn, mx = torch.full((a,), -bound), torch.full((a,), bound)
    res = mg.bwd_functionals(lambda: SquashedGaussPolicy(o, a, mn, mx, True), lambda: QNet(o, a, True),
                             lambda: DetPolicy(o, a, mn, mx, 0.1, True), env)
    seen = set()
    for name, t in res.items():
        for key, want, got in _unpacked(fx, name, t.detach()):
            assert np.array_equal(got, want), (env, key, float(np.abs(got.astype(np.float64) - want).max()))
            seen.add(key)
    assert seen == set(fx.files) and len(seen) >= 100


def _replay(fx, algo, env, drive):
    o, a, bound = CASES[env]
    B, seed = int(fx["B"]), int(fx["seed"])
    iters = fx["losses"].shape[0]
    for i in range(iters):
        noise = {"critic": fx[f"eps_critic_{i}"], "actor": [fx[f"eps_actor_{i}_{j}"] for j in range(2)],
                 "alpha": [fx[f"eps_alpha_{i}_{j}"] for j in range(2)]}
        yield i, fx[f"idx_{i}"], noise


@pytest.mark.parametrize("algo,env", [("sac", "hopper"), ("td3", "halfcheetah")])
def test_oracle_reproduces_trajectory(algo, env):
    o, a, bound = CASES[env]
    fx = np.load(os.path.join(HERE, "golden", f"traj_{algo}_{env}.npz"))
    hps = (Hps.td3 if algo == "td3" else Hps.sac)(batch_size=int(fx["B"]))
    torch.manual_seed(int(fx["seed"]))
    ag = RefAgent(o, a, [-bound] * a, [bound] * a, hps)
    for i, idx, noise in _replay(fx, algo, env, None):
        tn = {k: (torch.from_numpy(v) if not isinstance(v, list) else [torch.from_numpy(x) for x in v]) for k, v in noise.items()}
        b = ag.to_batch(fx["obs"][idx], fx["act"][idx], fx["rew"][idx], fx["nobs"][idx], fx["done"][idx])
        r = ag.iteration(b, i, tn)
        got = [float(r.get(k, float("nan"))) for k in ("loss/qf_loss", "loss/actor_loss", "loss/alpha_loss", "vitals/alpha")]
        np.testing.assert_allclose(got, fx["losses"][i], rtol=1e-4, atol=1e-5, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("algo,env", [("sac", "hopper"), ("td3", "halfcheetah")])
def test_engine_follows_golden_trajectory(algo, env):
    import sac_td3_cudagraphs_pytorch_amd as P
    from sac_td3_cudagraphs_pytorch_amd import _lib, schema
    o, a, bound = CASES[env]
    fx = np.load(os.path.join(HERE, "golden", f"traj_{algo}_{env}.npz"))
    B = int(fx["B"])
    hps = (Hps.td3 if algo == "td3" else Hps.sac)(batch_size=B)
    eng = P.Engine(P.Config.from_hps(hps, o, a, rb_capacity=1024, seed=0), [-bound] * a, [bound] * a)
    torch.manual_seed(int(fx["seed"]))   # same initial parameters as the fixture's agent (reference init order)
    actor, critics = schema.reference_initial_params(o, a, algo == "td3", True)
    for which, flat in ((_lib.ACTOR, actor), (_lib.ACTOR_TARGET, actor), (_lib.CRITICS, critics), (_lib.CRITICS_TARGET, critics)):
        eng.set_params(which, flat)
    eng.rb_extend(fx["obs"], fx["act"], fx["rew"], fx["nobs"], fx["done"])
    for i, idx, noise in _replay(fx, algo, env, None):
        eng.set_noise(_lib.SITE_CRITIC, noise["critic"])
        eng.rb_sample_with_indices(idx)
        eng.update_qnets()
        if i % 3 == 0:
            for j in range(2):
                eng.set_noise(_lib.SITE_ACTOR0, noise["actor"][j]); eng.set_noise(_lib.SITE_ALPHA0, noise["alpha"][j])
                eng.update_actor()
        eng.update_targ_nets(i + 1)
        m = eng.read_metrics()
        want = fx["losses"][i]
        keys = ["loss/qf_loss", "loss/actor_loss", "loss/alpha_loss", "vitals/alpha"]
        for k, w in zip(keys, want):
            if np.isfinite(w):   # (the fixture holds float64 printouts of float32 losses: 1e-6 of slack for that; then 1e-5 on the first
                                 #  iteration, 2e-5 flat afterwards -- the trajectory rule of tests/test_gpu_engine.py)
                tol = 1e-5 if i == 0 else 2e-5
                helpers.observe(f"engine_follows_golden_trajectory[{algo}-{env}]", f"iter {i} scalars", abs(m[k] - w) / (1.0 + abs(w)))
                np.testing.assert_allclose(m[k], w, rtol=tol + 1e-6, atol=tol, err_msg=f"iter {i} {k}")
