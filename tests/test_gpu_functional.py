"""GPU: functional checks beyond step-by-step parity -- the engine actually learns, checkpoints round-trip in the
reference's .pth schema, and long runs stay finite."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle.sac_td3_ref import Hps, QNet, SquashedGaussPolicy

pytestmark = pytest.mark.gpu
P = pytest.importorskip("sac_td3_cudagraphs_pytorch_amd")
from sac_td3_cudagraphs_pytorch_amd import _lib  # noqa: E402


def _agent(algo="sac", B=256, cap=20000, seed=0, **kw):
    o, a = 11, 3
    base = (Hps.td3 if algo == "td3" else Hps.sac)(batch_size=B, **kw)
    hps = SimpleNamespace(**{**base.__dict__, "cudagraphs": True, "rb_capacity": cap, "seed": seed})
    torch.manual_seed(seed)
    return P.Agent({"ob_shape": (4, o), "ac_shape": (4, a)}, np.full(a, -1.0, np.float32), np.full(a, 1.0, np.float32),
                   torch.device("cuda:0"), hps, P.ReplayBuffer(cap)), o, a


@pytest.mark.parametrize("algo", ["sac", "td3"])
def test_learns_a_one_step_control_task(algo):
    """Terminal one-step MDP: r = -|a - tanh(W s)|^2, done = True.  Q* = r, so the greedy policy is tanh(W s).
    From 20k random-action transitions the policy's error must fall far below a random policy's."""
    agent, o, a = _agent(algo, gamma=0.99)
    rng = np.random.default_rng(0)
    W = rng.standard_normal((o, a)).astype(np.float32) * 0.5
    n = 20000
    s = rng.standard_normal((n, o)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, a)).astype(np.float32)
    target = np.tanh(s @ W)
    r = -((act - target) ** 2).sum(1)
    for i in range(0, n, 2000):
        sl = slice(i, i + 2000)
        agent.rb.extend({"observations": s[sl], "actions": act[sl], "rewards": r[sl], "next_observations": s[sl],
                         "dones": np.ones(2000, bool)})
    test_s = rng.standard_normal((8, o)).astype(np.float32)

    def err():
        out = np.concatenate([agent.predict({"observations": test_s[i:i + 4]}, explore=False) for i in (0, 4)])
        return float(((out - np.tanh(test_s @ W)) ** 2).sum(1).mean())

    before = err()
    for i in range(3000):
        agent.iteration(i)
    after = err()
    m = agent.engine.read_metrics()
    assert all(np.isfinite(v) for v in m.values()), m
    assert agent.qnet_updates_so_far == 3000 and agent.actor_updates_so_far == 2000
    assert after < 0.25 * before and after < 0.3, (before, after, m)


def test_checkpoint_roundtrip_in_reference_schema(tmp_path):
    agent, o, a = _agent("sac", B=64, cap=1000)
    obs = torch.randn(500, o)
    agent.rb.extend({"observations": obs, "actions": torch.rand(500, a) * 2 - 1, "rewards": torch.randn(500),
                     "next_observations": torch.randn(500, o), "dones": torch.zeros(500, dtype=torch.bool)})
    for i in range(7):
        agent.iteration(i)
    agent.timesteps_so_far = 1234
    path = agent.save(tmp_path, sfx="best")
    ck = torch.load(path, weights_only=True)
    assert {"actor", "qnet1", "qnet2", "timesteps_so_far"} <= set(ck)
    # the reference's own module classes (as restated by the oracle, pinned to agents/nets.py) load these state_dicts
    pi = SquashedGaussPolicy(o, a, torch.full((a,), -1.0), torch.full((a,), 1.0))
    pi.load_state_dict(ck["actor"])                      # strict: same keys incl. action_scale / action_bias
    q1 = QNet(o, a)
    q1.load_state_dict(ck["qnet1"])
    want = agent.predict({"observations": obs[:4]}, explore=False)
    np.testing.assert_allclose(pi.get_action(obs[:4])["mode"].detach().numpy(), want, rtol=1e-5, atol=1e-5)
    # ... and a fresh engine resumes bit-exactly (targets, log_alpha, Adam moments and step counts included)
    other, _, _ = _agent("sac", B=64, cap=1000, seed=9)
    other.load_from_disk(path)
    assert other.timesteps_so_far == 1234
    for which in (_lib.ACTOR, _lib.CRITICS, _lib.ACTOR_TARGET, _lib.CRITICS_TARGET, _lib.LOG_ALPHA):
        assert np.array_equal(agent.engine.get_params(which), other.engine.get_params(which))
    for which in (_lib.ACTOR, _lib.CRITICS, _lib.LOG_ALPHA):
        m0, v0, t0 = agent.engine.get_adam_state(which)
        m1, v1, t1 = other.engine.get_adam_state(which)
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1) and t0 == t1


def test_long_run_with_ring_wraparound_stays_finite():
    agent, o, a = _agent("sac", B=256, cap=4096)
    rng = np.random.default_rng(1)
    for i in range(1500):
        if i % 10 == 0:   # 400 fresh rows every 10 iterations: the ring wraps several times
            n = 400
            agent.rb.extend({"observations": rng.standard_normal((n, o)), "actions": rng.uniform(-1, 1, (n, a)),
                             "rewards": rng.standard_normal(n), "next_observations": rng.standard_normal((n, o)),
                             "dones": rng.random(n) < 0.05})
        agent.iteration(i)
    assert len(agent.rb) == 4096
    m = agent.engine.read_metrics()
    assert all(np.isfinite(v) for v in m.values()) and 0 < m["vitals/alpha"] < 10, m
    for which in (_lib.ACTOR, _lib.CRITICS, _lib.CRITICS_TARGET):
        assert np.isfinite(agent.engine.get_params(which)).all()
