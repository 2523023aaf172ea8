"""GPU: functional checks beyond step-by-step parity -- the engine actually learns, checkpoints round-trip in the
reference's .pth schema, and long runs stay finite."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle.sac_td3_ref import DetPolicy, Hps, QNet, RefAgent, SquashedGaussPolicy

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


@pytest.mark.parametrize("algo", ["sac", "td3"])
def test_checkpoint_carries_the_reference_optimizer_keys(tmp_path, algo):
    """agents/agent.py:342-351,368-369: `actor_optimizer` / `q_optimizer` are torch.optim.Adam state_dicts (critics as the
    stacked [2, ...] tensors under one Adam, agents/agent.py:106-119), `hps` a plain mapping; the TD3 actor dict has the
    `exploration_noise` buffer (agents/nets.py:139-141).  Check: (i) strict load_state_dict into the oracle's modules (pinned
    to the reference's classes); (ii) torch.optim.Adam over the oracle's parameters accepts the optimiser state_dicts and
    ONE more oracle step equals ONE more engine step on the same batch and noise; (iii) a second engine that reads ONLY the
    reference's keys (engine_resume stripped) continues with the same Adam moments and step counts."""
    agent, o, a = _agent(algo, B=64, cap=1000)
    g = torch.Generator().manual_seed(3)
    rows = dict(observations=torch.randn(500, o, generator=g), actions=torch.rand(500, a, generator=g) * 2 - 1, rewards=torch.randn(500, generator=g),
                next_observations=torch.randn(500, o, generator=g), dones=torch.zeros(500, dtype=torch.bool))
    agent.rb.extend(rows)
    for i in range(6):
        agent.iteration(i)
    path = agent.save(tmp_path, sfx="best")
    ck = torch.load(path, weights_only=True)
    assert {"hps", "timesteps_so_far", "actor", "qnet1", "qnet2", "actor_optimizer", "q_optimizer"} <= set(ck)
    assert isinstance(ck["hps"], dict) and ck["hps"]["batch_size"] == 64 and ck["hps"]["prefer_td3_over_sac"] == (algo == "td3")
    lo, hi = torch.full((a,), -1.0), torch.full((a,), 1.0)
    hps = (Hps.td3 if algo == "td3" else Hps.sac)(batch_size=64)
    ref = RefAgent(o, a, lo, hi, hps)
    ref.actor.load_state_dict(ck["actor"])                                   # strict, incl. the TD3 exploration_noise buffer
    for qn, key in zip(ref.qnets, ("qnet1", "qnet2")):
        qn.load_state_dict(ck[key])
    if algo == "td3":
        assert isinstance(ref.actor, DetPolicy) and float(ck["actor"]["exploration_noise"]) == pytest.approx(0.1)
    # (ii) the optimiser state_dicts: actor as is; the critics' stacked [2, ...] entries split over the oracle's two modules
    ref.actor_optimizer.load_state_dict(ck["actor_optimizer"])
    qsd = ck["q_optimizer"]
    n_per = len(list(ref.qnets[0].parameters()))
    assert sorted(qsd["state"]) == list(range(n_per)) and qsd["state"][0]["exp_avg"].shape[0] == 2
    split = {"state": {}, "param_groups": [dict(qsd["param_groups"][0], params=list(range(2 * n_per)))]}
    for net in range(2):
        for i in range(n_per):
            st = qsd["state"][i]
            split["state"][net * n_per + i] = {"step": st["step"].clone(), "exp_avg": st["exp_avg"][net].clone(), "exp_avg_sq": st["exp_avg_sq"][net].clone()}
    ref.q_optimizer.load_state_dict(split)
    assert float(qsd["state"][0]["step"]) == 6 and float(ck["actor_optimizer"]["state"][0]["step"]) == 4
    assert qsd["param_groups"][0]["lr"] == pytest.approx(hps.qnets_lr) and ck["actor_optimizer"]["param_groups"][0]["lr"] == pytest.approx(hps.actor_lr)
    # one more critic step on both sides, same batch and noise
    b = {k: v[:64] for k, v in rows.items()}
    eps = torch.randn(64, a, generator=g)
    ex = ck["engine_resume"]
    ref.qnets_target.load_state_dict(ref.qnets.state_dict())
    from tests.test_gpu_engine import flat_critics, flat_actor
    eng = agent.engine
    eng.set_params(_lib.CRITICS_TARGET, flat_critics(ref, ref.qnets))            # both sides: targets := online critics
    if algo == "td3":
        ref.actor_target.load_state_dict(ref.actor.state_dict()); eng.set_params(_lib.ACTOR_TARGET, flat_actor(ref, ref.actor))
    else:
        ref.log_alpha.data.fill_(ex["log_alpha"])
    eng.load_batch(b["observations"], b["actions"], b["rewards"], b["next_observations"], b["dones"])
    eng.set_noise(_lib.SITE_CRITIC, eps)
    eng.update_qnets()
    out = ref.update_qnets(ref.to_batch(b["observations"], b["actions"], b["rewards"], b["next_observations"], b["dones"]), eps)
    np.testing.assert_allclose(eng.read_metrics()["loss/qf_loss"], float(out["loss/qf_loss"]), rtol=1e-5, atol=1e-5)
    from tests.helpers import assert_params_close
    assert_params_close(eng.get_params(_lib.CRITICS), flat_critics(ref, ref.qnets), hps.qnets_lr, 1, "critics after the 7th Adam step (moments from the checkpoint)")
    # (iii) the reference's keys alone restore the optimiser moments in a fresh engine
    ref_only = {k: v for k, v in ck.items() if k != "engine_resume"}
    p2 = tmp_path / "ref_keys_only.pth"
    torch.save(ref_only, p2)
    other, _, _ = _agent(algo, B=64, cap=1000, seed=5)
    other.load_from_disk(p2)
    m0, v0, t0 = ex["adam/critics"]["exp_avg"].numpy(), ex["adam/critics"]["exp_avg_sq"].numpy(), ex["adam/critics"]["step"]
    m1, v1, t1 = other.engine.get_adam_state(_lib.CRITICS)
    assert t1 == t0 == 6 and np.array_equal(m1, m0) and np.array_equal(v1, v0)
    ma, va, ta = other.engine.get_adam_state(_lib.ACTOR)
    assert ta == 4 and np.array_equal(ma, ex["adam/actor"]["exp_avg"].numpy())
    assert np.array_equal(other.engine.get_params(_lib.CRITICS_TARGET), other.engine.get_params(_lib.CRITICS))   # targets = clones (agents/agent.py:64,107)


def test_update_with_an_old_batch_handle_is_refused():
    """agents/agent.py:183,244 take the batch as an argument; here a handle names the engine's batch slot, so an update called with
    a handle from BEFORE the latest rb.sample() must fail loudly rather than train on the newest rows."""
    agent, o, a = _agent("sac", B=32, cap=400)
    agent.rb.extend({"observations": torch.randn(300, o), "actions": torch.rand(300, a) * 2 - 1, "rewards": torch.randn(300),
                     "next_observations": torch.randn(300, o), "dones": torch.zeros(300, dtype=torch.bool)})
    old = agent.rb.sample(32)
    new = agent.rb.sample(32)
    with pytest.raises(P.StaleBatchError):
        agent.update_qnets(old)
    with pytest.raises(P.StaleBatchError):
        agent.update_actor(old)
    agent.update_qnets(new); agent.update_actor(new)           # the current handle works, for both updates (orchestrator.py:341,348)
    rows = {k: np.array(new[k]) for k in ("observations", "actions", "rewards", "next_observations", "dones")}
    agent.iteration(0)                                         # the fused step draws its own sample: `new` is stale now
    with pytest.raises(P.StaleBatchError):
        agent.update_qnets(new)
    agent.update_qnets(rows)                                   # a caller-owned copy of the rows is always accepted
    assert np.isfinite(float(agent.engine.read_metrics()["loss/qf_loss"]))


def test_load_reports_hps_differences_like_the_reference(tmp_path):
    """agents/agent.py:411-415 warns with added / removed / changed config keys before loading; load_from_disk does the same
    against the checkpoint's own `hps` (a different gamma / lr must not pass silently)."""
    agent, o, a = _agent("sac", B=32, cap=200)
    path = agent.save(tmp_path, sfx="h")
    same, _, _ = _agent("sac", B=32, cap=200)
    same.load_from_disk(path)
    assert same.last_hps_diff == {"added": {}, "removed": {}, "changed": {}}
    other, _, _ = _agent("sac", B=32, cap=200, gamma=0.9, qnets_lr=5e-4)
    with pytest.warns(UserWarning, match="gamma"):
        other.load_from_disk(path)
    ch = other.last_hps_diff["changed"]
    assert ch["gamma"] == {"from": pytest.approx(0.99), "to": pytest.approx(0.9)} and "qnets_lr" in ch


def test_load_refuses_a_resume_blob_of_another_shape(tmp_path):
    agent, o, a = _agent("sac", B=32, cap=200)
    path = agent.save(tmp_path, sfx="x")
    ck = torch.load(path, weights_only=True)
    ck["engine_resume"]["adam/critics"]["exp_avg"] = ck["engine_resume"]["adam/critics"]["exp_avg"][:-8]
    torch.save(ck, path)
    before = agent.engine.get_params(_lib.CRITICS)
    with pytest.raises(ValueError):
        agent.load_from_disk(path)
    assert np.array_equal(agent.engine.get_params(_lib.CRITICS), before)        # nothing was written
    with pytest.raises(ValueError):
        agent.engine.set_adam_state(_lib.CRITICS, np.zeros(5, np.float32), np.zeros(5, np.float32), 1)


def test_shared_replay_device_path_equals_the_host_path():
    """launcher.SharedReplay on its device path (rows packed into the ring's record layout, one device slab, one k_rb_ingest
    reading device memory -- what every rank does after the RCCL all-gather) leaves the ring bit-identical to rb.extend on the
    host path, including the wrap-around; single rank here, the 2-rank exchange itself is covered over gloo on CPU."""
    from sac_td3_cudagraphs_pytorch_amd import launcher
    o, a, cap, B = 17, 6, 700, 64
    rows = dict(observations=torch.randn(1000, o), actions=torch.rand(1000, a) * 2 - 1, rewards=torch.randn(1000, 1),
                next_observations=torch.randn(1000, o), dones=torch.rand(1000, 1) < 0.1)
    rows["terminations"] = rows["dones"]
    rings = []
    for device_path in (True, False):
        eng = P.Engine(P.Config(ob_dim=o, ac_dim=a, batch_size=B, rb_capacity=cap, max_envs=4), [-1] * a, [1] * a)
        rb = P.ReplayBuffer(cap)
        rb._bind(eng)
        shared = launcher.SharedReplay(None, rb, device=torch.device("cuda:0"), every=3, force_device_path=device_path)
        assert (shared._engine is not None) == device_path
        for t in range(0, 1000, 4):                                  # 250 env steps of 4 envs: wraps the 700-row ring
            shared.extend({k: v[t:t + 4] for k, v in rows.items()})
        shared.flush()
        assert len(shared) == cap
        got = []
        for lo in range(0, cap - B + 1, B):
            eng.rb_sample_with_indices(np.arange(lo, lo + B))
            got.append(eng.read_batch())
        rings.append(got)
    for x, y in zip(*rings):
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    lay = eng.rb_layout()
    assert lay["record_floats"] % 16 == 0 and lay["capacity"] == cap and lay["next_obs_offset"] >= o + a
    with pytest.raises(P.EngineError):
        eng.rb_extend_device(0, 4)                                   # a null / host pointer is refused, not dereferenced


def test_sweep_launcher_trains_on_the_gpu(tmp_path):
    """The sweep runner for real: 2 seeds of the `debug` bundle (Hopper-v4 shapes, synthetic env), 2 worker processes (both on
    device 0 here), each job = loop.train with evaluation, tabular files, best checkpoint and a summary."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "sac_td3_cudagraphs_pytorch_amd.launcher", "--env_bundle", "debug", "--num_seeds", "2",
                          "--gpus", "2", "--device", "0", "--num_timesteps", "3000", "--learning_starts", "500", "--eval_every", "1000",
                          "--eval_steps", "2", "--batch_size", "64", "--rb_capacity", "10000", "--out", str(tmp_path)],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    jobs = [json.loads(ln[4:]) for ln in out.stdout.splitlines() if ln.startswith("JOB ")]
    assert sorted(j["seed"] for j in jobs) == [0, 1] and {j["gpu"] for j in jobs} == {0}
    for j in jobs:
        assert j["timesteps"] == 3004 and j["gradient_steps"] == 626 and j["actor_updates"] > 400     # orchestrator.py:317-352 counters
        assert all(np.isfinite(v) for v in j["final_metrics"].values()) and np.isfinite(j["best_eval_return"])
        files = set(os.listdir(j["dir"]))
        assert {"progress.json", "progress.csv", "summary.json", "ckpt_best.pth"} <= files
    sweep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("SWEEP ")][0][6:])
    assert sweep["jobs"] == 2 and sweep["gradient_steps"] == 2 * 626
