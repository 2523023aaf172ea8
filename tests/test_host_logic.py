"""CPU: host-side logic of the Python mirror that needs no GPU -- parameter schema, config mapping from the
reference's YAML keys, record layout arithmetic."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import sac_td3_cudagraphs_pytorch_amd as P
from sac_td3_cudagraphs_pytorch_amd import schema
from oracle.sac_td3_ref import DetPolicy, QNet, SquashedGaussPolicy


@pytest.mark.parametrize("o,a,ln", [(11, 3, True), (17, 6, False), (376, 17, True)])
def test_schema_matches_reference_state_dicts(o, a, ln):
    mn, mx = torch.full((a,), -1.0), torch.full((a,), 1.0)
    for net, in_dim, nh in ((SquashedGaussPolicy(o, a, mn, mx, ln), o, 2 * a), (DetPolicy(o, a, mn, mx, 0.1, ln), o, a),
                            (QNet(o, a, ln), o + a, 1)):
        sd = {k: v for k, v in net.state_dict().items() if k.startswith(("fc_stack", "head"))}
        assert [k for k, _ in schema.net_keys(in_dim, nh, ln)] == list(sd)          # same keys, same order
        assert schema.net_numel(in_dim, nh, ln) == sum(p.numel() for p in net.parameters())
        flat = schema.dict_to_flat(sd, in_dim, nh, ln)
        back = schema.flat_to_dict(flat, in_dim, nh, ln)
        assert all(np.array_equal(back[k], sd[k].detach().numpy()) for k in sd)


def test_param_counts_quoted_in_the_survey():
    # Hopper critic 70 913, SAC actor 71 430, TD3 actor 70 659; Humanoid 167 937 / 172 066
    assert schema.net_numel(14, 1, True) == 70_913 and schema.net_numel(11, 6, True) == 71_430
    assert schema.net_numel(11, 3, True) == 70_659
    assert schema.net_numel(393, 1, True) == 167_937 and schema.net_numel(376, 34, True) == 172_066


def test_reference_initial_params_are_orthogonal_and_seeded():
    torch.manual_seed(3)
    a1, c1 = schema.reference_initial_params(11, 3, False, True)
    torch.manual_seed(3)
    a2, c2 = schema.reference_initial_params(11, 3, False, True)
    assert np.array_equal(a1, a2) and np.array_equal(c1, c2)
    d = schema.flat_to_dict(a1, 11, 6, True)
    w2 = d["fc_stack.fc_block_2.fc.weight"]
    np.testing.assert_allclose(w2 @ w2.T, np.eye(256), atol=1e-5)               # nn.init.orthogonal_, gain 1
    assert not d["head.bias"].any() and (d["fc_stack.fc_block_1.ln.weight"] == 1).all()
    q = c1.reshape(2, -1)
    assert not np.array_equal(q[0], q[1])                                       # two independent critics


def test_config_from_reference_yaml_keys():
    sac = SimpleNamespace(cuda=True, cudagraphs=True, num_envs=4, layer_norm=True, actor_lr=3e-4, qnets_lr=1e-3, clip_norm=0.0,
                          segment_len=1, batch_size=256, gamma=0.99, rb_capacity=1000000, polyak=0.005,
                          prefer_td3_over_sac=False, bcq_style_targ_mix=False, actor_update_delay=2, crit_targ_update_freq=1,
                          alpha_init=0.2, autotune=True, log_alpha_lr=1e-3, seed=7)      # tasks/defaults/sac.yml
    c = P.Config.from_hps(sac, 11, 3)
    assert (c.batch_size, c.rb_capacity, c.max_envs, c.seed, c.use_graphs) == (256, 1000000, 4, 7, True)
    assert c.qnets_lr == pytest.approx(1e-3) and c.td3_std == pytest.approx(0.2)   # td3-only keys keep their defaults
    td3 = dict(prefer_td3_over_sac=True, bcq_style_targ_mix=True, qnets_lr=3e-4, actor_noise_std=0.1, targ_actor_smoothing=True,
               td3_std=0.2, td3_c=0.5, num_envs=8)                                 # td3.yml has no alpha_* keys
    c = P.Config.from_hps(td3, 17, 6, device_id=3)
    assert c.prefer_td3_over_sac and c.bcq_style_targ_mix and c.max_envs == 8 and c.device_id == 3
    assert c.alpha_init == pytest.approx(0.2) and c.crit_targ_update_freq == 1
    cc = c.to_c()
    assert cc.ob_dim == 17 and cc.ac_dim == 6 and cc.abi_version == 1 and cc.qnets_lr == pytest.approx(3e-4)


def test_compare_hps_reports_added_removed_changed_like_the_reference():
    """agents/agent.py:373-401 `compare_dictconfigs` (depth 1): added = only in the current config, removed = only in the saved one,
    changed = {"from": saved, "to": current}."""
    saved = {"gamma": 0.99, "polyak": 0.005, "actor_lr": 3e-4, "wandb_project": "x"}
    cur = {"gamma": 0.98, "polyak": 0.005, "actor_lr": 3e-4, "clip_norm": 0.5}
    d = P.Agent.compare_hps(saved, cur)
    assert d == {"added": {"clip_norm": 0.5}, "removed": {"wandb_project": "x"}, "changed": {"gamma": {"from": 0.99, "to": 0.98}}}
    assert P.Agent.compare_hps(cur, cur) == {"added": {}, "removed": {}, "changed": {}}


def test_batch_handle_goes_stale_when_the_slot_is_refilled():
    """A BatchHandle names the engine's ONE batch slot: after a later rb.sample() an older handle is refused (StaleBatchError)
    instead of silently standing for the newest sample.  (Host logic only: a stub engine counts the calls.)"""
    class StubEngine:
        def __init__(self):
            self.cfg = SimpleNamespace(batch_size=4)
            self.samples = 0

        def rb_sample(self):
            self.samples += 1

        def rb_len(self):
            return 10

        def read_batch(self):
            return {k: np.full((4, 1), self.samples, np.float32) for k in ("observations", "actions", "rewards", "next_observations", "dones", "index")}

    rb = P.ReplayBuffer(100)
    rb._bind(StubEngine())
    first = rb.sample(4)
    assert first._is_current() and float(first["observations"][0, 0]) == 1.0
    second = rb.sample(4)
    assert second._is_current() and not first._is_current()
    assert float(first["observations"][0, 0]) == 1.0          # rows already read back stay readable (they are host copies)
    third = rb.sample(4)
    with pytest.raises(P.StaleBatchError):
        second["observations"]                                 # never read while current: its rows are gone
    assert float(third["rewards"][0, 0]) == 3.0


@pytest.mark.parametrize("delay,td3,freq", [(2, False, 1), (2, True, 1), (1, False, 1), (2, False, 2), (0, False, 1)])
def test_run_iterations_splits_a_run_into_periods_a_cut_short_period_and_single_iterations(delay, td3, freq):
    """Engine.run_iterations (the loop body of orchestrator.py:337-352 for n iterations): whole periods of the actor schedule as one
    launch each, what is left behind the last whole period as ONE cut-short period (sactd3_step_prefix), iterations in front of the
    first period boundary one by one; no period graphs when the target update is gated (SAC with crit_targ_update_freq != 1) or
    there is no actor schedule (delay 0).  The launch sequence always covers exactly iterations i0 .. i0 + n - 1, actor updates at
    the multiples of delay + 1."""
    from sac_td3_cudagraphs_pytorch_amd.engine import Engine
    period = delay + 1
    can = delay > 0 and (td3 or freq == 1)
    for i0 in range(0, 7):
        for n in range(0, 11):
            calls = []
            stub = SimpleNamespace(cfg=SimpleNamespace(actor_update_delay=delay, prefer_td3_over_sac=td3, crit_targ_update_freq=freq),
                                   step=lambda a: calls.append(("step", bool(a))), step_period=lambda: calls.append(("period",)),
                                   step_prefix=lambda m: calls.append(("prefix", m)))
            assert Engine.run_iterations(stub, i0, n) == i0 + n
            i = i0
            for c in calls:                                     # replay the launches: which iterations, which of them with actor updates
                if c[0] == "period":
                    assert can and i % period == 0
                    i += period
                elif c[0] == "prefix":
                    assert can and i % period == 0 and 1 <= c[1] <= delay and c is calls[-1]
                    i += c[1]
                else:
                    assert c[1] == (i % period == 0)
                    i += 1
            assert i == i0 + n
            if can:                                             # nothing that could have been a period or a cut-short period went out singly
                assert not any(c[0] == "step" and c[1] for c in calls)
                assert sum(c[0] == "period" for c in calls) == max(0, (i0 + n) // period - (i0 + period - 1) // period)      # every whole period inside the run
            else:
                assert all(c[0] == "step" for c in calls)
