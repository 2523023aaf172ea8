"""Hand-derived backward/Adam/Polyak (oracle/manual_grads.py, the kernel plan) vs autograd oracle."""
import pytest
import torch

from oracle.manual_grads import ManualAgent
from oracle.sac_td3_ref import Hps, RefAgent
from tests.helpers import DIMS, assert_params_close, randomize_ln, synth_transitions


def _close(a, b, rtol=2e-4, atol=2e-6):
    # atol scales with the tensor's magnitude: fp32 sums in a different order
    atol = atol + 1e-5 * float(torch.as_tensor(b).abs().max())
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("algo,env,ln", [("sac", "hopper", True), ("sac", "hopper", False),
                                         ("td3", "halfcheetah", True), ("sac", "humanoid", True)])
def test_manual_matches_autograd(algo, env, ln):
    o, a, bound = DIMS[env]
    B = 64
    hps = (Hps.td3 if algo == "td3" else Hps.sac)(layer_norm=ln, batch_size=B)
    torch.manual_seed(0)
    ref = RefAgent(o, a, [-bound] * a, [bound] * a, hps)
    randomize_ln(ref)
    obs, act, rew, nobs, done = synth_transitions(B, o, a, bound, seed=3)
    b = ref.to_batch(obs, act, rew, nobs, done)
    g = torch.Generator().manual_seed(9)
    for it in range(3):
        e_c, e_a, e_l = (torch.randn(B, a, generator=g) for _ in range(3))
        man = ManualAgent(ref)  # re-sync each iteration: one step from identical state (incl. Adam moments)
        out = ref.update_qnets(b, e_c)
        ref.qnet_updates_so_far += 1
        loss = man.update_qnets(obs, act, rew, nobs, done.float(), e_c)
        _close(loss, out["loss/qf_loss"])
        _close(man.tr["targ_q"], ref.trace["targ_q"])
        names = [k for k, _ in ref.qnets[0].named_parameters()]
        flat = ref.trace["q_grads"]
        for i in range(2):
            for j, k in enumerate(names):
                _close(man.tr["q_grads"][i][k], flat[i * len(names) + j])
            for k, p in ref.qnets[i].named_parameters():
                assert_params_close(man.q[i][k], p.detach(), hps.qnets_lr, 1, k)
        out = ref.update_actor(b, e_a, e_l)
        mo = man.update_actor(obs, e_a, e_l)
        _close(mo["actor_loss"], out["loss/actor_loss"])
        for (k, p), gr in zip(ref.actor.named_parameters(), ref.trace["actor_grads"]):
            _close(man.tr["actor_grads"][k], gr)
            assert_params_close(man.actor[k], p.detach(), hps.actor_lr, 1, k)
        if algo == "sac":
            _close(mo["alpha_loss"], out["loss/alpha_loss"])
            _close(mo["alpha"], out["vitals/alpha"], rtol=1e-6, atol=1e-7)
        ref.update_targ_nets()
        man.update_targ_nets(ref.qnet_updates_so_far)
        for i in range(2):
            for k, p in ref.qnets_target[i].named_parameters():
                _close(man.q_t[i][k], p, rtol=1e-6, atol=1e-7)


def test_clip_norm():
    o, a, bound = DIMS["hopper"]
    B = 32
    hps = Hps.sac(batch_size=B, clip_norm=0.05)
    torch.manual_seed(1)
    ref = RefAgent(o, a, [-bound] * a, [bound] * a, hps)
    man = ManualAgent(ref)
    obs, act, rew, nobs, done = synth_transitions(B, o, a, bound, seed=4)
    e = torch.randn(B, a)
    ref.update_actor(ref.to_batch(obs, act, rew, nobs, done), e, e)
    man.update_actor(obs, e, e)
    for k, p in ref.actor.named_parameters():
        assert_params_close(man.actor[k], p.detach(), hps.actor_lr, 1, k)
