"""Shared synthetic-data helpers for the tests (shapes from SURVEY.md section 8d)."""
import numpy as np
import torch

DIMS = {"hopper": (11, 3, 1.0), "halfcheetah": (17, 6, 1.0), "humanoid": (376, 17, 0.4)}


def synth_transitions(n, o, a, bound, seed=0):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(n, o, generator=g)
    nobs = torch.randn(n, o, generator=g)
    act = (torch.rand(n, a, generator=g) * 2 - 1) * bound
    rew = torch.randn(n, generator=g)
    done = (torch.rand(n, generator=g) < 0.01)
    return obs, act, rew, nobs, done


def randomize_ln(agent, seed=1):
    """LN gamma/beta start at 1/0; perturb them (and biases) so their gradients are exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        nets = [agent.actor, *agent.qnets]
        for net in nets:
            for k, p in net.named_parameters():
                if k.endswith("ln.weight") or k.endswith("bias"):
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        agent.actor_target.load_state_dict(agent.actor.state_dict())
        agent.qnets_target.load_state_dict(agent.qnets.state_dict())
        # make targets differ from online nets
        for net in [agent.actor_target, *agent.qnets_target]:
            for p in net.parameters():
                p.add_(0.01 * torch.randn(p.shape, generator=g))


def assert_params_close(got, want, lr, steps, name="", atol=2e-5, rtol=1e-5, max_bad_frac=2e-3):
    """Post-Adam parameter parity.

    Adam's early steps are sign-like (p -= lr * g / (|g| + eps')), so an element whose gradient
    sits below the fp32 noise floor of its reduction (|g| <~ 1e-6 * max|g|) moves by +-lr in a
    direction that depends on summation order.  Those elements are legitimately unpredictable across
    implementations; everything else must agree tightly.  So: at most `max_bad_frac` of the elements
    may exceed (atol, rtol), and none may differ by more than 2 * lr * steps (+ atol).
    """
    got = torch.as_tensor(got, dtype=torch.float32).reshape(-1).cpu()
    want = torch.as_tensor(want, dtype=torch.float32).reshape(-1).cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    diff = (got - want).abs()
    bad = diff > (atol + rtol * want.abs())
    frac = bad.float().mean().item()
    assert frac <= max_bad_frac, f"{name}: {frac:.2e} of elements off (max diff {diff.max().item():.3e})"
    assert diff.max().item() <= 2.0 * lr * steps + atol, f"{name}: max diff {diff.max().item():.3e}"
