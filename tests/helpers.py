"""Shared synthetic-data helpers for the tests (shapes from SURVEY.md section 8d)."""
import numpy as np
import torch

DIMS = {"hopper": (11, 3, 1.0), "halfcheetah": (17, 6, 1.0), "humanoid": (376, 17, 0.4),
        # not environments of BASELINE.json: shapes that take the remaining branches of the narrow-head kernels (4 actions under SAC:
        # an 8-wide head with 16-float dQ/da partials in 8 tiles; 2 actions under TD3: 8-float partials in 16 tiles; 7 actions: the widest
        # action vector the folded dQ/da takes)
        "sac4": (9, 4, 1.0), "td3_2": (5, 2, 2.0), "td3_7": (13, 7, 1.0)}


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


# ---- observed parity deltas: tests report what they measured, the session writes it out (tests/conftest.py), and the
# tolerances in the tests are set from these records (<= 2x the observed value, never above north_star's 1e-5 per step)
OBSERVED = {}


def observe(test, key, value):
    """keep the LARGEST value seen for (test, key)"""
    d = OBSERVED.setdefault(test, {})
    d[key] = max(float(value), d.get(key, 0.0))


def param_keys(in_dim, nh, ln, nets=1):
    """[(name, start, stop)] over a flat parameter vector of `nets` networks back to back (sac-td3..._amd/schema.py order)"""
    from sac_td3_cudagraphs_pytorch_amd import schema
    out, off = [], 0
    for n in range(nets):
        for k, shp in schema.net_keys(in_dim, nh, ln):
            size = int(np.prod(shp))
            out.append((f"net{n}.{k}" if nets > 1 else k, off, off + size))
            off += size
    return out


def assert_params_close(got, want, lr, steps, name="", atol=2e-5, rtol=1e-5, max_bad_frac=1e-3, layout=None, vec_bad=1, record=None, max_lr_steps=None):
    """Post-Adam parameter parity.

    Adam's early steps are sign-like (p -= lr * g / (|g| + eps')), so an element whose gradient
    sits below the fp32 noise floor of its reduction (|g| <~ 1e-6 * max|g|) moves by +-lr in a
    direction that depends on summation order.  Those elements are legitimately unpredictable across
    implementations; everything else must agree tightly.  So: at most `max_bad_frac` of the elements
    may exceed (atol, rtol), and none may differ by more than 2 * lr * steps (+ atol).

    With `layout` = (in_dim, n_head, layer_norm, nets) the check is made PER state_dict KEY: a weight matrix may
    have `max_bad_frac` of its elements off, a vector (bias, LayerNorm affine, 1-row head) at most `vec_bad`
    elements -- a global fraction would let every bias and LN vector of a net (about 1 % of its elements) step
    the wrong way unnoticed.  `record` = test name: report the observed worst fractions (helpers.observe).
    `max_lr_steps`: the cap on any single element's difference in units of lr (default 2 x steps: every step the other way).
    """
    got = torch.as_tensor(got, dtype=torch.float32).reshape(-1).cpu()
    want = torch.as_tensor(want, dtype=torch.float32).reshape(-1).cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    diff = (got - want).abs()
    bad = diff > (atol + rtol * want.abs())
    cap = (2.0 * steps if max_lr_steps is None else max_lr_steps) * lr + atol
    assert diff.max().item() <= cap, f"{name}: max diff {diff.max().item():.3e} > {cap:.3e}"
    if layout is None:
        frac = bad.float().mean().item()
        assert frac <= max_bad_frac, f"{name}: {frac:.2e} of elements off (max diff {diff.max().item():.3e})"
        return
    keys = param_keys(*layout)
    assert keys[-1][2] == got.numel(), (name, keys[-1][2], got.numel())
    worst_mat, worst_vec = 0.0, 0
    for k, lo, hi in keys:
        nbad, n = int(bad[lo:hi].sum()), hi - lo
        is_vec = k.endswith(("bias", "ln.weight")) or n <= 512
        if is_vec:
            worst_vec = max(worst_vec, nbad)
            assert nbad <= vec_bad, f"{name} {k}: {nbad} of {n} elements off (max diff {diff[lo:hi].max().item():.3e})"
        else:
            worst_mat = max(worst_mat, nbad / n)
            assert nbad / n <= max_bad_frac, f"{name} {k}: {nbad / n:.2e} of elements off (max diff {diff[lo:hi].max().item():.3e})"
    if record:
        observe(record, f"{name}: max diff / lr", diff.max().item() / lr)
        observe(record, f"{name}: worst matrix bad fraction", worst_mat)
        observe(record, f"{name}: worst vector bad count", worst_vec)
