"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Hand-derived forward/backward/Adam/Polyak of the SAC/TD3 update, written WITHOUT
autograd and decomposed exactly the way the HIP engine's kernels are
(`csrc/kernels.hip`): NT/NN/TN GEMMs, LayerNorm+ReLU prologue, row-wise "tail"
kernels, flat Adam, flat lerp.  `tests/test_manual_grads.py` checks every
gradient and every post-step parameter here against `oracle/sac_td3_ref.py`
(real autograd + torch.optim.Adam); the GPU tests then use the intermediates
computed here (z1, dz2, ...) to localise a faulty kernel.

Follows: agents/agent.py:183-331 and agents/nets.py:52-234 of the reference, via
oracle/sac_td3_ref.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

LN_EPS = 1e-5
K1, K2 = "fc_stack.fc_block_1", "fc_stack.fc_block_2"


# ----------------------------------------------------------------- building blocks

def ln_relu_fwd(z, g, b):
    """returns h, (xhat, rstd, y).  g is None => no LayerNorm."""
    if g is None:
        return torch.relu(z), (None, None, z)
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + LN_EPS)
    xhat = (z - mu) * rstd
    y = xhat * g + b
    return torch.relu(y), (xhat, rstd, y)


def ln_relu_bwd(dh, saved, g):
    """returns dz, dgamma, dbeta (the last two None without LayerNorm)."""
    xhat, rstd, y = saved
    dy = dh * (y > 0)
    if g is None:
        return dy, None, None
    dg = (dy * xhat).sum(0)
    db = dy.sum(0)
    dxh = dy * g
    dz = rstd * (dxh - dxh.mean(-1, keepdim=True) - xhat * (dxh * xhat).mean(-1, keepdim=True))
    return dz, dg, db


def trunk_fwd(p: Dict[str, torch.Tensor], x):
    ln = f"{K1}.ln.weight" in p
    z1 = x @ p[f"{K1}.fc.weight"].T + p[f"{K1}.fc.bias"]
    h1, s1 = ln_relu_fwd(z1, p.get(f"{K1}.ln.weight"), p.get(f"{K1}.ln.bias"))
    z2 = h1 @ p[f"{K2}.fc.weight"].T + p[f"{K2}.fc.bias"]
    h2, s2 = ln_relu_fwd(z2, p.get(f"{K2}.ln.weight"), p.get(f"{K2}.ln.bias"))
    return h2, dict(x=x, z1=z1, h1=h1, s1=s1, z2=z2, h2=h2, s2=s2, ln=ln)


def trunk_bwd(p, c, dh2, need_param_grads=True) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """returns (grads dict, dx)."""
    g: Dict[str, torch.Tensor] = {}
    dz2, dg2, db2 = ln_relu_bwd(dh2, c["s2"], p.get(f"{K2}.ln.weight"))
    dh1 = dz2 @ p[f"{K2}.fc.weight"]
    dz1, dg1, db1 = ln_relu_bwd(dh1, c["s1"], p.get(f"{K1}.ln.weight"))
    dx = dz1 @ p[f"{K1}.fc.weight"]
    if need_param_grads:
        g[f"{K2}.fc.weight"] = dz2.T @ c["h1"]
        g[f"{K2}.fc.bias"] = dz2.sum(0)
        g[f"{K1}.fc.weight"] = dz1.T @ c["x"]
        g[f"{K1}.fc.bias"] = dz1.sum(0)
        if c["ln"]:
            g[f"{K2}.ln.weight"], g[f"{K2}.ln.bias"] = dg2, db2
            g[f"{K1}.ln.weight"], g[f"{K1}.ln.bias"] = dg1, db1
    c["dz2"], c["dz1"], c["dh1"] = dz2, dz1, dh1
    return g, dx


def critic_fwd(p, ob, ac):
    h2, c = trunk_fwd(p, torch.cat([ob, ac], -1))
    q = h2 @ p["head.weight"].T + p["head.bias"]      # [B, 1]
    return q.squeeze(-1), c


def critic_bwd(p, c, dq, need_param_grads=True):
    dh2 = dq[:, None] * p["head.weight"]               # [B, H]
    g, dx = trunk_bwd(p, c, dh2, need_param_grads)
    if need_param_grads:
        g["head.weight"] = (dq[:, None] * c["h2"]).sum(0, keepdim=True)
        g["head.bias"] = dq.sum().reshape(1)
    return g, dx


def tanh_gauss_fwd(u, eps, scale, bias):
    a = u.shape[-1] // 2
    mean, raw = u[:, :a], u[:, a:]
    t = torch.tanh(raw)
    log_std = -5.0 + 0.5 * (2.0 - (-5.0)) * (t + 1)
    std = torch.exp(log_std)
    x = mean + eps * std
    y = torch.tanh(x)
    action = y * scale + bias
    lp = -((x - mean) ** 2) / (2 * std ** 2) - torch.log(std) - math.log(math.sqrt(2 * math.pi))
    lp = lp - torch.log(scale * (1 - y * y) + 1e-6)
    logp = lp.sum(1)
    mode = torch.tanh(mean) * scale + bias
    return action, logp, mode, dict(t=t, std=std, y=y, eps=eps)


def tanh_gauss_bwd(dA, dlogp, s, scale):
    """dA [B,a] grad wrt action, dlogp [B] grad wrt log-prob -> du [B,2a]."""
    t, std, y, eps = s["t"], s["std"], s["y"], s["eps"]
    omy2 = 1 - y * y
    g0 = dA * scale * omy2 + dlogp[:, None] * (2 * scale * y * omy2) / (scale * omy2 + 1e-6)
    gmean = g0
    glogstd = g0 * eps * std - dlogp[:, None]
    graw = glogstd * 3.5 * (1 - t * t)
    return torch.cat([gmean, graw], -1)


def adam_step(p, g, m, v, t: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor, non-amsgrad, no weight decay; t is the NEW step count."""
    m = m + (1 - b1) * (g - m)
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    denom = torch.sqrt(v) / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def lerp(t, p, w):
    return t + w * (p - t)


# ----------------------------------------------------------------- whole updates

class ManualAgent:
    """State = plain dicts of tensors keyed like the reference state_dicts."""

    def __init__(self, ref):  # ref: oracle.sac_td3_ref.RefAgent (state is copied, not shared)
        cp = lambda sd: {k: v.detach().clone() for k, v in sd.items()}
        self.h = ref.hps
        self.actor = cp(dict(ref.actor.named_parameters()))
        self.actor_t = cp(dict(ref.actor_target.named_parameters()))
        self.q = [cp(dict(n.named_parameters())) for n in ref.qnets]
        self.q_t = [cp(dict(n.named_parameters())) for n in ref.qnets_target]
        self.scale, self.bias = ref.actor.action_scale.clone(), ref.actor.action_bias.clone()
        self.min_ac, self.max_ac = ref.min_ac.clone(), ref.max_ac.clone()
        self.log_alpha = None if ref.log_alpha is None else ref.log_alpha.detach().clone()
        self.targ_ent = -ref.ac_dim
        z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
        self.am, self.av, self.at = z(self.actor), z(self.actor), 0
        self.qm, self.qv, self.qt = [z(d) for d in self.q], [z(d) for d in self.q], 0
        self.lm = self.lv = torch.zeros(()); self.lt = 0
        self.tr: Dict[str, object] = {}
        self._load_adam(ref)

    def _load_adam(self, ref):
        """Copy torch.optim.Adam state (exp_avg, exp_avg_sq, step) so a ManualAgent can be re-synchronised
        with the autograd oracle at any point of a run."""
        def pull(opt, named, m, v):
            t = 0
            for k, p in named:
                st = opt.state.get(p)
                if st:
                    m[k], v[k], t = st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(st["step"])
            return t
        self.at = pull(ref.actor_optimizer, ref.actor.named_parameters(), self.am, self.av)
        for i, n in enumerate(ref.qnets):
            self.qt = pull(ref.q_optimizer, n.named_parameters(), self.qm[i], self.qv[i])
        if self.log_alpha is not None and self.h.autotune:
            st = ref.alpha_optimizer.state.get(ref.log_alpha)
            if st:
                self.lm, self.lv, self.lt = st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(st["step"])

    def _actor_fwd(self, p, ob, eps):
        h2, c = trunk_fwd(p, ob)
        u = h2 @ p["head.weight"].T + p["head.bias"]
        c["u"] = u
        if self.h.prefer_td3_over_sac:
            return torch.tanh(u) * self.scale + self.bias, None, c
        action, logp, _mode, s = tanh_gauss_fwd(u, eps, self.scale, self.bias)
        c["s"] = s
        return action, logp, c

    def update_qnets(self, ob, ac, rew, nob, done, eps):
        h = self.h
        if h.prefer_td3_over_sac:
            pi_next, _, _ = self._actor_fwd(self.actor_t, nob, None)
            if h.targ_actor_smoothing:
                n_ = (eps * h.td3_std).clamp(-h.td3_c, h.td3_c)
                a_next = torch.max(torch.min(pi_next + n_, self.max_ac), self.min_ac)
            else:
                a_next = pi_next
            logp_next = None
        else:
            a_next, logp_next, _ = self._actor_fwd(self.actor, nob, eps)
        qt = torch.stack([critic_fwd(p, nob, a_next)[0] for p in self.q_t], 0)
        qmin = torch.minimum(qt[0], qt[1])
        qp = 0.75 * qmin + 0.25 * torch.maximum(qt[0], qt[1]) if h.bcq_style_targ_mix else qmin
        if logp_next is not None:
            qp = qp - torch.exp(self.log_alpha) * logp_next
        y = rew + (1.0 - done) * h.gamma * qp
        B = ob.shape[0]
        loss = 0.0
        grads: List[Dict[str, torch.Tensor]] = []
        caches = []
        for p in self.q:
            q, c = critic_fwd(p, ob, ac)
            loss = loss + ((q - y) ** 2).mean()
            g, _ = critic_bwd(p, c, 2.0 * (q - y) / B)
            grads.append(g); caches.append(c); c["q"] = q
        self.qt += 1
        for i, p in enumerate(self.q):
            for k in p:
                p[k], self.qm[i][k], self.qv[i][k] = adam_step(p[k], grads[i][k], self.qm[i][k], self.qv[i][k],
                                                               self.qt, h.qnets_lr)
        self.tr.update(a_next=a_next, logp_next=logp_next, q_target=qt, targ_q=y, q_grads=grads,
                       q_caches=caches, qf_loss=loss)
        return loss

    def update_actor(self, ob, eps, eps_alpha):
        h = self.h
        B = ob.shape[0]
        a_pi, logp, c = self._actor_fwd(self.actor, ob, eps)
        qs, cs = zip(*[critic_fwd(p, ob, a_pi) for p in self.q])
        if h.prefer_td3_over_sac:
            loss = (-qs[0]).mean()
            dq = [torch.full((B,), -1.0 / B), torch.zeros(B)]
            dlogp = None
        else:
            alpha = torch.exp(self.log_alpha)
            first = qs[0] <= qs[1]                      # ties -> critic 0
            loss = (alpha * logp - torch.where(first, qs[0], qs[1])).mean()
            dq = [-(first.float()) / B, -((~first).float()) / B]
            dlogp = torch.full((B,), float(alpha) / B)
        o = ob.shape[1]
        dA = sum(critic_bwd(p, cc, d, need_param_grads=False)[1][:, o:] for p, cc, d in zip(self.q, cs, dq))
        if h.prefer_td3_over_sac:
            th = torch.tanh(c["u"])
            du = dA * self.scale * (1 - th * th)
        else:
            du = tanh_gauss_bwd(dA, dlogp, c["s"], self.scale)
        p = self.actor
        g, _ = trunk_bwd(p, c, du @ p["head.weight"])
        g["head.weight"] = du.T @ c["h2"]
        g["head.bias"] = du.sum(0)
        if h.clip_norm > 0:
            norm = torch.sqrt(sum((x * x).sum() for x in g.values()))
            coef = min(1.0, float(h.clip_norm / (norm + 1e-6)))
            g = {k: v * coef for k, v in g.items()}
        self.at += 1
        for k in p:
            p[k], self.am[k], self.av[k] = adam_step(p[k], g[k], self.am[k], self.av[k], self.at, h.actor_lr)
        self.tr.update(a_pi=a_pi, logp_pi=logp, q_pi=torch.stack(qs, 0), actor_grads=g, actor_cache=c,
                       actor_loss=loss, dA=dA, du=du)
        out = {"actor_loss": loss}
        if h.prefer_td3_over_sac:
            return out
        if h.autotune:
            _, logp2, _ = self._actor_fwd(self.actor, ob, eps_alpha)
            alpha = torch.exp(self.log_alpha)
            mean_term = (-logp2 - self.targ_ent).mean()
            out["alpha_loss"] = alpha * mean_term
            g_la = alpha * mean_term
            self.lt += 1
            self.log_alpha, self.lm, self.lv = adam_step(self.log_alpha, g_la, self.lm, self.lv, self.lt,
                                                         h.log_alpha_lr)
            self.tr["logp_alpha"] = logp2
        out["alpha"] = torch.exp(self.log_alpha)
        return out

    def update_targ_nets(self, qnet_updates_so_far: int):
        h = self.h
        if h.prefer_td3_over_sac or qnet_updates_so_far % h.crit_targ_update_freq == 0:
            for t, p in zip(self.q_t, self.q):
                for k in t:
                    t[k] = lerp(t[k], p[k], h.polyak)
            if h.prefer_td3_over_sac:
                for k in self.actor_t:
                    self.actor_t[k] = lerp(self.actor_t[k], self.actor[k], h.polyak)
