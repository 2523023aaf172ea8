"""The callers on either side of the update path, mirrored from the reference's orchestrator.py so that the engine can
be driven end to end without tensordict / torchrl / gymnasium being importable:

  segment()     rollout generator                      orchestrator.py:42-118   
  train()       the training loop's control flow       orchestrator.py:317-352 (+ counters :326,342,349)
  episode()     evaluation-episode generator           orchestrator.py:121-246 (lengths / returns only)

`env` is anything with gymnasium's vector-env protocol as the reference uses it: `reset(seed=) -> (obs[n, o], info)`,
`step(actions[n, a]) -> (next_obs, rewards[n], terminations[n], truncations[n], infos)` with autoreset and
`infos["final_observation"][k]` holding the true last observation of an env that just ended, and
`action_space.sample() -> actions[n, a]`.  `SyntheticVecEnv` below is a dependency-free stand-in of that protocol
(the image has no gymnasium / MuJoCo); it is a test double, not a port of any environment.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Generator, Optional

import numpy as np


def segment(env, agent, seed: int, segment_len: int, learning_starts: int, action_repeat: int) -> Generator[None, None, None]:
    """orchestrator.py:42-118, same order of operations and the same quirks:
    the env is seeded once (:53); random actions until `learning_starts` (:64-65); the action is refreshed every
    `action_repeat` steps (:62); the generator yields BEFORE stepping once `segment_len` steps were taken (:77-78);
    on truncation the stored next observation is the episode's true final observation (:86-89) while terminations keep
    the auto-reset observation; `dones` is the terminations array (:107-108); everything is float32 (:83,91-93,104-105)."""
    assert agent.rb is not None
    obs, _ = env.reset(seed=seed)
    obs = np.asarray(obs, np.float32)
    actions = None
    t = 0
    r = 0
    while True:
        if r % action_repeat == 0:
            if agent.timesteps_so_far < learning_starts:
                actions = env.action_space.sample()
            else:
                actions = agent.predict({"observations": obs}, explore=True)
        if t > 0 and t % segment_len == 0:
            yield
        next_obs, rewards, terminations, truncations, infos = env.step(actions)
        next_obs = np.asarray(next_obs, np.float32)
        real_next_obs = next_obs.copy()
        for idx, trunc in enumerate(np.array(truncations)):
            if trunc:
                real_next_obs[idx] = np.asarray(infos["final_observation"][idx], np.float32)
        terminations = np.asarray(terminations, bool).reshape(-1, 1)
        agent.rb.extend({
            "observations": obs,
            "next_observations": real_next_obs,
            "actions": np.asarray(actions, np.float32),
            "rewards": np.asarray(rewards, np.float32).reshape(-1, 1),
            "terminations": terminations,
            "dones": terminations,
        })
        obs = next_obs
        t += 1
        r += 1


def train(cfg: Any, env, agent, *, fused: bool = True, on_eval: Optional[Callable[[Any, int], None]] = None) -> Dict[str, float]:
    """Control flow of orchestrator.py:317-352 (no wandb / tqdm / checkpoint upload): interact, count, wait for
    `learning_starts`, then per iteration sample -> critic update -> (every delay+1 iterations) delay x actor
    update -> target update, with the reference's counters.  `fused=True` issues the whole iteration as one graph
    launch (Agent.iteration); `fused=False` makes the reference's individual calls.  Returns the last metrics."""
    seg_gen = segment(env, agent, cfg.seed, cfg.segment_len, cfg.learning_starts, cfg.action_repeat)
    i = 0
    tlog: Dict[str, Any] = {}
    while agent.timesteps_so_far <= cfg.num_timesteps:
        next(seg_gen)
        agent.timesteps_so_far += cfg.segment_len * cfg.num_envs
        if agent.timesteps_so_far <= cfg.learning_starts:
            i += 1
            continue
        if fused:
            agent.iteration(i)
        else:
            batch = agent.rb.sample(cfg.batch_size)
            tlog.update(agent.update_qnets(batch))
            agent.qnet_updates_so_far += 1
            if i % (cfg.actor_update_delay + 1) == 0:
                for _ in range(cfg.actor_update_delay):
                    tlog.update(agent.update_actor(batch))
                    agent.actor_updates_so_far += 1
            agent.update_targ_nets()
        if on_eval is not None and agent.timesteps_so_far % cfg.eval_every == 0:
            on_eval(agent, agent.timesteps_so_far)
        i += 1
    return agent.engine.read_metrics()


def episode(env, agent, seed: int) -> Generator[Dict[str, np.ndarray], None, None]:
    """orchestrator.py:121-246 without trajectory lists / pixels: one evaluation episode per `next()`, greedy
    actions (`explore=False`, :165-173), episode statistics taken from `infos["final_info"]` (:197-201), the env
    re-seeded per episode from a generator seeded with `seed` (:136-140,238)."""
    rng = np.random.default_rng(seed)

    def randomize_seed() -> int:
        return seed + rng.integers(2 ** 32 - 1, size=1).item()

    ob, _ = env.reset(seed=randomize_seed())
    while True:
        action = agent.predict({"observations": np.asarray(ob, np.float32)}, explore=False)
        ob, _reward, _termination, _truncation, infos = env.step(action)
        if "final_info" in infos:
            for info in infos["final_info"]:
                if info is None:
                    continue
                ep_len, ep_ret = float(np.asarray(info["episode"]["l"]).item()), float(np.asarray(info["episode"]["r"]).item())
            yield {"length": np.array(ep_len), "return": np.array(ep_ret)}
            ob, _ = env.reset(seed=randomize_seed())


class _Box:
    def __init__(self, low, high, n, rng):
        self.low, self.high, self._n, self._rng = low, high, n, rng

    def seed(self, seed):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high, (self._n, len(self.low))).astype(np.float32)


class SyntheticVecEnv:
    """A deterministic, dependency-free vector env with gymnasium-0.29 autoreset semantics: linear dynamics
    s' = A s + B a + noise, reward = -|s|^2/o - |a|^2/a, termination when |s|_inf > `term_at`, truncation after
    `horizon` steps; on either, `infos["final_observation"][k]` / `infos["final_info"][k]` are set and the returned
    observation is the next episode's first one."""

    def __init__(self, ob_dim: int, ac_dim: int, num_envs: int, horizon: int = 50, term_at: float = 4.0, bound: float = 1.0):
        self.o, self.a, self.n, self.horizon, self.term_at = ob_dim, ac_dim, num_envs, horizon, term_at
        g = np.random.default_rng(12345)
        self.A = (0.95 * np.eye(ob_dim) + 0.05 * g.standard_normal((ob_dim, ob_dim)) / np.sqrt(ob_dim)).astype(np.float32)
        self.Bm = (0.3 * g.standard_normal((ac_dim, ob_dim))).astype(np.float32)
        self._rng = np.random.default_rng(0)
        self.action_space = _Box(np.full(ac_dim, -bound, np.float32), np.full(ac_dim, bound, np.float32), num_envs,
                                 np.random.default_rng(0))
        self.num_envs = num_envs

    def _fresh(self, k):
        return self._rng.standard_normal((k, self.o)).astype(np.float32)

    def reset(self, seed=None):
        if seed is not None:
            self._rng = np.random.default_rng(seed)
        self.s = self._fresh(self.n)
        self.t = np.zeros(self.n, int)
        self.ret = np.zeros(self.n)
        return self.s.copy(), {}

    def step(self, actions):
        actions = np.clip(np.asarray(actions, np.float32).reshape(self.n, self.a), self.action_space.low, self.action_space.high)
        s2 = self.s @ self.A + actions @ self.Bm + 0.05 * self._rng.standard_normal((self.n, self.o)).astype(np.float32)
        rew = -(self.s ** 2).mean(1) - (actions ** 2).mean(1)
        self.t += 1
        self.ret += rew
        term = np.abs(s2).max(1) > self.term_at
        trunc = (self.t >= self.horizon) & ~term
        infos: Dict[str, Any] = {}
        ended = term | trunc
        if ended.any():
            infos["final_observation"] = np.array([s2[k].copy() if ended[k] else None for k in range(self.n)], dtype=object)
            infos["final_info"] = np.array([{"episode": {"l": np.array([self.t[k]]), "r": np.array([self.ret[k]])}} if ended[k] else None
                                            for k in range(self.n)], dtype=object)
            s2 = s2.copy()
            s2[ended] = self._fresh(int(ended.sum()))
            self.t[ended] = 0
            self.ret[ended] = 0.0
        self.s = s2.astype(np.float32)
        return self.s.copy(), rew.astype(np.float32), term, trunc, infos
