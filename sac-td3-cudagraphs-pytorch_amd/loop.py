"""The callers on either side of the update path, mirrored from the reference's orchestrator.py so that the engine can
be driven end to end without tensordict / torchrl / gymnasium being importable:

  Rollout / segment()   the acting side + its generator  orchestrator.py:42-118   
  train()       the training loop's control flow       orchestrator.py:317-352 (+ counters :326,342,349)
  episode()     evaluation-episode generator           orchestrator.py:121-246 (lengths / returns, trajectories with need_lists; no pixels)
  evaluate()    offline evaluation of a checkpoint     orchestrator.py:415-481 (trajectory files as .npz)
  Evaluator     the eval block of the loop             orchestrator.py:303-305,354-403 (rolling window, best model, speed)
  Tabular       key/value progress files               helpers/logger.py:93-150 (progress.json lines, progress.csv, text table)

`env` is anything with gymnasium's vector-env protocol as the reference uses it: `reset(seed=) -> (obs[n, o], info)`,
`step(actions[n, a]) -> (next_obs, rewards[n], terminations[n], truncations[n], infos)` with autoreset and
`infos["final_observation"][k]` holding the true last observation of an env that just ended, and
`action_space.sample() -> actions[n, a]`.  `SyntheticVecEnv` below is a dependency-free stand-in of that protocol
(the image has no gymnasium / MuJoCo); it is a test double, not a port of any environment.
"""
from __future__ import annotations

import json
import time
from collections import deque
from pathlib import Path
from typing import Any, Callable, Dict, Generator, Optional, TextIO

import numpy as np


class Rollout:
    """The acting side of one learner: the current observations of the vector env, the action in force and the write path
    into the replay ring.  Two half-steps, because the reference chooses the next action BEFORE it hands control to the learner
    and steps the env only afterwards (orchestrator.py:62-78): `choose()` then `advance()`.

    Reference behaviour kept (orchestrator.py:42-118): one seeded reset, never again (:53); uniformly random actions while
    `agent.timesteps_so_far < learning_starts` (:64-65), the policy's exploring action afterwards; an action stays in force for
    `action_repeat` steps (:62); a truncated env stores its true final observation as the transition's next observation while
    the rollout continues from the auto-reset one, terminated envs keep the auto-reset one (:86-89); the stored `dones` ARE the
    terminations (:107-108); every stored field is float32 / bool with `[n, 1]` rewards and flags (:83,91-93,104-105)."""

    def __init__(self, env, agent, seed: int, learning_starts: int, action_repeat: int):
        assert agent.rb is not None
        self.env, self.agent = env, agent
        self.learning_starts, self.action_repeat = learning_starts, action_repeat
        first, _ = env.reset(seed=seed)
        self.obs = np.asarray(first, np.float32)
        self.actions = None
        self.steps = 0

    def choose(self) -> None:
        if self.steps % self.action_repeat:
            return                                                   # the action in force is repeated
        if self.agent.timesteps_so_far < self.learning_starts:
            self.actions = self.env.action_space.sample()
        else:
            self.actions = self.agent.predict({"observations": self.obs}, explore=True)

    def advance(self) -> None:
        arrived, rewards, terminations, truncations, infos = self.env.step(self.actions)
        arrived = np.asarray(arrived, np.float32)
        stored_next = arrived
        cut = np.flatnonzero(np.asarray(truncations))
        if cut.size:                                                 # time-limit cuts: the episode's own last observation is stored
            stored_next = arrived.copy()
            for k in cut:
                stored_next[k] = np.asarray(infos["final_observation"][k], np.float32)
        ended = np.asarray(terminations, bool).reshape(-1, 1)
        self.agent.rb.extend({"observations": self.obs, "next_observations": stored_next,
                              "actions": np.asarray(self.actions, np.float32),
                              "rewards": np.asarray(rewards, np.float32).reshape(-1, 1),
                              "terminations": ended, "dones": ended})
        self.obs = arrived
        self.steps += 1


def segment(env, agent, seed: int, segment_len: int, learning_starts: int, action_repeat: int) -> Generator[None, None, None]:
    """orchestrator.py:42-118 as a generator over `Rollout`: control goes back to the caller every `segment_len` env steps, AFTER
    the next action has been chosen and BEFORE the env is stepped with it (:62-78) -- so the action that opens a segment was
    computed with the parameters of the previous one."""
    ro = Rollout(env, agent, seed, learning_starts, action_repeat)
    while True:
        ro.choose()
        if ro.steps and ro.steps % segment_len == 0:
            yield
        ro.advance()


def train(cfg: Any, env, agent, *, fused: bool = True, on_eval: Optional[Callable[[Any, int], None]] = None,
          evaluator: Optional["Evaluator"] = None) -> Dict[str, float]:
    """Control flow of orchestrator.py:317-352 (no wandb / tqdm / checkpoint upload): interact, count, wait for
    `learning_starts`, then per iteration sample -> critic update -> (every delay+1 iterations) delay x actor
    update -> target update, with the reference's counters.  `fused=True` issues the whole iteration as one graph
    launch (Agent.iteration); `fused=False` makes the reference's individual calls.  Every `eval_every` timesteps
    `evaluator` (the reference's eval block, :354-403) and/or `on_eval` run.  Returns the last metrics."""
    seg_gen = segment(env, agent, cfg.seed, cfg.segment_len, cfg.learning_starts, cfg.action_repeat)
    i = 0
    tlog: Dict[str, Any] = {}
    while agent.timesteps_so_far <= cfg.num_timesteps:
        if evaluator is not None:
            evaluator.maybe_start_clock(agent.timesteps_so_far)           # orchestrator.py:319-322
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
        if agent.timesteps_so_far % cfg.eval_every == 0:
            if evaluator is not None:
                evaluator(agent)
            if on_eval is not None:
                on_eval(agent, agent.timesteps_so_far)
        i += 1
    return agent.engine.read_metrics()


class Tabular:
    """Key/value progress writer with the file formats of the reference's logger (helpers/logger.py:93-150):
    `progress.json` holds one JSON object per dump, `progress.csv` one row per dump under a header that grows when a
    new key appears (earlier rows are padded), and an optional text stream gets a boxed two-column table."""

    def __init__(self, directory: Optional[Path] = None, stream: Optional[TextIO] = None, suffix: str = ""):
        self._kv: Dict[str, Any] = {}
        self._keys: list = []
        self._rows: list = []
        self._stream = stream
        self._json = self._csv = None
        if directory is not None:
            directory = Path(directory)
            directory.mkdir(parents=True, exist_ok=True)
            self._json = (directory / f"progress{suffix}.json").open("wt")
            self._csv = directory / f"progress{suffix}.csv"

    def record(self, key: str, val: Any) -> None:
        self._kv[key] = val.item() if isinstance(val, np.ndarray) and val.ndim == 0 else (float(val) if isinstance(val, np.floating) else val)

    def dump(self) -> Dict[str, Any]:
        kv, self._kv = self._kv, {}
        if not kv:
            return kv
        if self._json is not None:
            self._json.write(json.dumps(kv) + "\n")
            self._json.flush()
        if self._csv is not None:
            self._keys += [k for k in kv if k not in self._keys]
            self._rows.append(kv)
            with self._csv.open("wt") as f:            # a new column rewrites the (small) file, as the reference's writer does
                f.write(",".join(self._keys) + "\n")
                for row in self._rows:
                    f.write(",".join("" if row.get(k) is None else str(row[k]) for k in self._keys) + "\n")
        if self._stream is not None:
            cells = {k[:40]: (f"{v:<8.3g}" if isinstance(v, float) else str(v))[:40] for k, v in kv.items()}
            kw, vw = max(map(len, cells)), max(map(len, cells.values()))
            bar = "-" * (kw + vw + 7)
            self._stream.write("\n".join([bar] + [f"| {k.ljust(kw)} | {v.ljust(vw)} |" for k, v in cells.items()] + [bar]) + "\n")
            self._stream.flush()
        return kv

    def close(self) -> None:
        if self._json is not None:
            self._json.close()


class Evaluator:
    """The evaluation block of the training loop (orchestrator.py:354-403): every call plays `eval_steps` greedy
    episodes, reports the mean length / return over a rolling window of the last 20 x eval_steps episodes (:303-305,
    :364-367), saves the model as `ckpt_best` when the windowed return improves (:376-380) and computes the training speed
    in env steps per second with evaluation time excluded and a burn-in before the clock starts (:319-322,:392-397)."""

    def __init__(self, cfg: Any, eval_env, agent, tabular: Optional[Tabular] = None, ckpt_dir: Optional[Path] = None,
                 clock: Callable[[], float] = time.time):
        self.cfg, self.tabular, self.ckpt_dir, self._clock = cfg, tabular, ckpt_dir, clock
        self.ep_gen = episode(eval_env, agent, cfg.seed)
        window = 20 * cfg.eval_steps
        self.len_buff, self.ret_buff = deque(maxlen=window), deque(maxlen=window)
        self.start_time: Optional[float] = None
        self.burnin_ts: Optional[int] = None
        self.time_spent_eval = 0.0
        self.history: list = []

    def maybe_start_clock(self, timesteps_so_far: int) -> None:
        burn = getattr(self.cfg, "measure_burnin", 0)
        if self.start_time is None and timesteps_so_far >= burn + self.cfg.learning_starts:
            self.start_time, self.burnin_ts = self._clock(), timesteps_so_far

    def __call__(self, agent) -> Dict[str, float]:
        t0 = self._clock()
        for _ in range(self.cfg.eval_steps):
            ep = next(self.ep_gen)
            self.len_buff.append(float(ep["length"]))
            self.ret_buff.append(float(ep["return"]))
        out = {"timestep": int(agent.timesteps_so_far), "length": float(np.mean(np.asarray(self.len_buff, np.float32))),
               "return": float(np.mean(np.asarray(self.ret_buff, np.float32)))}
        if self.tabular is not None:
            for k, v in out.items():
                self.tabular.record(k, v)
        if out["return"] > agent.best_eval_ep_ret:
            agent.best_eval_ep_ret = out["return"]
            if self.ckpt_dir is not None:
                Path(self.ckpt_dir).mkdir(parents=True, exist_ok=True)
                agent.save(self.ckpt_dir, sfx="best")
            out["new_best"] = True
        out["replay_buffer_numel"] = len(agent.rb) if getattr(agent, "rb", None) is not None else 0
        self.time_spent_eval += self._clock() - t0
        if self.start_time is not None:
            train_time = self._clock() - self.start_time - self.time_spent_eval
            out["speed"] = (agent.timesteps_so_far - self.burnin_ts) / max(train_time, 1e-9)
            if self.tabular is not None:
                self.tabular.record("speed", out["speed"])
        if self.tabular is not None:
            self.tabular.dump()
        self.history.append(out)
        return out


class _Trajectory:
    """The per-step record of one evaluation episode (`need_lists=True`, orchestrator.py:143-148,179-195).  Columns are
    plain Python lists while the episode runs and become numpy arrays when it ends (the reference's reason: list.append is
    cheap).  Layout quirks kept on purpose: every entry keeps the vector-env's leading axis of size 1; `observations` is
    the reset observation followed by every non-final `new_ob`, `next_observations` every `new_ob` -- so the last one is
    the auto-reset observation of the NEXT episode, not the final observation of this one."""

    KEYS = ("observations", "actions", "next_observations", "rewards", "terminations", "dones")

    def __init__(self):
        self.cols: Dict[str, list] = {k: [] for k in self.KEYS}

    def begin(self, ob) -> None:
        self.cols = {k: [] for k in self.KEYS}
        self.cols["observations"].append(ob)

    def step(self, action, new_ob, reward, termination, done) -> None:
        c = self.cols
        c["next_observations"].append(new_ob)
        c["actions"].append(action)
        c["rewards"].append(reward)
        c["terminations"].append(termination)
        c["dones"].append(done)
        if not done:
            c["observations"].append(new_ob)

    def arrays(self) -> Dict[str, np.ndarray]:
        return {k: np.array(v) for k, v in self.cols.items()}


def episode(env, agent, seed: int, *, need_lists: bool = False) -> Generator[Dict[str, np.ndarray], None, None]:
    """orchestrator.py:121-246 without pixels: one evaluation episode per `next()` on a ONE-env vector env, greedy actions
    (`explore=False`, :165-173), episode statistics taken from `infos["final_info"]` (:197-201), the env re-seeded before
    every episode from a generator seeded with `seed` (:136-140,152,238).  `need_lists=True` adds the trajectory (see
    _Trajectory) to the yielded dict, as `evaluate()` consumes it (:446-457)."""
    rng = np.random.default_rng(seed)

    def fresh_episode():
        ob, _ = env.reset(seed=seed + rng.integers(2 ** 32 - 1, size=1).item())
        return ob

    traj = _Trajectory() if need_lists else None
    ob = fresh_episode()
    if traj:
        traj.begin(ob)
    while True:
        action = agent.predict({"observations": np.asarray(ob, np.float32)}, explore=False)
        new_ob, reward, termination, truncation, infos = env.step(action)
        done = bool(np.asarray(termination).any() or np.asarray(truncation).any())
        if traj:
            traj.step(action, new_ob, reward, termination, np.asarray(done) if np.ndim(termination) == 0 else np.asarray(termination) | np.asarray(truncation), )
        ob = new_ob
        if "final_info" in infos:
            stats = [i["episode"] for i in infos["final_info"] if i is not None][-1]
            out = traj.arrays() if traj else {}
            out["length"] = np.array(float(np.asarray(stats["l"]).item()))
            out["return"] = np.array(float(np.asarray(stats["r"]).item()))
            yield out
            ob = fresh_episode()
            if traj:
                traj.begin(ob)


def evaluate(cfg: Any, env, agent, name: str = "eval", tabular: Optional[Tabular] = None) -> Dict[str, float]:
    """orchestrator.py:415-481 without wandb / pixels: optionally restore a checkpoint (`cfg.load_ckpt`: a local .pth path
    for `agent.load_from_disk`; the reference downloads it from wandb, :430), play `cfg.num_episodes` greedy episodes, and
    with `cfg.gather_trajectories` write each one under `cfg.trajectory_dir / name` as `{i:03d}_L{length}_R{return}.npz`
    (the reference writes the same arrays with TensorDict.to_h5, :446-457; h5py / tensordict are not in this image) after
    checking that every column has `length` rows (:450-451) and casting float64 columns to float32 (:454-456).  Returns the
    mean length / return (:473-476, float32 means) and records them in `tabular` (:478-481)."""
    traj_dir = None
    if getattr(cfg, "gather_trajectories", False):
        traj_dir = Path(cfg.trajectory_dir) / name
        traj_dir.mkdir(parents=True, exist_ok=True)
    ckpt = getattr(cfg, "load_ckpt", None)
    if ckpt:
        agent.load_from_disk(Path(ckpt))
    ep_gen = episode(env, agent, cfg.seed, need_lists=traj_dir is not None)
    lens, rets = [], []
    for i in range(cfg.num_episodes):
        ep = next(ep_gen)
        lens.append(ep["length"])
        rets.append(ep["return"])
        if traj_dir is not None:
            n = int(ep["length"])
            cols = {k: v for k, v in ep.items() if k not in ("length", "return")}
            for k, v in cols.items():
                assert v.shape[0] == n, f"wrong array length for {k=}"
            cols = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in cols.items()}
            np.savez(traj_dir / f"{str(i).zfill(3)}_L{n}_R{int(ep['return'])}.npz", length=ep["length"], **{"return": ep["return"]}, **cols)
    out = {"length": float(np.asarray(lens, np.float32).mean()), "return": float(np.asarray(rets, np.float32).mean())}
    if tabular is not None:
        for k, v in out.items():
            tabular.record(k, v)
        tabular.dump()
    return out


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
