"""Drop-in mirror of the reference's learner object (agents/agent.py:21-331) on top of the HIP engine.

Same constructor, method names, argument meaning, counters and error behaviour (asserts / exceptions) as the
reference `Agent`, so orchestrator.py:258-412 can drive it unchanged; see INTEGRATION.md for the two-line
change in main.py.  Nothing here computes on the CPU: every method is a call into libsactd3_hip.so.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Mapping, Optional

import numpy as np

from . import _lib, schema
from .engine import Config, Engine


def _np(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class LazyMetric:
    """A loss value that stays on the device until somebody looks at it (the reference returns 0-dim device
    tensors from update_* and only materialises them at eval time, orchestrator.py:383)."""

    def __init__(self, agent: "Agent", key: str):
        self._agent, self._key = agent, key

    def item(self) -> float:
        return self._agent.engine.read_metrics()[self._key]

    __float__ = item

    def numpy(self):
        return np.float32(self.item())

    def __repr__(self):
        return f"LazyMetric({self._key})"


class BatchHandle(dict):
    """What `rb.sample()` returns: the batch lives in the engine's HBM batch slot; indexing a key reads it back
    (host sync) as the reference's TensorDict keys would (observations, actions, rewards, next_observations,
    terminations, dones, index)."""

    def __init__(self, engine: Engine):
        super().__init__()
        self._engine = engine
        self._cache: Optional[Dict[str, np.ndarray]] = None

    def __missing__(self, key):
        if self._cache is None:
            self._cache = self._engine.read_batch()
            self._cache["terminations"] = self._cache["dones"]
            for k in ("rewards", "dones", "terminations"):
                self._cache[k] = self._cache[k].reshape(-1, 1)
        return self._cache[key]


class ReplayBuffer:
    """TensorDictReplayBuffer(storage=LazyTensorStorage(capacity, device)) stand-in (main.py:167-171): built by
    the caller BEFORE the agent, bound to the engine's HBM ring when the agent receives it."""

    def __init__(self, capacity: int, device: Any = None):
        self.capacity, self.device = int(capacity), device
        self._engine: Optional[Engine] = None

    def _bind(self, engine: Engine):
        self._engine = engine

    def _need(self) -> Engine:
        assert self._engine is not None, "replay buffer is not attached to an Agent yet"
        return self._engine

    def extend(self, td: Mapping[str, Any]) -> None:
        """orchestrator.py:100-113: keys observations, next_observations, actions, rewards, terminations, dones."""
        done = td["dones"] if "dones" in td else td["terminations"]
        self._need().rb_extend(_np(td["observations"]), _np(td["actions"]), _np(td["rewards"]),
                               _np(td["next_observations"]), _np(done))

    def sample(self, batch_size: int) -> BatchHandle:
        eng = self._need()
        assert batch_size == eng.cfg.batch_size, "the engine is built for one batch size (hps.batch_size)"
        eng.rb_sample()
        return BatchHandle(eng)

    def __len__(self) -> int:
        return 0 if self._engine is None else self._engine.rb_len()


class Agent:
    """agents/agent.py:Agent, MI355X-native."""

    def __init__(self, net_shapes: Dict[str, tuple], min_ac: np.ndarray, max_ac: np.ndarray, device: Any,
                 hps: Any, rb: Optional[ReplayBuffer] = None, *, seed: Optional[int] = None,
                 init_params: bool = True, use_graphs: Optional[bool] = None):
        """`use_graphs`: None = the engine's own hipGraphs ON (whatever `hps.cudagraphs` says: that key steers the reference
        loop's CudaGraphModule wrappers, orchestrator.py:313-315, which must stay off around these methods)."""
        ob_dim, ac_dim = int(net_shapes["ob_shape"][-1]), int(net_shapes["ac_shape"][-1])
        self.device, self.hps = device, hps
        self.min_ac, self.max_ac = np.asarray(min_ac, np.float32), np.asarray(max_ac, np.float32)
        self.timesteps_so_far = 0
        self.actor_updates_so_far = 0
        self.qnet_updates_so_far = 0
        self.best_eval_ep_ret = -float("inf")  # updated by the orchestrator (orchestrator.py:376-379)
        dev_index = getattr(device, "index", None)
        over = dict(device_id=int(dev_index or 0))
        if seed is not None:
            over["seed"] = int(seed)
        if rb is not None:
            over["rb_capacity"] = rb.capacity
        if use_graphs is not None:
            over["use_graphs"] = bool(use_graphs)
        cfg = Config.from_hps(hps, ob_dim, ac_dim, **over)
        assert getattr(hps, "segment_len", 1) <= cfg.batch_size  # agents/agent.py:47
        self.ob_dim, self.ac_dim, self.td3, self.ln = ob_dim, ac_dim, cfg.prefer_td3_over_sac, cfg.layer_norm
        self.engine = Engine(cfg, self.min_ac, self.max_ac)
        self.rb = rb
        if rb is not None:
            rb._bind(self.engine)
        if init_params:  # agents/nets.py:34-49 under the caller's torch seed (main.py:146)
            actor, critics = schema.reference_initial_params(ob_dim, ac_dim, self.td3, self.ln)
            self.load_flat(actor, critics)

    # -- parameters
    def load_flat(self, actor: np.ndarray, critics: np.ndarray, also_targets: bool = True) -> None:
        e = self.engine
        e.set_params(_lib.ACTOR, actor)
        e.set_params(_lib.CRITICS, critics)
        if also_targets:  # agents/agent.py:64,107: targets start as clones
            e.set_params(_lib.ACTOR_TARGET, actor)
            e.set_params(_lib.CRITICS_TARGET, critics)

    def _nh(self) -> int:
        return self.ac_dim if self.td3 else 2 * self.ac_dim

    def state_dicts(self) -> Dict[str, Dict[str, np.ndarray]]:
        """{"actor", "qnet1", "qnet2"} with the reference's key names (agents/agent.py:346-348), LIVE weights."""
        e = self.engine
        scale, bias = (self.max_ac - self.min_ac) / 2.0, (self.max_ac + self.min_ac) / 2.0
        actor = schema.flat_to_dict(e.get_params(_lib.ACTOR), self.ob_dim, self._nh(), self.ln)
        actor["action_scale"], actor["action_bias"] = np.broadcast_to(scale, (self.ac_dim,)).copy(), np.broadcast_to(bias, (self.ac_dim,)).copy()
        q = e.get_params(_lib.CRITICS).reshape(2, -1)
        return {"actor": actor,
                "qnet1": schema.flat_to_dict(q[0], self.ob_dim + self.ac_dim, 1, self.ln),
                "qnet2": schema.flat_to_dict(q[1], self.ob_dim + self.ac_dim, 1, self.ln)}

    @property
    def alpha(self) -> Optional[float]:  # agents/agent.py:165-170
        return None if self.td3 else float(np.exp(self.engine.get_params(_lib.LOG_ALPHA)[0]))

    # -- the hot path
    def _stage(self, batch) -> None:
        if isinstance(batch, BatchHandle) or batch is None:
            return  # already in the engine's batch slot
        self.engine.load_batch(_np(batch["observations"]), _np(batch["actions"]), _np(batch["rewards"]),
                               _np(batch["next_observations"]), _np(batch["dones"]))

    def predict(self, in_td: Mapping[str, Any], *, explore: bool) -> np.ndarray:
        """agents/agent.py:172-181 -> np.ndarray[n, ac_dim] float32 on the host."""
        return self.engine.predict(_np(in_td["observations"]), explore)

    def update_qnets(self, batch) -> Dict[str, LazyMetric]:
        self._stage(batch)
        self.engine.update_qnets()
        return {"loss/qf_loss": LazyMetric(self, "loss/qf_loss")}

    def update_actor(self, batch) -> Dict[str, LazyMetric]:
        self._stage(batch)
        self.engine.update_actor()
        out = {"loss/actor_loss": LazyMetric(self, "loss/actor_loss")}
        if not self.td3:  # agents/agent.py:288-318
            if self.engine.cfg.autotune:
                out["loss/alpha_loss"] = LazyMetric(self, "loss/alpha_loss")
            out["vitals/alpha"] = LazyMetric(self, "vitals/alpha")
        return out

    def update_targ_nets(self) -> None:
        self.engine.update_targ_nets(self.qnet_updates_so_far)

    def iteration(self, i: int) -> None:
        """orchestrator.py:337-352 as ONE graph launch (sample, critic, delayed actor x N, Polyak); keeps the
        reference's counters."""
        do_actor = i % (self.engine.cfg.actor_update_delay + 1) == 0
        self.engine.step(do_actor)
        self.qnet_updates_so_far += 1
        if do_actor:
            self.actor_updates_so_far += self.engine.cfg.actor_update_delay

    # -- checkpoints (agents/agent.py:333-371), reference .pth schema with live critic weights
    def save(self, path: Path, sfx: Optional[str] = None) -> Path:
        import torch
        fname = f"ckpt_{sfx}" if sfx is not None else f".ckpt_{self.timesteps_so_far}ts"
        path = Path(path) / f"{fname}.pth"
        sds = {k: {kk: torch.from_numpy(np.ascontiguousarray(vv)) for kk, vv in v.items()} for k, v in self.state_dicts().items()}
        e = self.engine
        extra = {}
        for name, which in (("actor", _lib.ACTOR), ("critics", _lib.CRITICS), ("log_alpha", _lib.LOG_ALPHA)):
            m, v, step = e.get_adam_state(which)
            extra[f"adam/{name}"] = {"exp_avg": torch.from_numpy(m), "exp_avg_sq": torch.from_numpy(v), "step": step}
        extra["actor_target"] = torch.from_numpy(e.get_params(_lib.ACTOR_TARGET))
        extra["critics_target"] = torch.from_numpy(e.get_params(_lib.CRITICS_TARGET))
        extra["log_alpha"] = float(e.get_params(_lib.LOG_ALPHA)[0])
        torch.save({"timesteps_so_far": self.timesteps_so_far, **sds, "engine_resume": extra}, path)
        return path

    def load_from_disk(self, path: Path) -> None:
        import torch
        ck = torch.load(path, weights_only=True)
        if "timesteps_so_far" in ck:
            self.timesteps_so_far = ck["timesteps_so_far"]
        actor = schema.dict_to_flat(ck["actor"], self.ob_dim, self._nh(), self.ln)
        q = [schema.dict_to_flat(ck[k], self.ob_dim + self.ac_dim, 1, self.ln) for k in ("qnet1", "qnet2")]
        self.load_flat(actor, np.concatenate(q), also_targets="engine_resume" not in ck)
        ex = ck.get("engine_resume")
        if ex:
            e = self.engine
            e.set_params(_lib.ACTOR_TARGET, ex["actor_target"].numpy())
            e.set_params(_lib.CRITICS_TARGET, ex["critics_target"].numpy())
            e.set_params(_lib.LOG_ALPHA, np.array([ex["log_alpha"]], np.float32))
            for name, which in (("actor", _lib.ACTOR), ("critics", _lib.CRITICS), ("log_alpha", _lib.LOG_ALPHA)):
                st = ex[f"adam/{name}"]
                e.set_adam_state(which, st["exp_avg"].numpy(), st["exp_avg_sq"].numpy(), int(st["step"]))
