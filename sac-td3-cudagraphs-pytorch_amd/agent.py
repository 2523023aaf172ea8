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


class _DeviceScalar:
    """__cuda_array_interface__ view of ONE float32 in the engine's device memory (a metrics slot)."""

    def __init__(self, ptr: int):
        self.__cuda_array_interface__ = {"shape": (), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}


_METRIC_SLOT = {"loss/qf_loss": 0, "loss/actor_loss": 1, "loss/alpha_loss": 2, "vitals/alpha": 3}   # SACTD3_M_*


class StaleBatchError(RuntimeError):
    """A BatchHandle names THE batch slot of the engine, not a copy: once a later `rb.sample()` (or a caller-owned batch) has
    refilled the slot, an older handle no longer stands for the rows it was made for.  The reference's loop never keeps one
    (orchestrator.py:338-348 samples, updates, drops); a caller that does gets this error instead of a silent update on the
    newest sample."""


class BatchHandle(dict):
    """What `rb.sample()` returns: the batch lives in the engine's HBM batch slot; indexing a key reads it back
    (host sync) as the reference's TensorDict keys would (observations, actions, rewards, next_observations,
    terminations, dones, index)."""

    def __init__(self, engine: Engine, generation: int = 0):
        super().__init__()
        self._engine = engine
        self._generation = generation      # which rb.sample() filled the batch slot when this handle was made (see StaleBatchError)
        self._cache: Optional[Dict[str, np.ndarray]] = None

    def _is_current(self) -> bool:
        return getattr(self._engine, "_batch_generation", 0) == self._generation

    def __missing__(self, key):
        if self._cache is None:
            if not self._is_current():
                raise StaleBatchError("this batch handle is older than the engine's batch slot: a later rb.sample() / staged batch replaced its rows")
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
        eng._batch_generation = getattr(eng, "_batch_generation", 0) + 1
        return BatchHandle(eng, eng._batch_generation)

    def __len__(self) -> int:
        return 0 if self._engine is None else self._engine.rb_len()


class Agent:
    """agents/agent.py:Agent, MI355X-native."""

    def __init__(self, net_shapes: Dict[str, tuple], min_ac: np.ndarray, max_ac: np.ndarray, device: Any,
                 hps: Any, rb: Optional[ReplayBuffer] = None, *, seed: Optional[int] = None,
                 init_params: bool = True, use_graphs: Optional[bool] = None, metrics: str = "tensor"):
        """`use_graphs`: None = the engine's own hipGraphs ON (whatever `hps.cudagraphs` says: that key steers the reference
        loop's CudaGraphModule wrappers, orchestrator.py:313-315, which must stay off around these methods).
        `metrics`: "tensor" = update_* return 0-dim float32 CUDA tensors that alias the engine's metrics slots (zero copy; what
        `tlog.update(...)` of orchestrator.py:302,341,348 expects -- a TensorDict takes them as is, and they are only read at
        evaluation time, :383); "lazy" = host-side LazyMetric handles (no torch needed)."""
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
        self._metric_tensors: Optional[Dict[str, Any]] = None
        self._ext_stream = None
        if metrics != "lazy":   # 0-dim device tensors over the engine's metrics slots (what the reference's update_* return)
            try:
                import torch
                if torch.cuda.is_available():
                    st, mp = self.engine.device_handles()
                    dev = torch.device("cuda", cfg.device_id)
                    self._metric_tensors = {k: torch.as_tensor(_DeviceScalar(mp + 4 * i), device=dev) for k, i in _METRIC_SLOT.items()}
                    self._ext_stream = torch.cuda.ExternalStream(st, device=dev)
            except Exception:   # noqa: BLE001 -- no torch / no cuda array interface: the lazy host-side handles remain
                self._metric_tensors = None
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
        if self.td3:   # agents/nets.py:139-141: the TD3 actor registers its exploration sigma as a third buffer
            actor["exploration_noise"] = np.asarray(e.cfg.actor_noise_std, np.float32)
        q = e.get_params(_lib.CRITICS).reshape(2, -1)
        return {"actor": actor,
                "qnet1": schema.flat_to_dict(q[0], self.ob_dim + self.ac_dim, 1, self.ln),
                "qnet2": schema.flat_to_dict(q[1], self.ob_dim + self.ac_dim, 1, self.ln)}

    @property
    def alpha(self) -> Optional[float]:  # agents/agent.py:165-170
        return None if self.td3 else float(np.exp(self.engine.get_params(_lib.LOG_ALPHA)[0]))

    # -- the hot path
    def _stage(self, batch) -> None:
        if batch is None:
            return  # whatever is in the engine's batch slot
        if isinstance(batch, BatchHandle):
            if batch._engine is not self.engine:
                raise StaleBatchError("this batch handle belongs to another agent's replay buffer")
            if not batch._is_current():
                raise StaleBatchError("update called with an old batch handle: the engine's batch slot holds a later sample "
                                      "(keep the rows, e.g. dict(handle), to train on them again)")
            return  # already in the engine's batch slot
        self.engine._batch_generation = getattr(self.engine, "_batch_generation", 0) + 1   # a caller-owned batch replaces the slot
        self.engine.load_batch(_np(batch["observations"]), _np(batch["actions"]), _np(batch["rewards"]),
                               _np(batch["next_observations"]), _np(batch["dones"]))

    def predict(self, in_td: Mapping[str, Any], *, explore: bool) -> np.ndarray:
        """agents/agent.py:172-181 -> np.ndarray[n, ac_dim] float32 on the host."""
        return self.engine.predict(_np(in_td["observations"]), explore)

    def _results(self, keys) -> Dict[str, Any]:
        if self._metric_tensors is None:
            return {k: LazyMetric(self, k) for k in keys}
        # the tensors alias memory the engine's stream writes: whoever reads them on torch's current stream does so after the
        # work enqueued so far (an event wait between two streams, no host synchronisation)
        import torch
        torch.cuda.current_stream(self._ext_stream.device).wait_event(self._ext_stream.record_event())
        return {k: self._metric_tensors[k] for k in keys}

    def update_qnets(self, batch) -> Dict[str, Any]:
        self._stage(batch)
        self.engine.update_qnets()
        return self._results(["loss/qf_loss"])

    def update_actor(self, batch) -> Dict[str, Any]:
        self._stage(batch)
        self.engine.update_actor()
        keys = ["loss/actor_loss"]
        if not self.td3:  # agents/agent.py:288-318
            if self.engine.cfg.autotune:
                keys.append("loss/alpha_loss")
            keys.append("vitals/alpha")
        return self._results(keys)

    def update_targ_nets(self) -> None:
        self.engine.update_targ_nets(self.qnet_updates_so_far)

    def iteration(self, i: int) -> None:
        """orchestrator.py:337-352 as ONE graph launch (sample, critic, delayed actor x N, Polyak); keeps the
        reference's counters."""
        do_actor = i % (self.engine.cfg.actor_update_delay + 1) == 0
        self.engine.step(do_actor)
        self.engine._batch_generation = getattr(self.engine, "_batch_generation", 0) + 1      # the fused step drew a new sample
        self.qnet_updates_so_far += 1
        if do_actor:
            self.actor_updates_so_far += self.engine.cfg.actor_update_delay

    # -- checkpoints (agents/agent.py:333-371): the reference's .pth schema, with LIVE critic weights
    def _plain_hps(self) -> Dict[str, Any]:
        """`hps` as a plain mapping of builtin scalars (the reference stores its DictConfig object, agents/agent.py:343:
        only a pickling loader can read that back; a plain dict also loads with torch.load(weights_only=True))."""
        h = self.hps
        try:
            items = dict(h).items() if hasattr(h, "keys") else vars(h).items()
        except TypeError:
            items = {}
        return {str(k): v for k, v in items if isinstance(v, (bool, int, float, str)) or v is None}

    def _adam_state_dict(self, which: int, shapes, lr: float, stacked: int):
        """torch.optim.Adam.state_dict() of the optimiser that owns `which`, in the reference's form: parameters in
        `module.parameters()` order (= state_dict order), the twin critics as the dense-stacked [2, ...] tensors of
        agents/agent.py:106-119 under ONE Adam instance, per-parameter `step` / `exp_avg` / `exp_avg_sq`, torch's own
        `param_groups`.  Built by filling a real torch.optim.Adam, so the layout is whatever this torch version writes."""
        import torch
        m, v, step = self.engine.get_adam_state(which)
        m, v = m.reshape(max(stacked, 1), -1), v.reshape(max(stacked, 1), -1)
        params, off = [], 0
        mv = []
        for _, shp in shapes:
            n = int(np.prod(shp))
            full = ((stacked,) if stacked else ()) + tuple(shp)
            params.append(torch.nn.Parameter(torch.zeros(full)))
            take = (lambda a: a[:, off:off + n].reshape(full)) if stacked else (lambda a: a[0, off:off + n].reshape(full))
            mv.append((torch.from_numpy(np.ascontiguousarray(take(m))), torch.from_numpy(np.ascontiguousarray(take(v)))))
            off += n
        opt = torch.optim.Adam(params, lr=lr, betas=(self.engine.cfg.adam_beta1, self.engine.cfg.adam_beta2), eps=self.engine.cfg.adam_eps)
        if step > 0:                                   # torch creates the state lazily, at the first step
            for p_, (m_, v_) in zip(params, mv):
                opt.state[p_] = {"step": torch.tensor(float(step)), "exp_avg": m_, "exp_avg_sq": v_}
        return opt.state_dict()

    def save(self, path: Path, sfx: Optional[str] = None) -> Path:
        """agents/agent.py:333-358 (wandb upload aside): keys `hps`, `timesteps_so_far`, `actor`, `qnet1`, `qnet2`,
        `actor_optimizer`, `q_optimizer` as the reference writes them -- its own `load_from_disk` (:360-371) accepts the file --
        plus `engine_resume` (what the reference does not save: targets, log_alpha and its optimiser, every step count) for
        a bit-exact resume of this engine."""
        import torch
        fname = f"ckpt_{sfx}" if sfx is not None else f".ckpt_{self.timesteps_so_far}ts"
        path = Path(path) / f"{fname}.pth"
        sds = {k: {kk: torch.from_numpy(np.ascontiguousarray(vv)) for kk, vv in v.items()} for k, v in self.state_dicts().items()}
        e, c = self.engine, self.engine.cfg
        ck = {"hps": self._plain_hps(), "timesteps_so_far": self.timesteps_so_far, **sds,
              "actor_optimizer": self._adam_state_dict(_lib.ACTOR, schema.net_keys(self.ob_dim, self._nh(), self.ln), c.actor_lr, 0),
              "q_optimizer": self._adam_state_dict(_lib.CRITICS, schema.net_keys(self.ob_dim + self.ac_dim, 1, self.ln), c.qnets_lr, 2)}
        extra = {}
        for name, which in (("actor", _lib.ACTOR), ("critics", _lib.CRITICS), ("log_alpha", _lib.LOG_ALPHA)):
            m, v, step = e.get_adam_state(which)
            extra[f"adam/{name}"] = {"exp_avg": torch.from_numpy(m), "exp_avg_sq": torch.from_numpy(v), "step": step}
        extra["actor_target"] = torch.from_numpy(e.get_params(_lib.ACTOR_TARGET))
        extra["critics_target"] = torch.from_numpy(e.get_params(_lib.CRITICS_TARGET))
        extra["log_alpha"] = float(e.get_params(_lib.LOG_ALPHA)[0])
        ck["engine_resume"] = extra
        torch.save(ck, path)
        return path

    @staticmethod
    def _adam_from_state_dict(sd, stacked: int):
        """(exp_avg flat, exp_avg_sq flat, step) in the engine's layout from a torch Adam state_dict (see _adam_state_dict)."""
        st = sd["state"]
        if not st:
            return None
        idx = sorted(st)
        f = lambda key: np.concatenate([st[i][key].detach().cpu().numpy().astype(np.float32).reshape(max(stacked, 1), -1) for i in idx], 1).reshape(-1)
        return f("exp_avg"), f("exp_avg_sq"), int(round(float(st[idx[0]]["step"])))

    @staticmethod
    def compare_hps(saved: Mapping[str, Any], current: Mapping[str, Any]) -> Dict[str, Dict[str, Any]]:
        """agents/agent.py:373-401 (`compare_dictconfigs`): depth-1 comparison of two configs -> {"added", "removed", "changed"}
        (`added`: keys only the current config has; `removed`: keys only the saved one has; `changed`: {"from": saved, "to": current})."""
        diff: Dict[str, Dict[str, Any]] = {"added": {}, "removed": {}, "changed": {}}
        k1, k2 = set(saved.keys()), set(current.keys())
        for k in sorted(k2 - k1):
            diff["added"][k] = current[k]
        for k in sorted(k1 - k2):
            diff["removed"][k] = saved[k]
        for k in sorted(k1 & k2):
            if saved[k] != current[k]:
                diff["changed"][k] = {"from": saved[k], "to": current[k]}
        return diff

    def load_from_disk(self, path: Path) -> None:
        """agents/agent.py:360-371.  Reads files written by `save` above and any file in the reference's schema that a
        weights-only loader accepts (state_dicts + optimiser state_dicts of tensors and builtin scalars).  A .pth written
        by the reference itself pickles its OmegaConf DictConfig under `hps`: torch.load(weights_only=True) refuses it,
        and this loader is deliberately not loosened."""
        import torch
        ck = torch.load(path, weights_only=True)
        # the reference compares the saved run's config with the current one and reports added / removed / changed keys before
        # it loads (agents/agent.py:411-415, there against the wandb run's config); here against the checkpoint's own `hps`
        self.last_hps_diff = self.compare_hps(ck["hps"], self._plain_hps()) if isinstance(ck.get("hps"), dict) else None
        if self.last_hps_diff and any(self.last_hps_diff.values()):
            import warnings
            d = self.last_hps_diff
            warnings.warn(f"checkpoint hps differ from this agent's -- added: {d['added']}  removed: {d['removed']}  changed: {d['changed']}")
        if "timesteps_so_far" in ck:
            self.timesteps_so_far = ck["timesteps_so_far"]
        actor = schema.dict_to_flat(ck["actor"], self.ob_dim, self._nh(), self.ln)
        q = [schema.dict_to_flat(ck[k], self.ob_dim + self.ac_dim, 1, self.ln) for k in ("qnet1", "qnet2")]
        ex = ck.get("engine_resume")
        e = self.engine
        n_a, n_c = e.param_count(_lib.ACTOR), e.param_count(_lib.CRITICS)
        if ex:   # validate before touching the engine: a blob from other dims / layer_norm setting must not reach the C side
            for name, n in (("actor", n_a), ("critics", n_c), ("log_alpha", 1)):
                st = ex[f"adam/{name}"]
                if st["exp_avg"].numel() != n or st["exp_avg_sq"].numel() != n:
                    raise ValueError(f"engine_resume: adam/{name} holds {st['exp_avg'].numel()} values, this agent needs {n}")
            if ex["actor_target"].numel() != n_a or ex["critics_target"].numel() != n_c:
                raise ValueError("engine_resume: target networks of another shape")
        self.load_flat(actor, np.concatenate(q), also_targets=not ex)
        if ex:
            e.set_params(_lib.ACTOR_TARGET, ex["actor_target"].numpy())
            e.set_params(_lib.CRITICS_TARGET, ex["critics_target"].numpy())
            e.set_params(_lib.LOG_ALPHA, np.array([ex["log_alpha"]], np.float32))
            for name, which in (("actor", _lib.ACTOR), ("critics", _lib.CRITICS), ("log_alpha", _lib.LOG_ALPHA)):
                st = ex[f"adam/{name}"]
                e.set_adam_state(which, st["exp_avg"].numpy(), st["exp_avg_sq"].numpy(), int(st["step"]))
        else:    # the reference's own keys only (agents/agent.py:368-369): optimiser moments and step counts
            for key, which, stacked in (("actor_optimizer", _lib.ACTOR, 0), ("q_optimizer", _lib.CRITICS, 2)):
                got = self._adam_from_state_dict(ck[key], stacked) if key in ck else None
                if got is not None:
                    e.set_adam_state(which, *got)
