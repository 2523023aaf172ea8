"""Object wrapper over the C ABI (include/sactd3.h): numpy in, numpy out, exceptions for error codes."""
from __future__ import annotations

import ctypes as C
from dataclasses import asdict, dataclass
from typing import Dict, Optional

import numpy as np

from . import _lib
from ._lib import EngineError

_F = C.POINTER(C.c_float)


def _f32(x, shape=None) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if shape is not None:
        arr = arr.reshape(shape)
    return arr


def _fp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_F)


@dataclass
class Config:
    """Mirror of `sactd3_config`; defaults = tasks/defaults/sac.yml of the reference."""
    ob_dim: int = 0
    ac_dim: int = 0
    batch_size: int = 256
    rb_capacity: int = 1_000_000
    max_envs: int = 4
    prefer_td3_over_sac: bool = False
    layer_norm: bool = True
    autotune: bool = True
    bcq_style_targ_mix: bool = False
    targ_actor_smoothing: bool = True
    actor_update_delay: int = 2
    crit_targ_update_freq: int = 1
    use_graphs: bool = True
    device_id: int = 0
    actor_lr: float = 3e-4
    qnets_lr: float = 1e-3
    log_alpha_lr: float = 1e-3
    gamma: float = 0.99
    polyak: float = 0.005
    alpha_init: float = 0.2
    clip_norm: float = 0.0
    td3_std: float = 0.2
    td3_c: float = 0.5
    actor_noise_std: float = 0.1
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_eps: float = 1e-8
    seed: int = 0

    @staticmethod
    def from_hps(hps, ob_dim: int, ac_dim: int, **over) -> "Config":
        """Fill from any attribute- or key-style cfg (OmegaConf DictConfig, SimpleNamespace, dict, the
        oracle's Hps): only the keys the hot path reads, with the reference's branch-specific absences
        (td3.yml has no alpha_*/crit_targ_update_freq; sac.yml has no td3_*), agents/agent.py:47-139."""
        def get(k, default):
            if isinstance(hps, dict):
                return hps.get(k, default)
            try:
                v = getattr(hps, k)
            except Exception:
                return default
            return default if v is None else v
        c = Config(ob_dim=ob_dim, ac_dim=ac_dim)
        for k in ("batch_size", "rb_capacity", "prefer_td3_over_sac", "layer_norm", "autotune", "bcq_style_targ_mix",
                  "targ_actor_smoothing", "actor_update_delay", "crit_targ_update_freq", "actor_lr", "qnets_lr",
                  "log_alpha_lr", "gamma", "polyak", "alpha_init", "clip_norm", "td3_std", "td3_c", "actor_noise_std",
                  "seed"):
            setattr(c, k, type(getattr(c, k))(get(k, getattr(c, k))))
        c.max_envs = max(int(get("num_envs", 4)), 1)
        # NOT hps.cudagraphs: INTEGRATION.md runs the reference loop with `cudagraphs: false` so that orchestrator.py:313-315
        # does not wrap these methods in CudaGraphModule; the engine's own hipGraphs stay on unless `use_graphs=False` is passed
        for k, v in over.items():
            setattr(c, k, v)
        return c

    def to_c(self) -> _lib.CConfig:
        cc = _lib.CConfig()
        for k, v in asdict(self).items():
            setattr(cc, k, int(v) if isinstance(v, (bool, int)) else v)
        cc.abi_version = _lib.ABI_VERSION
        return cc


class Engine:
    """One learner on one MI355X.  All update calls are asynchronous on the engine's HIP stream."""

    def __init__(self, cfg: Config, min_ac, max_ac):
        self.lib = _lib.load_library()
        self.cfg = cfg
        self._h = C.c_void_p()
        lo = _f32(np.broadcast_to(np.asarray(min_ac, np.float32).reshape(-1), (cfg.ac_dim,)))
        hi = _f32(np.broadcast_to(np.asarray(max_ac, np.float32).reshape(-1), (cfg.ac_dim,)))
        cc = cfg.to_c()
        rc = self.lib.sactd3_create(C.byref(cc), _fp(lo), _fp(hi), C.byref(self._h))
        if rc != 0:
            msg = self.lib.sactd3_last_error(None)
            self._h = C.c_void_p()
            raise EngineError(f"sactd3_create failed ({rc}): {msg.decode() if msg else '?'}")
        # host-side staging for the per-env-step calls (rb_extend of num_envs rows, predict): numpy -> ctypes pointer conversion
        # is ~1-2 us per array; these arrays' pointers are made once
        n0, o, a = max(int(cfg.max_envs), 1), cfg.ob_dim, cfg.ac_dim
        self._st_n = n0
        self._st = [np.zeros((n0, o), np.float32), np.zeros((n0, a), np.float32), np.zeros(n0, np.float32),
                    np.zeros((n0, o), np.float32), np.zeros(n0, np.uint8), np.zeros((n0, o), np.float32), np.zeros((n0, a), np.float32)]
        self._st_p = [x.ctypes.data_as(C.POINTER(C.c_uint8) if x.dtype == np.uint8 else _F) for x in self._st]

    # -- plumbing
    def _ck(self, rc):
        if rc < 0:
            msg = self.lib.sactd3_last_error(self._h)
            raise EngineError(f"libsactd3_hip error {rc}: {msg.decode() if msg else '?'}")
        return rc

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.sactd3_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters
    def param_count(self, which: int) -> int:
        return int(self._ck(self.lib.sactd3_param_count(self._h, which)))

    def get_params(self, which: int) -> np.ndarray:
        out = np.empty(self.param_count(which), np.float32)
        self._ck(self.lib.sactd3_get_params(self._h, which, _fp(out)))
        return out

    def set_params(self, which: int, flat) -> None:
        flat = _f32(flat).reshape(-1)
        if flat.size != self.param_count(which):
            raise ValueError(f"expected {self.param_count(which)} floats, got {flat.size}")
        self._ck(self.lib.sactd3_set_params(self._h, which, _fp(flat)))

    def get_adam_state(self, which: int):
        n = self.param_count(which)
        m, v, step = np.empty(n, np.float32), np.empty(n, np.float32), C.c_int64(0)
        self._ck(self.lib.sactd3_get_adam_state(self._h, which, _fp(m), _fp(v), C.byref(step)))
        return m, v, int(step.value)

    def set_adam_state(self, which: int, m, v, step: int) -> None:
        m, v = _f32(m).reshape(-1), _f32(v).reshape(-1)
        n = self.param_count(which)
        if m.size != n or v.size != n:
            raise ValueError(f"expected {n} floats of exp_avg / exp_avg_sq, got {m.size} / {v.size}")
        self._ck(self.lib.sactd3_set_adam_state(self._h, which, _fp(m), _fp(v), int(step)))

    # -- replay
    def _rows(self, obs, act, rew, nobs, done):
        o, a = self.cfg.ob_dim, self.cfg.ac_dim
        obs, nobs = _f32(obs).reshape(-1, o), _f32(nobs).reshape(-1, o)
        n = obs.shape[0]
        act, rew = _f32(act).reshape(n, a), _f32(rew).reshape(n)
        done = np.ascontiguousarray(np.asarray(done).reshape(n) != 0, dtype=np.uint8)
        assert nobs.shape[0] == n
        return obs, act, rew, nobs, done, n

    def rb_extend(self, obs, act, rew, nobs, done) -> None:
        obs = np.asarray(obs)
        n = obs.size // self.cfg.ob_dim
        if 0 < n <= self._st_n and obs.size == n * self.cfg.ob_dim:      # an env step's rows: through the pre-bound staging arrays
            st, p = self._st, self._st_p
            st[0][:n] = obs.reshape(n, -1)
            st[1][:n] = np.asarray(act).reshape(n, -1)
            st[2][:n] = np.asarray(rew).reshape(n)
            st[3][:n] = np.asarray(nobs).reshape(n, -1)
            st[4][:n] = np.asarray(done).reshape(n) != 0
            self._ck(self.lib.sactd3_rb_extend(self._h, p[0], p[1], p[2], p[3], p[4], n))
            return
        obs, act, rew, nobs, done, n = self._rows(obs, act, rew, nobs, done)
        self._ck(self.lib.sactd3_rb_extend(self._h, _fp(obs), _fp(act), _fp(rew), _fp(nobs),
                                           done.ctypes.data_as(C.POINTER(C.c_uint8)), n))

    def rb_layout(self) -> Dict[str, int]:
        out = (C.c_int32 * 4)()
        self._ck(self.lib.sactd3_rb_layout(self._h, out))
        return dict(record_floats=int(out[0]), next_obs_offset=int(out[1]), next_obs_width=int(out[2]), capacity=int(out[3]))

    def pack_records(self, obs, act, rew, nobs, done) -> np.ndarray:
        """[n, record_floats] float32 in the ring's record layout (include/sactd3.h: sactd3_rb_layout)."""
        obs, act, rew, nobs, done, n = self._rows(obs, act, rew, nobs, done)
        lay, o, a = self.rb_layout(), self.cfg.ob_dim, self.cfg.ac_dim
        rec = np.zeros((n, lay["record_floats"]), np.float32)
        rec[:, :o], rec[:, o:o + a] = obs, act
        rec[:, lay["next_obs_offset"]:lay["next_obs_offset"] + o] = nobs
        tail = lay["next_obs_offset"] + lay["next_obs_width"]
        rec[:, tail], rec[:, tail + 1] = rew, done.astype(np.float32)
        return rec

    def rb_extend_device(self, device_ptr: int, n: int) -> None:
        """append n packed records that already live in device memory (kept alive by the caller until sync())."""
        self._ck(self.lib.sactd3_rb_extend_device(self._h, C.c_void_p(int(device_ptr)), int(n)))

    def rb_len(self) -> int:
        return int(self._ck(self.lib.sactd3_rb_len(self._h)))

    def rb_sample(self) -> None:
        self._ck(self.lib.sactd3_rb_sample(self._h))

    def rb_sample_with_indices(self, idx) -> None:
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64).reshape(-1))
        if idx.size != self.cfg.batch_size:
            raise ValueError(f"expected {self.cfg.batch_size} indices, got {idx.size}")
        self._ck(self.lib.sactd3_rb_sample_with_indices(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), idx.size))

    def load_batch(self, obs, act, rew, nobs, done) -> None:
        obs, act, rew, nobs, done, n = self._rows(obs, act, rew, nobs, done)
        self._ck(self.lib.sactd3_load_batch(self._h, _fp(obs), _fp(act), _fp(rew), _fp(nobs),
                                            done.ctypes.data_as(C.POINTER(C.c_uint8)), n))

    def read_batch(self) -> Dict[str, np.ndarray]:
        B, o, a = self.cfg.batch_size, self.cfg.ob_dim, self.cfg.ac_dim
        obs, act = np.empty((B, o), np.float32), np.empty((B, a), np.float32)
        rew, nobs = np.empty(B, np.float32), np.empty((B, o), np.float32)
        done, idx = np.empty(B, np.uint8), np.empty(B, np.int64)
        self._ck(self.lib.sactd3_read_batch(self._h, _fp(obs), _fp(act), _fp(rew), _fp(nobs),
                                            done.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            idx.ctypes.data_as(C.POINTER(C.c_int64))))
        return dict(observations=obs, actions=act, rewards=rew, next_observations=nobs, dones=done.astype(bool), index=idx)

    def rb_fill_synthetic(self, n: int, seed: int = 0) -> None:
        self._ck(self.lib.sactd3_rb_fill_synthetic(self._h, int(n), int(seed)))

    # -- noise
    def set_noise(self, site: int, eps) -> None:
        eps = _f32(eps).reshape(-1, self.cfg.ac_dim)
        self._ck(self.lib.sactd3_set_noise(self._h, site, _fp(eps), eps.shape[0]))

    def clear_noise(self, site: int = -1) -> None:
        self._ck(self.lib.sactd3_clear_noise(self._h, site))

    def read_noise(self, site: int, n: Optional[int] = None) -> np.ndarray:
        n = self.cfg.batch_size if n is None else n
        out = np.empty((n, self.cfg.ac_dim), np.float32)
        self._ck(self.lib.sactd3_read_noise(self._h, site, _fp(out), n))
        return out

    # -- updates
    def update_qnets(self) -> None:
        self._ck(self.lib.sactd3_update_qnets(self._h))

    def update_actor(self) -> None:
        self._ck(self.lib.sactd3_update_actor(self._h))

    def update_targ_nets(self, qnet_updates_so_far: int) -> None:
        self._ck(self.lib.sactd3_update_targ_nets(self._h, int(qnet_updates_so_far)))

    def step(self, do_actor: bool) -> None:
        self._ck(self.lib.sactd3_step(self._h, int(bool(do_actor))))

    def step_prefix(self, m: int) -> None:
        """the first m iterations of a period (the one with the actor updates + m - 1 critic-only ones) in one graph launch"""
        self._ck(self.lib.sactd3_step_prefix(self._h, int(m)))

    def step_period(self) -> None:
        """actor_update_delay + 1 iterations (actor updates in the first) as one graph launch."""
        self._ck(self.lib.sactd3_step_period(self._h))

    def instantiate_graphs(self) -> None:
        """capture + instantiate the step / period graphs now instead of at their first use (nothing is launched)."""
        self._ck(self.lib.sactd3_instantiate_graphs(self._h))

    def run_iterations(self, i0: int, n: int) -> int:
        """iterations i0 .. i0 + n - 1 of the loop (orchestrator.py:337-352 schedule: actor updates when i % (delay + 1) == 0),
        whole periods as one graph launch each, the rest one by one.  Returns i0 + n."""
        period = self.cfg.actor_update_delay + 1
        can = self.cfg.actor_update_delay > 0 and (self.cfg.prefer_td3_over_sac or self.cfg.crit_targ_update_freq == 1)
        i, end = i0, i0 + n
        while i < end:
            if can and i % period == 0 and i + period <= end:
                self.step_period()
                i += period
            elif can and i % period == 0:          # what is left behind the last whole period: its first end - i iterations, one launch
                self.step_prefix(end - i)
                i = end
            else:
                self.step(i % period == 0)
                i += 1
        return i

    def predict(self, obs, explore: bool) -> np.ndarray:
        obs = np.asarray(obs)
        n = obs.size // self.cfg.ob_dim
        if 0 < n <= self._st_n and obs.size == n * self.cfg.ob_dim:
            self._st[5][:n] = obs.reshape(n, -1)
            self._ck(self.lib.sactd3_predict(self._h, self._st_p[5], n, 1 if explore else 0, self._st_p[6]))
            return self._st[6][:n].copy()
        obs = _f32(obs).reshape(-1, self.cfg.ob_dim)
        out = np.empty((obs.shape[0], self.cfg.ac_dim), np.float32)
        self._ck(self.lib.sactd3_predict(self._h, _fp(obs), obs.shape[0], int(bool(explore)), _fp(out)))
        return out

    def read_metrics(self) -> Dict[str, float]:
        m = np.empty(_lib.NUM_METRICS, np.float32)
        self._ck(self.lib.sactd3_read_metrics(self._h, _fp(m)))
        return {"loss/qf_loss": float(m[0]), "loss/actor_loss": float(m[1]), "loss/alpha_loss": float(m[2]),
                "vitals/alpha": float(m[3])}

    def sync(self) -> None:
        self._ck(self.lib.sactd3_sync(self._h))

    def device_handles(self):
        """(hipStream_t of the engine, device address of the float32 metrics slots) as integers."""
        st, mp = C.c_void_p(), C.c_void_p()
        self._ck(self.lib.sactd3_device_handles(self._h, C.byref(st), C.byref(mp)))
        return int(st.value or 0), int(mp.value or 0)

    # -- introspection
    def debug_read(self, name: str) -> np.ndarray:
        n = int(self._ck(self.lib.sactd3_debug_read(self._h, name.encode(), None, 0)))
        out = np.empty(n, np.float32)
        self._ck(self.lib.sactd3_debug_read(self._h, name.encode(), _fp(out), n))
        return out

    def graph_kernel_count(self, which: int) -> int:
        return int(self._ck(self.lib.sactd3_graph_kernel_count(self._h, which)))

    def time_kernel(self, name: str, iters: int = 200) -> float:
        us = C.c_float(0)
        self._ck(self.lib.sactd3_time_kernel(self._h, name.encode(), iters, C.byref(us)))
        return float(us.value)

    def time_nodes(self, which, iters: int = 200):
        """[{name, us, flops, bytes, threads}] for every kernel node of: 0 / False a critic-only fused iteration, 1 / True one with
        the actor updates, 2 a whole period of the schedule as sactd3_step_period captures it (consumes the learner's state)."""
        cap = 128
        us, fl, by, th = np.zeros(cap, np.float32), np.zeros(cap, np.float64), np.zeros(cap, np.float64), np.zeros(cap, np.int64)
        names = C.create_string_buffer(cap * 128)
        dp = C.POINTER(C.c_double)
        n = self._ck(self.lib.sactd3_time_nodes(self._h, int(which), iters, cap, names, len(names), _fp(us),
                                                fl.ctypes.data_as(dp), by.ctypes.data_as(dp), th.ctypes.data_as(C.POINTER(C.c_int64))))
        labels = names.value.decode().split("\n")[:n]
        return [dict(name=labels[k], us=float(us[k]), flops=float(fl[k]), bytes=float(by[k]), threads=int(th[k])) for k in range(n)]

    def time_gather_sweep(self, batch: int, iters: int = 50):
        us, nbytes = C.c_float(0), C.c_double(0)
        self._ck(self.lib.sactd3_time_gather_sweep(self._h, batch, iters, C.byref(us), C.byref(nbytes)))
        return float(us.value), float(nbytes.value)
