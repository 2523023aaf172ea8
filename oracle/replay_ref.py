"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

numpy restatement of the integer/byte side of the path:

* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
  reference implementation is Random123 v1.x `philox.h`, a third-party algorithm that is NOT part of
  /root/reference).  Pinned by the known-answer vectors of Random123's `kat_vectors` (tests/test_oracle_replay.py).
* the engine's index draw / normal draw built on it (csrc/philox.h, kernels.h:philox_normal),
* the replay ring of torchrl's TensorDictReplayBuffer + LazyTensorStorage as the reference uses it
  (main.py:167-171, orchestrator.py:100-113,338,385): round-robin writes, uniform-with-replacement
  sampling over [0, len), per-key row gather.  torchrl's source is absent from this image, so these semantics
  come from its documentation and the reference's call sites -> PARITY UNPINNED for the sampler's stream
  (the engine documents its own Philox stream instead); the gather itself is pinned by being exact.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_INDEX, STREAM_FILL, STREAM_NOISE = 0x100, 0x200, 0x000


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """vectorised over equal-shaped uint32 arrays (counter words) and scalar keys -> 4 uint32 arrays"""
    c = [np.asarray(x, dtype=np.uint64) & MASK for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0 & MASK, p1 & MASK, n2 & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def sample_indices(seed: int, sample_ctr: int, batch: int, length: int) -> np.ndarray:
    """csrc/philox.h:philox_index for b = 0..batch-1"""
    b = np.arange(batch, dtype=np.uint64)
    r = philox4x32_10(np.full(batch, sample_ctr), 0, STREAM_INDEX, b >> np.uint64(2), seed & 0xFFFFFFFF, seed >> 32)
    word = np.choose((b & np.uint64(3)).astype(np.int64), r).astype(np.uint64)
    return ((word * np.uint64(length)) >> np.uint64(32)).astype(np.int64)


def _u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def normals(seed: int, ctr: int, site_code: int, n_elems: int) -> np.ndarray:
    """csrc/kernels.h:philox_normal for e = 0..n_elems-1 (float32 Box-Muller; compare with a small tolerance)"""
    e = np.arange(n_elems, dtype=np.uint64)
    r = philox4x32_10(np.full(n_elems, ctr), 0, STREAM_NOISE + site_code, e >> np.uint64(2), seed & 0xFFFFFFFF, seed >> 32)
    k = (e & np.uint64(3)).astype(np.int64)
    x0 = np.where(k < 2, r[0], r[2])
    x1 = np.where(k < 2, r[1], r[3])
    u1, u2 = _u01(x0), _u01(x1)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    th = (np.float32(6.283185307179586) * u2).astype(np.float32)
    return np.where(k & 1, rad * np.sin(th), rad * np.cos(th)).astype(np.float32)


class RingRef:
    """Round-robin ring with per-key storage, the way LazyTensorStorage keeps one tensor per key."""

    def __init__(self, capacity: int, ob_dim: int, ac_dim: int):
        self.cap, self.len, self.cursor = capacity, 0, 0
        self.obs = np.zeros((capacity, ob_dim), np.float32)
        self.nobs = np.zeros((capacity, ob_dim), np.float32)
        self.act = np.zeros((capacity, ac_dim), np.float32)
        self.rew = np.zeros(capacity, np.float32)
        self.done = np.zeros(capacity, bool)

    def extend(self, obs, act, rew, nobs, done):
        for i in range(len(obs)):
            c = self.cursor
            self.obs[c], self.act[c], self.rew[c], self.nobs[c], self.done[c] = obs[i], act[i], rew[i], nobs[i], done[i]
            self.cursor = (c + 1) % self.cap
            self.len = min(self.cap, self.len + 1)

    def gather(self, idx):
        idx = np.asarray(idx)
        assert (idx >= 0).all() and (idx < self.len).all()
        return dict(observations=self.obs[idx], actions=self.act[idx], rewards=self.rew[idx],
                    next_observations=self.nobs[idx], dones=self.done[idx], index=idx)
