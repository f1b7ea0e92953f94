"""Synthetic request-stream generators for the benchmark and the smoke paths (numpy, seeded).

These are *open-loop* streams shaped like the reference clients' traffic (the closed-loop
transaction drivers live in dint_amd/driver.py).  Keys follow Zipf(theta) over the key
space (BASELINE.json: Zipf-0.8) or the reference's own distributions.
"""
from __future__ import annotations

import numpy as np

from . import wire


class Zipf:
    """Zipf(theta) sampler over [0, n) by inverse-CDF approximation (Gray et al., SIGMOD'94),
    ranks scrambled by a multiplicative hash so hot keys are spread over the key space."""

    def __init__(self, n: int, theta: float, seed: int = 0):
        self.n, self.theta = int(n), float(theta)
        self.rng = np.random.default_rng(seed)
        if theta > 0:
            # zeta(n, theta) by Euler-Maclaurin for large n
            k = np.arange(1, min(self.n, 1_000_000) + 1, dtype=np.float64)
            z = (k ** -theta).sum()
            if self.n > 1_000_000:
                a, b = 1_000_000.0, float(self.n)
                z += (b ** (1 - theta) - a ** (1 - theta)) / (1 - theta) + 0.5 * (b ** -theta - a ** -theta)
            self.zetan = z
            self.zeta2 = 1.0 + 0.5 ** theta
            self.alpha = 1.0 / (1.0 - theta)
            self.eta = (1 - (2.0 / self.n) ** (1 - theta)) / (1 - self.zeta2 / self.zetan)

    def sample(self, size) -> np.ndarray:
        u = self.rng.random(size)
        if self.theta <= 0:
            return (u * self.n).astype(np.uint64)
        uz = u * self.zetan
        r = (self.n * (self.eta * u - self.eta + 1) ** self.alpha).astype(np.uint64)
        r = np.where(uz < 1.0, 0, np.where(uz < self.zeta2, 1, r))
        r = np.minimum(r, self.n - 1).astype(np.uint64)
        # scramble ranks -> keys (bijection on [0, n) is not needed: collisions only merge hot keys)
        return (r * np.uint64(0x9E3779B97F4A7C15) >> np.uint64(11)) % np.uint64(self.n)


def fasst_stream(n_req: int, key_space: int = 24_000_000, theta: float = 0.8, read_prop: float = 0.8,
                 seed: int = 0) -> np.ndarray:
    """FaSST-client-shaped stream (lock_fasst/caladan/client.cc:183-280 without retries):
    per transaction 5-10 keys: READ each, ACQUIRE_LOCK the write set (each key written with
    p = 1 - read_prop, trace_init.sh:21-23), re-READ every key to validate, COMMIT the write set."""
    rng = np.random.default_rng(seed)
    z = Zipf(key_space, theta, seed + 1)
    per_txn = 7.5 * 2 + 2 * 7.5 * (1 - read_prop)
    n_txn = int(n_req / per_txn) + 16
    nk = rng.integers(5, 11, n_txn)
    out = np.zeros(n_req + 64, wire.FASST_MSG)
    keys = z.sample(int(nk.sum())).astype("<u4")
    wr = rng.random(int(nk.sum())) >= read_prop
    pos = 0
    off = 0
    F = wire.Fasst
    for t in range(n_txn):
        k = np.sort(keys[off:off + nk[t]])
        w = k[wr[off:off + nk[t]]]
        off += nk[t]
        seq_t = np.concatenate([np.full(len(k), F.READ), np.full(len(w), F.ACQUIRE_LOCK),
                                np.full(len(k), F.READ), np.full(len(w), F.COMMIT)])
        seq_k = np.concatenate([k, w, k, w])
        m = len(seq_t)
        if pos + m > len(out):
            break
        out["type"][pos:pos + m] = seq_t
        out["lid"][pos:pos + m] = seq_k
        pos += m
        if pos >= n_req:
            break
    return out[:n_req]


def interleave(stream: np.ndarray, n_workers: int) -> np.ndarray:
    """Round-robin interleave `n_workers` contiguous slices of a stream, modelling that many
    concurrent closed-loop clients each with one outstanding request."""
    n = len(stream) // n_workers * n_workers
    return np.ascontiguousarray(stream[:n].reshape(n_workers, -1).T).reshape(-1)
