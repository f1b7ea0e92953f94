"""Key-stream helper of the synthetic workloads (numpy, seeded): Zipf(theta) over a key space
(BASELINE.json: Zipf-0.8).  The closed-loop clients themselves live in dint_amd/driver.py.
"""
from __future__ import annotations

import numpy as np


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
