"""Gymnasium spaces when gymnasium is installed, minimal stand-ins otherwise.

The reference declares ``spaces.Box`` / ``spaces.Discrete`` (env.py:89-95); gymnasium is an
optional dependency here because the hot path never touches it.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Discrete  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class Discrete:  # type: ignore[no-redef]
        def __init__(self, n: int, seed=None):
            self.n = int(n)
            self._rng = np.random.default_rng(seed)

        def sample(self) -> int:
            return int(self._rng.integers(self.n))

        def contains(self, x) -> bool:
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"
