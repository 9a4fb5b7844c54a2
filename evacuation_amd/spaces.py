"""Observation / action spaces.  gymnasium is optional: when it is importable its Box/Dict are used
(so gym wrappers accept our envs), otherwise these minimal stand-ins with the same attributes."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Dict  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32, seed=None):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(dict):
        """Keys are kept sorted, like gymnasium.spaces.Dict built from a plain dict."""

        def __init__(self, spaces=None):
            super().__init__(sorted((spaces or {}).items()))

        @property
        def spaces(self):
            return self

        def sample(self):
            return {k: s.sample() for k, s in self.items()}
