"""Load the *real* reference (cinemere/evacuation) from /root/reference -- build container only.

This module is used ONLY by tests/golden/make_golden.py (fixture generation) and by
tests/test_oracle_vs_reference.py (skipped when /root/reference is absent, e.g. on the
GPU box).  Nothing in the product (`evacuation_amd/`), in `bench.py` or in
`__graft_entry__.py` imports it.

Two routes, so that the arithmetic pinned by the fixtures never runs through code we wrote:

1. ``load_core()`` -- stub-free.  ``src/env/env/{area,pedestrians,statuses,distances,reward}.py``
   and ``src/env/constants.py`` only import numpy/scipy, but ``src/env/__init__.py`` pulls in
   ``gymnasium``/``wandb`` (absent in this image, no network).  We register *empty namespace
   packages* for ``src``, ``src.env`` and ``src.env.env`` whose ``__path__`` points at the
   reference directories, so the normal import machinery loads the reference's own files
   without executing the package ``__init__`` that needs gymnasium.  No reference symbol is
   replaced.

2. ``load_full()`` -- the whole ``setup_env()`` surface.  ``gymnasium`` and ``wandb`` are
   replaced by base-class *shells*: ``gymnasium.Env`` (a ``reset`` that does nothing and an
   ``unwrapped`` property), ``gymnasium.ObservationWrapper`` (forwards ``reset``/``step``
   through ``observation()``), ``spaces.Box`` (holds low/high/shape/dtype) and ``spaces.Dict``
   (a dict), ``wandb.log`` (no-op).  None of these carries arithmetic of the hot path; what
   they stand in for is control flow of third-party packages the reference does not vendor.
   Route 1 is used to cross-check that route 2 yields the same dynamics bit-for-bit.
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("EVAC_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "env", "env", "area.py"))


def _purge():
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        del sys.modules[name]


def load_core():
    """Stub-free import of the reference dynamics core.  Returns a namespace with
    area, pedestrians, statuses, distances, reward, constants modules."""
    _purge()
    for m in ("gymnasium", "gymnasium.spaces", "wandb"):
        sys.modules.pop(m, None)
    pk_src = types.ModuleType("src")
    pk_src.__path__ = [os.path.join(REFERENCE_ROOT, "src")]
    pk_env = types.ModuleType("src.env")
    pk_env.__path__ = [os.path.join(REFERENCE_ROOT, "src", "env")]
    pk_envenv = types.ModuleType("src.env.env")
    pk_envenv.__path__ = [os.path.join(REFERENCE_ROOT, "src", "env", "env")]
    sys.modules.update({"src": pk_src, "src.env": pk_env, "src.env.env": pk_envenv})
    ns = types.SimpleNamespace()
    ns.constants = importlib.import_module("src.env.constants")
    ns.distances = importlib.import_module("src.env.env.distances")
    ns.statuses = importlib.import_module("src.env.env.statuses")
    ns.reward = importlib.import_module("src.env.env.reward")
    ns.pedestrians = importlib.import_module("src.env.env.pedestrians")
    ns.area = importlib.import_module("src.env.env.area")
    return ns


def _install_shells():
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:
        def reset(self, seed=None, options=None):
            return None

        @property
        def unwrapped(self):
            return self

    class ObservationWrapper(Env):
        def __init__(self, env):
            self.env = env
            self.observation_space = env.observation_space
            self.action_space = env.action_space

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def reset(self, **kw):
            o, i = self.env.reset(**kw)
            return self.observation(o), i

        def step(self, a):
            o, r, t, tr, i = self.env.step(a)
            return self.observation(o), r, t, tr, i

    class Box:
        def __init__(self, low, high, shape, dtype):
            self.low = np.full(shape, low, dtype)
            self.high = np.full(shape, high, dtype)
            self.shape = shape
            self.dtype = dtype

    class Dict(dict):
        pass

    spaces.Box, spaces.Dict = Box, Dict
    gym.Env, gym.ObservationWrapper, gym.spaces = Env, ObservationWrapper, spaces
    wandb = types.ModuleType("wandb")
    wandb.log = lambda *a, **k: None
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "wandb": wandb})


def load_full():
    """Import the reference's public surface (setup_env, EnvConfig, EnvWrappersConfig, Status)."""
    _purge()
    _install_shells()
    import matplotlib

    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    env_pkg = importlib.import_module("src.env")
    return env_pkg


_LOGDIR = None


def log_dir() -> str:
    global _LOGDIR
    if _LOGDIR is None:
        _LOGDIR = tempfile.mkdtemp(prefix="evac_ref_logs_")
    return _LOGDIR


def status_codes(statuses) -> np.ndarray:
    """Enum object array -> int8 codes (VISCEK=1, FOLLOWER=2, EXITING=3, ESCAPED=4)."""
    return np.array([s.value for s in statuses], dtype=np.int8)
