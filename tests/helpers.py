"""Shared helpers for the oracle / parity tests (test infrastructure)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

from oracle import evac_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def traj_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))


def load_params(js) -> O.OracleParams:
    return O.OracleParams(**json.loads(str(js)))


def state_at(d, k, dtype=np.float64) -> O.OracleState:
    return O.OracleState(np.array(d["pos"][k], dtype=dtype), np.array(d["dir"][k], dtype=dtype),
                         np.array(d["status"][k], dtype=np.int8), np.array(d["agent_pos"][k], dtype=np.float32),
                         np.array(d["agent_dir"][k], dtype=np.float32), int(d["now"][k]))


def episode_record_files():
    """Fixtures that end with the reference's own per-episode record (env.py:114-127, captured by make_golden.capture_episode_record):
    the trajectories that ran to truncation, and the crafted episode that terminates with every pedestrian escaped."""
    out = [f for f in traj_files() if "episode_record" in np.load(f).files]
    extra = os.path.join(GOLDEN, "episode_all_escaped.npz")
    return out + ([extra] if os.path.exists(extra) else [])


def crafted_cases():
    d = np.load(os.path.join(GOLDEN, "crafted.npz"))
    for name in d["names"]:
        name = str(name)
        yield name, {k[len(name) + 2:]: d[k] for k in d.files if k.startswith(name + "__")}


OBS_VARIANTS = [(p, s, t) for p in ("abs", "rel") for s in ("no", "ohe", "cat") for t in ("Box", "Dict")
                if not (p == "abs" and s == "no" and t == "Dict")]


def check_observations(get, st: O.OracleState, eps: float, tol=1e-12):
    """Compare every observation variant of the oracle with the fixture's (``get(key)``)."""
    for a in (2, 3, 5):
        o = O.observe(st, "grav", alpha=a, eps=eps)
        for k, v in o.items():
            ref = get(f"obs_grav_a{a}__{k}")
            np.testing.assert_allclose(v, ref, rtol=tol, atol=tol, equal_nan=True, err_msg=f"grav a={a} {k}")
            assert np.asarray(v).dtype == ref.dtype, (k, np.asarray(v).dtype, ref.dtype)
    for pos, stat, typ in OBS_VARIANTS:
        o = O.observe(st, pos, stat, typ)
        name = f"obs_{pos}_{stat}_{typ.lower()}"
        if typ == "Box":
            ref = get(name)
            np.testing.assert_allclose(o, ref, rtol=tol, atol=tol, equal_nan=True, err_msg=name)
            assert o.dtype == ref.dtype and o.shape == ref.shape, (name, o.dtype, ref.dtype)
        else:
            for k, v in o.items():
                ref = get(f"{name}__{k}")
                np.testing.assert_allclose(v, ref, rtol=tol, atol=tol, equal_nan=True, err_msg=f"{name} {k}")
                assert np.asarray(v).dtype == ref.dtype, (name, k, np.asarray(v).dtype, ref.dtype)
