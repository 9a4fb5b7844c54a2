"""Philox4x32-10 restated in NumPy -- TEST INFRASTRUCTURE (see oracle/evac_oracle.py header).

The reference draws from NumPy's global MT19937 (pedestrians.py:17-18, area.py:124) with
data-dependent draw counts; that stream cannot be reproduced by thousands of parallel envs.  The
product therefore uses counter-based Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11; the Random123 library) keyed by (seed) with counter
(global_env_id, pedestrian, time, stream).  This file restates the generator and the three
stream definitions of evacuation_amd/csrc/evac_device.h so the tests can check the device draws
bit-exactly.  ``philox4x32_10`` itself is pinned by the Random123 known-answer vectors in
tests/test_philox.py."""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
STREAM_NOISE, STREAM_RESET, STREAM_ACTION = 0x4E4F4953, 0x52455345, 0x41435449
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised over the counter words (uint32 arrays of one shape); key words are scalars."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u01(x) -> np.ndarray:
    """24-bit uniform in [0,1), exact in f32 (device: u01)."""
    return (np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def usym(x) -> np.ndarray:
    """U[-1,1), exact in f32 (device: usym)."""
    return np.float32(2.0) * u01(x) - np.float32(1.0)


def _key(seed: int):
    return seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF


def reset_draws(seed: int, env_gid, n_ped: int, n_resets) -> np.ndarray:
    """[E,N,4] U(-1,1) reset draws (pos.x,pos.y,dir.x,dir.y) of envs ``env_gid`` at reset index
    ``n_resets`` (device: philox_reset_draw)."""
    env_gid = np.atleast_1d(np.asarray(env_gid, dtype=np.uint64))
    n_resets = np.broadcast_to(np.atleast_1d(np.asarray(n_resets, dtype=np.uint64)), env_gid.shape)
    i = np.arange(n_ped, dtype=np.uint64)[None, :]
    r = philox4x32_10(env_gid[:, None], i, n_resets[:, None], STREAM_RESET, *_key(seed))
    return np.stack([usym(w) for w in r], axis=-1)


def step_noise(seed: int, env_gid, n_ped: int, total, noise_coef: float) -> np.ndarray:
    """[E,N] angular noise of the step taken when the env's step counter is ``total`` (device:
    philox_noise): word (total & 3) of the block with counter (gid, ped, total >> 2, NOISE)."""
    env_gid = np.atleast_1d(np.asarray(env_gid, dtype=np.uint64))
    total = np.broadcast_to(np.atleast_1d(np.asarray(total, dtype=np.uint64)), env_gid.shape)
    i = np.arange(n_ped, dtype=np.uint64)[None, :]
    r = philox4x32_10(env_gid[:, None], i, (total >> np.uint64(2))[:, None], STREAM_NOISE, *_key(seed))
    sel = (total & np.uint64(3)).astype(np.int64)[:, None]
    w = np.choose(np.broadcast_to(sel, r[0].shape), r)
    return (u01(w) - np.float32(0.5)) * np.float32(noise_coef)


def random_action(seed: int, env_gid, total) -> np.ndarray:
    """[E,2] RandomAgent action U(-1,1)^2 (device: philox_action)."""
    env_gid = np.atleast_1d(np.asarray(env_gid, dtype=np.uint64))
    total = np.broadcast_to(np.atleast_1d(np.asarray(total, dtype=np.uint64)), env_gid.shape)
    r = philox4x32_10(env_gid, 0, total, STREAM_ACTION, *_key(seed))
    return np.stack([usym(r[0]), usym(r[1])], axis=-1)
