"""Timed CPU loops of the oracle for bench.py's cpu_baseline leg -- TEST/BENCH INFRASTRUCTURE.
Kept free of torch so that worker processes spawned for the many-core figure stay light."""
from __future__ import annotations

import os
import time

import numpy as np

from oracle import evac_oracle as O


def readme_loop(n_ped: int, seconds: float, seed: int = 0):
    """The reference's README loop (README.md:69-91): one env, RandomAgent, gravity observation, reset when
    the episode ends.  Returns (steps, elapsed_seconds)."""
    p = O.OracleParams(number_of_pedestrians=n_ped, is_new_exiting_reward=True)
    rng = np.random.default_rng(seed)
    st = O.env_reset(p, rng.uniform(-1, 1, (n_ped, 2)), rng.uniform(-1, 1, (n_ped, 2)))
    for _ in range(50):
        O.env_step(p, st, rng.uniform(-1, 1, 2).astype(np.float32), rng.uniform(-0.1, 0.1, n_ped))
    n = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(200):
            a = rng.uniform(-1, 1, 2).astype(np.float32)
            out = O.env_step(p, st, a, O.draw_step_noise(p, st, rng))
            O.observe(st, "grav", alpha=3, eps=p.eps)
            if out["terminated"] or out["truncated"]:
                st = O.env_reset(p, rng.uniform(-1, 1, (n_ped, 2)), rng.uniform(-1, 1, (n_ped, 2)))
        n += 200
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return n, dt


def _worker(args):
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    n_ped, seconds, seed = args
    return readme_loop(n_ped, seconds, seed)


def many_core(n_ped: int, seconds: float, procs: int):
    """`procs` independent envs, one worker process each (envs are independent, so this is how the
    reference would use many cores).  Returns (total_steps, wall_seconds)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        res = pool.map(_worker, [(n_ped, seconds, 100 + k) for k in range(procs)])
    wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return steps, busy, wall
