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


def usable_cores() -> int:
    """The cores THIS process may use (the job's CPU share), not the host's core count: the affinity mask, capped by the
    cgroup CPU quota where one is set (cpu.max of cgroup v2 / cfs_quota_us of v1)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.999)))
    return n


_THREAD_VARS = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "VECLIB_MAXIMUM_THREADS")


def _worker(args):
    n_ped, seconds, seed = args
    return readme_loop(n_ped, seconds, seed)


def many_core(n_ped: int, seconds: float, procs: int):
    """`procs` independent envs, one worker process each (envs are independent, so this is how the
    reference would use many cores).  The BLAS / OpenMP thread pools are pinned to one thread in the PARENT's environment
    before the pool is created -- the spawned workers import numpy with it in place (setting it inside a worker, after
    numpy is imported, has no effect).  Returns (total_steps, busiest worker's seconds, wall_seconds)."""
    import multiprocessing as mp
    saved = {k: os.environ.get(k) for k in _THREAD_VARS}
    os.environ.update({k: "1" for k in _THREAD_VARS})
    try:
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(procs) as pool:
            res = pool.map(_worker, [(n_ped, seconds, 100 + k) for k in range(procs)], chunksize=1)
        wall = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    steps = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return steps, busy, wall


def many_core_report(n_ped: int, seconds: float, procs: int, single_core_rate: float) -> dict:
    """bench.py's `cpu_baseline.many_core` object.  `cores` is the number of workers actually run (one per usable core);
    `per_worker_vs_single_core` says whether they really had a core each (1.0 = yes; well below = the job's CPU share is
    smaller than its affinity mask suggests -- cgroup quotas do not show in sched_getaffinity).
    One worker per core, each stepping ONE env: the reference's own parallelism is exactly that (independent single-env
    processes, run_scripts/run.sh:65-68).  A batched [E, N] NumPy restatement split over the cores is not
    built: it would be a second oracle, unpinned against the reference, and at N = 60 NumPy's per-call overhead -- what the
    batching removes -- is not what the GPU path is compared on."""
    steps, busy, wall = many_core(n_ped, seconds, procs)
    rate = steps / busy
    return {"value": rate, "unit": "env-steps/s", "cores": procs,
            "per_worker_env_steps_per_s": rate / procs,
            "per_worker_vs_single_core": (rate / procs) / single_core_rate if single_core_rate > 0 else None,
            "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
            "sample": f"{steps} steps by {procs} independent single-env worker processes (1 BLAS/OpenMP thread each), {busy:.1f} s each "
                      f"(wall {wall:.1f} s incl. process start-up); {usable_cores()} usable of {os.cpu_count()} host cores"}
