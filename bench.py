#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused evacuation step on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], weak scaling): n=60 pedestrians x 4096 envs PER GPU, gravity
observation (alpha=3), RandomAgent actions drawn on device (Philox), episodes of 2000 steps with
same-step autoreset.  One "step" = one env step of every env of the batch: leader move, Vicsek
update, statuses, rewards, flags, observation, autoreset -- all written to HBM every step.
Steps are issued as `--inner` steps per kernel launch (evac_rollout: the state stays in registers
between the steps of a launch; every step still reads its actions from / writes its outputs to
HBM).  `--mode step` times one evac_step launch per step instead (reported as `step_api` anyway).
With N > 1 ranks each rank owns 4096 envs (global env ids rank*4096..) and the packed
[obs|reward|flags] chunk is all-gathered over RCCL/xGMI on a side stream inside the timed region.

Rank 0 prints ONE JSON line (see the task contract); `roofline` is computed from the ALGORITHMIC
bytes (32N + 38 + 4*D per env-step) and the kernel's mean launch duration measured
with HIP events on the launching stream; `cpu_baseline` times the NumPy oracle (a port of the
reference's step) on one host core for a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy peak)
VALU_LANE_OPS_PEAK = 78.6e12    # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz

WORKLOADS = {
    # name: (n_ped, envs_per_gpu, wrapper kwargs, description)
    "c2": (60, 4096, dict(positions="grav", alpha=3), "C2: n=60 pedestrians x 4096 envs per GPU, gravity obs (alpha=3)"),
    "c3": (256, 1024, dict(positions="grav", alpha=3), "C3: n=256 pedestrians x 1024 envs per GPU, gravity obs (alpha=3)"),
    "c5": (1024, 32, dict(positions="rel", statuses="ohe", type="Box"), "C5: n=1024 pedestrians x 32 envs per GPU, rel-pos Box obs + ohe statuses"),
    "big": (60, 524288, dict(positions="grav", alpha=3), "roofline evidence: n=60 x 524288 envs (state 503 MB > 256 MiB Infinity Cache)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--inner", type=int, default=100, help="env steps per kernel launch (rollout mode)")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of the outputs (N>1)")
    ap.add_argument("--gather", default="obs", choices=["obs", "slab"],
                    help="what the ranks all-gather per chunk: the observation batch (north_star) or the whole packed record")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="worker processes for the many-core CPU figure (-1: min(32, cores); 0/1: skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    return ap.parse_args()


def cpu_baseline(n_ped: int, seconds: float, procs: int):
    """The NumPy oracle (a port of the reference's EvacuationEnv.step + GravityEncoding) stepped in a
    single-env RandomAgent loop on one host core, as the reference's README loop does; plus, as
    `many_core`, the same loop in `procs` independent worker processes."""
    from oracle import cpu_bench

    n, dt = cpu_bench.readme_loop(n_ped, seconds)
    out = {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} single-env steps (n={n_ped}, gravity obs, RandomAgent loop) of the NumPy oracle in {dt:.1f} s "
                     f"on 1 of {os.cpu_count()} host cores",
           "agent_updates_per_s": n * n_ped / dt}
    if procs > 1:
        try:
            steps, busy, wall = cpu_bench.many_core(n_ped, min(seconds, 6.0), procs)
            out["many_core"] = {"value": steps / busy, "unit": "env-steps/s", "cores": procs,
                                "sample": f"{steps} steps by {procs} independent single-env worker processes, {busy:.1f} s each "
                                          f"(wall {wall:.1f} s incl. process start-up) of {os.cpu_count()} host cores"}
        except Exception as exc:  # noqa: BLE001
            out["many_core"] = {"error": f"{type(exc).__name__}: {exc}"[:160]}
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # CPU baseline first (rank 0, N=1 only): its worker processes are spawned before this process has
    # touched the GPU, and the GPU measurement below runs on an otherwise idle host.
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        procs = min(32, os.cpu_count() or 1) if args.cpu_procs < 0 else args.cpu_procs
        cpu_base = cpu_baseline(WORKLOADS[args.workload][0], args.cpu_seconds, procs)

    import torch
    import torch.distributed as dist

    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; evacuation_amd has no CPU path")
    dev_index = int(os.environ.get("EVAC_BENCH_FORCE_DEVICE", local_rank))   # testing aid: several ranks on one GPU
    torch.cuda.set_device(dev_index)
    device = torch.device(f"cuda:{dev_index}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EVAC_BENCH_BACKEND", "nccl")                 # "nccl" IS RCCL on ROCm; gloo = testing aid
        import datetime
        limit = datetime.timedelta(seconds=300)       # a stuck collective must surface as an error, not hang the run
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=limit)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)

    import evacuation_amd as ea
    from evacuation_amd.distributed import ShardedEvacuationEnv

    n_ped, envs_per_gpu, wrap_kw, desc = WORKLOADS[args.workload]
    if args.envs:
        envs_per_gpu = args.envs
    total_envs = envs_per_gpu * world
    cfg = ea.EnvConfig(number_of_pedestrians=n_ped, is_new_exiting_reward=True, is_new_followers_reward=True,
                       intrinsic_reward_coef=0.0, max_timesteps=2000)          # SURVEY.md 8(d) synthetic inputs
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    seed = 0x5EED0000 + sorted(WORKLOADS).index(args.workload)
    env = ShardedEvacuationEnv(cfg, wrap, total_envs=total_envs, device=device, seed=seed)
    loc = env.local
    E, D = loc.num_envs, loc.obs_dim
    env.reset()
    K, W = args.steps, args.warmup
    inner = max(1, min(args.inner, K)) if args.mode == "rollout" else 1
    do_gather = world > 1 and not args.no_gather

    # preallocated, reused output chunks (the trainer's rollout buffer, rpo_agent.py:158-163): the kernel
    # writes one packed slab [T, E, D+3] = [obs | reward | terminated | truncated], which is also the
    # all-gather message
    def alloc(T):
        return {"slab": torch.empty((T, E, D + 3), dtype=torch.float32, device=device),
                "episode_stats": torch.zeros((T, E, 8), dtype=torch.float32, device=device)}
    bufs = [alloc(inner), alloc(inner)]
    tails = {}
    GW = D if args.gather == "obs" else D + 3                 # gathered words per env-step
    gathered = [torch.empty((world, inner, E, GW), dtype=torch.float32, device=device) for _ in range(2)] if do_gather else None
    gsrc = [torch.empty((inner, E, GW), dtype=torch.float32, device=device) for _ in range(2)] if (do_gather and args.gather == "obs") else None

    def gather_message(b, k):
        """The contiguous tensor this rank contributes: the slab itself, or its observation columns copied out
        (on the stream the caller is in -- the comm stream, off the compute stream's critical path)."""
        if args.gather == "slab":
            return b["slab"]
        gsrc[k & 1].copy_(b["slab"][..., :D])
        return gsrc[k & 1]
    comm = torch.cuda.Stream(device=device) if do_gather else None
    step_actions = torch.rand((E, 2), device=device) * 2 - 1

    from evacuation_amd.distributed import all_gather_envs, pack_outputs

    def run(n_steps, events=None):
        """Issue exactly n_steps env steps; returns the number of kernel launches."""
        done = 0
        k = 0
        pend = [None, None]
        while done < n_steps:
            t = min(inner, n_steps - done)
            b = bufs[k & 1]
            if pend[k & 1] is not None:                       # buffer reuse: its gather must be finished
                torch.cuda.current_stream().wait_event(pend[k & 1])
                pend[k & 1] = None
            if events is not None:
                ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            if args.mode == "rollout":
                if t != inner:
                    b = tails.setdefault(t, alloc(t))
                loc.rollout(t, out=b)
            else:
                loc.step(step_actions)
            if events is not None:
                ev1.record()
                events.append((ev0, ev1, t))
            if do_gather and args.mode == "rollout" and t == inner:
                ready = torch.cuda.Event(); ready.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(ready)
                    all_gather_envs(gather_message(b, k), out=gathered[k & 1])
                    fin = torch.cuda.Event(); fin.record(comm)
                pend[k & 1] = fin
            elif do_gather:
                if args.mode == "step":
                    msg = loc.obs if args.gather == "obs" else pack_outputs(loc.obs, loc.reward, loc.terminated, loc.truncated)
                else:
                    msg = b["slab"][..., :D].contiguous() if args.gather == "obs" else b["slab"]
                all_gather_envs(msg)
            done += t
            k += 1
        for p in pend:
            if p is not None:
                torch.cuda.current_stream().wait_event(p)
        return k

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Exercise the collective once before anything is timed.  If RCCL cannot gather on this node the
    # benchmark degrades to independent shards (and says so) instead of dying without a number.
    gather_note = None
    if do_gather:
        ok = torch.ones(1, device=device)
        try:
            all_gather_envs(gather_message(bufs[0], 0), out=gathered[0])
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            ok.zero_()
            gather_note = f"all-gather disabled: {type(exc).__name__}: {exc}"[:200]
        try:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() == 0:
                do_gather = False
        except Exception as exc:  # noqa: BLE001
            do_gather = False
            gather_note = gather_note or f"all-reduce failed: {type(exc).__name__}"
        if not do_gather and gather_note is None:
            gather_note = "all-gather disabled: failed on another rank"

    run(W)                                                    # untimed warm-up
    barrier()
    events = []
    t0 = time.perf_counter()
    launches = run(K, events)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # mean launch duration of the dominant kernel, from HIP events on the launching stream
    full = [(a.elapsed_time(b) * 1e-3, t) for a, b, t in events if t == inner]
    kernel_s = sum(d for d, _ in full) / max(1, len(full))
    bytes_per_env_step = loc.algorithmic_bytes_per_env_step
    bytes_per_launch = bytes_per_env_step * E * inner
    achieved = bytes_per_launch / kernel_s / 1e9
    lane_ops_per_env_step = 10.0 * n_ped * n_ped + 40.0 * n_ped          # SURVEY.md 8(d) op model

    # the per-step API (one evac_step launch per step, actions resident in HBM), for transparency
    step_api = None
    if rank == 0 and args.mode == "rollout":
        for _ in range(50):
            loc.step(step_actions)
        torch.cuda.synchronize()
        n_api = 500
        t1 = time.perf_counter()
        for _ in range(n_api):
            loc.step(step_actions)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        step_api = {"env_steps_per_s": E * n_api / dt, "us_per_step": dt / n_api * 1e6,
                    "note": "one evac_step launch per step from Python/ctypes, single GPU, no gather"}
        try:    # the same step captured once into a hipGraph and replayed (what a graph-captured trainer loop pays)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(10):
                        loc.step(step_actions)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_api // 10):
                graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            step_api["hipgraph_us_per_step"] = dt / (n_api // 10 * 10) * 1e6
            step_api["hipgraph_env_steps_per_s"] = E * (n_api // 10 * 10) / dt
        except Exception as exc:  # noqa: BLE001
            step_api["hipgraph_error"] = f"{type(exc).__name__}: {exc}"[:160]

    if rank == 0:
        traffic = None
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            ent = tj.get(f"{args.workload}:{args.mode}:{inner}:{E}")
            if ent:
                traffic = ent["hbm_bytes_per_launch"]
        except Exception:  # noqa: BLE001
            pass
        value = total_envs * K / elapsed
        out = {
            "metric": "env-steps/s (agent-updates/s) at n=60x4096 envs" if args.workload == "c2" else f"env-steps/s ({args.workload})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "n_pedestrians": n_ped, "envs_per_gpu": E, "total_envs": total_envs,
                       "obs": wrap_kw, "actions": "RandomAgent U(-1,1)^2 drawn on device (Philox4x32-10)",
                       "mode": args.mode, "steps_per_launch": inner, "launches": launches,
                       "parallelism": f"env-sharded x{world}" + (f", RCCL all-gather of the {'observation batch' if args.gather == 'obs' else '[obs|reward|flags] records'} per {inner}-step chunk, overlapped on a side stream" if do_gather else ""),
                       "gather_note": gather_note,
                       "max_timesteps": 2000, "autoreset": True},
            "agent_updates_per_s": value * n_ped,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "frac_of_measured_copy_peak": achieved / 6290.0, "traffic": traffic,
                         "kernel": f"k_rollout<{1 if n_ped <= 64 else 4 if n_ped <= 256 else 8 if n_ped <= 512 else 16}>" if args.mode == "rollout" else "k_step",
                         "kernel_ms_per_launch": kernel_s * 1e3, "algorithmic_bytes_per_env_step": bytes_per_env_step,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu_frac": lane_ops_per_env_step * E * inner / kernel_s / VALU_LANE_OPS_PEAK,
                         "note": "all-pairs O(N^2) work makes this kernel VALU-bound; valu_frac uses the 10*N^2+40*N lane-op model"},
            "step_api": step_api,
        }
        out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
