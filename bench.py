#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused evacuation step on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` without torchrun starts the N ranks itself (child processes, one per GPU; the
parent never touches a GPU) and fails if fewer than N join.

Workload (BASELINE.json configs[1], weak scaling): n=60 pedestrians x 4096 envs PER GPU, gravity
observation (alpha=3), RandomAgent actions drawn on device (Philox), episodes of 2000 steps with
same-step autoreset.  One "step" = one env step of every env of the batch: leader move, Vicsek
update, statuses, rewards, flags, observation, autoreset -- all written to HBM every step.

What is timed.  The cost of a step depends on the episode phase (the all-pairs loop runs over the
pedestrians that still move: all N after a reset, a third of them late in the episode), so a single
K-step block measures whichever phase it lands on.  The timed region is therefore a sequence of BLOCKS of
exactly K steps, each bracketed by barrier + torch.cuda.synchronize() on both sides and timed on its own
(max over ranks); consecutive blocks tile whole episodes (R = 2000/K blocks per sweep, --sweeps sweeps, at
least 20 blocks).  `ms_per_step` = mean over episode phases of the per-phase median block time / K, i.e. the
EPISODE-AVERAGE cost; `value` = total envs / that.  `blocks` in the line also gives the median / min / max
block, the dense block (all N moving, right after the reset) and the mid-episode block.  Steps are issued as
launches of min(--inner, K) steps (evac_rollout: the state stays in registers between the steps of a launch,
every step still writes its outputs to HBM).

`roofline` follows the task contract: ALGORITHMIC bytes per launch (32N + 38 + 4D per
env-step, times the env-steps of one launch) divided by the mean duration of the timed launches, measured
with HIP events on the launching stream, against the 8 TB/s HBM peak.  The kernel keeps its state in
registers, so this is an equivalent-bandwidth figure; the resource that actually binds is the VALU
(`binding_resource`, `valu_frac`), and `traffic` is the HBM traffic the PMC counters see
(profiles/traffic.json, bytes per env-step times the env-steps of one launch).
`cpu_baseline` times the NumPy oracle (a port of the reference's step) on the host cores.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy peak)
VALU_LANE_OPS_PEAK = 78.6e12    # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
EPISODE = 2000                  # max_timesteps of the synthetic workload

WORKLOADS = {
    # name: (n_ped, envs_per_gpu, wrapper kwargs, description)
    "c2": (60, 4096, dict(positions="grav", alpha=3), "C2: n=60 pedestrians x 4096 envs per GPU, gravity obs (alpha=3)"),
    "c3": (256, 1024, dict(positions="grav", alpha=3), "C3: n=256 pedestrians x 1024 envs per GPU, gravity obs (alpha=3)"),
    "c5": (1024, 32, dict(positions="rel", statuses="ohe", type="Box"), "C5: n=1024 pedestrians x 32 envs per GPU, rel-pos Box obs + ohe statuses"),
    "big": (60, 524288, dict(positions="grav", alpha=3), "roofline evidence: n=60 x 524288 envs (state 503 MB > 256 MiB Infinity Cache)"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--inner", type=int, default=100, help="env steps per kernel launch (rollout mode)")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"])
    ap.add_argument("--sweeps", type=int, default=0, help="passes over the episode (0: 3, or 1 when a block is a whole episode)")
    ap.add_argument("--blocks", type=int, default=0, help="override the number of timed K-step blocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-api", action="store_true", help="skip the one-launch-per-step side measurement")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of the outputs (N>1)")
    ap.add_argument("--gather", default="obs", choices=["obs", "slab"],
                    help="what the ranks all-gather per chunk: the observation batch (north_star) or the whole packed record")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="worker processes for the many-core CPU figure (-1: all host cores; 0/1: skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / gather control flow only, on CPU tensors over gloo (tests; prints no throughput)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` starts its own N ranks
# --------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv) -> int:
    """Parent of a self-launched multi-rank run.  Starts `--gpus` child processes of this script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torch.distributed.run would set), relays their
    output and returns non-zero unless EVERY rank exits cleanly.  The parent initialises no GPU (children are
    fresh processes, never an exec of a process that has touched the device)."""
    n = args.gpus
    if not args.dry_run and "EVAC_BENCH_FORCE_DEVICE" not in os.environ:
        import torch                                    # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to report a {n}-GPU number", file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               EVAC_BENCH_SELF_LAUNCHED="1")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e))
    rc = 0
    deadline = time.time() + float(os.environ.get("EVAC_BENCH_LAUNCH_TIMEOUT", "1500"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:                           # a rank died: the others would hang in a collective
                    q.terminate()
        if time.time() > deadline:
            rc = rc or 3
            for q in alive:
                q.kill()
        time.sleep(0.05)
    if rc != 0:
        print(f"bench.py: a rank failed (exit code {rc}); no {n}-GPU result", file=sys.stderr)
    return rc


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


def block_plan(K: int, sweeps: int, blocks: int):
    """(blocks per sweep, sweeps): consecutive K-step blocks tile the episode."""
    per_sweep = max(1, math.ceil(EPISODE / K))
    if sweeps <= 0:
        sweeps = 1 if per_sweep == 1 else 3
    while per_sweep * sweeps < 20:                     # at least 20 timed blocks
        sweeps += 1
    if blocks > 0:
        per_sweep, sweeps = blocks, 1
    return per_sweep, sweeps


def summarize_blocks(wall_s, phases, per_sweep, K):
    """Episode-average of the per-phase medians + descriptive figures.  wall_s[b] = block time (max over ranks)."""
    import statistics
    by_phase = {}
    for b, w in enumerate(wall_s):
        by_phase.setdefault(b % per_sweep, []).append(w)
    phase_median = [statistics.median(by_phase[k]) for k in sorted(by_phase)]
    avg = sum(phase_median) / len(phase_median)
    first_phase = {k: phases[k] for k in range(min(per_sweep, len(phases)))}
    dense_k = min(first_phase, key=lambda k: first_phase[k])                      # block that starts closest after a reset
    mid_k = min(first_phase, key=lambda k: abs(first_phase[k] - EPISODE // 2))
    return avg, {
        "timed_blocks": len(wall_s), "blocks_per_sweep": per_sweep, "steps_per_block": K,
        "episode_average_ms_per_step": avg / K * 1e3,
        "median_block_ms_per_step": statistics.median(wall_s) / K * 1e3,
        "min_block_ms_per_step": min(wall_s) / K * 1e3, "max_block_ms_per_step": max(wall_s) / K * 1e3,
        "dense": {"episode_phase": first_phase[dense_k], "ms_per_step": phase_median[dense_k] / K * 1e3},
        "mid_episode": {"episode_phase": first_phase[mid_k], "ms_per_step": phase_median[mid_k] / K * 1e3},
    }


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"bench.py --gpus {args.gpus} ...` or plain `python bench.py --gpus {args.gpus}`")
    if os.environ.get("EVAC_BENCH_FAIL_RANK") == str(rank):        # testing aid: a rank that dies before the rendezvous
        raise SystemExit(7)
    # CPU baseline first (rank 0, N=1 only): its worker processes are forked before this process has
    # touched the GPU, and the GPU measurement below runs on an otherwise idle host.
    cpu_base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not args.dry_run:
        procs = (os.cpu_count() or 1) if args.cpu_procs < 0 else args.cpu_procs
        cpu_base = cpu_baseline(WORKLOADS[args.workload][0], args.cpu_seconds, procs)

    import torch
    import torch.distributed as dist

    n_ped, envs_per_gpu, wrap_kw, desc = WORKLOADS[args.workload]
    if args.envs:
        envs_per_gpu = args.envs
    total_envs = envs_per_gpu * world
    K, W = args.steps, args.warmup
    inner = max(1, min(args.inner, K)) if args.mode == "rollout" else 1
    per_sweep, sweeps = block_plan(K, args.sweeps, args.blocks)

    if args.dry_run:
        return dry_run(args, rank, world, total_envs, envs_per_gpu, K, W, inner, per_sweep, sweeps)

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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: {dist.get_world_size()} ranks joined, --gpus {args.gpus} requested")

    import evacuation_amd as ea
    from evacuation_amd.distributed import ShardedEvacuationEnv, all_gather_envs, pack_outputs

    cfg = ea.EnvConfig(number_of_pedestrians=n_ped, is_new_exiting_reward=True, is_new_followers_reward=True,
                       intrinsic_reward_coef=0.0, max_timesteps=EPISODE)       # SURVEY.md 8(d) synthetic inputs
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    seed = 0x5EED0000 + sorted(WORKLOADS).index(args.workload)
    env = ShardedEvacuationEnv(cfg, wrap, total_envs=total_envs, device=device, seed=seed)
    loc = env.local
    E, D = loc.num_envs, loc.obs_dim
    env.reset()
    do_gather = world > 1 and not args.no_gather

    # preallocated, reused output chunks (the trainer's rollout buffer, rpo_agent.py:158-163): the kernel
    # writes one packed slab [T, E, D+3] = [obs | reward | terminated | truncated], which is also the
    # all-gather message
    def alloc(T):
        b = {"slab": torch.empty((T, E, D + 3), dtype=torch.float32, device=device),
             "episode_stats": torch.zeros((T, E, loc.stats_words), dtype=torch.float32, device=device)}
        b["launch"] = loc.rollout_launcher(T, b)          # pre-bound ctypes call: no per-launch Python argument work
        return b
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
    ev_pool = []

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
                ev0, ev1 = ev_pool.pop() if ev_pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev0.record()
            if args.mode == "rollout":
                if t != inner:
                    b = tails.get(t) or tails.setdefault(t, alloc(t))
                b["launch"]()
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

    def drain():
        """Poll the stream until the issued work is done, so that the synchronize() that closes a timed block returns at
        once: a blocking wait adds the host's sleep / wake-up latency (~10 us) to a block that is itself ~60 us."""
        st = torch.cuda.current_stream()
        while not st.query():
            pass

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

    run(W)                                                    # W untimed warm-up steps
    barrier()
    state0 = [t.clone() for t in (loc.ped, loc.status, loc.agent, loc.clock, loc.acc)]   # for the kernel-timing replay below
    n_blocks = per_sweep * sweeps
    wall, phases, block_events = [], [], []
    launches = 0
    # one launch per block and nothing to gather: issue it without run()'s bookkeeping (a few us of Python next to a 55 us kernel)
    one_launch = bufs[0]["launch"] if (args.mode == "rollout" and K == inner and not do_gather) else None
    for b in range(n_blocks):
        phases.append((W + b * K) % EPISODE)                  # RandomAgent episodes end by truncation at 2000
        barrier()
        t0 = time.perf_counter()
        if one_launch is not None:
            one_launch()                                      # EXACTLY K steps
            launches = 1
        else:
            launches = run(K)                                 # EXACTLY K steps (no event markers inside the timed region)
        drain()
        barrier()
        wall.append(time.perf_counter() - t0)
    if world > 1:
        tt = torch.tensor(wall, dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)             # per block: the slowest rank
        wall = [float(x) for x in tt.tolist()]
    block_s, blocks_info = summarize_blocks(wall, phases, per_sweep, K)

    # Duration of the dominant kernel's launches, from HIP events on the launching stream.  An event pair around ONE
    # short launch also times the ~7 us between the markers and the kernel (11 % of a 20-step launch), so the average
    # launch duration is taken over a replay of the first sweep's blocks (state restored) issued back to back between two events
    # (elapsed / launches: kernel + the ~1.5 us launch boundary) -- this is the figure a
    # `rocprofv3 --kernel-trace --stats` of this command reproduces; the per-launch pairs of the timed blocks give
    # the dense (all N moving) launch, corrected by the mean difference between the two measurements.
    def restore():
        barrier()
        for dst, src in zip((loc.ped, loc.status, loc.agent, loc.clock, loc.acc), state0):   # same phases as the first sweep
            dst.copy_(src)
        barrier()
    restore()
    for b in range(per_sweep):                                # replay 1: an event pair around every launch
        events = []
        run(K, events)
        block_events.append(events)
    barrier()
    per_launch = [a.elapsed_time(b) * 1e-3 for evs in block_events for a, b, t in evs if t == inner]
    restore()                                                 # replay 2: back to back between two events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    back_to_back = 0
    e0.record()
    for b in range(per_sweep):
        back_to_back += run(K)
    e1.record()
    barrier()
    tail_launches = per_sweep * (1 if K % inner else 0)           # launches of a shorter tail shape, if any
    kernel_s = e0.elapsed_time(e1) * 1e-3 / max(1, back_to_back) if not tail_launches else sum(per_launch) / max(1, len(per_launch))
    event_overhead_s = max(0.0, sum(per_launch) / max(1, len(per_launch)) - kernel_s) if not tail_launches else 0.0
    dense_b = min(range(min(per_sweep, n_blocks)), key=lambda k: phases[k])
    dense_l = [a.elapsed_time(b) * 1e-3 for a, b, t in block_events[dense_b] if t == inner]
    kernel_dense_s = (sorted(dense_l)[len(dense_l) // 2] - event_overhead_s) if dense_l else kernel_s
    full = per_launch
    if loc.team_error():                                      # a team barrier timed out somewhere above: the numbers are void
        raise SystemExit("bench.py: evac_team_error is set (a team rollout lost a member); results discarded")
    bytes_per_env_step = loc.algorithmic_bytes_per_env_step
    bytes_per_launch = bytes_per_env_step * E * inner
    achieved = bytes_per_launch / kernel_s / 1e9
    lane_ops_per_env_step = 10.0 * n_ped * n_ped + 40.0 * n_ped          # SURVEY.md 8(d) op model

    # the per-step API (one evac_step launch per step, actions resident in HBM), for transparency
    step_api = None
    if rank == 0 and args.mode == "rollout" and not args.no_step_api:
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
        traffic = traffic_src = None
        try:
            with open(args.traffic_json) as f:
                ent = json.load(f).get(f"{args.workload}:{args.mode}")
            if ent:
                traffic = ent["hbm_bytes_per_env_step"] * E * inner      # measured per env-step, scaled to one launch
                traffic_src = f'{ent.get("source")} ({ent.get("envs")} envs x {ent.get("steps_per_launch")} steps per launch)'
        except Exception:  # noqa: BLE001
            pass
        value = total_envs / (block_s / K)
        out = {
            "metric": "env-steps/s (agent-updates/s) at n=60x4096 envs" if args.workload == "c2" else f"env-steps/s ({args.workload})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": block_s / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "n_pedestrians": n_ped, "envs_per_gpu": E, "total_envs": total_envs,
                       "obs": wrap_kw, "actions": "RandomAgent U(-1,1)^2 drawn on device (Philox4x32-10)",
                       "mode": args.mode, "steps_per_launch": inner, "launches_per_block": launches,
                       "timing": "blocks of exactly K steps, each bracketed by barrier+synchronize, tiling whole episodes; "
                                 "ms_per_step = episode average of the per-phase median block",
                       "ranks_joined": dist.get_world_size() if world > 1 else 1,
                       "parallelism": f"env-sharded x{world}" + (f", RCCL all-gather of the {'observation batch' if args.gather == 'obs' else '[obs|reward|flags] records'} per {inner}-step chunk, overlapped on a side stream" if do_gather else ""),
                       "gather_note": gather_note,
                       "max_timesteps": EPISODE, "autoreset": True},
            "agent_updates_per_s": value * n_ped,
            "blocks": blocks_info,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": loc.kernel_variant(args.mode),
                         "kernel_ms_per_launch": kernel_s * 1e3, "kernel_launches_timed": back_to_back or len(full),
                         "kernel_ms_per_launch_event_pairs": sum(full) / max(1, len(full)) * 1e3,
                         "kernel_ms_per_launch_dense": kernel_dense_s * 1e3,
                         "frac_dense": bytes_per_launch / kernel_dense_s / 1e9 / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_env_step": bytes_per_env_step,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "binding_resource": "valu",
                         "valu_frac": lane_ops_per_env_step * E * inner / kernel_s / VALU_LANE_OPS_PEAK,
                         "hbm_traffic_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "note": "achieved = algorithmic bytes / launch time (task contract): an equivalent-bandwidth figure, the state "
                                 "stays in registers between the steps of a launch; the all-pairs O(N^2) work makes the kernel "
                                 "VALU-bound (valu_frac: 10*N^2+40*N lane-op model of SURVEY 8d against 78.6e12 lane-ops/s -- the model counts "
                                 "every ordered pair, the kernels only the rows that are needed against the pedestrians that still "
                                 "move, so it can exceed 1; DESIGN.md 5 has the measured instruction counts); hbm_traffic_frac "
                                 "is the counter-measured HBM traffic rate against the same peak"},
            "step_api": step_api,
        }
        out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def dry_run(args, rank, world, total_envs, envs_per_gpu, K, W, inner, per_sweep, sweeps):
    """The multi-rank control flow without a GPU: rendezvous over gloo, the block/barrier structure and the
    packed all-gather on CPU tensors.  Used by tests/test_bench_launcher_cpu.py; reports no throughput."""
    import torch
    import torch.distributed as dist

    from evacuation_amd.distributed import all_gather_envs, gathered_view, shard_range
    if world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(3)
    off, n_local = shard_range(total_envs, rank, world)
    gid = torch.arange(off, off + n_local, dtype=torch.float32)
    wall, phases = [], []
    ok = True
    for b in range(min(per_sweep * sweeps, 4)):
        phases.append((W + b * K) % EPISODE)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        slab = gid[None, :, None] + torch.zeros((min(inner, 4), n_local, 9))
        if world > 1:
            g, _ = all_gather_envs(slab)
            full = gathered_view(g)
            ok = ok and bool((full[0, :, 0] == torch.arange(total_envs, dtype=torch.float32)).all())
            dist.barrier()
        wall.append(time.perf_counter() - t0)
    if world > 1:
        tt = torch.tensor(wall, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_joined": dist.get_world_size() if world > 1 else 1,
                          "steps": K, "warmup": W, "total_envs": total_envs, "envs_per_gpu": envs_per_gpu,
                          "gather_in_global_env_order": ok, "blocks": len(wall), "value": None,
                          "self_launched": os.environ.get("EVAC_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 4


if __name__ == "__main__":
    raise SystemExit(main())
