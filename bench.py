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
pedestrians that still move: all N after a reset, a third of them late in the episode), so the timed region is
whole EPISODE SWEEPS: R = 2000/K launches of exactly K steps each (evac_rollout: the state stays in registers between
the steps of a launch, every step still writes its outputs to HBM), issued back to back -- no host sync and no
collective between the launches, the queue more than one launch deep; with N > 1 the gather of chunk j-1 runs under
chunk j (below).  A sweep is bracketed by barrier + torch.cuda.synchronize() on both sides and timed twice: by the
host clock (t1 after this rank's compute AND comm streams have drained) and by two HIP events on the launching
stream.  `ms_per_step` = median over --sweeps sweeps of (sweep wall time, MAX over ranks) / 2000 = the EPISODE-AVERAGE
cost of a step; `value` = total envs / that ("hipEvents around the step loop with no host sync
inside").  `blocks` keeps the per-block view of rounds 1-3 as a diagnostic: every K-step block bracketed by a device-idle
sync of its own (what a caller pays who waits for every launch), with the dense and mid-episode figures.

The gather (N > 1).  The only collective of the path is the all-gather of the returned observation batch.
`--gather-schedule pipelined` (default): the outputs are double-buffered and the gather of chunk j-1 is issued
right after the launch of chunk j -- INSIDE the timed block, on a side stream -- so that it runs under that
chunk's compute; with one launch per block (the driver's --steps 20) block b therefore computes K steps and
gathers block b-1's observations (the warm-up primes the pipeline, so every timed block carries exactly one
gather per launch and drains it before t1).  `--gather-schedule split`: a block's K steps go out as two K/2
launches and each half is gathered as soon as it is computed, all inside the block (the first half's gather
runs under the second half, the second half's is exposed).  RCCL's all-gather is the default; two forms without a
library collective write straight into the peers' hipIpc-mapped buffers: `--gather peer` (ONE hand-written kernel,
evac_peer_gather, storing the observation columns to all peers at once over xGMI; built to fit beside the rollout
workgroups, which hold every CU) and `--gather direct` (copy-engine writes, one peer after another; no CU at all).
A collective that fails is a hard error -- an N-GPU `value` is never printed without the gather traffic unless
--no-gather was given.

`roofline` follows the task contract: ALGORITHMIC bytes per launch (32N + 38 + 4D per
env-step, times the env-steps of one launch) divided by the mean duration of the timed launches, measured
with HIP events on the launching stream, against the 8 TB/s HBM peak.  The kernel keeps its state in
registers, so this is an equivalent-bandwidth figure (`equivalent_bandwidth`); the resource that actually binds is
the VALU: `valu_issue_frac` = counter-measured VALU wave-instructions per env-step (SQ_INSTS_VALU,
profiles/traffic.json) x 64 lanes x env-steps / kernel time against the plain fp32 issue rate of the chip (256 CU
x 4 SIMD x 16 lanes x 2.4 GHz = 39.3e12 lane-ops/s); `traffic` is the HBM traffic the PMC counters see
(bytes per env-step times the env-steps of one launch).
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
VALU_ISSUE_PEAK = 39.3e12       # lane-ops/s at one wave64 VALU instruction per 4 cycles per SIMD: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
VALU_PIPE_PEAK = 78.6e12        # ... per 2 cycles: what a SIMD's vector pipe takes from >= 4 issuing waves (MI355X_MICROARCH.md; tools/microbench/valu_rates.hip)
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
    ap.add_argument("--sweeps", type=int, default=0, help="passes over the episode (0: 11, or 1 when a block is a whole episode -- then raised until 20 blocks are timed)")
    ap.add_argument("--blocks", type=int, default=0, help="override the number of timed K-step blocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-api", action="store_true", help="skip the one-launch-per-step side measurement (and the side workloads: the headline kernel only)")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="skip the short runs of BASELINE configs 3 and 5 (one GPU's shard) and of the 524 288-env per-step run that ride "
                         "along with the default C2 line as `workloads`")
    ap.add_argument("--side-sweeps", type=int, default=20,
                    help="episode sweeps per side workload (rollout mode); their figure is the median of the LAST half (settled clocks)")
    ap.add_argument("--side-only", default="", choices=["", "c2_one_kernel", "c3", "c5_shard", "big_step"],
                    help="run ONLY this entry of `workloads` (exactly as the default line runs it) and print it as the JSON line: what "
                         "tools/final_run.sh puts under rocprofv3, so that a kernel trace covers the launches that entry times and no others")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the headline sweeps keep running the same sweeps for this much GPU time (0: skip): `sustained` in the line, "
                         "and `value` becomes the settled median when it differs from the headline sweeps' by more than 2 %%")
    ap.add_argument("--rollout-form", default="auto", choices=["auto", "one", "parts", "chain", "persist"],
                    help="how the handle issues a rollout (evac_options_t): one = ONE kernel per call on the launching stream, every launch "
                         "behind the one before; chain = consecutive launches alternately on two streams the handle owns, ordered per env on "
                         "the device (evac_options_t.chain; evac_join closes a sweep); parts = two half-batch kernels per call on those two "
                         "streams (evac_options_t.parts); auto = chain on a single GPU without gathers, one otherwise")
    ap.add_argument("--no-gather", action="store_true", help="skip the all-gather of the outputs (N>1)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the gather path -- process group, collective on the comm stream, double-buffered pipeline -- with "
                         "whatever world size there is, also 1 (a one-rank RCCL communicator): exercises the multi-GPU code on one GPU")
    ap.add_argument("--buffers", type=int, default=0,
                    help="output buffers the chunks rotate through (0: 4 with the pipelined gather -- the host may then run three "
                         "chunks ahead of the device before it has to wait for a gather --, else 2)")
    ap.add_argument("--device-wait", action="store_true",
                    help="order a buffer's reuse behind its gather with a device-side stream wait (a barrier packet in front of the "
                         "launch) instead of the host-side check")
    ap.add_argument("--gather", default="obs", choices=["obs", "slab", "direct", "peer", "auto"],
                    help="what the ranks all-gather per chunk and how: the observation batch (north_star) or the whole packed "
                         "record through RCCL; or the observation batch written into the peers' hipIpc-mapped buffers by one "
                         "peer-store kernel (peer) or by copy-engine writes (direct) -- no library collective; auto: peer if the "
                         "peer-store probe passes on every rank (decided before anything is timed), the RCCL observation gather if not")
    ap.add_argument("--no-alt-gather", action="store_true",
                    help="skip the two extra sweeps that time the OTHER gather form (peer-store kernel <-> RCCL) after the headline sweeps")
    ap.add_argument("--gather-schedule", default="pipelined", choices=["pipelined", "split"],
                    help="pipelined: the gather of chunk j-1 runs under the compute of chunk j (double-buffered, across blocks); "
                         "split: a one-launch block goes out as two K/2 launches, each gathered inside the block")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="worker processes for the many-core CPU figure (-1: the cores this process may run on; 0/1: skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / block and gather control flow only, on CPU tensors over gloo (tests; prints no throughput)")
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


CSRC_FILES = ("evac_api.hip", "evac_common.h", "evac_device.h", "evac_families.h", "evac_gather.h", "evac_subwave.h", "evac_team.h")


def csrc_sha16(root: str = ROOT) -> str:
    """First 16 hex digits of the SHA-256 over the kernel sources + the C ABI header: what profiles/traffic.json's counters
    were measured on must be what this run has built (tools/make_traffic_json.py stores the same figure)."""
    import hashlib
    h = hashlib.sha256()
    for name in CSRC_FILES:
        with open(os.path.join(root, "evacuation_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(root, "include", "evac.h"), "rb") as f:
        h.update(b"evac.h\0" + f.read())
    return h.hexdigest()[:16]


def load_traffic(path: str, key: str, kernel_variant: str, sources_sha16: str):
    """The counter-measured figures of `key` ("c2:rollout", ...) from profiles/traffic.json -- or None fields with a note when
    the entry was measured on other kernels than the ones loaded now (another kernel variant, or kernel sources that have
    changed since): stale counters must not ride along with a fresh timing."""
    out = {"hbm_bytes_per_env_step": None, "valu": None, "salu": None, "lds": None, "source": None, "note": None, "chain_record_bytes_per_env_launch": 0}
    try:
        with open(path) as f:
            ent = json.load(f).get(key)
    except Exception as exc:  # noqa: BLE001
        out["note"] = f"no counter file ({type(exc).__name__})"
        return out
    if not ent:
        out["note"] = f"no entry {key!r} in {os.path.basename(path)}"
        return out
    if ent.get("kernel_variant") != kernel_variant or ent.get("csrc_sha16") != sources_sha16:
        out["note"] = (f"profiles/traffic.json[{key!r}] was measured on kernel {ent.get('kernel_variant')!r} built from sources "
                       f"{ent.get('csrc_sha16')}; this run has {kernel_variant!r} from {sources_sha16}: counters withheld "
                       f"(re-run tools/profile_pmc.sh + tools/make_traffic_json.py)")
        return out
    out.update(hbm_bytes_per_env_step=ent["hbm_bytes_per_env_step"], valu=ent.get("valu_wave_insts_per_env_step"),
               salu=ent.get("salu_wave_insts_per_env_step"), lds=ent.get("lds_wave_insts_per_env_step"),
               source=f'{ent.get("source")} ({ent.get("envs")} envs x {ent.get("steps_per_launch")} steps per launch)')
    if ent.get("counted_variant") and ent["counted_variant"] != kernel_variant:
        # the counters serialise dispatches, and chained launches wait for each other: the kernel was counted in its plain launches
        # (same step loop) and the exchange record a chained launch reads and writes per env is added as its algorithmic size
        out["chain_record_bytes_per_env_launch"] = int(ent.get("chain_record_bytes_per_env_launch") or 0)
        if out["chain_record_bytes_per_env_launch"]:
            out["note"] = (f"counters collected on {ent['counted_variant']!r} (--rollout-form one: under --pmc every dispatch runs alone, chained launches "
                           f"would wait for each other); the chained form adds its exchange record, {out['chain_record_bytes_per_env_launch']} B per env "
                           f"and launch read + written in uncached memory, to `traffic`")
            out["source"] += "; + exchange records"
        else:
            out["note"] = (f"counters collected on {ent['counted_variant']!r} (--rollout-form one: the same step loop, one launch per call); the persistent "
                           f"kernel keeps the state in registers from call to call, so its HBM traffic per call is at most this")
    return out


def cpu_baseline(n_ped: int, seconds: float, procs: int):
    """The NumPy oracle (a port of the reference's EvacuationEnv.step + GravityEncoding) stepped in a
    single-env RandomAgent loop on one host core, as the reference's README loop does; plus, as
    `many_core`, the same loop in one worker process per core THIS process may run on."""
    from oracle import cpu_bench

    n, dt = cpu_bench.readme_loop(n_ped, seconds)
    allowed = cpu_bench.usable_cores()
    out = {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} single-env steps (n={n_ped}, gravity obs, RandomAgent loop) of the NumPy oracle in {dt:.1f} s "
                     f"on 1 core ({allowed} usable of {os.cpu_count()} host cores)",
           "agent_updates_per_s": n * n_ped / dt,
           # (the port stands in for the reference, which may not travel to the GPU box; its own step() as the survey timed it:)
           "reference_measured": {"value": 3177, "unit": "env-steps/s", "where": "BASELINE.md section 2: the reference's step() with gravity "
                                  "obs at n=60, survey container (Xeon 2.1 GHz), 1 thread", "port_vs_reference": None},
           "survey_row_8d_ii": "re-scoped: N independent single-env worker processes (many_core) instead of a batched [E, N] NumPy "
                               "restatement split over the cores -- one env per process is how the reference itself uses many cores "
                               "(run_scripts/run.sh:65-68), and a second, batched oracle would be unpinned against the reference"}
    out["reference_measured"]["port_vs_reference"] = out["value"] / 3177.0
    if procs > 1:
        try:
            out["many_core"] = cpu_bench.many_core_report(n_ped, min(seconds, 6.0), procs, n / dt)
        except Exception as exc:  # noqa: BLE001
            out["many_core"] = {"error": f"{type(exc).__name__}: {exc}"[:160]}
    return out


def block_plan(K: int, sweeps: int, blocks: int):
    """(blocks per sweep, sweeps): consecutive K-step blocks tile the episode."""
    per_sweep = max(1, math.ceil(EPISODE / K))
    if sweeps <= 0:
        sweeps = 1 if per_sweep == 1 else 11             # (median of >= 5 repeats; eleven 20-step sweeps are 50 ms of GPU time)
    while per_sweep * sweeps < 20:                     # at least 20 timed blocks
        sweeps += 1
    if blocks > 0:
        per_sweep, sweeps = blocks, 1
    return per_sweep, sweeps


def summarize_blocks(wall_s, phases, per_sweep, K):
    """Episode-average of the per-phase medians + descriptive figures.  wall_s[b] = block time (max over ranks),
    phases[b] = the episode step at which block b started.  Blocks are grouped by their ACTUAL phase: when K divides the
    episode the sweeps revisit the same phases (a median per phase); when it does not, every block is a phase of its own and
    the average runs over all of them (still whole sweeps of the episode)."""
    import statistics
    by_phase = {}
    for ph, w in zip(phases, wall_s):
        by_phase.setdefault(ph, []).append(w)
    keys = sorted(by_phase)
    phase_median = {k: statistics.median(by_phase[k]) for k in keys}
    avg = sum(phase_median.values()) / len(phase_median)
    dense_k = keys[0]                                                           # the block that starts closest after a reset
    mid_k = min(keys, key=lambda k: abs(k - EPISODE // 2))
    return avg, {
        "timed_blocks": len(wall_s), "blocks_per_sweep": per_sweep, "steps_per_block": K, "distinct_phases": len(keys),
        "episode_average_ms_per_step": avg / K * 1e3,
        "median_block_ms_per_step": statistics.median(wall_s) / K * 1e3,
        "min_block_ms_per_step": min(wall_s) / K * 1e3, "max_block_ms_per_step": max(wall_s) / K * 1e3,
        "dense": {"episode_phase": dense_k, "ms_per_step": phase_median[dense_k] / K * 1e3},
        "mid_episode": {"episode_phase": mid_k, "ms_per_step": phase_median[mid_k] / K * 1e3},
    }


def choose_gather(requested: str, probe: dict):
    """(form timed, alternative form timed for two extra sweeps or None).  `auto` = the peer-store kernel when the probe passed on
    every rank, RCCL's all-gather of the observation batch when it did not; the alternative is the other one of the two, when it
    can run (the copy-engine and whole-slab forms have none)."""
    peer_ok = bool(probe and probe.get("ok"))
    form = ("peer" if peer_ok else "obs") if requested == "auto" else requested
    alt = {"obs": "peer" if peer_ok else None, "peer": "obs"}.get(form)
    return form, alt


def gather_report(form, world, bytes_per_peer, alone_ms, chunks, launch_plain_ms, env=None, versions=None):
    """What a multi-rank line says about ITS gather, per rank, from timestamps alone -- a pure function
    (tests/test_bench_launcher_cpu.py feeds it synthetic ones).  `chunks`: one dict per chunk of the instrumented sweep with
    the ms timestamps `l0`, `l1` (launch of the chunk on the compute stream) and, when a gather rode under it (the gather of
    the previous chunk in the pipelined schedule), `g0`, `g1` of that gather on the comm stream; same clock.  `alone_ms`: the same
    gather timed on an idle device before the sweeps.  bytes_per_peer: what one peer sends this rank per chunk (one xGMI link)."""
    import statistics
    med = lambda v: (statistics.median(v) if v else None)  # noqa: E731
    with_g = [c for c in chunks if c.get("g0") is not None]
    under = [c["g1"] - c["g0"] for c in with_g]
    launch_with = [c["l1"] - c["l0"] for c in with_g]
    started_before_end = [c["g0"] < c["l1"] for c in with_g]
    overlap = [max(0.0, min(c["l1"], c["g1"]) - max(c["l0"], c["g0"])) for c in with_g]
    tail = [max(0.0, c["g1"] - c["l1"]) for c in with_g]            # the part of the gather that outlasts the launch it rode under
    gbps = lambda ms: (bytes_per_peer / (ms * 1e-3) / 1e9 if ms and ms > 0 else None)  # noqa: E731
    return {
        "form": form, "world": world, "chunks_instrumented": len(with_g),
        "bytes_received_per_chunk": bytes_per_peer * max(world - 1, 0), "bytes_per_link_per_chunk": bytes_per_peer,
        "gather_ms_alone": med(alone_ms), "gather_ms_under_compute": med(under),
        "launch_ms_with_gather": med(launch_with), "launch_ms_plain": launch_plain_ms,
        "launch_slowdown_under_gather": (med(launch_with) / launch_plain_ms) if (launch_with and launch_plain_ms) else None,
        "gather_started_before_rollout_ended": (sum(started_before_end) / len(started_before_end)) if started_before_end else None,
        "overlap_ms": med(overlap), "gather_tail_after_launch_ms": med(tail),
        "link_GBps_alone": gbps(med(alone_ms)), "link_GBps_under_compute": gbps(med(under)),
        "versions": versions, "env": env,
        "checks_design_estimates": {"link_GBps_under_compute": "DESIGN 6: 48-60 GB/s per link assumed",
                                    "gather_started_before_rollout_ended": "DESIGN 6: the gather kernel is co-resident with the rollout's workgroups (1.0 = always)",
                                    "launch_slowdown_under_gather": "DESIGN 6: <= 1.10",
                                    "gather_ms_under_compute": "DESIGN 6: 33-41 us wire time + ~10 us latency at 8 GPUs, 20-step chunks"},
    }


def comm_environment():
    """Versions and the environment variables that shape the collective, for the multi-rank line."""
    import torch
    vers = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    try:
        vers["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as exc:  # noqa: BLE001
        vers["rccl"] = f"unknown ({type(exc).__name__})"
    keys = ("NCCL_", "RCCL_", "HSA_", "HIP_", "ROCR_", "GPU_MAX_HW_QUEUES", "TORCH_NCCL_", "CUDA_VISIBLE", "ROCM_")
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith(keys)}
    return vers, env


def chunk_sizes(K: int, inner: int, schedule: str, gather: bool):
    """The launches of one K-step block.  `split` turns a one-launch block into two halves so that the first half's gather
    has compute to hide under."""
    if gather and schedule == "split" and K <= inner and K >= 2:
        return [K - K // 2, K // 2]
    out, done = [], 0
    while done < K:
        t = min(inner, K - done)
        out.append(t)
        done += t
    return out


class ChunkPipeline:
    """Issues the launches of the timed blocks and the gathers of their outputs; the ONE implementation of the block
    structure, used with HIP streams by main() and with CPU stand-ins by dry_run() (tests/test_bench_launcher_cpu.py reads
    its trace).

    Chunks are numbered across blocks; chunk j computes into buffer j % nbuf.  lag = 1 (pipelined): after the launch of chunk j
    the gather of chunk j-1 is issued (it reads buffer (j-1) % nbuf while chunk j writes another).  lag = 0 (split): chunk j's
    own gather is issued right after its launch.  Before a chunk reuses a buffer the compute stream waits for the gather
    that read it: with nbuf = 2 launch j+1 waits for gather j-1, which was issued only after launch j -- on a device that the
    rollout fills, the gather's kernels find no free CU under launch j and the two serialise; with nbuf = 3 (the default of the
    pipelined schedule) launch j+1 waits for gather j-2, which had launch j's whole duration.
    `drain()` = everything issued so far, compute and gathers, has completed on this rank."""

    def __init__(self, launch, gather, wait_gather, drain_compute, drain_gather, lag=1, trace=None, nbuf=2):
        self.launch, self.gather, self.wait_gather = launch, gather, wait_gather
        self.drain_compute, self.drain_gather = drain_compute, drain_gather
        self.lag = lag
        self.nbuf = nbuf
        self.trace = trace
        self.next = 0                   # next chunk number
        self.sizes = {}                 # chunk -> steps (for its gather)
        self.ungathered = None          # the chunk whose gather has not been issued yet (lag = 1)
        self.inflight = {}              # buffer -> gather token of the last gather that read it

    def _log(self, *ev):
        if self.trace is not None:
            self.trace.append(ev)

    def _issue_gather(self, j):
        self._log("gather", j)
        self.inflight[j % self.nbuf] = self.gather(j, self.sizes.pop(j))

    def run_block(self, sizes):
        """Issue one block: exactly sum(sizes) env steps."""
        for t in sizes:
            j = self.next
            self.next += 1
            tok = self.inflight.pop(j % self.nbuf, None)
            if tok is not None:                       # buffer reuse: the gather that read buffer j % nbuf must be finished
                self._log("wait_gather_of_buffer", j % self.nbuf)
                self.wait_gather(tok)
            self._log("launch", j)
            self.launch(j, t)
            self.sizes[j] = t
            if self.gather is None:
                self.sizes.pop(j)
                continue
            if self.lag == 0:
                self._issue_gather(j)
            else:
                if self.ungathered is not None:
                    self._issue_gather(self.ungathered)
                self.ungathered = j

    def drain(self):
        self._log("drain")
        self.drain_compute()
        if self.gather is not None:
            self.drain_gather()

    def flush(self):
        """After the last block: the gather still owed (pipelined), untimed."""
        if self.gather is not None and self.ungathered is not None:
            self._issue_gather(self.ungathered)
            self.ungathered = None
        self.drain()


# --------------------------------------------------------------------------------------------------
# side workloads: BASELINE configs 3 and 5 (one GPU's shard) and the roofline-evidence run, inside the driver's line
# --------------------------------------------------------------------------------------------------
SIDE_WORKLOADS = (
    # key in `workloads`, WORKLOADS name, mode, steps per launch
    ("c3", "c3", "rollout", 100),
    ("c5_shard", "c5", "rollout", 100),
    ("big_step", "big", "step", 1),
)


def side_workload(name: str, mode: str, inner: int, sweeps: int, device, traffic_json: str, options=None):
    """One BASELINE config beside the headline: whole episode sweeps (2000 steps each, launches back to back, bracketed by
    torch.cuda.synchronize and two HIP events on the launching stream) of a fresh batch, after one untimed sweep's worth of
    warm-up launches (200 steps).  Returns the entry of `workloads` in the bench line.  `value` = envs / (median of the LAST half of
    the sweeps / 2000) -- the clocks settle over the first sweeps (C5 slowed 8.9 -> 10.5 ms per sweep over 200 ms) --,
    `roofline` as for the headline: ALGORITHMIC bytes per launch / the mean launch of the median sweep (HIP events), counter
    traffic from profiles/traffic.json under the same guard.  The per-step run of the 524 288-env batch (state 503 MB: every
    step reads and writes it through HBM) is timed over ONE episode sweep of evac_step launches."""
    import statistics

    import torch

    import evacuation_amd as ea

    n_ped, E, wrap_kw, desc = WORKLOADS[name]
    cfg = ea.EnvConfig(number_of_pedestrians=n_ped, is_new_exiting_reward=True, is_new_followers_reward=True,
                       intrinsic_reward_coef=0.0, max_timesteps=EPISODE)
    env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(**wrap_kw), num_envs=E, device=device,
                                  seed=0x5EED0000 + sorted(WORKLOADS).index(name), options=options)
    env.reset()
    stream = torch.cuda.current_stream(device)
    if mode == "rollout":
        out = {"slab": torch.empty((inner, E, env.obs_dim + 3), dtype=torch.float32, device=device),
               "episode_stats": torch.zeros((inner, E, env.stats_words), dtype=torch.float32, device=device)}
        go = env.rollout_launcher(inner, out, stream=stream)
        launches = EPISODE // inner
        warm = max(1, 200 // inner)
    else:
        actions = torch.rand((E, 2), device=device) * 2 - 1
        go = env.step_launcher(actions, stream=stream)
        launches, warm, sweeps = EPISODE, 20, 1
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea0.record(stream)                                        # (every launch of this entry, warm-up included: what a kernel trace of the process averages)
    for _ in range(warm):
        go()
    env.join(stream)                                          # (the handle's own streams: the warm-up ends here)
    torch.cuda.synchronize()
    wall, dev = [], []
    for _ in range(sweeps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(launches):
            go()
        env.join(stream)
        e1.record(stream)
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
        dev.append(e0.elapsed_time(e1) * 1e-3)
    env.join(stream)
    ea1.record(stream)
    torch.cuda.synchronize()
    all_launches = warm + sweeps * launches
    if env.team_error():
        raise SystemExit(f"bench.py: evac_team_error is set in side workload {name}; results discarded")
    settled = slice(len(wall) // 2, None)                      # the last half of the sweeps
    sweep_s = statistics.median(wall[settled])
    kernel_s = statistics.median(dev[settled]) / launches
    variant = env.kernel_variant(mode)
    tr = load_traffic(traffic_json, f"{name}:{mode}", variant, csrc_sha16())
    bytes_per_env_step = env.algorithmic_bytes_per_env_step
    bytes_per_launch = bytes_per_env_step * E * inner
    achieved = bytes_per_launch / kernel_s / 1e9
    traffic = (tr["hbm_bytes_per_env_step"] * E * inner + tr["chain_record_bytes_per_env_launch"] * E) if tr["hbm_bytes_per_env_step"] is not None else None
    value = E / (sweep_s / EPISODE)
    env.close()
    return {
        "workload": desc, "mode": mode, "envs": E, "n_pedestrians": n_ped, "steps_per_launch": inner,
        "value": value, "unit": "env-steps/s", "agent_updates_per_s": value * n_ped, "ms_per_step": sweep_s / EPISODE * 1e3,
        "sweeps": {"timed": sweeps, "launches_per_sweep": launches, "wall_ms": [x * 1e3 for x in wall], "hip_event_ms": [x * 1e3 for x in dev],
                   "value_from": f"median of sweeps {len(wall) // 2}..{len(wall) - 1} (the last half)", "gpu_ms_timed_total": sum(dev) * 1e3,
                   "value_first_half": E * EPISODE / statistics.median(wall[:max(1, len(wall) // 2)]),
                   "value_min_median_max": [E * EPISODE / max(wall), value, E * EPISODE / min(wall)]},
        "kernel": variant, "kernels_in_flight": 1 if "persistent" in variant else max(1, env.own_streams), "kernel_ms_per_launch": kernel_s * 1e3,
        "profile_check": {"launches": all_launches, "sum_of_sweeps_ms_per_launch": sum(dev) * 1e3 / (sweeps * launches),
                          "first_launch_to_last_ms_per_launch": ea0.elapsed_time(ea1) / all_launches,
                          "note": "for a reader with the kernel trace of `bench.py --side-only <this entry>`: the trace's average duration of this kernel "
                                  "over ALL its launches (warm-up and every sweep) is to be compared with sum_of_sweeps_ms_per_launch (the sweeps' own "
                                  "HIP-event time over their launches; first_launch_to_last also holds the idle gaps between sweeps); "
                                  "kernel_ms_per_launch -- what `value` and `roofline` use -- is the median of the LAST half of the sweeps"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": tr["source"], "traffic_note": tr["note"],
                     "hbm_traffic_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                     "algorithmic_bytes_per_env_step": bytes_per_env_step, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "valu_wave_insts_per_env_step": tr["valu"], "salu_wave_insts_per_env_step": tr["salu"],
                     "valu_pipe_frac": (tr["valu"] * 64.0 * E * inner / kernel_s / VALU_PIPE_PEAK) if tr["valu"] else None,
                     "equivalent_bandwidth": mode == "rollout"},
    }


def side_options(ea, mode):
    """The options of a side workload's handle: as the headline's on one GPU -- chained rollout launches wherever the library offers them
    (the CU-wide kernels of one- and four-wave envs: C3; the teams of C5 and the per-step runs are plain handles)."""
    return ea.KernelOptions(chain=2) if mode == "rollout" else None


def side_workloads(args, device):
    """`workloads` of the default line: each entry measured as side_workload() says; an entry that fails carries the error."""
    import evacuation_amd as ea
    out = {}
    try:      # the headline batch, the driver's K steps per launch, issued as ONE kernel per rollout call (what rounds 1-5 timed)
        out["c2_one_kernel"] = side_workload("c2", "rollout", max(1, min(args.inner, args.steps)), args.side_sweeps, device, args.traffic_json,
                                             options=ea.KernelOptions(parts=1, chain=0))
        out["c2_one_kernel"]["note"] = ("the headline's batch and launch shape as plain launches (evac_options_t.chain = 0, parts = 1): one kernel per "
                                        "rollout call on the launching stream, every launch behind the one before -- what rounds 1-5 timed")
    except SystemExit:
        raise
    except Exception as exc:  # noqa: BLE001
        out["c2_one_kernel"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    for key, name, mode, inner in SIDE_WORKLOADS:
        try:
            out[key] = side_workload(name, mode, inner, args.side_sweeps, device, args.traffic_json, options=side_options(ea, mode))
        except SystemExit:
            raise
        except Exception as exc:  # noqa: BLE001
            out[key] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.side_only:                                         # one entry of `workloads`, exactly as the default line runs it
        import torch
        import evacuation_amd as ea
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; evacuation_amd has no CPU path")
        device = torch.device("cuda:0")
        torch.cuda.set_device(device)
        if args.side_only == "c2_one_kernel":
            entry = side_workload("c2", "rollout", max(1, min(args.inner, args.steps)), args.side_sweeps, device, args.traffic_json,
                                  options=ea.KernelOptions(parts=1, chain=0))
        else:
            key, name, mode, inner = next(w for w in SIDE_WORKLOADS if w[0] == args.side_only)
            entry = side_workload(name, mode, inner, args.side_sweeps, device, args.traffic_json, options=side_options(ea, mode))
        print(json.dumps({"side_only": args.side_only, **entry}), flush=True)
        return 0
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
        from oracle import cpu_bench
        procs = cpu_bench.usable_cores() if args.cpu_procs < 0 else args.cpu_procs
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
    use_dist = world > 1 or args.force_gather                                  # (--force-gather: a one-rank communicator)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
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
    from evacuation_amd.distributed import (DirectGather, PeerStoreGather, ShardedEvacuationEnv, agree_all, all_gather_envs,
                                             pack_outputs, peer_store_probe, side_stream)

    cfg = ea.EnvConfig(number_of_pedestrians=n_ped, is_new_exiting_reward=True, is_new_followers_reward=True,
                       intrinsic_reward_coef=0.0, max_timesteps=EPISODE)       # SURVEY.md 8(d) synthetic inputs
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    seed = 0x5EED0000 + sorted(WORKLOADS).index(args.workload)
    # How a rollout is issued (evac_options_t.parts; results do not depend on it): with gathers the chunk pipeline orders its
    # streams by events recorded behind ONE launch per chunk, so those runs keep one kernel per launch.
    form_req = args.rollout_form
    if form_req == "auto":
        form_req = "persist" if (world == 1 and not use_dist and args.mode == "rollout") else "one"
    if form_req != "one" and use_dist and not args.no_gather:
        raise SystemExit("bench.py: --rollout-form parts / chain with gathers is not supported (the gather pipeline waits on one launch per chunk)")
    kopts = {"one": ea.KernelOptions(parts=1, chain=0), "parts": ea.KernelOptions(parts=2, chain=0),
             "chain": ea.KernelOptions(chain=1), "chain_auto": ea.KernelOptions(chain=-1),
             "persist": ea.KernelOptions(chain=2)}[form_req]          # (the library falls back to chained, then plain launches where it cannot: config.rollout_form says what ran)
    env = ShardedEvacuationEnv(cfg, wrap, total_envs=total_envs, device=device, seed=seed, options=kopts)
    loc = env.local
    E, D = loc.num_envs, loc.obs_dim
    env.reset()
    do_gather = use_dist and not args.no_gather
    gather_rollout = do_gather and args.mode == "rollout"
    sizes = chunk_sizes(K, inner, args.gather_schedule, gather_rollout)
    inner = sizes[0]                                           # the launch shape the roofline block describes
    lag = 0 if (args.gather_schedule == "split") else 1
    # Which gather is timed is decided HERE, before anything is timed, by all ranks together: the peer-store probe (hipIpc
    # mappings of every rank's buffer, peer access, a checked store pattern; never raises, every stage agreed over the host) runs
    # whatever form was asked for -- its result is part of the line --; `--gather auto` takes the peer-store kernel when it passed
    # on every rank and RCCL otherwise.  Nothing is restarted or re-executed: a rank that cannot map a peer simply says so.
    probe = peer_store_probe(device) if gather_rollout else None
    form, alt_form = choose_gather(args.gather, probe)
    if args.no_alt_gather or not gather_rollout or lag == 0:
        alt_form = None
    forms = [form] + ([alt_form] if alt_form else [])

    # Preallocated, reused output chunks (the trainer's rollout buffer, rpo_agent.py:158-163): the kernel writes one packed
    # slab [T, E, D+3] = [obs | reward | terminated | truncated] per launch shape and buffer of the ring.
    compute = torch.cuda.current_stream(device)               # every launch of this benchmark goes to this stream
    comm = side_stream(device, beside=compute) if do_gather else None      # (a stream on ANOTHER hardware queue than `compute`)
    nbuf = args.buffers if args.buffers > 0 else (4 if (gather_rollout and lag == 1) else 2)
    GW = D + 3 if form == "slab" else D                       # gathered words per env-step (obs, direct, peer: the observation columns)
    chunks = {}

    def chunk_bufs(t, parity):
        key = (t, parity)
        b = chunks.get(key)
        if b is None:
            b = {"slab": torch.empty((t, E, D + 3), dtype=torch.float32, device=device),
                 "episode_stats": torch.zeros((t, E, loc.stats_words), dtype=torch.float32, device=device)}
            b["launch"] = loc.rollout_launcher(t, b, stream=compute)   # pre-bound ctypes call: no per-launch Python argument work
            if gather_rollout:
                b["ready"], b["fin"] = torch.cuda.Event(), torch.cuda.Event()   # (reused: creating two events per gather cost 5 us of host time)
                b["gathered"] = torch.empty((world, t, E, GW), dtype=torch.float32, device=device)
                if "obs" in forms or "direct" in forms:
                    b["gsrc"] = torch.empty((t, E, GW), dtype=torch.float32, device=device)
                if "direct" in forms:
                    b["direct"] = DirectGather(b["gsrc"], b["gathered"])     # peers' `gathered` buffers mapped through hipIpc
                if "peer" in forms:                          # ... and written by one kernel, columns picked on the way
                    b["peer"], perr = PeerStoreGather.try_build(b["slab"], D, b["gathered"],    # (all ranks: the gather, or the same error)
                                                                 _inject_failure=os.environ.get("EVAC_BENCH_FAIL_PEER_BUILD") == str(rank))
                    if perr:
                        raise PeerFormUnavailable(perr)
            chunks[key] = b
        return b

    class PeerFormUnavailable(RuntimeError):
        """The peer-store gather cannot be built / failed its self-test -- on SOME rank; every rank raises it together."""

    def without_peer_form(why):
        """The peer-store form is out, on all ranks alike (`why` is the agreed first error): the alternative form of a line is a
        diagnostic and simply goes; `--gather auto` falls back to RCCL -- before anything is timed --; an explicit `--gather peer` is a
        hard error (returns False)."""
        nonlocal form, alt_form
        if form == "peer" and args.gather != "auto":
            return False
        if form == "peer":
            form, alt_form = "obs", None
            run["form"] = form
            run["auto_fell_back"] = why
        else:
            alt_form = None
            run["alternative_dropped"] = why
        forms[:] = [form]
        for b in chunks.values():
            b.pop("peer", None)
        chunks.clear()                                        # (rebuilt for the forms that are left)
        return True

    step_actions = torch.rand((E, 2), device=device) * 2 - 1

    run = {"form": form, "diag": None}      # the gather form in use; diag: {chunk: {...events}} while the instrumented sweep runs

    def launch(j, t):
        if args.mode == "rollout":
            b = chunk_bufs(t, j % nbuf)
            if run["diag"] is not None:       # (instrumented sweep only: a timing event pair around the launch)
                ev = run["diag"].setdefault(j, {})
                ev["l0"], ev["l1"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev["l0"].record(compute)
                b["launch"]()
                ev["l1"].record(compute)
            else:
                b["launch"]()
            if gather_rollout:
                # the chunk's outputs are complete HERE on the compute stream.  (Rounds 2-3 recorded this event inside gather(),
                # i.e. after the NEXT launch had been enqueued: the gather of chunk j-1 then waited for launch j to finish and
                # never ran under it -- tools/gather_cost.py, profiles/r04_j_force_gather_world1.txt.)
                b["ready"].record(compute)
        else:
            loc.step(step_actions)

    def gather(j, t):
        """Issue the gather of chunk j's outputs on the comm stream; returns the event that marks its end."""
        if args.mode == "step":
            msg = loc.obs if args.gather in ("obs", "auto") else pack_outputs(loc.obs, loc.reward, loc.terminated, loc.truncated)
            all_gather_envs(msg)                              # (per-step API: one small collective per step, on the compute stream)
            return None
        # (the comm stream is torch's CURRENT stream for the whole pipeline -- see run_pipeline_on_comm below --, so that no
        # stream switch is paid per gather: the collective and the column copy go to the current stream, the launches are bound
        # to `compute`.  Host cost of one gather: tools/gather_cost.py.)
        b = chunk_bufs(t, j % nbuf)
        fin = b["fin"]
        comm.wait_event(b["ready"])                           # recorded right behind chunk j's launch (launch())
        ev = None
        if run["diag"] is not None:                           # (instrumented sweep only: a timing event pair around the gather)
            ev = run["diag"].setdefault(j, {})
            ev["g0"], ev["g1"] = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev["g0"].record(comm)
        issue_gather(b, run["form"])
        if ev is not None:
            ev["g1"].record(comm)
        fin.record(comm)
        return fin

    def issue_gather(b, f):
        """The gather of one chunk's outputs in form `f`, enqueued on the comm stream (torch's current stream here)."""
        if f == "slab":
            all_gather_envs(b["slab"], out=b["gathered"])
        elif f == "peer":
            b["peer"].issue(comm)                             # one launch: the observation columns to every peer
        else:                                                 # the observation columns, copied out on the comm stream
            b["gsrc"].copy_(b["slab"][..., :D])
            if f == "direct":
                b["direct"].issue(comm)                       # world copy-engine writes into the peers' buffers
            else:
                all_gather_envs(b["gsrc"], out=b["gathered"])

    def wait_gather(fin):
        """Before a chunk reuses a buffer: the gather that read it is done.  Checked on the HOST (a query; a host-side wait only
        if the host has run `nbuf - 1` chunks ahead of the device) -- a device-side `compute.wait_event(fin)` puts a barrier
        packet between two rollout launches and costs 7-10 us per chunk on this runtime whether or not the event has fired
        (profiles/r04_j_force_gather_world1.txt: 58.6 -> 48.2 us per chunk).  --device-wait restores it."""
        if fin is None:
            return
        if args.device_wait:
            compute.wait_event(fin)
        elif not fin.query():
            fin.synchronize()

    def drain_compute():
        """Everything issued to the device so far is done.  (hipDeviceSynchronize spins here; a hipStreamQuery loop costs the
        same wait plus ~2.5 us on the NEXT launch call -- the query leaves a marker behind: tools/block_overhead.py.)"""
        if loc.own_streams:
            loc.join()                                        # (no-op when the sweep has joined already; see barrier())
        torch.cuda.synchronize()

    def drain_gather():
        comm.synchronize()

    pipe = ChunkPipeline(launch, gather if do_gather else None, wait_gather, drain_compute, drain_gather if do_gather else None, lag=lag,
                         nbuf=nbuf)

    def barrier():
        """Opens a timed region (and, being the next one's opening, closes the previous one outside its timed region)."""
        if use_dist:
            dist.barrier()
        if loc.own_streams:                                   # (everything the handle's own streams hold belongs to the region that ends here)
            loc.join()
        torch.cuda.synchronize()

    # Every (chunk size, parity) buffer set is created HERE, on all ranks in the same order: the peer-mapped gathers rendezvous
    # (hipIpc handles through all_gather_object, a barrier) when they are built, which must not happen inside a timed region.
    # Then the collective is exercised once before anything is timed.  A collective that fails is a hard error: an N-GPU value
    # without the gather traffic is not the benchmark (use --no-gather to measure independent shards on purpose).
    all_sizes = sorted(set(sizes) | ({min(inner, W - d) for d in range(0, W, inner)} if W > 0 else set()))
    while True:
        try:
            if args.mode == "rollout":
                for t_ in all_sizes:
                    for par_ in range(nbuf):
                        chunk_bufs(t_, par_)
            if gather_rollout:
                b0 = chunk_bufs(sizes[0], 0)
                for f_ in list(forms):
                    # every form's first gather ends with an agreement over the host: a rank whose gather failed does not
                    # leave the others waiting in the warm-up's collectives
                    err = None
                    try:
                        if f_ == "direct":
                            b0["direct"].issue(torch.cuda.current_stream())
                            b0["direct"].self_test()
                        elif f_ == "peer":
                            b0["peer"].self_test()
                        else:
                            all_gather_envs(b0["slab"] if f_ == "slab" else b0["gsrc"], out=b0["gathered"])
                        torch.cuda.synchronize()
                    except Exception as exc:  # noqa: BLE001
                        err = f"{type(exc).__name__}: {exc}"
                    bad = agree_all(err)
                    if bad and f_ == "peer":
                        raise PeerFormUnavailable(bad)
                    if bad:
                        raise RuntimeError(bad)
            break
        except PeerFormUnavailable as exc:
            if without_peer_form(str(exc)):
                continue                                      # build again without it (nothing has been timed yet)
            raise SystemExit(f"bench.py: rank {rank}: --gather peer: {exc}; no {world}-GPU result (--gather auto falls back to RCCL)") from exc
        except Exception as exc:  # noqa: BLE001
            if not gather_rollout:
                raise
            raise SystemExit(f"bench.py: rank {rank}: the all-gather failed ({type(exc).__name__}: {exc}); no {world}-GPU result "
                             f"(--no-gather measures independent shards)") from exc

    # With gathers the COMM stream is torch's current stream from here to the end of the timed parts (the launches are bound to
    # `compute`): the collectives and the column copies then need no stream switch per chunk.
    # (only the ROLLOUT pipeline: in --mode step the step launches and the per-step collective go to the current stream, which
    # must stay `compute` -- the sweep's events are recorded there)
    if gather_rollout:
        torch.cuda.set_stream(comm)
    # W untimed warm-up steps through the same pipeline (so that, pipelined, the first timed launch has a gather to carry)
    w_done = 0
    while w_done < W:
        t = min(inner, W - w_done)
        pipe.run_block([t])
        w_done += t
    pipe.drain()
    barrier()
    # one chunk's gather ALONE on an idle device, in every form that will be timed (untimed region; the data is whatever the
    # warm-up left): what the same gather takes under a rollout launch is measured in the instrumented sweep below
    alone_ms = {f_: [] for f_ in forms} if gather_rollout else {}
    for f_ in alone_ms:
        b0 = chunk_bufs(sizes[0], 0)
        for _ in range(5):
            barrier()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(comm)
            issue_gather(b0, f_)
            a1.record(comm)
            comm.synchronize()
            alone_ms[f_].append(a0.elapsed_time(a1))
    barrier()
    state0 = [t.clone() for t in (loc.ped, loc.status, loc.agent, loc.clock, loc.acc)]   # for the per-launch replay below
    ws0 = loc.workspace.clone() if loc.workspace is not None else None
    phase0 = W % EPISODE

    def progress(what):
        """One line on stderr per phase: a run that stops writing is taken to be hung by the GPU box's watchdog, and a hang is then located."""
        if rank == 0:
            print(f"bench.py: {what}", file=sys.stderr, flush=True)

    # ---- the headline: whole episode sweeps, launches back to back, no host sync inside ----
    import statistics
    progress(f"warm-up done; {loc.kernel_variant(args.mode)}; timing {sweeps} sweeps")
    uniform = all(t == inner for t in sizes)
    launches_per_sweep = per_sweep * len(sizes)
    sweep_wall, sweep_dev = [], []
    for sw in range(sweeps):
        barrier()                                             # opening bracket: all ranks present, device idle
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(compute)
        for b in range(per_sweep):
            pipe.run_block(sizes)                             # EXACTLY K steps (+ one gather per launch), nothing waited for
        loc.join(compute)                                     # (two parts: `compute` waits for the handle's own streams; else a no-op)
        e1.record(compute)
        pipe.drain()                                          # this rank's compute AND gathers are done
        sweep_wall.append(time.perf_counter() - t0)           # local t1; no collective inside the timed region
        sweep_dev.append(e0.elapsed_time(e1) * 1e-3)
    pipe.flush()
    barrier()
    sweep_wall_this_rank = list(sweep_wall)                    # (before the max over ranks below: gather_report.per_rank[r].headline)
    # ---- steady state: the SAME sweeps again until --sustain-seconds of GPU time have passed.  The headline's
    # eleven sweeps are ~50 ms; clocks and power settle over seconds.  All ranks run the same number of sweeps (from the max-over-ranks
    # median above); nothing else changes: same brackets, same launches, same gathers.
    sustain_wall, sustain_dev = [], []
    progress(f"headline sweeps done: {[round(x * 1e3, 3) for x in sweep_wall]} ms")
    if args.sustain_seconds > 0:
        med0 = statistics.median(sweep_wall)
        if use_dist:
            tt = torch.tensor([med0], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            med0 = float(tt.item())
        n_sus = max(10, min(4000, int(math.ceil(args.sustain_seconds / max(med0, 1e-6)))))
        for sw in range(n_sus):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(compute)
            for b in range(per_sweep):
                pipe.run_block(sizes)
            loc.join(compute)
            e1.record(compute)
            pipe.drain()
            sustain_wall.append(time.perf_counter() - t0)
            sustain_dev.append(e0.elapsed_time(e1) * 1e-3)
        pipe.flush()
        barrier()
        if use_dist:
            tt = torch.tensor(sustain_wall, dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sustain_wall = [float(x) for x in tt.tolist()]
    if use_dist:
        tt = torch.tensor(sweep_wall, dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)             # per sweep: the slowest rank
        sweep_wall = [float(x) for x in tt.tolist()]
    steps_per_sweep = per_sweep * K
    sweep_s = statistics.median(sweep_wall)
    kernel_s = statistics.median(sweep_dev) / launches_per_sweep      # mean launch of a sweep (kernel + the ~1.5 us launch boundary)
    sweeps_run = sweeps + len(sustain_wall)
    sustained = None
    if sustain_wall:
        allw, alld = sweep_wall + sustain_wall, sweep_dev + sustain_dev          # in the order they ran
        cum, first = 0.0, []
        total_dev = sum(alld)
        for w_, d_ in zip(allw, alld):                      # the sweeps that START within the first 50 ms of GPU time ...
            if cum < 0.050 or not first:
                first.append(w_)
            cum += d_
        cum, last = 0.0, []
        for w_, d_ in zip(reversed(allw), reversed(alld)):  # ... and those that END within the last 500 ms
            if cum < 0.500 or not last:
                last.append(w_)
            cum += d_
        last.reverse()
        v_first = total_envs * steps_per_sweep / statistics.median(first)
        v_last = total_envs * steps_per_sweep / statistics.median(last)
        sustained = {"gpu_seconds": total_dev, "sweeps": len(allw), "first_50ms": v_first, "last_500ms": v_last,
                     "sweeps_in_first_50ms": len(first), "sweeps_in_last_500ms": len(last), "drift": v_last / v_first - 1.0,
                     "last_500ms_min_max": [total_envs * steps_per_sweep / max(last), total_envs * steps_per_sweep / min(last)],
                     "ms_per_sweep_every_50th": [round(x * 1e3, 4) for x in allw[::50]],
                     "value_is_settled_median": False,
                     "note": "the headline sweeps followed by the same sweeps until --sustain-seconds of GPU time: env-steps/s over the "
                             "sweeps of the first 50 ms and of the last 500 ms of GPU time (medians), drift = last / first - 1"}
        if abs(v_last / (total_envs * steps_per_sweep / sweep_s) - 1.0) > 0.02:
            # the settled figure is the one that holds: `value`, `ms_per_step` and the roofline's launch time are taken from it
            sustained["value_is_settled_median"] = True
            sustained["value_of_the_headline_sweeps"] = total_envs * steps_per_sweep / sweep_s
            k = len(last)
            sweep_s = statistics.median(last)
            kernel_s = statistics.median(alld[-k:]) / launches_per_sweep

    # ---- the gather explains itself (N > 1, or --force-gather): one INSTRUMENTED sweep per form -- a timing event pair around
    # every launch (compute stream) and around every gather (comm stream) --, and two plain sweeps of the alternative form.
    # Untimed as far as the headline goes; all ranks run the same sweeps (the forms were agreed before the warm-up).
    def extra_sweep(instrumented):
        barrier()
        ref = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        run["diag"] = {} if instrumented else None
        t0 = time.perf_counter()
        ref.record(compute)
        for b in range(per_sweep):
            pipe.run_block(sizes)
        e1.record(compute)
        pipe.drain()
        wall_ = time.perf_counter() - t0
        pipe.flush()
        barrier()
        diag_, run["diag"] = run["diag"], None
        chunks_ = []
        if diag_:
            for j in sorted(diag_):
                c = {"l0": ref.elapsed_time(diag_[j]["l0"]), "l1": ref.elapsed_time(diag_[j]["l1"])}
                g = diag_.get(j - lag)                        # the gather that rode under launch j: chunk j - 1's (pipelined)
                if g is not None and "g0" in g and lag == 1:
                    c["g0"], c["g1"] = ref.elapsed_time(g["g0"]), ref.elapsed_time(g["g1"])
                chunks_.append(c)
        return wall_, ref.elapsed_time(e1) * 1e-3, chunks_

    gather_runs = {}
    if gather_rollout and rank == 0:
        # (should a diagnostic below take the process down, the measurement is in the log at least)
        print(f"bench.py: headline sweeps done ({form}): {total_envs * steps_per_sweep / sweep_s:.6g} env-steps/s over {world} rank(s), "
              f"sweeps {[round(x * 1e3, 3) for x in sweep_wall]} ms; the gather diagnostics follow", file=sys.stderr, flush=True)
    if gather_rollout:
        for f_ in forms:
            run["form"] = f_
            rec = {"sweep_wall_s": [], "sweep_dev_s": []}
            if f_ != form:
                for _ in range(2):
                    w_, d_, _c = extra_sweep(False)
                    rec["sweep_wall_s"].append(w_)
                    rec["sweep_dev_s"].append(d_)
                    sweeps_run += 1
            w_, d_, rec["chunks"] = extra_sweep(True)
            sweeps_run += 1
            gather_runs[f_] = rec
        run["form"] = form

    # ---- diagnostic 1: one sweep of blocks, each bracketed by a device-idle sync of its own (the headline of rounds 1-3) ----
    progress(f"sustained sweeps done ({len(sustain_wall)}); block diagnostic")
    wall, phases, t_call = [], [], []
    one_launch = chunk_bufs(K, 0)["launch"] if (args.mode == "rollout" and sizes == [K] and not do_gather) else None
    for b in range(per_sweep):
        phases.append((phase0 + (sweeps_run * per_sweep + b) * K) % EPISODE)     # RandomAgent episodes end by truncation at 2000
        barrier()
        t0 = time.perf_counter()
        if one_launch is not None:
            one_launch()
            t_call.append(time.perf_counter() - t0)           # (the launch call alone: reported, not subtracted)
            drain_compute()
        else:
            pipe.run_block(sizes)
            pipe.drain()
        wall.append(time.perf_counter() - t0)
    pipe.flush()
    barrier()
    if use_dist:
        tt = torch.tensor(wall, dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = [float(x) for x in tt.tolist()]
    block_s, blocks_info = summarize_blocks(wall, phases, per_sweep, K)
    blocks_info["note"] = ("diagnostic: ONE sweep of K-step blocks, each opened by barrier + synchronize and closed by the rank's own "
                           "drain (launch call + start / completion latency of a lone launch included); not the headline")
    if t_call:
        blocks_info["launch_call_us_median"] = sorted(t_call)[len(t_call) // 2] * 1e6
    if os.environ.get("EVAC_BENCH_DUMP") and rank == 0:      # diagnostic: the raw sweeps and blocks
        with open(os.environ["EVAC_BENCH_DUMP"], "w") as f:
            json.dump({"sweep_wall_s": sweep_wall, "sweep_dev_s": sweep_dev, "wall_s": wall, "phase": phases, "launch_call_s": t_call}, f)

    if gather_rollout:
        torch.cuda.set_stream(compute)
    # ---- diagnostic 2: an event pair around every launch of one sweep (state restored, no gathers): the dense launch ----
    def restore():
        barrier()
        for dst, src in zip((loc.ped, loc.status, loc.agent, loc.clock, loc.acc), state0):   # same phases as the first sweep
            dst.copy_(src)
        if ws0 is not None:                                   # ... and the same load schedule: loads as they were, dealt again
            loc.workspace.copy_(ws0)
            loc.rebind_workspace()
        barrier()

    def replay_launch(t):
        if args.mode == "rollout":
            chunk_bufs(t, 0)["launch"]()
        else:
            loc.step(step_actions)

    progress("replay of one sweep with an event pair per launch")
    restore()
    per_launch = []
    for b in range(per_sweep):
        for t in sizes:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            replay_launch(t)
            loc.join()                                        # (two parts: the pair brackets both kernels of the round)
            ev1.record()
            per_launch.append((b, t, ev0, ev1))
    barrier()
    dense_b = min(range(per_sweep), key=lambda k: (phase0 + k * K) % EPISODE)
    full = [a.elapsed_time(z) * 1e-3 for b, t, a, z in per_launch if t == inner]
    dense_l = [a.elapsed_time(z) * 1e-3 for b, t, a, z in per_launch if t == inner and b == dense_b]
    sweep_launch_s = kernel_s                                 # mean launch of the median timed sweep (with gathers: incl. their contention and the host's buffer waits)
    if not uniform:
        kernel_s = sum(full) / max(1, len(full))
    # (an event pair around ONE short launch also times the few us between the markers and the kernel: corrected by the mean
    # difference between the pairs and the back-to-back sweep)
    event_overhead_s = max(0.0, sum(full) / max(1, len(full)) - kernel_s) if (uniform and not gather_rollout) else 0.0
    if gather_rollout and uniform:
        #  with gathers in the timed sweeps the span between their two events is not a kernel figure (gather kernels
        # competing for CUs, host-side waits before a buffer is reused): the roofline block describes the KERNEL, taken from
        # the replayed sweep without gathers; the sweep's own mean launch is reported beside it as launch_ms_with_gather
        kernel_s = sum(full) / max(1, len(full))
    kernel_s = max(kernel_s, 1e-9)                            # (never divide by a zero span)
    kernel_dense_s = max((sorted(dense_l)[len(dense_l) // 2] - event_overhead_s) if dense_l else kernel_s, 1e-9)
    back_to_back = launches_per_sweep * sweeps
    # two parts: what each of the two concurrent kernels takes, from an event pair on each of the handle's own streams around one
    # more sweep (the period of that stream's launches: kernel + its boundary) -- what rocprofv3's kernel trace shows per kernel
    part_ms = None
    if loc.own_streams and args.mode == "rollout":
        barrier()
        pev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in loc.part_streams()]
        for (a_, _z), s_ in zip(pev, loc.part_streams()):
            a_.record(s_)
        for b in range(per_sweep):
            for t in sizes:
                chunk_bufs(t, 0)["launch"]()
        for (_a, z_), s_ in zip(pev, loc.part_streams()):
            z_.record(s_)
        loc.join()
        barrier()
        part_ms = [a_.elapsed_time(z_) / launches_per_sweep for a_, z_ in pev]
    # ---- per-rank account of the gather(s), gathered on rank 0 ----
    gather_info = None
    if gather_rollout:
        vers, envv = comm_environment()
        #  this rank's OWN clock over the headline sweeps next to the job's max-over-ranks figure: one slow rank shows
        mine = {"rank": rank, "device": dev_index,
                "headline": {"sweep_wall_ms_this_rank": [x * 1e3 for x in sweep_wall_this_rank],
                             "value_if_every_rank_were_this_one": total_envs * steps_per_sweep / statistics.median(sweep_wall_this_rank),
                             "env_steps_per_s_of_this_rank": E * steps_per_sweep / statistics.median(sweep_wall_this_rank)}}
        for f_, rec in gather_runs.items():
            rep = gather_report(f_, world, inner * E * (D + 3 if f_ == "slab" else D) * 4, alone_ms.get(f_, []), rec["chunks"], kernel_s * 1e3,
                                env=envv if f_ == form else None, versions=vers if f_ == form else None)
            if rec["sweep_wall_s"]:
                rep["sweep_wall_ms"] = [x * 1e3 for x in rec["sweep_wall_s"]]
                rep["value_this_rank"] = total_envs * steps_per_sweep / statistics.median(rec["sweep_wall_s"])
            mine[f_] = rep
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        gather_info = {"timed_form": form, "requested": args.gather, "alternative_form": alt_form, "peer_store_probe": probe,
                       "schedule": args.gather_schedule, "per_rank": everyone,
                       "auto_fell_back": run.get("auto_fell_back"), "alternative_dropped": run.get("alternative_dropped"),
                       "note": "per rank, from one instrumented sweep per form (timing event pairs around every launch and every gather; "
                               "not the headline sweeps): gather_ms_alone = one chunk's gather on an idle device; "
                               "gather_ms_under_compute / launch_ms_with_gather = the same gather under the next chunk's launch and "
                               "that launch; launch_ms_plain = the mean launch of the replayed sweep without gathers; "
                               "gather_started_before_rollout_ended = share of chunks whose gather began before the launch it rode "
                               "under had ended (co-residency of the gather kernel with the rollout's workgroups); link_GBps = bytes "
                               "one peer sends this rank per chunk / gather time; the alternative form: two plain sweeps + one "
                               "instrumented (value_this_rank from the plain ones)"}
        if alt_form and alt_form in gather_runs and gather_runs[alt_form]["sweep_wall_s"]:
            tt = torch.tensor(gather_runs[alt_form]["sweep_wall_s"], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            gather_info["alternative_value"] = total_envs * steps_per_sweep / statistics.median([float(x) for x in tt.tolist()])
    if loc.team_error():                                      # a team barrier timed out somewhere above: the numbers are void
        raise SystemExit("bench.py: evac_team_error is set (a team rollout lost a member); results discarded")
    bytes_per_env_step = loc.algorithmic_bytes_per_env_step
    bytes_per_launch = bytes_per_env_step * E * inner
    achieved = bytes_per_launch / kernel_s / 1e9

    # the per-step API (one evac_step launch per step, actions resident in HBM), for transparency
    progress("per-step API")
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
        go = loc.step_launcher(step_actions, stream=torch.cuda.current_stream())     # the same step with its arguments bound once
        for _ in range(50):
            go()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_api):
            go()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        step_api["launcher_us_per_step"] = dt / n_api * 1e6
        step_api["launcher_env_steps_per_s"] = E * n_api / dt
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
        # ... and through the SyncVectorEnv-shaped host face the reference's UNMODIFIED trainer would use (rpo_agent.py:193-196:
        # NumPy actions up, NumPy observations / rewards / flags down, one stream synchronisation per step)
        try:
            host = ea.HostVectorEnv(loc, copy=True)
            act_np = step_actions.cpu().numpy()
            for _ in range(20):
                host.step(act_np)
            n_host = 200
            t1 = time.perf_counter()
            for _ in range(n_host):
                host.step(act_np)
            dt = time.perf_counter() - t1
            step_api["host_vector_env_us_per_step"] = dt / n_host * 1e6
            step_api["host_vector_env_env_steps_per_s"] = E * n_host / dt
            step_api["host_vector_env_zero_copy"] = host.zero_copy
            staged = ea.HostVectorEnv(loc, copy=True, zero_copy=False)     # round 4's form: an asynchronous copy each way
            for _ in range(20):
                staged.step(act_np)
            t1 = time.perf_counter()
            for _ in range(n_host):
                staged.step(act_np)
            step_api["host_vector_env_staged_us_per_step"] = (time.perf_counter() - t1) / n_host * 1e6
        except Exception as exc:  # noqa: BLE001
            step_api["host_vector_env_error"] = f"{type(exc).__name__}: {exc}"[:160]

    # BASELINE configs 3 and 5 (one GPU's shard) and the 524 288-env per-step run, a few episode sweeps each: in the driver's line
    # (VERDICT r04 item 2).  Only beside the default single-GPU C2 rollout line; every entry is a measurement of its own batch.
    side = None
    if (rank == 0 and world == 1 and args.workload == "c2" and args.mode == "rollout" and not args.envs and not args.no_side_workloads
            and not args.no_step_api and not do_gather):        # (--no-step-api = "the headline kernel only": profiling and A/B runs)
        progress("side workloads")
        side = side_workloads(args, device)

    persistent = "persistent" in loc.kernel_variant(args.mode)
    in_flight = 1 if persistent else max(1, loc.own_streams)
    if rank == 0:
        tr = load_traffic(args.traffic_json, f"{args.workload}:{args.mode}", loc.kernel_variant(args.mode), csrc_sha16())
        traffic = (tr["hbm_bytes_per_env_step"] * E * inner + tr["chain_record_bytes_per_env_launch"] * E
                   if tr["hbm_bytes_per_env_step"] is not None else None)                 # per env-step, scaled to one launch
        traffic_src, valu_insts = tr["source"], tr["valu"]
        step_s = sweep_s / steps_per_sweep                         # the episode-average cost of one step of the whole batch
        value = total_envs / step_s
        if not do_gather:
            gather_desc = ""
        elif args.mode == "step":
            gather_desc = ", RCCL all-gather of the step outputs after every step"
        else:
            coll = "RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + ", host-staged: testing aid"
            what = {"obs": f"observation batch ({coll})", "slab": f"[obs|reward|flags] records ({coll})",
                    "direct": "observation batch (copy-engine peer writes over hipIpc, no CU-resident collective kernel)",
                    "peer": "observation batch (one peer-store kernel, evac_peer_gather, into hipIpc-mapped buffers; no library collective)"}[form]
            how = ("issued after the NEXT chunk's launch, inside the timed block (double-buffered; block b carries block b-1's gather)"
                   if lag == 1 else "each chunk gathered as soon as it is computed, inside its block")
            gather_desc = f", all-gather of the {what} per {inner}-step chunk on a side stream, {how}"
        out = {
            "metric": "env-steps/s (agent-updates/s) at n=60x4096 envs" if args.workload == "c2" else f"env-steps/s ({args.workload})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "n_pedestrians": n_ped, "envs_per_gpu": E, "total_envs": total_envs,
                       "obs": wrap_kw, "actions": "RandomAgent U(-1,1)^2 drawn on device (Philox4x32-10)",
                       "mode": args.mode, "steps_per_launch": inner, "launches_per_block": len(sizes),
                       "timing": "whole episode sweeps of 2000/K launches of exactly K steps each, issued back to back (no host sync, no "
                                 "barrier collective inside; N > 1: the gather of chunk j-1 on a second hardware queue under chunk j, a "
                                 "chunk's buffer reused only once the host has seen its gather finished -- a query, the host at most "
                                 "gather_buffers - 1 chunks ahead of the device); per sweep: barrier + synchronize (all "
                                 "ranks, device idle) -> t0 -> the launches (+ gathers) -> the rank drains its compute and comm streams "
                                 "-> t1, max over ranks; ms_per_step = median sweep / 2000 = the episode-average step; `blocks`: the "
                                 "per-block view (a device-idle sync around every K-step launch) as a diagnostic",
                       "sweeps": {"timed": sweeps, "launches_per_sweep": launches_per_sweep, "steps_per_sweep": steps_per_sweep,
                                  "wall_ms": [x * 1e3 for x in sweep_wall], "hip_event_ms": [x * 1e3 for x in sweep_dev],
                                  "gpu_ms_timed_total": sum(sweep_dev) * 1e3,
                                  "value_min_median_max": [total_envs * steps_per_sweep / max(sweep_wall), value,
                                                           total_envs * steps_per_sweep / min(sweep_wall)]},
                       "timing_method": "r06: median of whole-episode sweeps (11 by default for K-step blocks), back-to-back launches, followed "
                                        "by the same sweeps for --sustain-seconds of GPU time (`sustained`; `value` = the settled median when "
                                        "the two differ by more than 2 %); "
                                        "roofline.kernel_ms_per_launch = median sweep (HIP events) / launches -- with gathers in the "
                                        "sweeps: the mean launch of a replayed sweep WITHOUT gathers (the sweep's own figure is "
                                        "roofline.launch_ms_with_gather)",
                       "gather": (form if gather_rollout else None), "gather_requested": (args.gather if do_gather else None),
                       "gather_schedule": (args.gather_schedule if gather_rollout else None),
                       "gather_buffers": (nbuf if gather_rollout else None),
                       "ranks_joined": dist.get_world_size() if use_dist else 1,
                       "collective_backend": (dist.get_backend() if use_dist else None),
                       "parallelism": f"env-sharded x{world}" + gather_desc,
                       "max_timesteps": EPISODE, "autoreset": True},
            "agent_updates_per_s": value * n_ped,
            "blocks": blocks_info,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": tr["note"],
                         "salu_wave_insts_per_env_step": tr["salu"], "lds_wave_insts_per_env_step": tr["lds"],
                         "kernel": loc.kernel_variant(args.mode),
                         "kernels_in_flight": in_flight, "round_ms": kernel_s * 1e3,
                         "profile_check": {
                             "sweeps_all": len(sweep_dev) + len(sustain_dev),
                             "mean_period_ms_all_sweeps": (sum(sweep_dev) + sum(sustain_dev)) * 1e3 / max(1, (len(sweep_dev) + len(sustain_dev)) * launches_per_sweep),
                             "expected_kernel_trace_avg_duration_ms": (launches_per_sweep if persistent else max(1, loc.own_streams)) * (sum(sweep_dev) + sum(sustain_dev)) * 1e3
                                                                      / max(1, (len(sweep_dev) + len(sustain_dev)) * launches_per_sweep),
                             "note": ("for a reader with the kernel trace of this command: ONE PERSISTENT rollout kernel per sweep carries all its "
                                      f"{launches_per_sweep} calls, so the DURATION of a resident kernel is the sweep (expected_kernel_trace_avg_duration_ms) and "
                                      "the call period is that over the calls of a sweep.  The trace lists further kernels of the same name: the FINISHER that "
                                      "every evac_join enqueues behind the resident kernel (a few us: it finds nothing left to run) and the single-call kernels "
                                      "of the warm-up and of the diagnostics -- so the per-name AVERAGE of a kernel_stats table is not the sweep; "
                                      "tools/chain_timeline.py separates them (profiles/*_timeline.txt: the whole-sweep kernels' mean and median duration).  "
                                      "mean_period_ms_all_sweeps is over every timed sweep (headline + sustained), round_ms the settled median `value` uses") if persistent else
                                     "for a reader with the kernel trace of this command: a CHAINED launch is enqueued behind its queue's previous launch and "
                                     "ends two launch periods later (its queue carries every second launch), so the trace's average DURATION of the "
                                     "rollout kernel is kernels_in_flight x the launch period, and the period itself is the trace's start-to-start "
                                     "distance of consecutive launches (tools/chain_trace.sh prints both); mean_period_ms_all_sweeps is over "
                                     "every timed sweep (headline + sustained), round_ms the settled median `value` uses"},
                         "part_stream_ms_per_launch": part_ms,
                         "kernel_ms_per_launch": kernel_s * 1e3, "kernel_launches_timed": back_to_back if (uniform and not gather_rollout) else len(full),
                         "launch_ms_with_gather": (sweep_launch_s * 1e3 if gather_rollout else None),
                         "kernel_ms_per_launch_event_pairs": sum(full) / max(1, len(full)) * 1e3,
                         "kernel_ms_per_launch_dense": kernel_dense_s * 1e3,
                         "frac_dense": bytes_per_launch / kernel_dense_s / 1e9 / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_env_step": bytes_per_env_step,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "equivalent_bandwidth": True,
                         "binding_resource": "valu",
                         "valu_issue_frac": (valu_insts * 64.0 * E * inner / kernel_s / VALU_ISSUE_PEAK) if valu_insts else None,
                         "valu_wave_insts_per_env_step": valu_insts,
                         "valu_issue_peak_lane_ops_per_s": VALU_ISSUE_PEAK,
                         "valu_pipe_frac": (valu_insts * 64.0 * E * inner / kernel_s / VALU_PIPE_PEAK) if valu_insts else None,
                         "valu_pipe_peak_lane_ops_per_s": VALU_PIPE_PEAK,
                         "hbm_traffic_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "note": "achieved = algorithmic bytes / launch time (task contract): an EQUIVALENT-bandwidth figure, the state "
                                 "stays in registers between the steps of a launch and the counter-measured HBM traffic (traffic, "
                                 "hbm_traffic_frac) is a few per cent of it; the resource that binds is the VALU: valu_issue_frac = "
                                 "SQ_INSTS_VALU per env-step (rocprofv3 --pmc, profiles/traffic.json) x 64 lanes x env-steps per launch / "
                                 "kernel time against 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3e12 lane-ops/s (one wave64 VALU "
                                 "instruction per 4 cycles per SIMD: about what ONE wave per SIMD can issue); valu_pipe_frac = the same "
                                 "against 78.6e12 (one per 2 cycles: the vector pipe's own rate, reached only with four or more waves of "
                                 "a SIMD issuing -- a packed-f32 instruction takes 4 of those cycles and counts as one here)"},
            "step_api": step_api,
        }
        if sustained is not None:
            out["sustained"] = sustained
        ro = loc.resolved_options()
        out["config"]["rollout_form"] = {
            "requested": args.rollout_form, "parts": loc.num_parts, "chain": ro.chain, "own_streams": loc.own_streams,
            "note": ("evac_options_t.chain = 2: ONE PERSISTENT KERNEL per join -- the first rollout call after a join starts the rollout kernel on a stream the handle "
                     "owns, and every call (that one included) is a 64-byte COMMAND (steps, slab, episode records) the host writes into a ring in device "
                     "memory; the resident kernel runs the call's steps, writes its slab and takes the next command with the state still in registers: no "
                     "launch boundary, no prologue, no hand-off between calls; evac_join -- here: the end of every sweep -- posts STOP, the waves store their "
                     "state and the kernel ends.  Every call still runs exactly its K steps into its own slab.  roofline.achieved = the batch's algorithmic "
                     "bytes per call / the call PERIOD (round_ms = kernel_ms_per_launch = sweep / calls, kernel start and STOP included); a kernel trace shows "
                     "one kernel per sweep whose duration is the sweep") if ro.chain == 2 else
                    ("evac_options_t.chain = 1: rollout call g goes to stream g & 1 of two streams the handle owns; a launch waits PER ENV, on the "
                     "device, for the launch before it (a generation word in an exchange record per env) and its queue waits until every workgroup "
                     "of that launch has started; at most two launches overlap; a sweep is closed by evac_join on the timing stream.  "
                     "roofline.achieved = the batch's algorithmic bytes per launch / the launch PERIOD (round_ms = kernel_ms_per_launch = sweep / "
                     "launches); a kernel's own duration, as a kernel trace shows it, is about two periods (part_stream_ms_per_launch = each "
                     "stream's time per launch of the whole chain, i.e. half its own launches' period)") if ro.chain == 1 else
                    (("evac_options_t.parts = 2: every rollout call issues envs [0, E/2) and [E/2, E) as two kernels on two streams the "
                      "handle owns; a sweep is closed by evac_join on the timing stream; roofline.achieved = the whole batch's "
                      "algorithmic bytes per round of launches / the round's period (round_ms = kernel_ms_per_launch), "
                      "part_stream_ms_per_launch = the launch period of each of the two streams") if loc.num_parts > 1 else
                     "one kernel per rollout call on the launching stream" +
                     (" (an N > 1 line runs plain launches beside its gathers -- the gather of chunk j - 1 under chunk j needs CUs that a resident or "
                      "a waiting chained kernel would hold --; the like-for-like N = 1 line for a scaling ratio is `--rollout-form one`, not the "
                      "N = 1 default's persistent kernel)" if world > 1 else ""))}
        out["cpu_baseline"] = cpu_base
        if side is not None:
            out["workloads"] = side
        if gather_info is not None:
            out["gather_report"] = gather_info
        print(json.dumps(out), flush=True)
    env.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def dry_run(args, rank, world, total_envs, envs_per_gpu, K, W, inner, per_sweep, sweeps):
    """The multi-rank control flow without a GPU: rendezvous over gloo, the block structure of main() -- the SAME
    ChunkPipeline, with CPU stand-ins for the launches and the real (gloo) all-gather -- and the packed all-gather in
    global env order.  Rank 0 prints the trace of its first blocks: tests/test_bench_launcher_cpu.py asserts that no
    barrier lies inside a timed region and that, pipelined, the gather of chunk j-1 is issued after the launch of chunk j.
    Reports no throughput."""
    import torch
    import torch.distributed as dist

    from evacuation_amd.distributed import all_gather_envs, gathered_view, shard_range
    if world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(3)
    do_gather = world > 1 and not args.no_gather
    if do_gather and os.environ.get("EVAC_BENCH_FAIL_GATHER") == str(rank):     # testing aid: a collective that fails on one rank
        print(f"bench.py: rank {rank}: the all-gather failed (injected); no {world}-GPU result", file=sys.stderr)
        raise SystemExit(5)
    sizes = chunk_sizes(K, inner, args.gather_schedule, do_gather)
    lag = 0 if args.gather_schedule == "split" else 1
    off, n_local = shard_range(total_envs, rank, world)
    gid = torch.arange(off, off + n_local, dtype=torch.float32)
    bufs = {}
    trace, state = [], {"ok": True, "gathers": 0}

    nbuf = args.buffers if args.buffers > 0 else (4 if (do_gather and lag == 1) else 2)

    stamps = {}                         # chunk -> host-clock stamps of its stand-in launch and gather (ms): the multi-rank line's report, dry

    def launch(j, t):
        t0_ = time.perf_counter() * 1e3
        bufs[j % nbuf] = gid[None, :, None] + 1000.0 * j + torch.zeros((min(t, 4), n_local, 9))   # chunk j's "outputs"
        stamps.setdefault(j, {}).update(l0=t0_, l1=time.perf_counter() * 1e3)

    def gather(j, t):
        t0_ = time.perf_counter() * 1e3
        g, _ = all_gather_envs(bufs[j % nbuf])
        full = gathered_view(g)
        state["ok"] = state["ok"] and bool((full[0, :, 0] == torch.arange(total_envs, dtype=torch.float32) + 1000.0 * j).all())
        state["gathers"] += 1
        stamps.setdefault(j, {}).update(g0=t0_, g1=time.perf_counter() * 1e3)
        return j

    pipe = ChunkPipeline(launch, gather if do_gather else None, lambda tok: None, lambda: None, lambda: None, lag=lag, trace=trace, nbuf=nbuf)
    w_done = 0
    while w_done < W:
        t = min(sizes[0], W - w_done)
        pipe.run_block([t])
        w_done += t
    pipe.drain()
    wall = []
    n_timed = min(per_sweep * sweeps, 4)
    for b in range(n_timed):
        if world > 1:
            dist.barrier()
        trace.append(("barrier",))
        trace.append(("t0", b))
        t0 = time.perf_counter()
        pipe.run_block(sizes)
        pipe.drain()
        wall.append(time.perf_counter() - t0)
        trace.append(("t1", b))
    pipe.flush()
    if world > 1:
        dist.barrier()
        tt = torch.tensor(wall, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    # the multi-rank line's account of its gather, assembled exactly as main() does it -- one dict per rank through all_gather_object,
    # the form decided by choose_gather from a probe result -- from the stand-ins' host-clock stamps (no GPU: the probe says so)
    probe = {"ok": False, "stage": "skipped", "error": "dry run: no GPU", "world": world}
    form, alt_form = choose_gather(args.gather, probe)
    report = None
    if do_gather:
        chunks = []
        for j in sorted(stamps):
            c = {"l0": stamps[j]["l0"], "l1": stamps[j]["l1"]}
            g = stamps.get(j - lag)
            if lag == 1 and g is not None and "g0" in g:
                c["g0"], c["g1"] = g["g0"], g["g1"]
            chunks.append(c)
        mine = {"rank": rank, form: gather_report(form, world, min(sizes[0], 4) * n_local * 9 * 4, [], chunks, None)}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        report = {"timed_form": form, "requested": args.gather, "alternative_form": alt_form, "peer_store_probe": probe, "per_rank": everyone,
                  "auto_fell_back": None, "alternative_dropped": None}     # (the keys of the real line: no peer form is built in a dry run)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "gather_report": report, "ranks_joined": dist.get_world_size() if world > 1 else 1,
                          "steps": K, "warmup": W, "total_envs": total_envs, "envs_per_gpu": envs_per_gpu,
                          "gather_in_global_env_order": state["ok"], "gathers": state["gathers"], "blocks": len(wall),
                          "launches_per_block": len(sizes), "chunk_sizes": sizes, "gather_schedule": args.gather_schedule,
                          "trace": [list(ev) for ev in trace], "value": None,
                          "self_launched": os.environ.get("EVAC_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if state["ok"] else 4


def _main_with_fallback():
    """`--rollout-form auto` takes the persistent kernel on one GPU.  Should a run in that form ever end with the library's ABORTED code (the
    chained launches' bounded waits are the only source left; the persistent kernel has none), the line is measured again with chained
    launches, in a CHILD process (this one has an aborted handle; no exec once the GPU is in use)."""
    try:
        return main()
    except Exception as exc:  # noqa: BLE001
        from evacuation_amd import _lib
        auto = "--rollout-form" not in " ".join(sys.argv[1:])
        if not (isinstance(exc, _lib.EvacError) and exc.code == _lib.ERR_TEAM_ABORTED and auto and int(os.environ.get("WORLD_SIZE", "1")) == 1):
            raise
        print(f"bench.py: the run was aborted ({exc}); measuring with chained launches instead", file=sys.stderr, flush=True)
        import subprocess
        return subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--rollout-form", "chain"]).returncode


if __name__ == "__main__":
    raise SystemExit(_main_with_fallback())
