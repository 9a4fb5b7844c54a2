"""bench.py's own multi-rank launcher, driven on CPU (gloo, world 2) through `--dry-run`:
`python bench.py --gpus N` without torchrun must start N ranks itself, report n_gpus == N only when N ranks
joined, and exit non-zero when a rank is missing (the reference's fan-out being replaced is
gym.vector.SyncVectorEnv, /root/reference/src/agents/rpo_agent.py:123-126)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_plain_invocation_spawns_its_own_ranks():
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                   # rank 0 prints ONE line
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["self_launched"] is True
    assert d["steps"] == 20 and d["warmup"] == 5
    assert d["total_envs"] == 2 * d["envs_per_gpu"]    # weak scaling: the per-rank shard is fixed
    assert d["gather_in_global_env_order"] is True


def test_the_multi_rank_line_carries_every_ranks_account_of_the_gather():
    """`gather_report.per_rank` is gathered from ALL ranks (all_gather_object), the form that is timed is decided
    by choose_gather from the probe -- here, without a GPU, the probe says no, so `--gather auto` falls back to the RCCL form."""
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run", "--gather", "auto"])
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json_lines(r.stdout)[0]["gather_report"]
    assert rep["requested"] == "auto" and rep["timed_form"] == "obs" and rep["alternative_form"] is None
    assert rep["peer_store_probe"]["ok"] is False and rep["peer_store_probe"]["stage"] == "skipped"
    assert rep["auto_fell_back"] is None and rep["alternative_dropped"] is None       # (keys of the real line; nothing was built here)
    assert [x["rank"] for x in rep["per_rank"]] == [0, 1]
    for x in rep["per_rank"]:
        mine = x["obs"]
        assert mine["world"] == 2 and mine["chunks_instrumented"] >= 4 and mine["gather_ms_under_compute"] > 0
        assert mine["bytes_received_per_chunk"] == mine["bytes_per_link_per_chunk"] > 0       # one peer
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run", "--no-gather"])
    assert json_lines(r.stdout)[0]["gather_report"] is None


def _blocks_of(trace):
    """Split a dry-run trace into the event lists of its timed blocks (between t0 and t1) and what lies outside."""
    blocks, outside, cur = [], [], None
    for ev in trace:
        if ev[0] == "t0":
            cur = []
        elif ev[0] == "t1":
            blocks.append(cur)
            cur = None
        elif cur is not None:
            cur.append(tuple(ev))
        else:
            outside.append(tuple(ev))
    return blocks, outside


def test_timed_block_has_no_barrier_and_carries_the_previous_blocks_gather():
    """per rank, opening barrier -> t0 -> launches -> drain compute and gather -> local t1; the closing
    barrier is outside the timed region; with one launch per block (the driver's --steps 20) the gather of block b-1 is issued
    after the launch of block b, inside block b (double-buffered), and drained before t1."""
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_lines(r.stdout)[0]
    assert d["gather_schedule"] == "pipelined" and d["chunk_sizes"] == [20] and d["launches_per_block"] == 1
    blocks, outside = _blocks_of(d["trace"])
    assert len(blocks) == d["blocks"] == 4
    assert ("barrier",) in outside                           # the brackets exist ...
    launched = []
    for evs in blocks:
        kinds = [e[0] for e in evs]
        assert "barrier" not in kinds                        # ... but never inside a timed region
        assert kinds[-1] == "drain"                          # the block ends with the rank draining compute + gather
        (j,) = [e[1] for e in evs if e[0] == "launch"]
        (g,) = [e[1] for e in evs if e[0] == "gather"]
        assert g == j - 1                                    # the previous block's outputs ...
        assert evs.index(("gather", g)) > evs.index(("launch", j))   # ... gathered under this block's compute
        launched.append(j)
    assert launched == list(range(launched[0], launched[0] + 4))
    assert launched[0] >= 1                                  # the warm-up primed the pipeline: block 0 has a gather to carry
    # every chunk is gathered exactly once, the last one by the untimed flush
    gathered = [e[1] for e in d["trace"] if e[0] == "gather"]
    assert gathered == list(range(launched[-1] + 1)) and d["gathers"] == len(gathered)
    assert d["gather_in_global_env_order"] is True


def test_split_schedule_gathers_each_half_inside_its_block():
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "10", "--dry-run", "--gather-schedule", "split"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_lines(r.stdout)[0]
    assert d["chunk_sizes"] == [10, 10] and d["launches_per_block"] == 2
    blocks, _ = _blocks_of(d["trace"])
    for evs in blocks:
        kinds = [e[0] for e in evs]
        assert "barrier" not in kinds and kinds[-1] == "drain"
        ls = [e[1] for e in evs if e[0] == "launch"]
        gs = [e[1] for e in evs if e[0] == "gather"]
        assert len(ls) == 2 and gs == ls                     # both halves gathered inside the block that computed them
        assert evs.index(("gather", ls[0])) < evs.index(("launch", ls[1]))   # the first half's gather runs under the second half


def test_a_failing_collective_is_a_hard_error():
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"], {"EVAC_BENCH_FAIL_GATHER": "1"})
    assert r.returncode != 0
    assert not json_lines(r.stdout)                    # never an N-GPU value without the gather traffic
    assert "all-gather failed" in r.stderr
    # ... unless the run asked for independent shards
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run", "--no-gather"], {"EVAC_BENCH_FAIL_GATHER": "1"})
    assert r.returncode == 0 and json_lines(r.stdout)[0]["gathers"] == 0


def test_chunk_pipeline_orders_buffer_reuse_after_the_gather_that_read_it():
    sys.path.insert(0, ROOT)
    import bench
    log = []
    pipe = bench.ChunkPipeline(launch=lambda j, t: log.append(("L", j, t)), gather=lambda j, t: log.append(("G", j, t)) or ("tok", j),
                               wait_gather=lambda tok: log.append(("W",) + tok), drain_compute=lambda: log.append(("DC",)),
                               drain_gather=lambda: log.append(("DG",)), lag=1)
    pipe.run_block([5, 5, 3])
    pipe.drain()
    pipe.flush()
    # chunk 2 reuses buffer 0, which gather 0 read: the compute stream waits for it first
    assert log == [("L", 0, 5), ("L", 1, 5), ("G", 0, 5), ("W", "tok", 0), ("L", 2, 3), ("G", 1, 5), ("DC",), ("DG",),
                   ("G", 2, 3), ("DC",), ("DG",)]
    # three buffers: launch 3 reuses buffer 0 and waits for gather 0 -- which had launch 2's whole duration, not launch 1's tail
    log.clear()
    pipe = bench.ChunkPipeline(launch=lambda j, t: log.append(("L", j, t)), gather=lambda j, t: log.append(("G", j, t)) or ("tok", j),
                               wait_gather=lambda tok: log.append(("W",) + tok), drain_compute=lambda: log.append(("DC",)),
                               drain_gather=lambda: log.append(("DG",)), lag=1, nbuf=3)
    pipe.run_block([5, 5, 5, 5, 5])
    pipe.flush()
    assert log == [("L", 0, 5), ("L", 1, 5), ("G", 0, 5), ("L", 2, 5), ("G", 1, 5), ("W", "tok", 0), ("L", 3, 5), ("G", 2, 5),
                   ("W", "tok", 1), ("L", 4, 5), ("G", 3, 5), ("G", 4, 5), ("DC",), ("DG",)]
    assert bench.chunk_sizes(20, 100, "pipelined", True) == [20] and bench.chunk_sizes(2000, 100, "pipelined", True) == [100] * 20
    assert bench.chunk_sizes(20, 100, "split", True) == [10, 10] and bench.chunk_sizes(20, 100, "split", False) == [20]
    assert bench.chunk_sizes(250, 100, "split", True) == [100, 100, 50]


def test_a_missing_rank_is_a_hard_error():
    r = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"], {"EVAC_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not json_lines(r.stdout)                    # no result line for a run that lost a rank
    assert "rank failed" in r.stderr


def test_world_size_mismatch_is_a_hard_error():
    # under a launcher (WORLD_SIZE set) the rank count must equal --gpus: never report a 1-rank number as N GPUs
    r = run_bench(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = run_bench(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29999"})
    assert r.returncode != 0


def test_torchrun_style_environment_is_honoured():
    # what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` provides, without the launcher
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = {k: v for k, v in os.environ.items()}
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "40", "--warmup", "8", "--dry-run"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    lines = json_lines(outs[0][0])
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["self_launched"] is False
    assert not json_lines(outs[1][0])                  # only rank 0 reports


def test_block_plan_tiles_the_episode():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.block_plan(20, 0, 0) == (100, 11)     # the driver's --steps 20: 100 blocks per episode sweep, 11 sweeps
    assert bench.block_plan(2000, 0, 0) == (1, 20)     # default: a block is a whole episode, 20 of them
    assert bench.block_plan(500, 0, 0)[0] == 4 and bench.block_plan(500, 0, 0)[0] * bench.block_plan(500, 0, 0)[1] >= 20
    per, sw = bench.block_plan(20, 0, 7)
    assert per * sw == 7
    avg, info = bench.summarize_blocks([1.0, 3.0, 1.2, 2.8], [5, 1005, 5, 1005], 2, 10)
    assert abs(avg - (1.1 + 2.9) / 2) < 1e-12 and info["dense"]["episode_phase"] == 5 and info["mid_episode"]["episode_phase"] == 1005
    # K does not divide the episode: blocks are grouped by their ACTUAL phase, never by block index
    ph = [(5 + b * 300) % 2000 for b in range(14)]
    avg, info = bench.summarize_blocks([1.0] * 14, ph, 7, 300)
    assert info["distinct_phases"] == len(set(ph)) == 14 and abs(avg - 1.0) < 1e-12
