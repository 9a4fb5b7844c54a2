"""GPU: RCCL runs under this code before a multi-GPU box does it for the first time.  backend "nccl" IS
RCCL on ROCm; a communicator of ONE rank executes the same calls as one of eight -- init_process_group(device_id=),
all_gather_into_tensor on the comm stream, the events between the streams, record_stream -- only the wire is missing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_sharded_env_gathers_through_a_one_rank_rccl_communicator():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_world1_worker.py")], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("extra", [[], ["--gather", "slab"], ["--gather", "peer"], ["--gather", "auto"], ["--gather-schedule", "split"], ["--device-wait", "--buffers", "2"]])
def test_bench_force_gather_runs_the_pipeline_with_rccl(extra):
    """bench.py --force-gather: world size 1, backend nccl, the SAME ChunkPipeline as an N-GPU run -- the gather of chunk j - 1
    on the comm stream under chunk j, buffer-reuse waits, the drain of both streams -- and a bench line that says so."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-gather", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
           "--no-step-api", "--sweeps", "1", "--sustain-seconds", "0.02"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "nccl" and line["config"]["ranks_joined"] == 1
    assert "all-gather" in line["config"]["parallelism"] and line["value"] > 1e8
    assert line["config"]["gather_schedule"] == ("split" if "split" in extra else "pipelined")
    # the line explains its gather: the probe, the form timed, and per rank what the gather cost
    rep = line["gather_report"]
    assert rep["peer_store_probe"]["ok"] is True and rep["peer_store_probe"]["stage"] == "done", rep["peer_store_probe"]
    want = {"slab": "slab", "peer": "peer", "auto": "peer"}.get(extra[1] if extra[:1] == ["--gather"] else "", "obs")
    assert rep["timed_form"] == want == line["config"]["gather"] and len(rep["per_rank"]) == 1
    mine = rep["per_rank"][0][want]
    for k in ("gather_ms_alone", "launch_ms_plain", "bytes_per_link_per_chunk", "versions", "env"):
        assert mine[k] is not None, k
    assert mine["versions"]["rccl"] and mine["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    if "split" not in extra:                                 # pipelined: chunk j - 1's gather rides under launch j
        for k in ("gather_ms_under_compute", "launch_ms_with_gather", "gather_started_before_rollout_ended", "link_GBps_under_compute"):
            assert mine[k] is not None and mine[k] >= 0, k
        assert mine["chunks_instrumented"] >= 90
        if want in ("obs", "peer"):                          # ... and the OTHER form ran two plain sweeps + an instrumented one
            other = "peer" if want == "obs" else "obs"
            assert rep["alternative_form"] == other and rep["alternative_value"] > 1e8
            assert rep["per_rank"][0][other]["gather_ms_under_compute"] is not None
    assert line["roofline"]["launch_ms_with_gather"] > 0 and line["roofline"]["kernel_ms_per_launch"] > 0


def test_side_stream_really_runs_beside_the_compute_stream():
    """HIP deals its streams onto a few hardware queues; two streams that share one serialise (round 4: every gather of the forced
    one-rank run sat on the rollout's queue).  side_stream() hands out a stream that was timed against the compute stream."""
    import torch
    from evacuation_amd.distributed import side_stream
    dev = torch.device("cuda:0")
    for _ in range(6):                       # use up a few queue assignments first, as a process group would
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            torch.zeros(8, device=dev)
    compute = torch.cuda.current_stream(dev)
    comm = side_stream(dev, beside=compute)
    assert comm.cuda_stream != compute.cuda_stream
    spin = 400_000

    def timed(second):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(compute)
        torch.cuda._sleep(spin)
        if second is not None:
            with torch.cuda.stream(second):
                torch.cuda._sleep(spin)
            compute.wait_stream(second)
        e1.record(compute)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    timed(comm)
    one, pair = min(timed(None) for _ in range(3)), min(timed(comm) for _ in range(3))
    assert pair < 1.5 * one, (one, pair)
