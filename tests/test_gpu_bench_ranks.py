"""bench.py's multi-rank timed loop on real HIP streams: two ranks sharing the ONE GPU of the test box (gloo rendezvous --
RCCL refuses two ranks on one device), rollout kernels on the compute stream, the gather of the previous chunk on the comm
stream.  Checks the control flow end to end (launcher, pipeline, gather buffers, per-block max over ranks, one JSON line);
the copy-engine gather (--gather direct: hipIpc-mapped peer buffers) is exercised the same way, both "peers" living on this
GPU.  RCCL itself first runs on the driver's multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=600, env_extra=None, expect_rc0=True):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(EVAC_BENCH_FORCE_DEVICE="0", EVAC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--blocks", "6",
                        "--envs", "512", "--no-step-api", "--sustain-seconds", "0.02"] + extra, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    if not expect_rc0:
        assert r.returncode != 0
        return r.stderr
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


@pytest.mark.parametrize("extra", [[], ["--gather-schedule", "split"], ["--gather", "slab"], ["--gather", "direct"], ["--gather", "peer"], ["--gather", "auto"]])
def test_two_ranks_on_one_gpu(extra):
    d = _run(extra)
    # the line explains its gather -- with TWO real processes: the staged peer-store probe (hipIpc mappings of each
    # other's buffers, a checked store pattern, an agreement over the host after every stage) and one account per rank
    rep = d["gather_report"]
    assert rep["peer_store_probe"]["ok"] is True and rep["peer_store_probe"]["world"] == 2 and rep["peer_store_probe"]["devices"] == [0, 0]
    assert [x["rank"] for x in rep["per_rank"]] == [0, 1]
    form = rep["timed_form"]
    assert form == {"slab": "slab", "direct": "direct", "peer": "peer", "auto": "peer"}.get(extra[1] if extra[:1] == ["--gather"] else "", "obs") == d["config"]["gather"]
    for x in rep["per_rank"]:
        #  every rank's own clock over the headline sweeps: a slow rank shows; the job's value is the slowest rank's
        # (`value` itself may be the settled median of the sustained sweeps: only the headline sweeps' own figures are compared)
        assert x["headline"]["env_steps_per_s_of_this_rank"] > 0 and len(x["headline"]["sweep_wall_ms_this_rank"]) == d["config"]["sweeps"]["timed"]
        assert x["headline"]["value_if_every_rank_were_this_one"] >= 0.999 * d["config"]["sweeps"]["value_min_median_max"][0]
        assert x[form]["world"] == 2 and x[form]["gather_ms_alone"] > 0 and x[form]["bytes_received_per_chunk"] == x[form]["bytes_per_link_per_chunk"] > 0
    if "split" not in extra and form in ("obs", "peer"):
        assert rep["alternative_form"] == ("peer" if form == "obs" else "obs") and rep["alternative_value"] > 0
    assert d["n_gpus"] == 2 and d["config"]["ranks_joined"] == 2 and d["config"]["total_envs"] == 1024
    assert d["value"] > 0 and d["blocks"]["timed_blocks"] == 6
    assert d["config"]["launches_per_block"] == (2 if "split" in extra else 1)
    assert "all-gather" in d["config"]["parallelism"]
    assert d["roofline"]["frac"] > 0


def test_a_rank_that_cannot_build_the_peer_form_takes_nobody_down():
    """The peer-store gather is built in stages that every rank closes with an agreement over the host: when ONE rank cannot
    export its buffer (injected), the alternative form of a line is dropped, `--gather auto` falls back to RCCL's form before
    anything is timed, and an explicit `--gather peer` is a hard error on every rank -- no rank is left waiting in a collective."""
    fail = {"EVAC_BENCH_FAIL_PEER_BUILD": "1"}
    d = _run([], env_extra=fail)                                     # default: RCCL's form timed, the peer form its alternative
    rep = d["gather_report"]
    assert rep["timed_form"] == "obs" and rep["alternative_form"] is None and "alternative_value" not in rep
    assert "rank 1" in rep["alternative_dropped"] and "injected" in rep["alternative_dropped"] and rep["auto_fell_back"] is None
    assert d["value"] > 0 and d["config"]["ranks_joined"] == 2
    d = _run(["--gather", "auto"], env_extra=fail)
    rep = d["gather_report"]
    assert rep["peer_store_probe"]["ok"] is True                     # (the probe passed: it is the build of the real buffers that failed)
    assert rep["timed_form"] == "obs" == d["config"]["gather"] and "rank 1" in rep["auto_fell_back"]
    assert d["value"] > 0
    err = _run(["--gather", "peer"], env_extra=fail, expect_rc0=False, timeout=300)
    assert "--gather peer" in err and "injected" in err
