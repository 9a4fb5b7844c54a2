"""bench.py's guard on profiles/traffic.json: the PMC counters that ride along in the bench line --
`roofline.traffic`, `valu_wave_insts_per_env_step` ... -- were measured in an earlier profiling run; they are reported only
when that run profiled the kernel variant and the kernel SOURCES this run has built, and withheld (None + a note) otherwise."""
import json
import os

import bench


def _entry(variant, sha):
    return {"c2:rollout": {"kernel": "k_rollout", "hbm_bytes_per_env_step": 183.0, "envs": 4096, "steps_per_launch": 20,
                           "valu_wave_insts_per_env_step": 240.0, "salu_wave_insts_per_env_step": 69.0, "lds_wave_insts_per_env_step": 24.0,
                           "source": "gpurun_out/x", "kernel_variant": variant, "csrc_sha16": sha}}


def test_matching_entry_is_reported(tmp_path):
    sha = bench.csrc_sha16()
    p = tmp_path / "traffic.json"
    p.write_text(json.dumps(_entry("k_rollout_default_config<V>", sha)))
    tr = bench.load_traffic(str(p), "c2:rollout", "k_rollout_default_config<V>", sha)
    assert tr["hbm_bytes_per_env_step"] == 183.0 and tr["valu"] == 240.0 and tr["salu"] == 69.0 and tr["lds"] == 24.0
    assert tr["note"] is None and "4096 envs x 20 steps" in tr["source"]


def test_mismatching_entry_nulls_the_fields(tmp_path):
    sha = bench.csrc_sha16()
    p = tmp_path / "traffic.json"
    p.write_text(json.dumps(_entry("k_rollout_default_config<V>", "0123456789abcdef")))          # other kernel sources
    tr = bench.load_traffic(str(p), "c2:rollout", "k_rollout_default_config<V>", sha)
    assert tr["hbm_bytes_per_env_step"] is None and tr["valu"] is None and tr["salu"] is None and "withheld" in tr["note"]
    p.write_text(json.dumps(_entry("k_rollout<other family>", sha)))                              # another kernel variant
    tr = bench.load_traffic(str(p), "c2:rollout", "k_rollout_default_config<V>", sha)
    assert tr["hbm_bytes_per_env_step"] is None and "withheld" in tr["note"]
    legacy = _entry(None, None)                                                                   # an entry of the round-3 format
    del legacy["c2:rollout"]["kernel_variant"], legacy["c2:rollout"]["csrc_sha16"]
    p.write_text(json.dumps(legacy))
    assert bench.load_traffic(str(p), "c2:rollout", "k_rollout_default_config<V>", sha)["valu"] is None
    assert bench.load_traffic(str(p), "c3:rollout", "v", sha)["note"].startswith("no entry")
    assert bench.load_traffic(str(tmp_path / "absent.json"), "c2:rollout", "v", sha)["note"].startswith("no counter file")


def test_source_hash_follows_the_kernel_sources(tmp_path):
    root = tmp_path / "r"
    (root / "evacuation_amd" / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    for name in bench.CSRC_FILES:
        src = os.path.join(bench.ROOT, "evacuation_amd", "csrc", name)
        (root / "evacuation_amd" / "csrc" / name).write_bytes(open(src, "rb").read())
    (root / "include" / "evac.h").write_bytes(open(os.path.join(bench.ROOT, "include", "evac.h"), "rb").read())
    assert bench.csrc_sha16(str(root)) == bench.csrc_sha16()
    with open(root / "evacuation_amd" / "csrc" / "evac_device.h", "ab") as f:
        f.write(b"\n// touched\n")
    assert bench.csrc_sha16(str(root)) != bench.csrc_sha16()


# ---- the multi-rank line explains its gather: the pure parts, on the CPU ----
def test_gather_auto_is_decided_by_the_peer_store_probe():
    assert bench.choose_gather("auto", {"ok": True}) == ("peer", "obs")           # probe passed everywhere: the peer-store kernel, RCCL as the alternative
    assert bench.choose_gather("auto", {"ok": False, "stage": "map", "error": "rank 3: hipIpcOpenMemHandle"}) == ("obs", None)
    assert bench.choose_gather("auto", None) == ("obs", None)
    assert bench.choose_gather("obs", {"ok": True}) == ("obs", "peer")            # RCCL timed (the default), the peer kernel for two extra sweeps
    assert bench.choose_gather("obs", {"ok": False}) == ("obs", None)
    assert bench.choose_gather("peer", {"ok": True}) == ("peer", "obs")
    assert bench.choose_gather("slab", {"ok": True}) == ("slab", None) and bench.choose_gather("direct", {"ok": True}) == ("direct", None)


def test_gather_report_from_synthetic_timestamps():
    """Eight ranks, 20-step chunks of 4096 envs x 6 observation words: 1.97 MB per link and chunk.  Launch j runs [100 j, 100 j + 45] us,
    the gather of chunk j - 1 starts 6 us into it and takes 40 us: co-resident, 4 us of it in the launch's own time, the
    launch 45 us against 42 plain."""
    per_link = 20 * 4096 * 6 * 4
    chunks = [{"l0": 0.0, "l1": 0.045}]                                           # the first chunk carries no gather
    for j in range(1, 11):
        l0 = 0.1 * j
        chunks.append({"l0": l0, "l1": l0 + 0.045, "g0": l0 + 0.006, "g1": l0 + 0.046})
    rep = bench.gather_report("obs", 8, per_link, [0.030, 0.031, 0.029], chunks, 0.042, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"}, versions={"rccl": "2.x"})
    assert rep["chunks_instrumented"] == 10 and rep["world"] == 8 and rep["form"] == "obs"
    assert rep["bytes_per_link_per_chunk"] == per_link and rep["bytes_received_per_chunk"] == 7 * per_link
    assert abs(rep["gather_ms_alone"] - 0.030) < 1e-12 and abs(rep["gather_ms_under_compute"] - 0.040) < 1e-9
    assert abs(rep["launch_ms_with_gather"] - 0.045) < 1e-9 and rep["launch_ms_plain"] == 0.042
    assert abs(rep["launch_slowdown_under_gather"] - 0.045 / 0.042) < 1e-9
    assert rep["gather_started_before_rollout_ended"] == 1.0
    assert abs(rep["overlap_ms"] - 0.039) < 1e-9 and abs(rep["gather_tail_after_launch_ms"] - 0.001) < 1e-9
    assert abs(rep["link_GBps_under_compute"] - per_link / 0.040e-3 / 1e9) < 1e-6 and rep["link_GBps_alone"] > rep["link_GBps_under_compute"]
    assert set(rep["checks_design_estimates"]) <= set(rep)                        # every estimate of DESIGN 6 names a field of this report
    # a gather that only starts when the rollout has ended (not co-resident): the report says so
    late = [{"l0": 0.1 * j, "l1": 0.1 * j + 0.045, "g0": 0.1 * j + 0.046, "g1": 0.1 * j + 0.08} for j in range(1, 6)]
    rep = bench.gather_report("obs", 8, per_link, [0.03], late, 0.042)
    assert rep["gather_started_before_rollout_ended"] == 0.0 and rep["overlap_ms"] == 0.0
    # no instrumented chunk (world 1 without --force-gather never gets here; defensive): None, not a crash
    rep = bench.gather_report("peer", 1, per_link, [], [], None)
    assert rep["gather_ms_under_compute"] is None and rep["launch_slowdown_under_gather"] is None and rep["bytes_received_per_chunk"] == 0


def test_sweeps_default_to_eleven_for_k_step_blocks():
    assert bench.block_plan(20, 0, 0) == (100, 11)              # the driver's --steps 20: 1100 launches, ~50 ms of GPU time
    assert bench.block_plan(2000, 0, 0) == (1, 20)              # whole-episode blocks: at least 20 of them
    assert bench.block_plan(20, 3, 0) == (100, 3)
