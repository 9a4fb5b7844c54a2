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
