#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_pmc.sh output directory.
HBM bytes per launch = FETCH_SIZE[KiB] * 1024 * 2  (gfx950 reports half of a coalesced read stream,
MI355X_MICROARCH.md 'HBM') + WRITE_SIZE[KiB] * 1024; separate --pmc passes, mean over the dispatches
of the timed kernel, divided by the env-steps of one launch: bench.py scales it back to whatever launch shape it
runs (`roofline.traffic`).
Every entry also records WHAT it was measured on -- the kernel variant string of the handle (evac_kernel_variant, asked on the
GPU box: run this script there) and the hash of the kernel sources (bench.csrc_sha16) -- and bench.py withholds the counters when
either differs from what it runs (bench.load_traffic).
usage: make_traffic_json.py <pmc_dir> <workload:mode> <envs> <steps_per_launch> [...]"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def kernel_variant(key, envs, chain=2):
    """The variant string bench.py will see for this workload (needs the GPU: the team / CU-wide choice depends on the device)."""
    import evacuation_amd as ea
    workload, mode = key.split(":")
    n_ped, _, wrap_kw, _ = bench.WORKLOADS[workload]
    cfg = ea.EnvConfig(number_of_pedestrians=n_ped, is_new_exiting_reward=True, is_new_followers_reward=True,
                       intrinsic_reward_coef=0.0, max_timesteps=bench.EPISODE)
    # (the options bench.py's headline takes on one GPU: chained launches where the library offers them -- the C2 rollout; the side
    # workloads and the per-step runs are plain handles)
    opts = ea.KernelOptions(chain=chain) if mode == "rollout" else None
    env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(**wrap_kw), num_envs=int(envs), options=opts)
    env.reset()
    if mode == "rollout":
        env.rollout(2)                      # (the team kernels' residency check runs at the first rollout)
    v = env.kernel_variant(mode)
    env.close()
    return v


out_path = os.path.join(ROOT, "profiles", "traffic.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data = {k: v for k, v in data.items() if "hbm_bytes_per_env_step" in v}     # drop entries of the round-1 format
# (SQ_INSTS_*: wave-instructions per dispatch, divided by the env-steps of a launch -- bench.py's valu_issue_frac)
args = sys.argv[1:]
for d, key, envs, inner in zip(args[0::4], args[1::4], args[2::4], args[3::4]):
    kern = "k_rollout" if key.endswith(":rollout") else "k_step"
    vals = {"FETCH_SIZE": [], "WRITE_SIZE": [], "SQ_INSTS_VALU": [], "SQ_INSTS_SALU": [], "SQ_INSTS_LDS": []}
    for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row.get("Kernel_Name", "") and row["Counter_Name"] in vals:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    fetch = sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
    write = sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
    per_launch = fetch * 1024 * 2 + write * 1024
    data[key] = {"kernel": kern, "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write,
                 "envs": int(envs), "steps_per_launch": int(inner),
                 "hbm_bytes_per_launch": per_launch, "hbm_bytes_per_env_step": per_launch / (int(envs) * int(inner)),
                 "note": "FETCH_SIZE doubled per the gfx950 correction; dispatches averaged: %d" % len(vals["FETCH_SIZE"]),
                 "valu_wave_insts_per_env_step": (sum(vals["SQ_INSTS_VALU"]) / len(vals["SQ_INSTS_VALU"]) / (int(envs) * int(inner))) if vals["SQ_INSTS_VALU"] else None,
                 "salu_wave_insts_per_env_step": (sum(vals["SQ_INSTS_SALU"]) / len(vals["SQ_INSTS_SALU"]) / (int(envs) * int(inner))) if vals["SQ_INSTS_SALU"] else None,
                 "lds_wave_insts_per_env_step": (sum(vals["SQ_INSTS_LDS"]) / len(vals["SQ_INSTS_LDS"]) / (int(envs) * int(inner))) if vals["SQ_INSTS_LDS"] else None,
                 "source": os.path.relpath(d, ROOT), "kernel_variant": kernel_variant(key, envs), "csrc_sha16": bench.csrc_sha16()}
    # rollouts are counted in their PLAIN launches (tools/final_run.sh passes --rollout-form one to the counter passes: rocprofv3 --pmc
    # runs every dispatch alone, and a chained launch waits for its predecessor); where bench.py's headline chains its launches the
    # entry says so and carries the exchange record's size (evac_common.h Xchg<T>: 20 T + 256 bytes, read once and written once)
    plain = kernel_variant(key, envs, chain=0) if key.endswith(":rollout") else data[key]["kernel_variant"]
    if plain != data[key]["kernel_variant"]:
        lanes = 256 if "4 waves/env" in plain else 64
        data[key]["counted_variant"] = plain
        # (chained launches move an exchange record per env and launch; a persistent kernel keeps the state in registers: nothing to add)
        data[key]["chain_record_bytes_per_env_launch"] = 2 * (20 * lanes + 256) if "chained" in data[key]["kernel_variant"] else 0
    print(key, data[key])
json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
