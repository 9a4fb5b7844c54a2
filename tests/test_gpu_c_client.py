"""The C ABI is usable without torch: examples/c_api_demo.cpp (raw hipMalloc buffers) must produce the very
same rollout slab as the Python host (FNV-1a checksum over the bytes)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for chunk in (np.frombuffer(b, dtype=np.uint8),):
        for x in chunk.tolist():
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_client_matches_python_host():
    import evacuation_amd as ea
    from evacuation_amd import build
    exe = os.path.join(ROOT, "examples", "c_api_demo")
    if not os.path.exists(exe):
        subprocess.run([build.hipcc_path(), "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "examples", "c_api_demo.cpp"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "evacuation_amd"), "-levac",
                        "-Wl,-rpath," + os.path.join(ROOT, "evacuation_amd"), "-o", exe], check=True)
    E, N, T, seed = 64, 60, 24, 0x1234
    out = subprocess.run([exe, str(E), str(N), str(T), hex(seed)], check=True, capture_output=True, text=True).stdout
    rec = json.loads(out.strip().splitlines()[-1])
    cfg = ea.EnvConfig(number_of_pedestrians=N, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
    env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=seed)
    env.reset()
    slab = env.rollout(T)["slab"].cpu().numpy()
    assert rec["obs_dim"] == 6 and rec["num_envs"] == E
    assert int(rec["slab_fnv1a"], 16) == _fnv1a(slab.tobytes())
