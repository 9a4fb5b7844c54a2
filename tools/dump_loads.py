"""Per-env (viscek, follower, exiting) counts at several episode phases of the bench workloads -> gpurun_out/loads_<tag>.npz
(input of tools/schedule_study.py, which compares env-to-SIMD dealing rules offline).  Run on the GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

out = {}
for tag, n, E, wrap in (("c2", 60, 4096, dict(positions="grav")), ("c3", 256, 1024, dict(positions="grav"))):
    env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True), ea.EnvWrappersConfig(**wrap), num_envs=E, seed=1)
    env.reset()
    rows = []
    for phase in range(0, 2000, 50):
        st = env.get_state()["status"].cpu().numpy()
        rows.append(np.stack([(st == 1).sum(1), (st == 2).sum(1), (st == 3).sum(1)], axis=1))
        env.rollout(50)
    torch.cuda.synchronize()
    out[tag] = np.stack(rows).astype(np.int16)        # [phase, env, (viscek, follower, exiting)]
    env.close()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/loads.npz", **out)
print({k: v.shape for k, v in out.items()})
