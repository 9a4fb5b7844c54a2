"""Which share of the env-launches of an episode runs packed (csrc/evac_packed.h), by steps per launch.  Run on the GPU box."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

E = 4096
for T in (20, 50, 100):
    env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
    env.reset()
    prev, line = 0, []
    for k in range(2000 // T):
        env.rollout(T)
        if (k + 1) * T % 200 == 0:
            torch.cuda.synchronize()
            now = int(env.pack_stats[0])
            line.append(f"{(now - prev) / (E * (200 // T)):.2f}")
            prev = now
    print(f"T={T:3d}: packed share per 200 steps of the episode: " + " ".join(line))
    env.close()
