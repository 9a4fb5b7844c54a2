"""Fixed cost of a rollout launch: kernel time (HIP events around back-to-back launches) against steps per launch.
Run on the GPU box."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

E = 4096
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
print(env.kernel_variant())
env.reset()
env.rollout(1000); torch.cuda.synchronize()          # mid-episode, loads spread out
state = env.get_state()
res = []
for T in (1, 2, 5, 10, 20, 50, 100):
    out = {"slab": torch.empty((T, E, 9), device="cuda"), "episode_stats": torch.zeros((T, E, 10), device="cuda")}
    launch = env.rollout_launcher(T, out)
    env.set_state(**state); torch.cuda.synchronize()
    n = max(4, 400 // T)
    for _ in range(3): launch()
    env.set_state(**state); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    res.append((T, us))
    print(f"T={T:4d}: {us:8.2f} us per launch, {us / T:6.3f} us per step")
T = np.array([r[0] for r in res], float); U = np.array([r[1] for r in res])
b, a = np.polyfit(T[2:], U[2:], 1)
print(f"fit over T >= 5: {a:.2f} us fixed + {b:.3f} us per step")
