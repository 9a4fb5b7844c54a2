"""C2 (N = 60 x 4096 envs): where over the episode do 20-step launches lose against 100-step launches?  From the same state at
every 100th step: five 20-step launches back to back against one 100-step launch (HIP events, median of 5 repeats).  GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea  # noqa: E402

E = 4096
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                              ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=1)
print(env.kernel_variant(), flush=True)
env.reset()
env.rollout(2000)            # the batch's second episode on: the benchmark's state
torch.cuda.synchronize()
outs = {T: {"slab": torch.empty((T, E, 9), device="cuda"), "episode_stats": torch.zeros((T, E, 10), device="cuda")} for T in (20, 100)}
launch = {T: env.rollout_launcher(T, outs[T]) for T in (20, 100)}


def timed(T, n, state, reps=5):
    res = []
    for _ in range(reps):
        env.set_state(**state)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            launch[T]()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(res))


tot = {20: 0.0, 100: 0.0}
print("   t0   5 x 20 steps [us]   1 x 100 steps [us]   difference per 20-step launch [us]")
for b in range(20):
    state = env.get_state()
    a = timed(20, 5, state)
    c = timed(100, 1, state)
    tot[20] += a
    tot[100] += c
    print(f"{b * 100:5d} {a:12.1f} {c:18.1f} {(a - c) / 5:18.2f}", flush=True)
    env.set_state(**state)
    launch[100]()
    torch.cuda.synchronize()
print(f"episode: {tot[20] / 2000:.3f} us per step with 20-step launches, {tot[100] / 2000:.3f} with 100-step launches")
