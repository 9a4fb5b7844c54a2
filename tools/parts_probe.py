"""evac_options_t.parts: the headline batch (N = 60 x 4096 envs, 20 steps per launch, whole episodes) as ONE handle issuing one
kernel per rollout (parts = 1) or two half-batch kernels on its own two streams (parts = 2 / automatic).  GPU box:
python tools/parts_probe.py [steps_per_launch] [sweeps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SWEEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
E = 4096
envs = {p: ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=0x5EED0001, options=ea.KernelOptions(parts=p)) for p in (1, -1)}
res = {}
for rep in range(SWEEPS):
    for p, env in envs.items():          # interleaved: the same box, the same clocks
        if rep == 0:
            env.reset()
            out = {"slab": torch.empty((T, E, env.obs_dim + 3), device=env.device), "episode_stats": torch.zeros((T, E, env.stats_words), device=env.device)}
            env._launch = env.rollout_launcher(T, out)
            for _ in range(2000 // T):
                env._launch()
            env.join(); torch.cuda.synchronize()
        n = 2000 // T
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            env._launch()
        env.join()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        res.setdefault(p, []).append(us)
        print(f"parts={env.num_parts} sweep {rep}: {us:6.2f} us per {T}-step round of {E} envs = {E * T / us * 1e6:.3e} env-steps/s  [{env.kernel_variant()}]", flush=True)
for p, v in res.items():
    v = sorted(v)
    print(f"parts option {p:2d}: median {v[len(v) // 2]:6.2f} us per round, min {v[0]:.2f}, max {v[-1]:.2f}")
a, b = sorted(res[1]), sorted(res[-1])
print(f"=> two parts / one kernel: {a[len(a) // 2] / b[len(b) // 2]:.3f}x")
