"""Sub-wave kernels (2 or 4 envs per wave) vs one wave per env, small rooms.  python tools/subwave_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evacuation_amd as ea

for n, E in ((10, 16384), (16, 16384), (30, 8192), (32, 8192)):
    row = []
    for flag in ("0", "1"):
        env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True),
                                      ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=1, options=ea.KernelOptions(subwave=int(flag)))
        env.reset()
        out = env.rollout(100)
        env.rollout(100, out=out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            env.rollout(100, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row.append(E * 1000 / dt)
        env.close()
    print(f"N={n:3d} E={E:6d}: one wave per env {row[0]:.3e} env-steps/s | sub-wave {row[1]:.3e} env-steps/s | x{row[1]/row[0]:.2f}")
