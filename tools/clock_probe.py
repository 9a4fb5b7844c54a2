"""Shader clock while a workload runs: a one-wave kernel on a side stream reads s_memtime (shader clock) and s_memrealtime (100 MHz)
at both ends of ~2 ms; printed every 50 ms next to the sweep times of chained / plain rollouts.  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
chain = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
E, T = 4096, 20
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                              ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=0x5EED0001, options=ea.KernelOptions(chain=chain))
env.reset()
out = {"slab": torch.empty((T, E, env.obs_dim + 3), device=env.device), "episode_stats": torch.zeros((T, E, env.stats_words), device=env.device)}
go = env.rollout_launcher(T, out)
import subprocess
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        s = [l.split(":")[-1].strip() for l in o.splitlines() if "sclk" in l or "Package Power" in l]
        return " ".join(s)
    except Exception as e:
        return str(e)
t_last = time.time()
for sw in range(sweeps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        go()
    env.join(); e1.record()
    if sw % 40 == 39:
        torch.cuda.synchronize()
        print(f"sweep {sw}: {e0.elapsed_time(e1) * 10:.2f} us per round   rocm-smi (idle moment): {smi()}", flush=True)
torch.cuda.synchronize()
print(env.kernel_variant(), "error word", env.team_error())
