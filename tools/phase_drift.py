"""Why do the headline's sweeps slow down over the first half second (4.44 -> 4.92 ms)?  Clocks, or the batch itself: envs whose
episode TERMINATES (every pedestrian escaped) before the truncation at 2000 steps are reset early, so over many sweeps the envs'
episode phases spread out and every launch carries some freshly reset (dense) envs.  Prints, every 25 sweeps, the sweep time, the
spread of Time.now over the batch at the sweep boundary and the in-kernel clock ratio is left to tools/stamps.sh.  GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

T, E = 20, 4096
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=0x5EED0001, options=ea.KernelOptions(parts=parts))
env.reset()
out = {"slab": torch.empty((T, E, env.obs_dim + 3), device=env.device), "episode_stats": torch.zeros((T, E, env.stats_words), device=env.device)}
go = env.rollout_launcher(T, out)
for sw in range(400):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(2000 // T):
        go()
    env.join()
    e1.record()
    torch.cuda.synchronize()
    if sw % 25 == 0 or sw < 4:
        now = env.clock[:, 0].float()
        resets = env.clock[:, 1].float()
        mv = (env.status != 4).sum(dim=1).float()
        print(f"sweep {sw:3d}: {e0.elapsed_time(e1):6.3f} ms   Time.now at the boundary: min {int(now.min())} median {int(now.median())} max {int(now.max())}, "
              f"envs not at phase 0: {int((now != 0).sum())}; resets per env: min {int(resets.min())} max {int(resets.max())}; "
              f"pedestrians still inside: mean {mv.mean():.1f} max {int(mv.max())}", flush=True)
