"""How many candidates would an x-sorted window scan visit?  (design study for the pair loop)
For each env: K = max over FOLLOWER/VISCEK pedestrians of the number of moving pedestrians whose x lies
within 0.1 on one side.  An x-sorted scan needs K iterations (2 pairs each); all-pairs needs N."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evacuation_amd as ea

n, E = int(sys.argv[1]) if len(sys.argv) > 1 else 60, 4096
cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, max_timesteps=2000)
env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset()
for t in (0, 100, 300, 600, 1000, 1500, 1999):
    done = getattr(env, "_t", 0)
    if t > done:
        env.rollout(t - done); env._t = t
    st = env.get_state()
    x = st["pos"][..., 0]; s = st["status"]
    moving = s != 4
    fv = (s == 1) | (s == 2)
    dx = x[:, None, :] - x[:, :, None]                     # [E, i, j] = xj - xi
    mj = moving[:, None, :]
    right = ((dx > 0) & (dx < 0.1) & mj).sum(-1)
    left = ((dx < 0) & (dx > -0.1) & mj).sum(-1)
    k = torch.maximum(right, left) * fv
    K = k.max(dim=1).values.float()
    mean_pairs = ((dx.abs() < 0.1) & mj).sum(-1).float()[fv].mean()
    print(f"t={t:5d}  K mean={K.mean():.1f} p50={K.median():.0f} p90={K.quantile(0.9):.0f} p99={K.quantile(0.99):.0f} max={K.max():.0f}"
          f"  | mean candidates per fv lane={mean_pairs:.1f}  | status counts V/F/E/X = "
          f"{(s==1).float().sum(1).mean():.1f}/{(s==2).float().sum(1).mean():.1f}/{(s==3).float().sum(1).mean():.1f}/{(s==4).float().sum(1).mean():.1f}")
