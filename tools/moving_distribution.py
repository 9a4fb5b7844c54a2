"""Distribution over envs of the pedestrians still moving (the pair-loop length) at several episode phases, and what it
means for a launch that lasts as long as its slowest SIMD (4 random envs per SIMD).  Run on the GPU box."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

E, n = 4096, 60
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset()
rng = np.random.default_rng(0)
for phase in range(0, 2000, 100):
    st = env.get_state()["status"].cpu().numpy()
    moving = ((st >= 1) & (st <= 3)).sum(1)
    cost = 210 + 5 * ((moving + 3) // 4 * 4)                      # VALU instructions per step (model)
    perm = rng.permutation(E).reshape(-1, 4)
    simd = cost[perm].sum(1)
    srt = np.sort(cost)
    bal = (srt[:E // 4] + srt[E // 4:E // 2][::-1] + srt[E // 2:3 * E // 4] + srt[3 * E // 4:][::-1])
    print(f"t={phase:5d} moving mean {moving.mean():5.1f} p5 {np.percentile(moving, 5):4.0f} p95 {np.percentile(moving, 95):4.0f} max {moving.max():3d} | "
          f"cost mean {cost.mean():6.1f} max {cost.max():4d} | SIMD sum/4: random max {simd.max() / 4:6.1f} ({simd.max() / 4 / cost.mean():.3f}x mean), "
          f"balanced max {bal.max() / 4:6.1f} ({bal.max() / 4 / cost.mean():.3f}x)")
    env.rollout(100)
    torch.cuda.synchronize()

# status composition of the moving pedestrians (which rows the neighbour sum really needs: with enslaving_degree = 1 a
# follower's own Vicsek mean is multiplied by 0, area.py:139-142)
for n2, E2, wrap in ((60, 4096, dict(positions="grav")), (256, 1024, dict(positions="grav")), (1024, 32, dict(positions="rel", statuses="ohe", type="Box"))):
    env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n2, is_new_exiting_reward=True), ea.EnvWrappersConfig(**wrap), num_envs=E2, seed=1)
    env.reset()
    for phase in range(0, 2000, 200):
        st = env.get_state()["status"].cpu().numpy()
        v, f, x = (st == 1).sum(1), (st == 2).sum(1), (st == 3).sum(1)
        print(f"N={n2} t={phase:5d} viscek mean {v.mean():6.1f} max {v.max():4d} zero in {100.0 * (v == 0).mean():5.1f}% | follower mean {f.mean():6.1f} max {f.max():4d} | exiting mean {x.mean():5.1f}")
        env.rollout(200)
        torch.cuda.synchronize()
    env.close()
