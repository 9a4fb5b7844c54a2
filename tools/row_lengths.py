"""C5 (N=1024): per env, how many tile entries the needed rows of the cell-list sweep visit (3 x 3 cells around the row's cell)
-- the serial length of the slowest lane bounds the sweep.  Run on the GPU box."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

E, n = 32, 1024
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box"), num_envs=E, seed=1)
env.reset()
for phase in range(0, 2000, 200):
    s = env.get_state()
    pos, st = s["pos"].cpu().numpy(), s["status"].cpu().numpy()
    out = []
    for e in range(E):
        mv = (st[e] >= 1) & (st[e] <= 3)
        cx = np.clip(((pos[e, :, 0] + 1) * 8).astype(int), 0, 15); cy = np.clip(((pos[e, :, 1] + 1) * 8).astype(int), 0, 15)
        grid = np.zeros((18, 18), int)
        np.add.at(grid, (cx[mv] + 1, cy[mv] + 1), 1)
        nb = sum(grid[1 + dx:17 + dx, 1 + dy:17 + dy] for dx in (-1, 0, 1) for dy in (-1, 0, 1))
        rows = st[e] == 1
        cand = nb[cx[rows], cy[rows]] if rows.any() else np.zeros(1, int)
        out.append((int(rows.sum()), int(cand.max()), float(cand.mean()), int((cand > 64).sum())))
    out = np.array(out)
    worst = out[:, 1].argmax()
    print(f"t={phase:5d} rows mean {out[:,0].mean():6.1f} | longest row per env: mean {out[:,1].mean():6.1f} max {out[:,1].max():4.0f} (env {worst}: {out[worst,0]:.0f} rows, mean row {out[worst,2]:.1f}, rows > 64: {out[worst,3]:.0f}) | rows > 64 per env mean {out[:,3].mean():.1f}")
    env.rollout(200); torch.cuda.synchronize()
