"""Design study for windowed sweeps: with the tile sorted by x strip (S strips over the room) and the needed
rows dealt to waves 64 at a time in tile order, how many columns would a wave's UNION window hold, against the whole tile that the
all-pairs sweeps of Wave<4> (C3) and Team<K> (C5) visit today?  Per episode phase: the mean env, and the heaviest (most rows x
columns) env of the batch -- which is the one that ends a launch.  Run on the GPU box:  python tools/window_union.py c3|c5 [S]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n, E, wrap = {"c3": (256, 1024, dict(positions="grav")), "c5": (1024, 32, dict(positions="rel", statuses="ohe", type="Box")),
              "c2": (60, 4096, dict(positions="grav"))}[wl]
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                              ea.EnvWrappersConfig(**wrap), num_envs=E, seed=1)
env.reset()
margin = int(np.ceil(0.1 / (2.0 / S)))          # strips a neighbour can be away
step = 50
print(f"{wl}: N={n} E={E} strips={S} (width {2.0 / S:.4f}, margin {margin} strips); work = rows-of-64 x columns visited, per env-step")
for t in range(0, 2000, step):
    st = env.get_state()
    pos, status = st["pos"].cpu().numpy(), st["status"].cpu().numpy()
    now, win, rows_l, cols_l = [], [], [], []
    for e in range(E):
        mv = (status[e] >= 1) & (status[e] <= 3)
        x = pos[e, mv, 0]
        need = status[e][mv] == 1
        strip = np.clip(((x + 1.0) * (S / 2.0)).astype(int), 0, S - 1)
        order = np.argsort(strip, kind="stable")
        strip_s, need_s = strip[order], need[order]
        n_cols = len(strip_s)
        start = np.searchsorted(strip_s, np.arange(S + 1))      # first tile slot of every strip
        rows = np.nonzero(need_s)[0]
        w = 0
        for k in range(0, len(rows), 64):
            blk = strip_s[rows[k:k + 64]]
            lo, hi = max(blk.min() - margin, 0), min(blk.max() + margin, S - 1)
            w += start[hi + 1] - start[lo]
        now.append(((len(rows) + 63) // 64) * n_cols)
        win.append(w)
        rows_l.append(len(rows)); cols_l.append(n_cols)
    now, win = np.array(now, float), np.array(win, float)
    h = int(np.argmax(now))
    print(f"t={t:5d} rows mean {np.mean(rows_l):6.1f} cols mean {np.mean(cols_l):6.1f} | all-pairs work mean {now.mean():8.0f} windowed {win.mean():8.0f} "
          f"({win.sum() / max(now.sum(), 1):.2f}) | heaviest env: rows {rows_l[h]:4d} cols {cols_l[h]:4d} work {now[h]:7.0f} windowed {win[h]:7.0f} ({win[h] / max(now[h], 1):.2f})"
          f" | max windowed {win.max():7.0f}")
    env.rollout(step)
    torch.cuda.synchronize()
