"""C5 shard (N = 1024 x 32 envs): the time of each 100-step launch of an episode for every family the batch could run on --
teams of 16 / 8 / 4 / 2 CUs per env, one 16-wave workgroup per env (cell list).  Would a launch late in the episode (15 rows, 140
columns left per env) be better served by a smaller team?  GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea  # noqa: E402


def run(team, E=32, N=1024, inner=100, episodes=3):
    env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=N, max_timesteps=2000),
                                  ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box"), num_envs=E, seed=1,
                                  options=ea.KernelOptions(team=int(team)))
    env.reset()
    out = env.rollout(inner)
    n = 2000 // inner
    for _ in range(n - 1):
        env.rollout(inner, out=out)              # a whole episode: the batch is at t = 0 of its second one afterwards
    torch.cuda.synchronize()
    ms = np.zeros((episodes, n))
    for ep in range(episodes):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for b in range(n):
            env.rollout(inner, out=out)
            ev[b + 1].record()
        torch.cuda.synchronize()
        ms[ep] = [ev[b].elapsed_time(ev[b + 1]) for b in range(n)]
    name = env.kernel_variant("rollout")
    env.close()
    return name, np.median(ms, axis=0) * 1000.0 / inner


def main():
    rows = {}
    for team in (16, 8, 4, 2, 0):
        name, us = run(team)
        rows[team] = us
        print(f"EVAC_TEAM={team:2d} {name}: episode mean {us.mean():6.2f} us per step", flush=True)
    print("\nus per step by 100-step launch of the episode (median of 3 episodes):")
    print("t0    " + "".join(f"{('team ' + str(k)) if k else 'cells':>9}" for k in rows))
    for b in range(len(rows[8])):
        print(f"{b * 100:5d} " + "".join(f"{rows[k][b]:9.2f}" for k in rows))
    best = np.minimum.reduce([rows[k] for k in rows])
    print(f"\nbest family per launch: {best.mean():.2f} us per step against team 8's {rows[8].mean():.2f}")


if __name__ == "__main__":
    main()
