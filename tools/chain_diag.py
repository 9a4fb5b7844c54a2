"""Diagnostics of the chained launches: sweeps of back-to-back 20-step launches of the headline batch; on an error the abort line
of the workspace says what the first wave to give up waited for.  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
from evacuation_amd import _lib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n, E = 60, 4096
cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
b = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=0x5EED0001, options=ea.KernelOptions(chain=1))
b.reset()
out = {"slab": torch.empty((T, E, b.obs_dim + 3), device=b.device), "episode_stats": torch.zeros((T, E, b.stats_words), device=b.device)}
go = b.rollout_launcher(T, out)
nb = int(b.lib.evac_workspace_bytes(b._h))
n_l = 2000 // T
times = []
try:
    for k in range(sweeps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        t0 = time.perf_counter()
        for _ in range(n_l):
            go()
        t_host = time.perf_counter() - t0
        b.join(); e1.record(); torch.cuda.synchronize()
        times.append((e0.elapsed_time(e1) * 1e3 / n_l, t_host * 1e6 / n_l))
        if b.team_error(sync=False):
            raise RuntimeError("error word set")
    print("ok:", " ".join(f"{g:.1f}/{h:.1f}" for g, h in times[::6]), "(GPU us per round / host us per launch call)")
except Exception as exc:
    torch.cuda.synchronize()
    ws = b.workspace[nb - 256:nb].view(torch.int32).cpu().tolist()
    nz = [(i, v) for i, v in enumerate(ws) if v]
    print(f"FAILED in sweep {len(times)}: {type(exc).__name__}: {str(exc)[:120]}; non-zero int32 words of the workspace tail (index, value): {nz}; launches so far ~{len(times) * n_l}")
