"""How long does a free-running f32 GPU episode track the f64 oracle?  Same reset draws, actions and
per-pedestrian noise on both sides; reports, per env, the first step with |pos_gpu - pos_oracle| > 1e-5 or a
status mismatch.  (Teacher-forced single steps always agree to ~1e-7; this measures chaotic error growth.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import evacuation_amd as ea
from oracle import evac_oracle as O

n, E, T = 60, 24, 600
rng = np.random.default_rng(0)
p = O.OracleParams(number_of_pedestrians=n, is_new_exiting_reward=True, max_timesteps=10**6)
cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, max_timesteps=10**6)
env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav"), num_envs=E, autoreset=False)
draws = rng.uniform(-1, 1, (E, n, 4)).astype(np.float32)
env.reset(draws=draws)
states = [O.env_reset(p, draws[e, :, 0:2].astype(np.float64), draws[e, :, 2:4].astype(np.float64)) for e in range(E)]
first = np.full(E, T + 1)
worst_before = np.zeros(E)
for t in range(T):
    act = rng.uniform(-1, 1, (E, 2)).astype(np.float32)
    nz = rng.uniform(-0.1, 0.1, (E, n)).astype(np.float32)
    env.step(act, noise=nz)
    st = env.get_state()
    pos = st["pos"].cpu().numpy(); status = st["status"].cpu().numpy()
    for e in range(E):
        O.env_step(p, states[e], act[e], nz[e].astype(np.float64))
        if first[e] > T:
            err = np.abs(pos[e] - states[e].pos).max()
            if err > 1e-5 or (status[e] != states[e].status).any():
                first[e] = t + 1
            else:
                worst_before[e] = max(worst_before[e], err)
tracked = np.minimum(first, T)
print(f"free-running f32 GPU vs f64 oracle, N={n}, {E} envs, {T} steps, identical draws/actions/noise:")
print(f"  first divergence (>1e-5 or status mismatch): min {tracked.min()}  median {int(np.median(tracked))}  max {tracked.max()}"
      f"  | envs that tracked all {T} steps: {(first > T).sum()} of {E}")
print(f"  max |pos error| while tracking: median {np.median(worst_before):.2e}  max {worst_before.max():.2e}")
