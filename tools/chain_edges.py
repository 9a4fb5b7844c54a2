"""What a launch adds to the heaviest env's chain under chained launches (diagnostic build: hipcc ... -DEVAC_STEP_TIMES
-o tools/ab_libs/libevac_steptimes.so, loaded through EVAC_LIB).  As tools/launch_edges.py, for a handle with options.chain = 1 (or
0: argv[3]): the 16 waves of workgroups 0 and 100 note the 100 MHz clock at entry, after the deal + wait + record load + first action
block, at the top of their first step, after their last step and behind their record stores.  GPU box."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea  # noqa: E402
from evacuation_amd import _lib  # noqa: E402

lib = _lib.load()
E = 4096
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
chain = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                              ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=1, options=ea.KernelOptions(chain=chain))
print(env.kernel_variant(), f"T = {T}, after {warm} sweeps of 2000 steps (phases mixed)")
env.reset()
out = {"slab": torch.empty((T, E, env.obs_dim + 3), device=env.device), "episode_stats": torch.zeros((T, E, env.stats_words), device=env.device)}
launch = env.rollout_launcher(T, out)
for _ in range(warm):
    for _ in range(2000 // T):
        launch()
    env.join(); torch.cuda.synchronize()
n = 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    launch()
env.join()
e1.record()
torch.cuda.synchronize()
print(f"{n} launches back to back: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch")
buf = (C.c_ulonglong * (64 * 2 * 16 * 8))()
assert lib.evac_debug_launch_marks(buf) == 0
m = np.array(buf[:], dtype=np.int64).reshape(64, 2, 16, 8)
order = np.argsort(m[:, 1, 0, 0])            # launches by entry time of workgroup 100's wave 0
m = m[order][-(n - 2):]
for g, name in ((1, "workgroup 100"), (0, "workgroup 0 (the lightest envs; deals the launch after next)")):
    a = m[:, g].astype(np.float64) * 0.01    # us
    entry, loop, done, exit_, init, act, state = a[..., 0], a[..., 1], a[..., 2], a[..., 3], a[..., 5], a[..., 6], a[..., 7]
    med = lambda x: float(np.median(x))      # noqa: E731
    print(f"-- {name}: medians over {len(a)} launches [us]")
    print(f"   period: first entry(g+1) - first entry(g)                          {med(np.diff(entry.min(axis=1))):6.2f}")
    print(f"   entry -> F::init done                                  mean wave {med((init - entry).mean(axis=1)):6.2f}   slowest {med((init - entry).max(axis=1)):6.2f}")
    print(f"   F::init -> deal, WAIT, record load, first action block  mean wave {med((act - init).mean(axis=1)):6.2f}   slowest {med((act - init).max(axis=1)):6.2f}   shortest {med((act - init).min(axis=1)):6.2f}")
    print(f"   -> top of the first step                               mean wave {med((loop - act).mean(axis=1)):6.2f}   slowest {med((loop - act).max(axis=1)):6.2f}")
    print(f"   the {T} steps                                          mean wave {med((done - loop).mean(axis=1)):6.2f}   slowest {med((done - loop).max(axis=1)):6.2f}")
    print(f"   last step -> behind the stores                         mean wave {med((exit_ - done).mean(axis=1)):6.2f}   slowest {med((exit_ - done).max(axis=1)):6.2f}")
    print(f"   first entry -> last exit (the workgroup's life)                    {med(exit_.max(axis=1) - entry.min(axis=1)):6.2f}")
    # the chain of the workgroup's LAST wave: its exit in launch g against the last first-step of launch g + 1 (the same slot carries
    # nearly the same envs from launch to launch: the load order changes slowly)
    print(f"   last exit(g) -> last wave's first step(g+1)                        {med(loop.max(axis=1)[1:] - exit_.max(axis=1)[:-1]):6.2f}")
    print(f"   last wave's first step(g) -> last wave's first step(g+1)           {med(np.diff(loop.max(axis=1))):6.2f}")
    print(f"   entry(g+1) of this workgroup - last exit(g)  (< 0: resident before its predecessor is done) {med(entry.min(axis=1)[1:] - exit_.max(axis=1)[:-1]):6.2f}")
env.close()
