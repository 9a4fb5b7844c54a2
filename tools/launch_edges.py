"""What a launch boundary of the C2 rollout is made of (diagnostic build: hipcc ... -DEVAC_STEP_TIMES -o tools/ab_libs/libevac_steptimes.so,
loaded through EVAC_LIB).  The 16 waves of workgroups 0 and 100 note the 100 MHz clock at kernel entry, at the top of their first step,
after their last step and behind their state write-back; 40 launches of T steps go back to back.  Printed per workgroup: entry -> first
step (prologue: dispatch order, kernel arguments, permutation, state loads, first Philox block), the steps, last step -> exit
(epilogue), and the gap between a launch's LAST exit and the next launch's FIRST entry on this CU.  GPU box."""
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
t_start = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                              ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=1)
print(env.kernel_variant(), f"T = {T}, from t = {t_start}")
env.reset()
env.rollout(2000)
for _ in range(t_start // 100):
    env.rollout(100)
torch.cuda.synchronize()
out = env.rollout(T)
launch = env.rollout_launcher(T, out)
for _ in range(8):
    launch()
torch.cuda.synchronize()
n = 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    launch()
e1.record()
torch.cuda.synchronize()
print(f"{n} launches back to back: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch")
buf = (C.c_ulonglong * (64 * 2 * 16 * 8))()
assert lib.evac_debug_launch_marks(buf) == 0
m = np.array(buf[:], dtype=np.int64).reshape(64, 2, 16, 8)
order = np.argsort(m[:, 1, 0, 0])            # launches by entry time of workgroup 100's wave 0
m = m[order][-n:]                            # the n launches just timed
for g, name in ((1, "workgroup 100"), (0, "workgroup 0 (the lightest envs; deals the next launch)")):
    a = m[:, g].astype(np.float64) * 0.01    # us
    entry, loop, done, exit_ = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    first_entry, last_exit = entry.min(axis=1), exit_.max(axis=1)
    print(f"-- {name}: medians over {n} launches [us]")
    print(f"   entry spread of the 16 waves             {np.median(entry.max(axis=1) - first_entry):6.2f}")
    print(f"   entry -> top of the first step            mean wave {np.median((loop - entry).mean(axis=1)):6.2f}   slowest {np.median((loop - entry).max(axis=1)):6.2f}")
    for k, what in ((4, "permutation entry here"), (5, "F::init done"), (6, "first action block drawn (needs the env's clock word)"), (7, "state loads retired")):
        print(f"     entry -> {what:55s} mean wave {np.median((a[..., k] - entry).mean(axis=1)):6.2f}   slowest {np.median((a[..., k] - entry).max(axis=1)):6.2f}")
    print(f"   first entry -> last wave's first step    {np.median(loop.max(axis=1) - first_entry):6.2f}")
    print(f"   the {T} steps                            mean wave {np.median((done - loop).mean(axis=1)):6.2f}   slowest {np.median((done - loop).max(axis=1)):6.2f}")
    print(f"   last step -> behind the state stores      mean wave {np.median((exit_ - done).mean(axis=1)):6.2f}   slowest {np.median((exit_ - done).max(axis=1)):6.2f}")
    print(f"   first entry -> last exit                 {np.median(last_exit - first_entry):6.2f}")
    print(f"   last exit -> next launch's first entry   {np.median(first_entry[1:] - last_exit[:-1]):6.2f}")
    print(f"   first entry -> next launch's first entry {np.median(np.diff(first_entry)):6.2f}")

span = (C.c_ulonglong * (64 * 256 * 2))()
assert lib.evac_debug_launch_span(span) == 0
sp = np.array(span[:], dtype=np.int64).reshape(64, 256, 2).astype(np.float64) * 0.01
sp = sp[np.argsort(sp[:, 0, 0])][-n:]
first_in, last_in, first_out, last_out = sp[..., 0].min(axis=1), sp[..., 0].max(axis=1), sp[..., 1].min(axis=1), sp[..., 1].max(axis=1)
print(f"-- all 256 workgroups (wave 0 of each): medians over {n} launches [us]")
print(f"   first entry -> last entry (dispatch of the grid)          {np.median(last_in - first_in):6.2f}")
print(f"   first exit -> last exit (how unevenly the workgroups end)  {np.median(last_out - first_out):6.2f}")
print(f"   last exit -> the next launch's first entry (THE BOUNDARY) {np.median(first_in[1:] - last_out[:-1]):6.2f}")
print(f"   last exit -> the next launch's last entry                 {np.median(last_in[1:] - last_out[:-1]):6.2f}")
print(f"   first entry -> last exit (the slowest workgroup ends it)   {np.median(last_out - first_in):6.2f}")
print(f"   the longest workgroup (its own entry -> exit)              {np.median((sp[..., 1] - sp[..., 0]).max(axis=1)):6.2f}")
print(f"   launch period                                             {np.median(np.diff(first_in)):6.2f}")
