"""The rhythm of the chained launches (diagnostic build -DEVAC_STEP_TIMES through EVAC_LIB): for the last 64 launches, per workgroup
the entry / exit of its wave 0 (all 256 workgroups), and for workgroups 0 and 100 every wave's entry, first step, last step, exit.
Prints the period, how long waves wait for their env (entry -> first step) and how long their 20 steps take.  GPU box."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
from evacuation_amd import _lib
lib = _lib.load()
E, T = 4096, 20
chain = int(sys.argv[1]) if len(sys.argv) > 1 else 1
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 30
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                              ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=0x5EED0001, options=ea.KernelOptions(chain=chain))
env.reset()
out = {"slab": torch.empty((T, E, env.obs_dim + 3), device=env.device), "episode_stats": torch.zeros((T, E, env.stats_words), device=env.device)}
go = env.rollout_launcher(T, out)
clk = torch.zeros((64, 2), dtype=torch.int64, device=env.device)
side = torch.cuda.Stream()
lib.evac_debug_clock.argtypes = [C.c_void_p, C.c_void_p]
for sw in range(warm):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for j in range(100):
        go()
        if sw == warm - 1 and j % 10 == 5:          # the shader clock while the launches run (a one-wave kernel beside them)
            lib.evac_debug_clock(C.c_void_p(clk[j // 10].data_ptr()), C.c_void_p(side.cuda_stream))
    env.join(); e1.record(); torch.cuda.synchronize()
ck = clk[:10].cpu().numpy().astype(np.float64)
print("shader clock during the last sweep [MHz]:", " ".join(f"{c / max(r, 1) * 100:.0f}" for c, r in ck))
print(f"{env.kernel_variant()}: last sweep {e0.elapsed_time(e1) * 10:.2f} us per round; error word {env.team_error()}")
span = (C.c_ulonglong * (64 * 256 * 2))()
assert lib.evac_debug_launch_span(span) == 0
sp = np.array(span[:], dtype=np.int64).reshape(64, 256, 2).astype(np.float64) * 0.01
order = np.argsort(sp[:, :, 0].min(axis=1))
sp = sp[order][4:60]                      # launches in time order, edges dropped
t0 = sp[:, :, 0].min()
first_in, last_in, first_out, last_out = sp[..., 0].min(axis=1), sp[..., 0].max(axis=1), sp[..., 1].min(axis=1), sp[..., 1].max(axis=1)
dur = sp[..., 1] - sp[..., 0]
print(f"launch period (first entry to next launch's first entry): median {np.median(np.diff(first_in)):.1f} us; last exit to last exit {np.median(np.diff(last_out)):.1f}")
print(f"dispatch of a grid (first entry -> last entry): median {np.median(last_in - first_in):.1f} us, p90 {np.percentile(last_in - first_in, 90):.1f}")
print(f"first exit -> last exit: {np.median(last_out - first_out):.1f} us; a workgroup's residence (entry -> exit of its wave 0): median {np.median(dur):.1f}, p10 {np.percentile(dur, 10):.1f}, p90 {np.percentile(dur, 90):.1f}, max {np.median(dur.max(axis=1)):.1f}")
print(f"overlap: next launch's first entry - this launch's last exit: median {np.median(first_in[1:] - last_out[:-1]):.1f} us (negative = overlap); next-but-one: {np.median(first_in[2:] - last_out[:-2]):.1f}")
for k in (20, 21):
    o = np.argsort(sp[k, :, 0])
    print(f"  launch {k}: entries at " + " ".join(f"{sp[k, j, 0] - first_in[k]:.0f}" for j in o[::16]) + "   exits at " + " ".join(f"{sp[k, j, 1] - first_in[k]:.0f}" for j in o[::16]))
buf = (C.c_ulonglong * (64 * 2 * 16 * 8))()
assert lib.evac_debug_launch_marks(buf) == 0
m = np.array(buf[:], dtype=np.int64).reshape(64, 2, 16, 8).astype(np.float64) * 0.01
m = m[np.argsort(m[:, 1, 0, 0])][4:60]
for g, name in ((1, "workgroup 100"), (0, "workgroup 0")):
    a = m[:, g]
    entry, loop, done, exit_ = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    wait = loop - entry; steps = done - loop
    print(f"{name}: per wave, medians over launches: entry->first step (wait + state load + Philox) " + " ".join(f"{x:.0f}" for x in np.median(wait, axis=0)))
    print(f"{' ' * len(name)}  the 20 steps                                                        " + " ".join(f"{x:.0f}" for x in np.median(steps, axis=0)))
    print(f"{' ' * len(name)}  slowest wave's steps: median {np.median(steps.max(axis=1)):.1f} us; its wait {np.median(wait[np.arange(len(wait)), steps.argmax(axis=1)]):.1f} us; workgroup residence {np.median(exit_.max(axis=1) - entry.min(axis=1)):.1f}")
