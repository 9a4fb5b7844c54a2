"""Progress of the 16 waves of workgroup 0 inside ONE rollout launch (diagnostic build: hipcc ... -DEVAC_STEP_TIMES -o
tools/ab_libs/libevac_steptimes.so, loaded through EVAC_LIB):
s_memrealtime at the top of every step.  With the load schedule workgroup 0 carries the heaviest envs of the batch in its
waves 0..3 (one per SIMD: the SIMDs' oldest waves), the lightest in waves 12..15.  GPU box."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
from evacuation_amd import _lib
lib = _lib.load()
E, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 100
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset()
for _ in range(12):
    env.rollout(100)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
out = env.rollout(T)
for rep in range(2):
    env.rollout(T, out=out); torch.cuda.synchronize()
    lib.evac_debug_step_times(buf)
    a = np.array(buf[:], dtype=np.int64).reshape(16, 128)[:, :T]
    t0 = a[:, 0].min()
    rel = (a - t0) * 0.01                                        # us since the first wave started
    print(f"launch {rep}: per wave: start, time of step T/2, time of the last step top [us]; median ns per step")
    for w in range(16):
        d = np.diff(a[w]) * 10.0
        print(f"  wave {w:2d} (SIMD {w % 4}): start {rel[w, 0]:6.1f}  mid {rel[w, T // 2]:7.1f}  last {rel[w, -1]:7.1f}   median {np.median(d):6.0f} ns/step, max {d.max():6.0f} at step {int(d.argmax())}")
    if rep == 1:
        for w in (0, 1, 2, 3):
            d = np.diff(a[w]) * 10.0
            slow = np.nonzero(d > 1.5 * np.median(d))[0]
            print(f"  wave {w}: slow steps (> 1.5 x median) at", [(int(i), round(float(rel[w, i]), 1), int(d[i])) for i in slow][:30])
