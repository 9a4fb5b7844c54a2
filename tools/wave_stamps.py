"""Per-phase cycles of each of the 16 waves of workgroup 0 in ONE rollout launch (diagnostic build: hipcc ... -DEVAC_STAMP
-DEVAC_STAMP_WAVES -o tools/ab_libs/libevac_wavestamps.so, loaded through EVAC_LIB).  With the load schedule workgroup 0
carries the heaviest envs of the batch in its waves 0..3: this is the heavy wave's own breakdown -- the chain that ends the
launch -- next to the light waves'.  GPU box."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
from evacuation_amd import _lib
lib = _lib.load()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N, E = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (60, 4096)
which = sys.argv[4] if len(sys.argv) > 4 else "slowest"      # or "first": workgroup 0
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=N, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
print(env.kernel_variant("rollout"))
env.reset()
names = ["action+noise", "leader+pre-pair", "tile write", "pair loop", "head/move", "classify+reduce", "reward/flags", "reset/obs/stores"]
buf = (C.c_ulonglong * 256)()
out = env.rollout(T)
done = T
for target in (0, 600, 1200, 1800):
    while done + T <= target:
        env.rollout(T, out=out); done += T
    # pass 1 finds the workgroup whose wave 0 lived longest; the launch is repeated from the same state with that workgroup reporting
    state, ws = env.get_state(), (env.workspace.clone() if env.workspace is not None else None)
    slow = C.c_ulonglong(0)
    lib.evac_debug_stamp_block(0, None)
    env.rollout(T, out=out); torch.cuda.synchronize()
    lib.evac_debug_stamp_block(0, C.byref(slow))
    block = int(slow.value & 0xfffff) if which == "slowest" else 0
    env.set_state(**state)
    if ws is not None: env.workspace.copy_(ws)
    lib.evac_debug_stamp_block(block, None)
    env.rollout(T, out=out); done += T; torch.cuda.synchronize()
    lib.evac_debug_wave_stamps(buf)
    print(f"-- workgroup {block} ({which}); slowest of pass 1: workgroup {slow.value & 0xfffff}, wave 0 alive {(slow.value >> 20) * 0.01:.1f} us")
    a = np.array(buf[:], dtype=np.float64).reshape(16, 16)
    print(f"== steps {done - T}..{done} of the episode; cycles per step per wave of the reporting workgroup (lifetime {a[:, 9].max() * 0.01:.1f} us max)")
    print("   wave  total  " + "  ".join(f"{n:>16s}" for n in names) + "   sub 12..15")
    for w in range(16):
        print(f"   {w:4d} {a[w, :8].sum() / T:6.0f}  " + "  ".join(f"{a[w, k] / T:16.0f}" for k in range(8)) + "   " + " ".join(f"{a[w, k] / T:.0f}" for k in range(12, 16))
              + f"   clock {a[w, 8] / max(a[w, 9], 1) * 100:.0f} MHz")
