#!/bin/bash
# Per-phase s_memtime breakdown of k_rollout (diagnostic build, never shipped).
#   bash tools/stamps.sh build      (here)      ;   gpurun -- bash tools/stamps.sh run
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = "build" ]; then
  mkdir -p "$ROOT/tools/ab_libs"
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -fPIC -shared -DEVAC_STAMP \
    "$ROOT/evacuation_amd/csrc/evac_api.hip" -o "$ROOT/tools/ab_libs/libevac_stamp.so" && echo built
else
  EVAC_LIB="$ROOT/tools/ab_libs/libevac_stamp.so" python3 - <<PY
import ctypes as C, sys, torch
sys.path.insert(0, "$ROOT")
import evacuation_amd as ea
from evacuation_amd import _lib
lib = _lib.load()
n, E, T = int("${2:-60}"), int("${3:-4096}"), 100
SKIP = int("${4:-5}")
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
print(env.kernel_variant())
env.reset()
buf = (C.c_ulonglong * 16)()
names = ["0 action fetch + noise Philox", "1 leader + pre-pair per-lane", "2 tile write / binning", "3 neighbour loop (+ result exchange)",
         "4 heading/blend/move/reflect", "5 classify + reductions", "6 rewards/flags", "7 reset check + obs + stores"]
for phase in range(4):
    env.rollout(T * SKIP) if SKIP else None; torch.cuda.synchronize()
    lib.evac_debug_stamps(buf)
    env.rollout(T); torch.cuda.synchronize()
    lib.evac_debug_stamps(buf)
    waves = E * (1 if n <= 64 else 2 if n <= 128 else 4 if n <= 256 else 8 if n <= 512 else 16)
    if "CUs/env" in env.kernel_variant(): waves = E * 16                                                 # team kernels: only the 16 ped waves of an env are stamped
    tot = sum(buf[:8])
    print(f"-- steps {phase*(SKIP+1)*100+SKIP*100}..{phase*(SKIP+1)*100+SKIP*100+100}: {tot / waves / T:.0f} cycles per wave-step; shader clock {buf[8] / max(1, buf[9]) * 100:.0f} MHz, "
          f"{buf[9] / waves / T * 10:.0f} ns per wave-step (s_memrealtime); wave lifetime per step: fastest {((1 << 64) - 1 - buf[11]) / T * 10:.0f} ns, slowest {buf[10] / T * 10:.0f} ns")
    for k in range(8):
        print(f"   {names[k]:34s} {buf[k] / waves / T:8.1f} cycles/wave-step  {100.0 * buf[k] / tot:5.1f} %")
    if any(buf[12:16]):   # sub-phases a family stamps on its own (they are NOT part of the total: their time is taken out of the phase they sit in)
        print("   sub-phases 12..15: " + "  ".join(f"{buf[k] / waves / T:.0f}" for k in range(12, 16)))
PY
fi
