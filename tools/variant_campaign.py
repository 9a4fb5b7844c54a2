"""A longer run of tests/test_gpu_variants_sweep.py's check with fresh random cases (rooms of every family, biased to the team
sizes): every scheduling / decomposition device on against all of them off, bit for bit.  usage: variant_campaign.py [cases] [seed]
GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import evacuation_amd as ea
import test_gpu_variants_sweep as S

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0, fam = time.time(), {}
for c in range(cases):
    n = int(rng.choice([17, 33, 60, 64, 100, 130, 256, 300, 513, 600, 777, 1000, 1024, 1024]))
    if n > 512:
        E = int(rng.choice([2, 5, 8, 9, 16, 17, 24, 32, 40]))          # team sizes 16 / 8 / 4 by what the batch leaves free
    elif n > 64:
        E = int(rng.integers(3, 70))
    else:
        E = int(rng.integers(20, 400))
    mode = str(rng.choice(["grav", "grav", "relbox", "absdict"]))
    ens = float(rng.choice([1.0, 1.0, 0.5, 0.1]))
    seed = int(rng.integers(0, 1 << 30))
    S.test_all_devices_on_equals_all_off.__wrapped__(ea, n, E, mode, ens, seed) if hasattr(S.test_all_devices_on_equals_all_off, "__wrapped__") else \
        S.test_all_devices_on_equals_all_off(ea, n, E, mode, ens, seed)
    fam[(n > 512, n > 64)] = fam.get((n > 512, n > 64), 0) + 1
    if c % 10 == 9:
        print(f"{c + 1} cases ok ({time.time() - t0:.0f} s)", flush=True)
print(f"CAMPAIGN OK: {cases} cases, {fam.get((True, True), 0)} with teams, {fam.get((False, True), 0)} multi-wave, {fam.get((False, False), 0)} one-wave / sub-wave")
