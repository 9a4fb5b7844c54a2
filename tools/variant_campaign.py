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
# ... and the forms with launches in flight (evac_options_t.chain = 1 / 2): bursts of launches of random lengths without a join against one
# plain launch after the other (tests/test_gpu_variants_sweep.py::test_launches_in_flight_equal_one_launch_after_the_other), fresh random shapes
flying = int(sys.argv[3]) if len(sys.argv) > 3 else cases // 3
t1, forms = time.time(), {1: 0, 2: 0}
for c in range(flying):
    if rng.integers(0, 3) == 0:
        n, E = int(rng.choice([130, 200, 256])), 4 * int(rng.integers(2, 64))
    else:
        n, E = int(rng.choice([33, 48, 60, 64])), 16 * int(rng.integers(2, 64))
    mode = str(rng.choice(["grav", "grav", "relbox", "absdict"]))
    ens = float(rng.choice([1.0, 1.0, 0.5]))
    form = int(rng.integers(1, 3))
    S.test_launches_in_flight_equal_one_launch_after_the_other(ea, n, E, mode, ens, form, int(rng.integers(0, 1 << 30)))
    forms[form] += 1
    if c % 25 == 24:
        print(f"{c + 1} bursts-in-flight cases ok ({time.time() - t1:.0f} s)", flush=True)
print(f"CAMPAIGN OK: {flying} cases with launches in flight: {forms[1]} chained, {forms[2]} persistent")
