"""Long-run equality of the scheduling / decomposition devices against the plain kernels: many episodes, rewards, flags and
final state compared bit for bit (any race in the team exchange or the per-env LDS barriers would show here).
Run on the GPU box:  python tools/soak_variants.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
SW = ("EVAC_CU_WIDE", "EVAC_TEAM", "EVAC_WORKSPACE")


def make(cfg, wrap, E, **env):
    from evacuation_amd.options import from_switches       # (create-time options by the names of their diagnostic switches)
    return ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=123, options=from_switches(**env))


ok = True
for name, n, E, wrap_kw, T in (("C5 teams of 8 CUs", 1024, 32, dict(positions="rel", statuses="ohe", type="Box"), 100),
                               ("C5 teams of 4 CUs", 1024, 64, dict(positions="grav", alpha=3), 50),
                               ("teams of 16 CUs", 1024, 16, dict(positions="rel", statuses="ohe", type="Box"), 100),
                               ("teams of 2 CUs (N = 700)", 700, 128, dict(positions="grav", alpha=3), 100),
                               ("C3 CU-wide", 256, 1024, dict(positions="grav", alpha=3), 100),
                               ("C2 CU-wide + schedule", 60, 4096, dict(positions="grav", alpha=3), 100),
                               ("C2 CU-wide, 20-step launches", 60, 4096, dict(positions="grav", alpha=3), 20)):
    cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=700)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    a = make(cfg, wrap, E, EVAC_CU_WIDE=0, EVAC_TEAM=0, EVAC_WORKSPACE=0)
    b = make(cfg, wrap, E)
    a.reset(); b.reset()
    t0 = time.time()
    bad = 0
    for k in range(STEPS // T):
        ra, rb = a.rollout(T), b.rollout(T)
        same = torch.equal(ra["slab"].view(torch.int32), rb["slab"].view(torch.int32)) and \
            torch.equal(ra["episode_stats"].view(torch.int32), rb["episode_stats"].view(torch.int32))
        bad += 0 if same else 1
    torch.cuda.synchronize()
    sa, sb = a.get_state(), b.get_state()
    state_ok = all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(a.acc, b.acc)
    err = b.team_error()
    print(f"{name:32s} {b.kernel_variant():70s} {STEPS} steps in {time.time() - t0:5.1f} s: launches that differ {bad}, final state equal {state_ok}, team_error {err}")
    ok = ok and bad == 0 and state_ok and err == 0
    a.close(); b.close()
# Chained launches (evac_options_t.chain): R launches in flight back to back -- overlapping on two queues, ordered per env by the
# exchange records -- then one join; every launch's slab and episode records against the plain handle's, bit for bit; every few
# rounds a per-step call in between (the chain joins, exports, and restarts from an import).
for name, n, E, wrap_kw, T, opts in (("C2 chained, 20-step launches", 60, 4096, dict(positions="grav", alpha=3), 20, dict()),
                                     ("C2 chained, 7-step launches", 60, 4096, dict(positions="grav", alpha=3), 7, dict()),
                                     ("N = 40 x 512 chained (Box obs)", 40, 512, dict(positions="rel", statuses="ohe", type="Box"), 10, dict(cu_wide=1)),
                                     # one persistent kernel per join (chain = 2): the same pattern, every launch a command of the resident kernel's ring
                                     ("C2 persistent, 20-step calls", 60, 4096, dict(positions="grav", alpha=3), 20, dict(chain=2)),
                                     ("C2 persistent, 7-step calls", 60, 4096, dict(positions="grav", alpha=3), 7, dict(chain=2)),
                                     ("C3 persistent, 20-step calls", 256, 1024, dict(positions="grav", alpha=3), 20, dict(chain=2)),
                                     ("N = 40 x 512 persistent (Box obs)", 40, 512, dict(positions="rel", statuses="ohe", type="Box"), 10, dict(cu_wide=1, chain=2)),
                                     ("C5 teams of 8, persistent", 1024, 32, dict(positions="rel", statuses="ohe", type="Box"), 20, dict(chain=2)),
                                     ("teams of 16, persistent (grav)", 1024, 9, dict(positions="grav", alpha=3), 7, dict(chain=2))):
    cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=700)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    a = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=123, options=ea.KernelOptions(cu_wide=0, workspace=False))
    b = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=123, options=ea.KernelOptions(**{"chain": 1, **opts}))
    a.reset(); b.reset()
    R = 16
    outs = [{"slab": torch.empty((T, E, a.obs_dim + 3), device=b.device), "episode_stats": torch.zeros((T, E, b.stats_words), device=b.device)} for _ in range(R)]
    goes = [b.rollout_launcher(T, o) for o in outs]
    act = torch.rand((E, 2), device=b.device) * 2 - 1
    t0 = time.time()
    bad = rounds = 0
    for k in range(max(1, STEPS // (T * R))):
        refs = [a.rollout(T) for _ in range(R)]
        for g in goes:
            g()
        b.join()
        torch.cuda.synchronize()
        for o, r in zip(outs, refs):
            done = (r["terminated"] != 0) | (r["truncated"] != 0)
            same = torch.equal(o["slab"].view(torch.int32), r["slab"].view(torch.int32)) and \
                torch.equal(o["episode_stats"].view(torch.int32)[done], r["episode_stats"].view(torch.int32)[done])
            bad += 0 if same else 1
        rounds += 1
        if k % 200 == 199:
            print(f"   ... {name}: {rounds * R * T} steps, launches that differ so far {bad}", flush=True)
        if k % 5 == 4:                            # something else than a plain rollout: the chain restarts behind it
            ra, rb = a.step(act), b.step(act)
            bad += 0 if all(torch.equal(x, y) for x, y in zip(ra[:4], rb[:4])) else 1
    sa, sb = a.get_state(), b.get_state()
    state_ok = all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(a.acc, b.acc)
    err = b.team_error()
    print(f"{name:32s} {b.kernel_variant():70s} {rounds * R * T} steps in {time.time() - t0:5.1f} s: launches that differ {bad}, final state equal {state_ok}, error word {err}")
    ok = ok and bad == 0 and state_ok and err == 0
    a.close(); b.close()
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
