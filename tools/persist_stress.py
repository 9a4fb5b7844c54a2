"""Stress of the persistent form's idle exit (evac_options_t.chain = 2): calls separated by RANDOM host pauses around the kernel's idle bound
(~150 us), so that kernels leave, are leaving, or are still resident when the next command is posted -- every combination of the race between
a wave's last poll and the host's write -- with joins and device-wide waits thrown in; every call's slab and the final state against plain
launches, bit for bit.  GPU box: python tools/persist_stress.py [calls]"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
random.seed(20261005)
ok = True
for name, n, E, wrap_kw, T in (("C2 corner of the device", 60, 64, dict(positions="grav", alpha=3), 3),
                               ("C2 every CU", 60, 4096, dict(positions="grav", alpha=3), 5),
                               ("C3 four-wave envs", 256, 256, dict(positions="grav", alpha=3), 3),
                               ("C5 teams of 8", 1024, 32, dict(positions="rel", statuses="ohe", type="Box"), 2),
                               ("teams of 16, grav", 1024, 11, dict(positions="grav", alpha=3), 2)):
    cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=300)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    cw = 1 if n <= 256 else -1
    a = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=77, options=ea.KernelOptions(cu_wide=cw))
    b = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=77, options=ea.KernelOptions(cu_wide=cw, chain=2))
    assert "persistent" in b.kernel_variant(), b.kernel_variant()
    a.reset(); b.reset()
    R = 32
    outs = [{"slab": torch.empty((T, E, a.obs_dim + 3), device=b.device)} for _ in range(R)]
    goes = [b.rollout_launcher(T, o) for o in outs]
    t0, bad, done, joins, syncs = time.time(), 0, 0, 0, 0
    while done < CALLS:
        k = random.randint(1, R)
        refs = [a.rollout(T)["slab"].clone() for _ in range(k)]
        torch.cuda.synchronize()
        for j in range(k):
            goes[j]()
            r = random.random()
            if r < 0.45:
                pass                                           # back to back
            elif r < 0.85:
                t_end = time.perf_counter() + random.uniform(20e-6, 400e-6)      # around the idle bound
                while time.perf_counter() < t_end:
                    pass
            elif r < 0.95:
                time.sleep(random.uniform(0.0005, 0.003))      # well beyond it
            else:
                torch.cuda.synchronize(); syncs += 1           # a device-wide wait WITHOUT a join
        b.join(); joins += 1
        torch.cuda.synchronize()
        for j in range(k):
            bad += 0 if torch.equal(outs[j]["slab"].view(torch.int32), refs[j].view(torch.int32)) else 1
        done += k
    sa, sb = a.get_state(), b.get_state()
    state_ok = all(torch.equal(sa[x], sb[x]) for x in sa)
    err = b.team_error()
    print(f"{name:26s} {done} calls of {T} steps, {joins} joins, {syncs} device-wide waits without a join, {time.time() - t0:5.1f} s: calls that differ {bad}, "
          f"final state equal {state_ok}, error word {err}", flush=True)
    ok = ok and bad == 0 and state_ok and err == 0
    a.close(); b.close()
print("PERSIST STRESS", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
