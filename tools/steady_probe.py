"""The headline geometry in its STEADY STATE (episode phases mixed: tools/phase_drift.py), 20 steps per launch:
  a) one handle, one kernel per launch          b) one handle, two parts on its own streams
  c) TWO independent 4096-env batches alternating on two streams (what filling every CU all the time is worth)
  d) 65 536 envs in one handle (the chip's throughput bound for this kernel: 16 rounds of workgroups per launch)
Every batch is first run for `warm` sweeps so that its envs' phases have spread.  GPU box: python tools/steady_probe.py [warm] [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
from evacuation_amd.distributed import side_stream

WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
dev = torch.device("cuda:0")


def make(E, seed, **opt):
    env = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed, options=ea.KernelOptions(**opt))
    env.reset()
    out = {"slab": torch.empty((T, E, env.obs_dim + 3), device=dev), "episode_stats": torch.zeros((T, E, env.stats_words), device=dev)}
    return env, out


def timed(name, envs_launch, joins, E_total, sweeps=12, warm=WARM):
    n = 2000 // T
    res = []
    for sw in range(warm + sweeps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            for go in envs_launch:
                go()
        for j in joins:
            j()
        e1.record()
        torch.cuda.synchronize()
        if sw in (0, 1, 2) or sw >= warm:
            res.append(e0.elapsed_time(e1))
        if sw in (0, 1, 2):
            print(f"   {name}: sweep {sw}: {res[-1] * 1e3 / n:6.2f} us per {T}-step round", flush=True)
    tail = sorted(res[3:])
    med = tail[len(tail) // 2]
    print(f"{name}: synchronised phases (first sweeps) {sorted(res[:3])[1] * 1e3 / n:6.2f} us per round -> steady state (after {warm} sweeps) "
          f"{med * 1e3 / n:6.2f} us per round = {E_total * 2000 / (med * 1e-3):.3e} env-steps/s", flush=True)


which = sys.argv[3] if len(sys.argv) > 3 else "abcd"
if "a" in which:
    env, out = make(4096, 0x5EED0001, parts=1)
    timed("a) one handle, one kernel per launch      ", [env.rollout_launcher(T, out)], [env.join], 4096)
    env.close()
if "b" in which:
    env, out = make(4096, 0x5EED0001, parts=2)
    timed("b) one handle, two parts (own streams)    ", [env.rollout_launcher(T, out)], [env.join], 4096)
    env.close()
if "e" in which:
    env, out = make(4096, 0x5EED0001, chain=1)
    timed("e) one handle, CHAINED launches           ", [env.rollout_launcher(T, out)], [env.join], 4096)
    print("   team/chain error word:", env.team_error(), env.kernel_variant())
    env.close()
if "f" in which or "g" in which or "q" in which:          # BASELINE config 3 (N = 256 x 1024 envs, four waves per env): plain, chained, persistent
    cfg3 = ea.EnvConfig(number_of_pedestrians=256, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
    for tag, opt in (("f) C3 one handle, one kernel per launch  ", dict(chain=0)), ("g) C3 one handle, CHAINED launches       ", dict(chain=1)),
                     ("q) C3 ONE PERSISTENT KERNEL per sweep    ", dict(chain=2))):
        if tag[0] not in which:
            continue
        env = ea.BatchedEvacuationEnv(cfg3, wrap, num_envs=1024, seed=0x5EED0003, options=ea.KernelOptions(**opt))
        env.reset()
        out = {"slab": torch.empty((T, 1024, env.obs_dim + 3), device=dev), "episode_stats": torch.zeros((T, 1024, env.stats_words), device=dev)}
        timed(tag, [env.rollout_launcher(T, out)], [env.join], 1024, sweeps=8, warm=min(WARM, 60))
        print("   error word:", env.team_error(), env.kernel_variant())
        env.close()
if "c" in which:
    s1 = torch.cuda.current_stream(); s2 = side_stream(dev, beside=s1)
    (ea_, oa), (eb, ob) = make(4096, 1, parts=1), make(4096, 2, parts=1)
    la, lb = ea_.rollout_launcher(T, oa, stream=s1), eb.rollout_launcher(T, ob, stream=s2)
    timed("c) TWO batches of 4096 on two streams     ", [la, lb], [lambda: s1.wait_stream(s2)], 8192)
    ea_.close(); eb.close()
if "d" in which:
    env, out = make(65536, 3, parts=1)
    timed("d) 65 536 envs, one handle                ", [env.rollout_launcher(T, out)], [env.join], 65536, sweeps=5, warm=min(WARM, 250))
    env.close()
# the 256-thread workgroups (cu_wide = 0: four one-wave envs per workgroup, no pace keeping, no deal): alone, in throughput mode
if "j" in which:
    env, out = make(4096, 0x5EED0001, parts=1, cu_wide=0)
    timed("j) 256-thread workgroups, one kernel/launch", [env.rollout_launcher(T, out)], [env.join], 4096)
    print("  ", env.kernel_variant())
    env.close()
if "h" in which:
    s1 = torch.cuda.current_stream(); s2 = side_stream(dev, beside=s1)
    (ea_, oa), (eb, ob) = make(4096, 1, parts=1, cu_wide=0), make(4096, 2, parts=1, cu_wide=0)
    la, lb = ea_.rollout_launcher(T, oa, stream=s1), eb.rollout_launcher(T, ob, stream=s2)
    timed("h) 256-thread WGs: TWO batches, two streams", [la, lb], [lambda: s1.wait_stream(s2)], 8192)
    ea_.close(); eb.close()
if "i" in which:
    env, out = make(65536, 3, parts=1, cu_wide=0)
    timed("i) 256-thread WGs: 65 536 envs, one handle ", [env.rollout_launcher(T, out)], [env.join], 65536, sweeps=5, warm=min(WARM, 250))
    env.close()
if "k" in which:
    env, out = make(4096, 0x5EED0001, chain=1, cu_wide=0)
    timed("k) 256-thread workgroups, CHAINED launches ", [env.rollout_launcher(T, out)], [env.join], 4096)
    print("   team/chain error word:", env.team_error(), env.kernel_variant())
    env.close()
# BASELINE config 3 in throughput mode: what keeping every CU busy is worth for the four-wave kernels
if "l" in which or "m" in which:
    cfg3 = ea.EnvConfig(number_of_pedestrians=256, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)

    def make3(E, seed, **opt):
        env = ea.BatchedEvacuationEnv(cfg3, wrap, num_envs=E, seed=seed, options=ea.KernelOptions(**opt))
        env.reset()
        return env, {"slab": torch.empty((T, E, env.obs_dim + 3), device=dev), "episode_stats": torch.zeros((T, E, env.stats_words), device=dev)}
    if "l" in which:
        s1 = torch.cuda.current_stream(); s2 = side_stream(dev, beside=s1)
        (ea_, oa), (eb, ob) = make3(1024, 1, chain=0), make3(1024, 2, chain=0)
        la, lb = ea_.rollout_launcher(T, oa, stream=s1), eb.rollout_launcher(T, ob, stream=s2)
        timed("l) C3: TWO batches of 1024 on two streams ", [la, lb], [lambda: s1.wait_stream(s2)], 2048, sweeps=8, warm=min(WARM, 60))
        ea_.close(); eb.close()
    if "m" in which:
        env, out = make3(8192, 3, chain=0)
        timed("m) C3: 8192 envs, one handle              ", [env.rollout_launcher(T, out)], [env.join], 8192, sweeps=4, warm=min(WARM, 40))
        print("  ", env.kernel_variant())
        env.close()
if "p" in which:
    env, out = make(4096, 0x5EED0001, chain=2)
    timed("p) ONE PERSISTENT KERNEL per sweep         ", [env.rollout_launcher(T, out)], [env.join], 4096)
    print("   error word:", env.team_error(), env.kernel_variant())
    env.close()
