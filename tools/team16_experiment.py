"""What would 16 CUs do for the DENSE envs of a C5 shard?  The 16 heaviest envs of a 32-env batch late in an episode (and a fresh
batch of 16 in the dense first steps), stepped by teams of 8 CUs and by teams of 16 CUs; the launch lasts as long as its slowest
env either way.  Bit-identity of the two against the cell-list kernels is checked on the way.  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

def make(E, team, seed=3):
    return ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=1024, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000),
                                   ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box"), num_envs=E, seed=seed,
                                   options=ea.KernelOptions(team=int(team)))

def us_per_step(env, T=100, reps=3):
    out = env.rollout(T)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(T, out=out); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / T)
    return best

big = make(32, 8)
big.reset()
for phase in (0, 600, 1200):
    while int(big.get_state()["now"][0]) < phase:
        big.rollout(100)
    st = big.get_state()
    s_ = st["status"].to(torch.int32); moving = ((s_ >= 1) & (s_ <= 3)).sum(1)
    viscek = (s_ == 1).sum(1)
    order = torch.argsort(moving * 1024 + viscek, descending=True)[:16]
    sub = {k: v[order].contiguous() for k, v in st.items()}
    res = {}
    outs = {}
    for team in (0, 8, 16):
        e = make(16, team)
        e.reset(); e.set_state(**sub)
        r = e.rollout(50)
        outs[team] = r["slab"].clone()
        e.set_state(**sub)
        res[team] = us_per_step(e)
        name = e.kernel_variant("rollout")
        e.close()
        print(f"  phase {phase:5d}: EVAC_TEAM={team:2d} {name:64s} {res[team]:7.2f} us per step (16 heaviest envs: moving {int(moving[order].min())}..{int(moving[order].max())}, viscek {int(viscek[order].min())}..{int(viscek[order].max())})")
    same = torch.equal(outs[0].view(torch.int32), outs[8].view(torch.int32)) and torch.equal(outs[0].view(torch.int32), outs[16].view(torch.int32))
    print(f"  phase {phase:5d}: teams of 8 and of 16 bit-identical to the cell-list kernels over 50 steps: {same};  16 CUs against 8: x{res[8] / res[16]:.2f}")
print("32 envs, teams of 8 (the shard as shipped):", f"{us_per_step(big):.2f} us per step at phase {int(big.get_state()['now'][0])}")
