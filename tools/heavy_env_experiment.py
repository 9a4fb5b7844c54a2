"""Is a late-episode C2 launch bounded by its heaviest env (one wave running at the issue rate of a lone wave) rather than by
the SIMDs' total work?  Times 100-step rollout launches (4096 envs, N = 60) at episode phase ~1000 with (A) the natural load
distribution, (B) every env with more than 24 moving pedestrians lightened, (C) = B plus ONE env reset to all-60-moving,
(D) = B plus 64 dense envs.  Run on the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

E, n = 4096, 60
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset()
print(env.kernel_variant())
for _ in range(10):
    env.rollout(100)
torch.cuda.synchronize()
keep = {k: v.clone() for k, v in env.get_state().items()}
buf = env.rollout(100)


def timed(tag, reps=6):
    # restore the same state before every launch so that the phase does not drift
    ts = []
    for _ in range(reps):
        env.set_state(**cur)
        env.rebind_workspace()
        env.rollout(100, out=buf)            # lets k_schedule see the loads
        env.set_state(**cur)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(100, out=buf); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10.0)          # us per step
    st = cur["status"]
    moving = ((st >= 1) & (st <= 3)).sum(1)
    visc = (st == 1).sum(1)
    print(f"{tag:48s} {np.median(ts):6.3f} us/step (min {min(ts):.3f})  moving mean {moving.float().mean():5.1f} max {int(moving.max())}  viscek mean {visc.float().mean():5.1f} max {int(visc.max())}")


cur = {k: v.clone() for k, v in keep.items()}
timed("A natural distribution at t=1000")
st = cur["status"].clone()
moving = ((st >= 1) & (st <= 3)).sum(1)
heavy = moving > 24
print("envs with more than 24 moving:", int(heavy.sum()))
mv = (st >= 1) & (st <= 3)
rank = torch.cumsum(mv.int(), dim=1)
drop = heavy[:, None] & mv & (rank > 8)                        # heavy envs keep their first 8 moving pedestrians, the rest "escape"
st[drop] = 4
cur["status"] = st
timed("B heavy envs lightened (8 moving left)")
for k in (1, 64, 1024):
    c2 = {kk: v.clone() for kk, v in cur.items()}
    c2["status"] = cur["status"].clone()
    idx = torch.arange(k, device=st.device) * (E // k)
    c2["status"][idx] = 1                                       # every pedestrian VISCEK: dense, every row needed
    c2["pos"] = cur["pos"].clone()
    c2["pos"][idx] = (torch.rand((k, n, 2), device=st.device) * 1.6 - 0.8)
    c2["dir"] = cur["dir"].clone()
    d = torch.randn((k, n, 2), device=st.device)
    c2["dir"][idx] = 0.01 * d / d.norm(dim=-1, keepdim=True)
    save = cur
    cur = c2
    timed(f"C = B + {k} dense env(s) (60 VISCEK)")
    cur = save
