import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
E, n = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 60
T = int(sys.argv[3]) if len(sys.argv) > 3 else 7
sync_each = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=45, is_new_exiting_reward=True, is_new_followers_reward=True)
wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
one = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1))
ch = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1, chain=1))
one.reset(); ch.reset()
print(ch.kernel_variant())
R = 16
outs = [{"slab": torch.empty((T, E, one.obs_dim + 3), device=ch.device), "episode_stats": torch.zeros((T, E, ch.stats_words), device=ch.device)} for _ in range(R)]
goes = [ch.rollout_launcher(T, o) for o in outs]
refs = []
for j in range(R):
    refs.append(one.rollout(T))
    goes[j]()
    if sync_each:
        ch.join(); torch.cuda.synchronize()
ch.join(); torch.cuda.synchronize()
print("error word", ch.team_error())
for j in range(R):
    d = (outs[j]["slab"] != refs[j]["slab"])
    if d.any():
        idx = d.nonzero()
        t, e, c = idx[0].tolist()
        print(f"launch {j}: {int(d.sum())} words differ, {int(d.any(dim=2).any(dim=0).sum())} envs; first at t={t} env={e} col={c}: {outs[j]['slab'][t,e].tolist()} vs {refs[j]['slab'][t,e].tolist()}")
    else:
        print(f"launch {j}: equal")
sa, sb = one.get_state(), ch.get_state()
for k in sa:
    print(k, "state equal" if torch.equal(sa[k], sb[k]) else f"state differs in {int((sa[k] != sb[k]).sum())}")
