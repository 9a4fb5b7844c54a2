import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
E, T, n = 64, 3, 60
cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=45, is_new_exiting_reward=True, is_new_followers_reward=True)
wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
one = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1))
ch = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=11, options=ea.KernelOptions(cu_wide=1, chain=1))
one.reset(); ch.reset()
torch.cuda.synchronize()
sa, sb = one.get_state(), ch.get_state()
print("after reset:", {k: bool(torch.equal(sa[k], sb[k])) for k in sa})
for j in range(3):
    a = one.rollout(T); b = ch.rollout(T)
    torch.cuda.synchronize()
    print(f"launch {j}: slab equal {bool(torch.equal(a['slab'], b['slab']))}")
    if not torch.equal(a['slab'], b['slab']):
        d = (a['slab'] != b['slab']).nonzero()[0].tolist()
        print("   first diff at", d, a['slab'][d[0], d[1]].tolist(), b['slab'][d[0], d[1]].tolist())
    sa, sb = one.get_state(), ch.get_state()
    for k in sa:
        if not torch.equal(sa[k], sb[k]):
            dd = (sa[k] != sb[k]).nonzero()
            print(f"   state {k}: {len(dd)} differ; first {dd[0].tolist()}: {sa[k][tuple(dd[0][:2].tolist())].tolist()} vs {sb[k][tuple(dd[0][:2].tolist())].tolist()}")
