"""Does the peer-store gather kernel run BESIDE a rollout launch that holds every CU?  One GPU: the C2 rollout (4096 envs x 20
steps) on the compute stream, evac_peer_gather of the other chunk's records (8 "peers", all buffers local; few workgroups so
that it lasts about as long as a link-bound gather over xGMI would) on a side stream.  Wall time of the pair against the two
alone: overlapped ~ max, serialised ~ sum.  GPU box."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
from evacuation_amd import _lib

lib = _lib.load()
E, T, D, W = 4096, 20, 6, 8
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, max_timesteps=2000), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset(); env.rollout(1000); torch.cuda.synchronize()
state = env.get_state()
comp, comm = torch.cuda.current_stream(), torch.cuda.Stream()
out = [{"slab": torch.randn((T, E, D + 3), device="cuda"), "episode_stats": torch.zeros((T, E, 10), device="cuda")} for _ in range(2)]
launch = env.rollout_launcher(T, out[0], stream=comp)
bufs = [torch.empty((W, T, E, D), device="cuda") for _ in range(W)]
ptrs = (C.c_void_p * W)(*[b.data_ptr() for b in bufs])
src = C.c_void_p(out[1]["slab"].data_ptr())
def gather(wgs, stream):
    rc = lib.evac_peer_gather(src, T * E, D + 3, D, ptrs, W, 0, wgs, C.c_void_p(stream.cuda_stream)); assert rc == 0
def timed(fn, n=200):
    ts = []
    for _ in range(n):
        env.set_state(**state); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
print(env.kernel_variant("rollout"))
for wgs in (8, 2, 1):
    t_r = timed(lambda: launch())
    t_g = timed(lambda: gather(wgs, comm))
    t_b = timed(lambda: (launch(), gather(wgs, comm)))
    t_s = timed(lambda: (launch(), gather(wgs, comp)))          # same stream: serialised by construction
    print(f"wgs_per_peer {wgs}: rollout alone {t_r:6.1f} us, gather alone {t_g:6.1f} us, both (side stream) {t_b:6.1f} us, both (one stream) {t_s:6.1f} us"
          f"  -> overlap hides {100 * (t_s - t_b) / max(t_g - 5, 1):4.0f} % of the gather")
