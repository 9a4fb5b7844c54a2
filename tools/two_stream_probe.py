"""Feasibility probe for launches of ONE batch chained through per-env flags instead of the queue's kernel boundary: what do two
hardware queues give when adjacent launches do not have to wait for each other?  TWO independent env batches A and B (4096 envs each:
every launch wants all 256 CUs), stepped alternately -- on ONE stream (A_j B_j A_j+1 ...: each launch behind the queue's ~4 us kernel
boundary) against TWO streams (A_j on s1, B_j on s2: a workgroup of the other stream's launch takes a CU the moment the workgroup
before it leaves, and the boundary + prologue of one batch hide behind the other's steps).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import evacuation_amd as ea  # noqa: E402
from evacuation_amd.distributed import side_stream  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E = 4096
cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
envs = [ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=s) for s in (1, 2)]
dev = envs[0].device
s1 = torch.cuda.current_stream(dev)
s2 = side_stream(dev, beside=s1)
outs = []
for e in envs:
    e.reset()
    outs.append(e.rollout(T))
one = [e.rollout_launcher(T, out=o, stream=s1) for e, o in zip(envs, outs)]
two = [envs[0].rollout_launcher(T, out=outs[0], stream=s1), envs[1].rollout_launcher(T, out=outs[1], stream=s2)]
torch.cuda.synchronize()
n = 2000 // T
import time  # noqa: E402
for rep in range(3):
    for name, launch in (("one stream ", one), ("two streams", two)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(n):
            launch[0]()
            launch[1]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} sweep {rep}: {dt * 1e6 / (2 * n):6.2f} us per {T}-step launch of 4096 envs (whole episode, two batches alternating)", flush=True)
print("outputs finite:", all(bool(torch.isfinite(o["slab"]).all()) for o in outs))
