"""Timing probe (tools/microbench/anyorder.hip showed what hipExtAnyOrderLaunch does on gfx950 for whole-chip grids): TWO env
batches A and B stepped alternately on one stream, A_j B_j A_j+1 B_j+1 ..., so that adjacent launches are independent -- with
any-order launches (EVAC_ANYORDER=1 in the experiment library) adjacent kernels may overlap at their edges without a data race
in practice.  Per-launch time against the default launches: what removing the kernel-boundary gap is worth.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evacuation_amd as ea
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
E = 4096
envs = [ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=s) for s in (1, 2)]
outs = []
for e in envs:
    e.reset(); outs.append(e.rollout(T))
launch = [e.rollout_launcher(T, out=o) for e, o in zip(envs, outs)]
torch.cuda.synchronize()
n = 2000 // T
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(n):
        launch[0](); launch[1]()
    e1.record(); torch.cuda.synchronize()
    print(f"EVAC_ANYORDER={os.environ.get('EVAC_ANYORDER', '0')} sweep {rep}: {e0.elapsed_time(e1) * 1e3 / (2 * n):.2f} us per {T}-step launch (two batches alternating)")
ok = all(bool(torch.isfinite(o["slab"]).all()) for o in outs)
print("outputs finite:", ok)
