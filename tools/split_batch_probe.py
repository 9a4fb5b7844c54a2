"""ONE batch of 4096 envs (N = 60) as two / four INDEPENDENT handles (env_id_offset: the same global env ids, the same trajectories) whose
rollout launches go to streams of their own -- the envs of a batch do not depend on each other, only the launches of one handle do.  Against
the single handle on one stream.  20 steps per launch, whole episodes.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evacuation_amd as ea
from evacuation_amd.distributed import side_stream
T = 20
cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
def run(name, envs, streams):
    outs = []
    for e in envs:
        e.reset(); outs.append(e.rollout(T))
    launch = [e.rollout_launcher(T, out=o, stream=s) for e, o, s in zip(envs, outs, streams)]
    n = 2000 // T
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for j in range(n):
            for l in launch: l()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tot = sum(e.num_envs for e in envs)
        print(f"{name}: sweep {rep}: {dt * 1e6 / n:6.2f} us per {T} steps of {tot} envs = {tot * 2000 / dt:.3e} env-steps/s  [{envs[0].kernel_variant('rollout')}]", flush=True)
s1 = torch.cuda.current_stream(); s2 = side_stream(torch.device("cuda:0"), beside=s1)
run("one handle, 4096 envs, one stream        ", [ea.BatchedEvacuationEnv(cfg, wrap, num_envs=4096, seed=1)], [s1])
halves = [ea.BatchedEvacuationEnv(cfg, wrap, num_envs=2048, seed=1, env_id_offset=o, options=ea.KernelOptions(cu_wide=1)) for o in (0, 2048)]
run("two handles of 2048 envs, one stream     ", halves, [s1, s1])
run("two handles of 2048 envs, two streams    ", halves, [s1, s2])
halves = [ea.BatchedEvacuationEnv(cfg, wrap, num_envs=2048, seed=1, env_id_offset=o, options=ea.KernelOptions(cu_wide=0)) for o in (0, 2048)]
run("... 256-thread workgroups, two streams   ", halves, [s1, s2])
quarters = [ea.BatchedEvacuationEnv(cfg, wrap, num_envs=1024, seed=1, env_id_offset=o, options=ea.KernelOptions(cu_wide=0)) for o in (0, 1024, 2048, 3072)]
s3 = side_stream(torch.device("cuda:0"), beside=s2); s4 = torch.cuda.Stream()
run("... four handles of 1024, four streams   ", quarters, [s1, s2, s3, s4])
