"""What one evac_step launch costs and what it would cost without its arithmetic (diagnostic -DEVAC_ABLATE builds through
EVAC_LIB): pre-bound launcher, 4096 envs of 60 pedestrians, mid-episode.  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea
E = 4096
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset(); env.rollout(600); torch.cuda.synchronize()
act = torch.rand((E, 2), device="cuda") * 2 - 1
go = env.step_launcher(act, stream=torch.cuda.current_stream())
for _ in range(100): go()
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): go()
torch.cuda.synchronize()
print(f"{os.environ.get('EVAC_LIB', 'libevac.so'):40s} {env.kernel_variant('step'):60s} {(time.perf_counter() - t0) / n * 1e6:6.2f} us per step")
