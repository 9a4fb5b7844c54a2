"""Where the wall time of ONE timed block of the driver's benchmark goes (bench.py --steps 20: one 20-step launch between two
synchronisations): time until the launch call returns, time until the stream reports idle, against the kernel's own duration
(HIP events around back-to-back launches).  GPU box.  usage: block_overhead.py [T]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evacuation_amd as ea

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] == "side":       # everything on a non-default stream
    torch.cuda.set_stream(torch.cuda.Stream())
E = 4096
env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, max_timesteps=2000), ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=1)
env.reset()
env.rollout(1000); torch.cuda.synchronize()
out = {"slab": torch.empty((T, E, 9), device="cuda"), "episode_stats": torch.zeros((T, E, 10), device="cuda")}
st = torch.cuda.current_stream()
launch = env.rollout_launcher(T, out, stream=st)
hip = C.CDLL("libamdhip64.so")
hip.hipStreamQuery.argtypes = [C.c_void_p]
sq, sp = hip.hipStreamQuery, C.c_void_p(st.cuda_stream)
for _ in range(20): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): launch()
e1.record(); torch.cuda.synchronize()
kern = e0.elapsed_time(e1) * 1e3 / 50
pc = time.perf_counter
def run(poll, n=300):
    call, tot = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = pc(); launch(); t1 = pc(); poll(); t2 = pc()
        call.append(t1 - t0); tot.append(t2 - t0)
    return np.median(call) * 1e6, np.median(tot) * 1e6, np.percentile(tot, 10) * 1e6
def poll_torch():
    while not st.query(): pass
def poll_hip():
    while sq(sp) != 0: pass
def poll_sync():
    torch.cuda.synchronize()
print(f"stream {st.cuda_stream:#x} T={T}: kernel {kern:.1f} us per launch (events, back to back); env: " + " ".join(f"{k}={os.environ[k]}" for k in ("HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_INTERRUPT", "GPU_MAX_HW_QUEUES", "HIP_LAUNCH_BLOCKING") if k in os.environ))
for name, f in (("torch stream.query() loop", poll_torch), ("hipStreamQuery loop (ctypes)", poll_hip), ("torch.cuda.synchronize()", poll_sync), ("stream.synchronize()", st.synchronize)):
    c, t, p10 = run(f)
    print(f"  {name:32s}: launch call {c:5.1f} us, block {t:6.1f} us (p10 {p10:6.1f}) -> {t - kern:5.1f} us over the kernel")
