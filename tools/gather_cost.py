"""What one gather of a 20-step chunk costs on the HOST and on the DEVICE, per gather form, in a communicator of one rank
(bench.py --force-gather shows the sum: the rollout launches of a sweep with the gather of chunk j - 1 issued after launch j).
Host: wall time of the issuing calls alone (nothing waited for).  Device: HIP events around the gather on an idle device.
GPU box."""
import datetime, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import evacuation_amd as ea
from evacuation_amd.distributed import PeerStoreGather, all_gather_envs

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
E, T, D = 4096, 20, 6
slab = torch.rand((T, E, D + 3), device=dev)
gsrc = torch.empty((T, E, D), device=dev)
g_obs = torch.empty((1, T, E, D), device=dev)
g_slab = torch.empty((1, T, E, D + 3), device=dev)
comm = torch.cuda.Stream(device=dev)
peer = PeerStoreGather(slab, D, g_obs)

def obs():
    gsrc.copy_(slab[..., :D]); all_gather_envs(gsrc, out=g_obs)
def whole():
    all_gather_envs(slab, out=g_slab)
def peer_k():
    peer.issue(comm)
def bench_form(name, fn, n=300):
    for _ in range(20):
        with torch.cuda.stream(comm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ready = torch.cuda.Event(); ready.record()
        with torch.cuda.stream(comm):
            comm.wait_event(ready); fn()
            fin = torch.cuda.Event(); fin.record(comm)
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(comm):
        e0.record(comm)
        for _ in range(50): fn()
        e1.record(comm)
    torch.cuda.synchronize()
    print(f"{name:28s} host {host * 1e6:6.1f} us per gather (event + stream switch + issue)   device {e0.elapsed_time(e1) * 1e3 / 50:6.1f} us per gather on an idle device")
bench_form("obs: column copy + RCCL", obs)
bench_form("slab: RCCL", whole)
bench_form("peer-store kernel", peer_k)
t0 = time.perf_counter()
for _ in range(300):
    ready = torch.cuda.Event(); ready.record()
    with torch.cuda.stream(comm):
        comm.wait_event(ready)
        fin = torch.cuda.Event(); fin.record(comm)
print(f"{'(the bracket alone)':28s} host {(time.perf_counter() - t0) / 300 * 1e6:6.1f} us")
torch.cuda.synchronize(); dist.destroy_process_group()
