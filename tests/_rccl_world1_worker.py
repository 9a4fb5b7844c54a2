"""Worker of tests/test_gpu_rccl_world1.py (a process of its own: the process group is global state).  A ONE-rank RCCL
communicator (backend "nccl" on ROCm) under ShardedEvacuationEnv: the collective of the multi-GPU path -- comm stream,
event hand-off, all_gather_into_tensor, record_stream bookkeeping, gathered_view -- executed on one GPU."""
import datetime
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import evacuation_amd as ea  # noqa: E402
from evacuation_amd.distributed import PeerStoreGather, all_gather_envs, gathered_view, pack_outputs  # noqa: E402


def main():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    E, n = 96, 60
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=35, is_new_exiting_reward=True)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    env = ea.ShardedEvacuationEnv(cfg, wrap, total_envs=E, device=dev, seed=9, force_collective=True)
    twin = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, device=dev, seed=9)
    assert env.collective and env.comm_stream is not None
    env.reset(); twin.reset()
    # rollout_gathered: chunk k's gather (comm stream) overlaps chunk k + 1's compute; several chunks, autoresets inside
    pend = None
    for k in range(4):
        ro, nxt = env.rollout_gathered(20)
        ref = twin.rollout(20)
        if pend is not None:
            full = env.wait(pend[0])
            assert tuple(full.shape) == (20, E, env.obs_dim + 3)
            assert torch.equal(full, pend[1]), f"chunk {k - 1}: gathered slab differs"
        pend = (nxt, ref["slab"].clone())
        assert torch.equal(ro["slab"], ref["slab"])
    assert torch.equal(env.wait(pend[0]), pend[1])
    # the per-step form: pack + all-gather of one step's outputs
    acts = torch.rand((E, 2), device=dev) * 2 - 1
    obs, rew, term, trunc, info, slab = env.step(acts)
    o2, r2, t2, u2, _ = twin.step(acts)
    assert tuple(slab.shape) == (E, env.obs_dim + 3) and torch.equal(slab, pack_outputs(o2, r2, t2, u2))
    # all_gather_envs on a side stream with an explicit output buffer (bench.py's form), and the peer-store gather's one-rank case
    g, _ = all_gather_envs(ro["slab"])
    assert torch.equal(gathered_view(g), ro["slab"])
    gathered = torch.zeros((1,) + tuple(ro["slab"].shape[:-1]) + (env.obs_dim,), device=dev)
    ps = PeerStoreGather(ro["slab"], env.obs_dim, gathered)
    ps.self_test()
    ps.issue()
    torch.cuda.synchronize()
    assert torch.equal(gathered[0], ro["slab"][..., :env.obs_dim])
    env.close(); twin.close()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
