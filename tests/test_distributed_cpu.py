"""world_size-2 gloo tests of the sharding / packed all-gather logic (CPU; the per-rank HIP step is
covered by the -m gpu tests, including test_sharded_handles_reproduce_single_handle)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from evacuation_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_local_outputs(offset, n_local, T, d):
    """Deterministic stand-in for a shard's rollout outputs: a function of the GLOBAL env id."""
    gid = torch.arange(offset, offset + n_local, dtype=torch.float32)
    t = torch.arange(T, dtype=torch.float32)[:, None]
    obs = (gid[None, :, None] * 10 + t[:, :, None] * 1000 + torch.arange(d, dtype=torch.float32)[None, None, :])
    reward = -(gid[None, :] + t)
    term = ((gid[None, :] + t) % 3 == 0)
    trunc = ((gid[None, :] + t) % 5 == 0)
    return obs, reward, term.to(torch.uint8), trunc.to(torch.uint8)


def _worker(rank, world, port, total, T, d, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        off, n_local = D.shard_range(total, rank, world)
        obs, rew, te, tr = _fake_local_outputs(off, n_local, T, d)
        slab = D.pack_outputs(obs, rew, te, tr)
        g, _ = D.all_gather_envs(slab)
        full = D.gathered_view(g)
        # single-step form ([E_local, C])
        g1, _ = D.all_gather_envs(slab[0])
        full1 = D.gathered_view(g1)
        q.put((rank, full.numpy(), full1.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gather_restores_global_env_order(world):
    total, T, d = 8, 3, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    obs, rew, te, tr = _fake_local_outputs(0, total, T, d)       # what one process owning all envs has
    want = D.pack_outputs(obs, rew, te, tr).numpy()
    for rank, full, full1 in res:
        np.testing.assert_array_equal(full, want)
        np.testing.assert_array_equal(full1, want[0])
    o, r, a, b = D.unpack_outputs(torch.from_numpy(want))
    assert (o == obs).all() and (r == rew).all() and (a == te.bool()).all() and (b == tr.bool()).all()


def test_shard_range():
    assert D.shard_range(32768, 3, 8) == (3 * 4096, 4096)
    assert D.shard_range(256, 7, 8) == (224, 32)
    with pytest.raises(ValueError):
        D.shard_range(10, 0, 4)
