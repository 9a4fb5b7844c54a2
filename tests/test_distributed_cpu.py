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


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        everybody_fine = D.agree_all(None)
        one_failed = D.agree_all("out of handles" if rank == 1 else None)
        # the staged build of the peer-store gather: rank 1 cannot export its buffer (injected) -- every rank gets the same
        # error back and nobody is left in a collective (CPU tensors: the stages in front of the failure are what runs here)
        slab, gathered = torch.zeros((4, 9)), torch.zeros((world, 4, 6))
        g, err = D.PeerStoreGather.try_build(slab, 6, gathered, _inject_failure=(rank == 1))
        dist.barrier()
        q.put((rank, everybody_fine, one_failed, g is None, err))
    finally:
        dist.destroy_process_group()


def test_a_failure_on_one_rank_is_everybodys_answer():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, fine, failed, no_gather, err in res:
        assert fine is None
        assert failed == "rank 1: out of handles"
        assert no_gather and err.startswith("rank 1: RuntimeError: injected failure")


def test_host_env_reissues_an_output_array_only_when_nobody_holds_it():
    """HostVectorEnv(copy=True) takes the arrays it hands out from a pool (a fresh 147 KB array per step can cost an mmap and its
    page faults): the pool logic alone, on the CPU."""
    import types
    from evacuation_amd.host_env import HostVectorEnv
    stub = types.SimpleNamespace(_POOL=HostVectorEnv._POOL, _pools=[[None] * HostVectorEnv._POOL for _ in range(4)], _pool_next=[0, 0, 0, 0],
                                 _free_refcount=HostVectorEnv._free_refcount)
    fresh = lambda kind=0: HostVectorEnv._fresh(stub, kind, (8,), np.float32)  # noqa: E731
    assert len({id(fresh()) for _ in range(40)}) <= HostVectorEnv._POOL          # dropped at once: the pool's arrays go round
    kept = []
    for k in range(20):
        a = fresh()
        a[:] = k
        kept.append(a)
    assert [int(a[0]) for a in kept] == list(range(20))                          # kept: never re-issued
    view = fresh()[2:4]                                                          # a view holds its base
    base = view.base
    del kept
    assert all(fresh() is not base for _ in range(3 * HostVectorEnv._POOL))
    t = torch.from_numpy(fresh(1))                                               # ... and so does a tensor made from one
    assert all(fresh(1).ctypes.data != t.data_ptr() for _ in range(3 * HostVectorEnv._POOL))
