"""Env-sharded data parallelism: one process per GPU, a contiguous block of envs per rank, and ONE
collective -- an all-gather of the packed step outputs -- because the envs are fully independent
(no cross-env term anywhere in the reference's area.py).

The reference has no distributed code at all (its "vector env" is gym.vector.SyncVectorEnv stepping
3 envs sequentially, rpo_agent.py:123-126); this is the MI355X-native replacement for scaling it.

Layout of the gathered slab: float32 [..., E_total, D + 3] = [obs(D) | reward | terminated | truncated]
in GLOBAL env order (rank r owns envs [r*E_local, (r+1)*E_local)).  Philox streams are keyed by the
global env id (evac_create's env_id_offset), so a sharded run reproduces the single-GPU run bit for
bit.  With backend "nccl" (= RCCL on ROCm) the gather runs over xGMI; the sizes are small
(C4: 4096 envs x 9 floats = 144 KiB per rank per step) so it is latency-bound and is issued on a
side stream, chunked over T steps, overlapping the next chunk's compute.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous equal shards; total_envs must divide evenly (all_gather_into_tensor needs equal sizes)."""
    if total_envs % world_size != 0:
        raise ValueError(f"num_envs={total_envs} is not divisible by world_size={world_size}")
    per = total_envs // world_size
    return rank * per, per


def pack_outputs(obs: torch.Tensor, reward: torch.Tensor, terminated: torch.Tensor, truncated: torch.Tensor,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[..., E, D] obs + [..., E] reward/flags -> one float32 slab [..., E, D+3] (a single message)."""
    d = obs.shape[-1]
    shape = tuple(obs.shape[:-1]) + (d + 3,)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=obs.device)
    out[..., :d] = obs
    out[..., d] = reward
    out[..., d + 1] = terminated.to(torch.float32)
    out[..., d + 2] = truncated.to(torch.float32)
    return out


def unpack_outputs(slab: torch.Tensor):
    d = slab.shape[-1] - 3
    return slab[..., :d], slab[..., d], slab[..., d + 1] != 0, slab[..., d + 2] != 0


def side_stream(device, beside=None, candidates: int = 8) -> "torch.cuda.Stream":
    """A stream of ``device`` whose kernels really run BESIDE those of ``beside`` (default: the current stream).  HIP maps its
    streams onto a handful of hardware queues (4 by default) in the order in which they are first used, and two streams that share
    a queue execute strictly one after the other -- with the process group's and the allocator's streams created first, a fresh
    ``torch.cuda.Stream()`` landed on the compute stream's own queue in bench.py's forced-gather runs of round 4 and the gather never
    ran under the rollout (profiles/r04_j_force_gather_world1.txt).  So: time two one-thread spin kernels (``torch.cuda._sleep``),
    one on each stream; a pair that takes as long as one of them overlaps.  The first candidate that does is returned (the others are
    dropped; their queue assignments stay used up, which is what moves the next candidate on).  Falls back to the last candidate."""
    device = torch.device(device)
    beside = beside if beside is not None else torch.cuda.current_stream(device)
    spin = 400_000                                   # cycles: ~0.2 ms
    def pair_ms(cand):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        e0.record(beside)
        with torch.cuda.stream(beside):
            torch.cuda._sleep(spin)
        with torch.cuda.stream(cand):
            torch.cuda._sleep(spin)
        beside.wait_stream(cand)
        e1.record(beside)
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)
    with torch.cuda.device(device):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(beside):
            torch.cuda._sleep(spin)                  # (warm-up: the kernel's first launch loads its code object)
        torch.cuda.synchronize(device)
        e0.record(beside)
        with torch.cuda.stream(beside):
            torch.cuda._sleep(spin)
        e1.record(beside)
        torch.cuda.synchronize(device)
        one = e0.elapsed_time(e1)
        cand = None
        for _ in range(max(1, candidates)):
            cand = torch.cuda.Stream(device=device)
            pair_ms(cand)                            # (first use: the stream gets its queue here)
            if pair_ms(cand) < 1.5 * one:
                break
    return cand


def all_gather_envs(local: torch.Tensor, env_dim: int = -2, group=None, out: Optional[torch.Tensor] = None,
                    async_op: bool = False):
    """All-gather ``local`` ([..., E_local, C]) along its env dimension into global env order.

    RCCL's all_gather_into_tensor concatenates along dim 0, so for time-major chunks [T, E_local, C]
    the gathered buffer is [world, T, E_local, C]; ``gathered_view`` turns it into [T, E_total, C]
    without a copy of the payload being needed by callers that index by (rank, env)."""
    world = dist.get_world_size(group)
    local = local.contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    # concatenated-along-dim-0 form: accepted by both RCCL and gloo
    flat_out = out.view((world * local.shape[0],) + tuple(local.shape[1:]))
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # testing aid (several ranks sharing one GPU rendezvous over gloo, which gathers host tensors): stage through the host
        host = torch.empty(flat_out.shape, dtype=flat_out.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=group)
        flat_out.copy_(host)
        return out, None
    work = dist.all_gather_into_tensor(flat_out, local, group=group, async_op=async_op)
    return out, work


def gathered_view(gathered: torch.Tensor, env_dim: int = -2) -> torch.Tensor:
    """[world, ..., E_local, C] -> [..., world*E_local, C] (global env order)."""
    world = gathered.shape[0]
    nd = gathered.dim()
    env_axis = env_dim % (nd - 1) + 1           # axis of E_local in `gathered`
    perm = list(range(1, env_axis)) + [0] + list(range(env_axis, nd))
    g = gathered.permute(perm)                    # [..., world, E_local, C...]
    shape = list(g.shape)
    k = env_axis - 1
    return g.reshape(shape[:k] + [world * shape[k + 1]] + shape[k + 2:])


class DirectGather:
    """All-gather of one rank-local tensor by COPY-ENGINE peer writes instead of an RCCL kernel (on the
    fully connected xGMI mesh every rank can write its shard to all peers at once).

    Why: the rollout workgroups occupy every CU for the whole launch (one 1024-thread workgroup per CU, all vector registers),
    so a CU-resident collective kernel that is to overlap the compute has nowhere to run until the rollout's first workgroups
    retire; a device-to-device ``hipMemcpyAsync`` into a peer's buffer is executed by the SDMA engines and overlaps by
    construction.

    How: every rank owns ``gathered`` = [world, *src.shape]; the buffers are exchanged once as hipIpc handles (PyTorch's own
    CUDA-IPC tensor sharing: ``torch.multiprocessing.reductions.reduce_tensor`` through ``all_gather_object``; needs
    HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver, which the image exports) and ``issue(stream)`` enqueues world copies
    ``peer_gathered[r][my_rank] <- src`` on ``stream``.  The event the caller records afterwards marks the end of this
    rank's OUTGOING copies; a consumer of ``gathered`` needs all ranks to have passed theirs (a barrier / the next step's
    rendezvous) -- bench.py's blocks end with exactly that.  RCCL stays the default path (north_star); this one is
    ``bench.py --gather direct``."""

    def __init__(self, src: torch.Tensor, gathered: torch.Tensor, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert src.is_contiguous() and gathered.is_contiguous()
        assert tuple(gathered.shape) == (self.world,) + tuple(src.shape) and gathered.dtype == src.dtype
        self.src, self.gathered = src, gathered
        self.peers, self._keep = _map_peer_buffers(gathered, group)      # rank r's `gathered`, mapped here

    def issue(self, stream=None) -> None:
        """Enqueue the world copies on ``stream`` (default: the current stream), the farthest peers first, own copy last.
        Copies of ONE stream execute one after another (each takes its link alone); PeerStoreGather writes to all peers at once."""
        ctx = torch.cuda.stream(stream) if stream is not None else _null_context()
        with ctx:
            for k in range(1, self.world + 1):
                r = (self.rank + k) % self.world
                self.peers[r][self.rank].copy_(self.src, non_blocking=True)

    def self_test(self) -> None:
        """Every rank fills its source with a rank-coded pattern, gathers, and checks what the peers wrote."""
        keep = self.src.clone()
        self.src.fill_(float(self.rank + 1))
        self.issue()
        torch.cuda.synchronize(self.src.device)
        dist.barrier(self.group)
        got = self.gathered.reshape(self.world, -1)[:, 0].cpu()
        want = torch.arange(1, self.world + 1, dtype=got.dtype)
        ok = bool((got == want).all()) and bool((self.gathered.reshape(self.world, -1)[:, -1].cpu() == want).all())
        self.src.copy_(keep)
        dist.barrier(self.group)
        if not ok:
            raise RuntimeError(f"DirectGather self-test failed on rank {self.rank}: got {got.tolist()}, expected {want.tolist()}")


def agree_all(err, group=None) -> Optional[str]:
    """Every rank hands in what went wrong locally (or None); every rank gets back the first failure ANYWHERE (None: all fine).
    One host-side ``all_gather_object``: a rank that failed still takes part, so nobody is left waiting in a collective."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    errs = [None] * world
    dist.all_gather_object(errs, None if err is None else f"rank {rank}: {err}"[:240], group=group)
    bad = [e for e in errs if e is not None]
    return bad[0] if bad else None


def _exchange_handles(gathered: torch.Tensor, group=None):
    """hipIpc handles of every rank's `gathered` buffer (PyTorch's CUDA-IPC tensor sharing), by rank."""
    from torch.multiprocessing.reductions import reduce_tensor
    handles = [None] * dist.get_world_size(group)
    dist.all_gather_object(handles, reduce_tensor(gathered), group=group)
    return handles


def _open_handles(handles, gathered: torch.Tensor, group=None):
    """The peers' buffers mapped into this process (this rank's own entry is `gathered` itself).  May raise (hipIpcOpenMemHandle)."""
    rank = dist.get_rank(group)
    return [gathered if r == rank else rebuild(*rebuild_args) for r, (rebuild, rebuild_args) in enumerate(handles)]


def _map_peer_buffers(gathered: torch.Tensor, group=None):
    """Every rank's `gathered` buffer mapped into this process (hipIpc through PyTorch's CUDA-IPC tensor sharing); returns the
    list of tensors by rank (this rank's own entry is `gathered` itself) and what must be kept alive with them."""
    handles = _exchange_handles(gathered, group)
    peers = _open_handles(handles, gathered, group)
    dist.barrier(group)                  # every rank has mapped every buffer before anybody writes
    return peers, handles


def _enable_peer_access(mine: int, others) -> None:
    """Stores issued by a kernel of device `mine` into buffers of the devices `others`: hipDeviceEnablePeerAccess must succeed
    (or report that access is enabled already) for every one of them -- a store to a peer without access faults the GPU, so
    anything else is an error here, not something for a self-test to find."""
    import ctypes as C
    others = sorted(set(others) - {mine})
    if not others:
        return
    hip = C.CDLL("libamdhip64.so")               # (OSError propagates: PyTorch-ROCm has loaded this library already)
    hip.hipDeviceEnablePeerAccess.restype = C.c_int
    hip.hipGetErrorString.restype = C.c_char_p
    HIP_SUCCESS, HIP_ERROR_PEER_ACCESS_ALREADY_ENABLED = 0, 704
    with torch.cuda.device(mine):
        for d in others:
            rc = hip.hipDeviceEnablePeerAccess(C.c_int(d), C.c_uint(0))
            if rc == HIP_ERROR_PEER_ACCESS_ALREADY_ENABLED:
                hip.hipGetLastError()            # (clear the sticky error)
            elif rc != HIP_SUCCESS:
                msg = hip.hipGetErrorString(C.c_int(rc))
                raise RuntimeError(f"hipDeviceEnablePeerAccess(device {d}) from device {mine} failed with "
                                   f"{rc} ({msg.decode() if msg else '?'}): no peer stores to that rank")


def peer_store_probe(device, group=None, rows: int = 256, row_words: int = 9, take: int = 6) -> dict:
    """Can this group gather by peer stores (PeerStoreGather)?  Run by ALL ranks together, before anything is timed; never
    raises and never leaves a rank behind in a collective: the three stages -- (1) every rank's buffer mapped into every other
    rank through hipIpc, (2) peer access enabled towards the devices that own them, (3) a rank- and position-coded pattern stored
    by evac_peer_gather and checked slice by slice -- each end with an agreement over the host (``all_gather_object``), and the
    first stage that fails ANYWHERE ends the probe everywhere.  Returns ``{"ok", "stage", "error", "world", "devices",
    "store_ms"}``: what ``bench.py --gather auto`` decides by, and what every multi-rank bench line reports whatever gather it
    times.  (A store to an unmapped address would fault the GPU rather than raise; stage 2 is strict for that reason.)"""
    import time
    device = torch.device(device)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    info = {"ok": False, "stage": "map", "error": None, "world": world, "devices": None, "store_ms": None}

    def agree(err):
        errs = [None] * world
        dist.all_gather_object(errs, None if err is None else f"rank {rank}: {err}"[:240], group=group)
        bad = [e for e in errs if e is not None]
        if bad:
            info["error"] = bad[0]
        return not bad

    err, peers, handles, g = None, None, None, None
    try:
        from torch.multiprocessing.reductions import reduce_tensor
        slab = torch.zeros((rows, row_words), dtype=torch.float32, device=device)
        gathered = torch.zeros((world, rows, take), dtype=torch.float32, device=device)
        #  the export itself can fail on ONE rank (hipIpcGetMemHandle): it is made outside the collective and a failure
        # travels as None, as in PeerStoreGather.try_build -- nobody is left waiting in all_gather_object
        mine = None
        try:
            mine = reduce_tensor(gathered)
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        handles = [None] * world
        dist.all_gather_object(handles, mine, group=group)
        if not agree(err):
            return info
        try:
            peers = _open_handles(handles, gathered, group)
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        if not agree(err):
            return info
        devs = [None] * world
        dist.all_gather_object(devs, int(device.index if device.index is not None else torch.cuda.current_device()), group=group)
        info["devices"] = devs
        info["stage"] = "peer_access"
        try:
            _enable_peer_access(slab.device.index, {p.device.index for p in peers})
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        if not agree(err):
            return info
        info["stage"] = "store"
        try:
            g = PeerStoreGather(slab, take, gathered, group=group, _mapped=(peers, handles))
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        if not agree(err):                      # (a rank whose constructor raised must not meet the others inside self_test's barrier)
            return info
        try:
            t0 = time.perf_counter()
            g.self_test()
            info["store_ms"] = (time.perf_counter() - t0) * 1e3
        except Exception as exc:  # noqa: BLE001  (self_test's own barriers are passed by every rank before it compares)
            err = f"{type(exc).__name__}: {exc}"
        if not agree(err):
            return info
        info["ok"], info["stage"] = True, "done"
        return info
    except Exception as exc:  # noqa: BLE001  (the exchange itself: all ranks see the same failure or the process group is gone)
        info["error"] = f"rank {rank}: {type(exc).__name__}: {exc}"[:240]
        return info


class PeerStoreGather:
    """All-gather of the observation columns of a rank-local record slab by ONE hand-written kernel that stores into every
    peer's buffer over xGMI (``evac_peer_gather``, csrc/evac_gather.h) -- no library collective, no staging copy of the columns.

    ``slab`` [..., row_words] (the rollout kernel's packed records), ``take_words`` leading columns of every record,
    ``gathered`` [world, ..., take_words] on every rank; the buffers are mapped once (hipIpc, as DirectGather).  The kernel is
    built to run beside a rollout launch that holds every CU (29 vector registers per lane against the 64 the rollout leaves
    free per SIMD lane), so ``issue(stream)`` on a side stream overlaps the next chunk's compute.  The event the caller records
    afterwards marks the end of this rank's OUTGOING stores; a consumer of ``gathered`` needs all ranks to have passed theirs.
    ``bench.py --gather peer``."""

    def __init__(self, slab: torch.Tensor, take_words: int, gathered: torch.Tensor, group=None, wgs_per_peer: int = 8, _mapped=None):
        import ctypes as C
        from . import _lib
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        assert slab.is_contiguous() and gathered.is_contiguous() and slab.dtype == gathered.dtype == torch.float32
        self.row_words, self.take = int(slab.shape[-1]), int(take_words)
        self.rows = slab.numel() // self.row_words
        assert tuple(gathered.shape) == (self.world,) + tuple(slab.shape[:-1]) + (self.take,), (gathered.shape, slab.shape, take_words)
        self.slab, self.gathered, self.wgs = slab, gathered, int(wgs_per_peer)
        self.peers, self._keep = _mapped if _mapped is not None else _map_peer_buffers(gathered, group)   # (_mapped: peer_store_probe)
        _enable_peer_access(self.slab.device.index, {p.device.index for p in self.peers})
        self._lib = _lib.load()
        self._ptrs = (C.c_void_p * self.world)(*[p.data_ptr() for p in self.peers])
        self._src = C.c_void_p(slab.data_ptr())
        self._C = C

    @classmethod
    def try_build(cls, slab: torch.Tensor, take_words: int, gathered: torch.Tensor, group=None, wgs_per_peer: int = 8, _inject_failure: bool = False):
        """``(gather, None)`` on every rank, or ``(None, first error anywhere)`` on every rank: the constructor in stages (handles
        made and exchanged, peers' buffers opened, peer access enabled), each closed by ``agree_all`` -- a rank whose
        hipIpcGetMemHandle / hipIpcOpenMemHandle / hipDeviceEnablePeerAccess fails does not leave the others in a collective.
        For callers that have another way to gather (bench.py: the alternative form of a line; ``--gather auto``)."""
        from torch.multiprocessing.reductions import reduce_tensor
        world = dist.get_world_size(group)
        err, mine, peers = None, None, None
        try:
            if _inject_failure:                  # (testing aid: this rank cannot export its buffer)
                raise RuntimeError("injected failure (hipIpcGetMemHandle)")
            mine = reduce_tensor(gathered)
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        handles = [None] * world
        dist.all_gather_object(handles, mine, group=group)
        bad = agree_all(err, group)
        if bad:
            return None, bad
        try:
            peers = _open_handles(handles, gathered, group)
            _enable_peer_access(slab.device.index, {p.device.index for p in peers})
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
        bad = agree_all(err, group)             # (also the barrier: every rank has mapped every buffer before anybody writes)
        if bad:
            return None, bad
        try:
            g = cls(slab, take_words, gathered, group=group, wgs_per_peer=wgs_per_peer, _mapped=(peers, handles))
        except Exception as exc:  # noqa: BLE001
            g, err = None, f"{type(exc).__name__}: {exc}"
        bad = agree_all(err, group)
        return (None, bad) if bad else (g, None)

    def issue(self, stream=None) -> None:
        # (evac_peer_gather takes no handle and launches on the CURRENT device: make it the slab's, whatever the caller has current)
        with torch.cuda.device(self.slab.device):
            st = stream if stream is not None else torch.cuda.current_stream(self.slab.device)
            rc = self._lib.evac_peer_gather(self._src, self.rows, self.row_words, self.take, self._ptrs, self.world, self.rank, self.wgs,
                                            self._C.c_void_p(st.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"evac_peer_gather failed with status {rc}")

    def self_test(self) -> None:
        """Every rank fills its slab with a rank- and position-coded pattern, gathers, and checks every slice it received.
        Every rank passes every barrier whatever happens to it locally (a failed launch is remembered and raised at the end), so
        a failure on one rank cannot leave the others waiting."""
        keep = self.slab.clone()
        flat = self.slab.view(-1, self.row_words)
        base = torch.arange(flat.numel(), dtype=torch.float32, device=self.slab.device).view_as(flat) % 4093.0
        flat.copy_(base + 5000.0 * (self.rank + 1))
        torch.cuda.synchronize(self.slab.device)
        dist.barrier(self.group)
        err = None
        try:
            self.issue()
            torch.cuda.synchronize(self.slab.device)
        except Exception as exc:  # noqa: BLE001
            err = exc
        dist.barrier(self.group)
        got = self.gathered.view(self.world, -1, self.take)
        ok = err is None
        for r in range(self.world):
            want = base[:, : self.take] + 5000.0 * (r + 1)
            ok = ok and bool(torch.equal(got[r], want))
        self.slab.copy_(keep)
        torch.cuda.synchronize(self.slab.device)
        dist.barrier(self.group)
        if err is not None:
            raise RuntimeError(f"PeerStoreGather self-test: the store kernel failed on rank {self.rank}: {err}") from err
        if not ok:
            raise RuntimeError(f"PeerStoreGather self-test failed on rank {self.rank}")


class _null_context:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class ShardedEvacuationEnv:
    """``total_envs`` envs sharded over the ranks of a torch.distributed group; each rank steps its
    shard with a BatchedEvacuationEnv and the packed outputs are all-gathered."""

    def __init__(self, env_config, wrap_config=None, total_envs: int = 1, device=None, seed: int = 0, group=None,
                 autoreset: bool = True, force_collective: bool = False, options=None):
        """``force_collective``: run the collective (comm stream, event hand-off, ``all_gather_into_tensor``) also in a group of
        ONE rank -- the multi-GPU code path exercised on a single GPU (tests/test_gpu_rccl_world1.py)."""
        from .vector_env import BatchedEvacuationEnv
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world_size > 1 or (bool(force_collective) and dist.is_initialized())
        self.total_envs = int(total_envs)
        self.offset, self.local_envs = shard_range(self.total_envs, self.rank, self.world_size)
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}"
        self.local = BatchedEvacuationEnv(env_config, wrap_config, num_envs=self.local_envs, device=device, seed=seed,
                                          env_id_offset=self.offset, autoreset=autoreset, options=options)
        self.obs_dim = self.local.obs_dim
        self.comm_stream = side_stream(self.local.device) if self.collective else None      # (one that overlaps the compute stream)

    def reset(self, **kw):
        return self.local.reset(**kw)

    def step(self, actions_local, gather: bool = True):
        """Step the local shard; returns the local outputs and (if gather) the global slab
        [E_total, D+3] in global env order."""
        obs, rew, term, trunc, info = self.local.step(actions_local)
        if not gather:
            return obs, rew, term, trunc, info, None
        slab = pack_outputs(obs, rew, term, trunc)
        if not self.collective:
            return obs, rew, term, trunc, info, slab
        g, _ = all_gather_envs(slab, group=self.group)
        return obs, rew, term, trunc, info, gathered_view(g)

    def rollout_gathered(self, n_steps: int, actions=None, prev=None):
        """One T-step rollout launch on the compute stream, then the all-gather of its packed
        outputs on the comm stream.  Returns (local rollout dict, pending) where
        ``pending = (gathered [world,T,E_local,D+3], event)``; pass it back as ``prev`` (or call
        ``wait``) before reading it.  The gather of chunk k overlaps the compute of chunk k+1."""
        ro = self.local.rollout(n_steps, actions=actions)
        slab = ro["slab"]            # the kernel already wrote the packed [obs | reward | flags] record
        if not self.collective:
            return ro, (slab.unsqueeze(0), None)
        compute = torch.cuda.current_stream(self.local.device)
        # the gathered buffer is allocated on the COMPUTE stream (where wait() / gathered_view consume it) and lent to
        # the comm stream for the collective: both directions are recorded with the caching allocator
        g = torch.empty((self.world_size,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
        ready = torch.cuda.Event()
        ready.record(compute)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ready)
            all_gather_envs(slab, group=self.group, out=g)
            done = torch.cuda.Event()
            done.record(self.comm_stream)
        slab.record_stream(self.comm_stream)
        g.record_stream(self.comm_stream)
        return ro, (g, done)

    def wait(self, pending):
        g, ev = pending
        if ev is not None:
            torch.cuda.current_stream(self.local.device).wait_event(ev)
        return gathered_view(g)

    def close(self):
        self.local.close()
