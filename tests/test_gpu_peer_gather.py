"""evac_peer_gather (include/evac.h, csrc/evac_gather.h): the all-gather of the observation columns as peer stores.  Here
every "peer" buffer lives on the one GPU of the test box (the kernel does not care where a destination pointer points); the
hipIpc mapping between processes is covered by tests/test_gpu_bench_ranks.py (--gather peer)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from evacuation_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("rows,row_words,take,world,wgs", [
    (20 * 512, 9, 6, 4, 8),          # the bench's records: observation columns of [obs | reward | flags]
    (20 * 512, 9, 9, 2, 0),          # the whole record (contiguous), default workgroup count
    (1000, 13, 10, 3, 5),            # run-time column count, sizes that are no multiple of anything
    (7, 9, 6, 8, 8),                 # fewer elements than one workgroup
    (4096 * 20, 9, 6, 8, 8),         # the driver's chunk: 4096 envs x 20 steps
])
def test_every_rank_slice_lands_in_every_buffer(lib, rows, row_words, take, world, wgs):
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(rows + take)
    slabs = [torch.randn((rows, row_words), generator=g).to(dev) for _ in range(world)]
    bufs = [torch.full((world, rows, take), float("nan"), device=dev) for _ in range(world)]
    ptrs = (C.c_void_p * world)(*[b.data_ptr() for b in bufs])
    st = torch.cuda.current_stream()
    for r in range(world):                                    # "rank" r writes its slice into all buffers
        rc = lib.evac_peer_gather(C.c_void_p(slabs[r].data_ptr()), rows, row_words, take, ptrs, world, r, wgs, C.c_void_p(st.cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    want = torch.stack([s[:, :take] for s in slabs])
    for b in bufs:
        assert torch.equal(b, want)


def test_arguments_are_checked(lib):
    import torch
    from evacuation_amd import _lib
    dev = torch.device("cuda:0")
    slab, buf = torch.zeros((8, 9), device=dev), torch.zeros((2, 8, 6), device=dev)
    ptrs = (C.c_void_p * 2)(buf.data_ptr(), buf.data_ptr())
    src = C.c_void_p(slab.data_ptr())
    assert lib.evac_peer_gather(src, 8, 9, 10, ptrs, 2, 0, 8, None) == _lib.ERR_INVALID_ARGUMENT      # more columns than the record has
    assert lib.evac_peer_gather(src, 8, 9, 6, ptrs, 2, 2, 8, None) == _lib.ERR_INVALID_ARGUMENT       # rank outside the world
    assert lib.evac_peer_gather(src, 8, 9, 6, ptrs, 17, 0, 8, None) == _lib.ERR_INVALID_ARGUMENT      # more peers than the kernel takes
    assert lib.evac_peer_gather(None, 8, 9, 6, ptrs, 2, 0, 8, None) == _lib.ERR_INVALID_ARGUMENT
    nul = (C.c_void_p * 2)(buf.data_ptr(), None)
    assert lib.evac_peer_gather(src, 8, 9, 6, nul, 2, 0, 8, None) == _lib.ERR_INVALID_ARGUMENT
