"""The exchange layer (include/epievo_mi355x_comm.h -> libepv_rccl.so) on its own, through ctypes:
  * loopback transport -- three ranks sharing device 0, what EPV_DEVICES=0,0,0 uses: the halo
    exchange moves every rank's edge buffers to the right neighbour's receive buffers and the
    all-gather concatenates the ranks' pieces in rank order;
  * RCCL -- one rank created from a unique id (ncclGetUniqueId / ncclCommInitRank, the
    one-process-per-GPU entry) and one created by ncclCommInitAll: all-gather and the (empty)
    exchange of a one-rank world, stream synchronisation, teardown.
More than one RCCL rank needs more than one GPU (RCCL refuses two ranks on a device)."""
import ctypes as C

import numpy as np
import pytest

from epievo_amd import _build
from epievo_amd.workloads import simulate

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def _lib():
    L = C.CDLL(_build.COMM_SO)
    L.epv_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.epv_comm_init_rank.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.epv_comm_get_unique_id.argtypes = [vp]
    L.epv_comm_destroy.argtypes = [vp]
    L.epv_comm_destroy.restype = None
    L.epv_comm_is_rccl.argtypes = [vp]
    L.epv_comm_rank.argtypes = [vp]
    L.epv_comm_world.argtypes = [vp]
    L.epv_comm_exchange.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64]
    L.epv_comm_all_gather.argtypes = [vp, vp, vp, C.c_uint64]
    L.epv_comm_sync.argtypes = [vp]
    L.epv_comm_last_error.argtypes = [vp]
    L.epv_comm_last_error.restype = C.c_char_p
    return L


def _dev():
    from epievo_amd.sampler import DeviceSampler
    model, tree, fp = simulate("tree", 64, seed=1)
    d = DeviceSampler(0)
    d.set_tree(tree); d.set_model(model); d.upload_paths(fp, 16)
    return d


def test_loopback_three_ranks_on_one_gpu():
    L, d = _lib(), _dev()
    n = 3
    comms = (vp * n)()
    devs = (C.c_int * n)(0, 0, 0)
    assert L.epv_comm_init_all(n, devs, comms) == 0
    assert [L.epv_comm_rank(comms[r]) for r in range(n)] == [0, 1, 2]
    assert L.epv_comm_world(comms[0]) == 3 and L.epv_comm_is_rccl(comms[0]) == 0
    nb = 4096
    bufs = [[d.alloc(nb) for _ in range(4)] for _ in range(n)]      # send prev, recv prev, send next, recv next
    for r in range(n):
        d.write(bufs[r][0], 0, np.full(nb // 8, 100.0 + r))         # what rank r sends to r-1
        d.write(bufs[r][2], 0, np.full(nb // 8, 200.0 + r))         # ... and to r+1
    assert L.epv_comm_group_start() == 0
    for r in range(n):
        assert L.epv_comm_exchange(comms[r], bufs[r][0].ptr, bufs[r][1].ptr, nb, bufs[r][2].ptr, bufs[r][3].ptr, nb) == 0
    assert L.epv_comm_group_end() == 0
    for r in range(n):
        assert L.epv_comm_sync(comms[r]) == 0
        if r > 0:
            assert np.all(d.read(bufs[r][1], 0, nb // 8) == 200.0 + (r - 1))     # from the left neighbour
        if r < n - 1:
            assert np.all(d.read(bufs[r][3], 0, nb // 8) == 100.0 + (r + 1))     # from the right neighbour
    piece = [d.alloc(1024) for _ in range(n)]
    gathered = [d.alloc(1024 * n) for _ in range(n)]
    for r in range(n):
        d.write(piece[r], 0, np.arange(128, dtype=np.float64) + 1000 * r)
    assert L.epv_comm_group_start() == 0
    for r in range(n):
        assert L.epv_comm_all_gather(comms[r], piece[r].ptr, gathered[r].ptr, 1024) == 0
    assert L.epv_comm_group_end() == 0
    want = np.concatenate([np.arange(128.0) + 1000 * r for r in range(n)])
    for r in range(n):
        assert L.epv_comm_sync(comms[r]) == 0
        assert np.array_equal(d.read(gathered[r], 0, 128 * n), want)
    # outside a group bracket a multi-rank loopback call is refused, not half-executed
    assert L.epv_comm_all_gather(comms[0], piece[0].ptr, gathered[0].ptr, 1024) != 0
    for r in range(n):
        L.epv_comm_destroy(comms[r])
    d.close()


@pytest.mark.parametrize("how", ["init_all", "init_rank"])
def test_rccl_one_rank(how):
    L, d = _lib(), _dev()
    comm = vp()
    if how == "init_all":
        comms = (vp * 1)()
        devs = (C.c_int * 1)(0)
        assert L.epv_comm_init_all(1, devs, comms) == 0
        comm = comms[0]
    else:
        uid = C.create_string_buffer(128)
        assert L.epv_comm_get_unique_id(uid) == 0
        assert L.epv_comm_init_rank(0, 1, 0, uid, C.byref(comm)) == 0
        comm = comm.value
    assert L.epv_comm_is_rccl(comm) == 1 and L.epv_comm_world(comm) == 1
    piece, gathered = d.alloc(2048), d.alloc(2048)
    d.write(piece, 0, np.arange(256, dtype=np.float64))
    assert L.epv_comm_group_start() == 0
    assert L.epv_comm_all_gather(comm, piece.ptr, gathered.ptr, 2048) == 0, L.epv_comm_last_error(comm)
    assert L.epv_comm_exchange(comm, None, None, 0, None, None, 0) == 0          # no neighbours: nothing to do
    assert L.epv_comm_group_end() == 0
    assert L.epv_comm_sync(comm) == 0, L.epv_comm_last_error(comm)
    assert np.array_equal(d.read(gathered, 0, 256), np.arange(256.0))
    L.epv_comm_destroy(comm)
    d.close()
