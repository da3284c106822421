"""libbtcomm.so (include/btcomm.h) on the GPU: the RCCL exchange steps.  The box of the GPU tests has one GPU, so the collectives are
exercised with a one-rank communicator (every call must then behave as its definition says for world_size = 1); with two or more GPUs
visible a two-process run checks the real exchange (all-reduce sums, gather in rank order, all-to-all routing)."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _comm_lib():
    from bayestyper_amd import lib

    dll = C.CDLL(os.path.join(ROOT, "bayestyper_amd", "libbtcomm.so"))
    vp = C.c_void_p
    dll.bt_comm_unique_id.argtypes = [vp]
    dll.bt_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    dll.bt_comm_destroy.argtypes = [vp]
    dll.bt_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    dll.bt_comm_allreduce_hist.argtypes = [vp, vp, C.c_uint64]
    dll.bt_comm_gather_summaries.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
    dll.bt_comm_alltoallv_matches.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]
    dll.bt_comm_allgatherv.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
    return lib, dll


def _rank_body(rank, world, id_bytes, queue):
    lib, dll = _comm_lib()
    ctx = lib.Ctx(rank)
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)(*id_bytes)
    lib.check(dll.bt_comm_init(ctx.h, ident, rank, world, C.byref(h)))
    r, w = C.c_int(), C.c_int()
    lib.check(dll.bt_comm_rank(h, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (rank, world)
    # all-reduce: rank r contributes (r + 1) * arange
    hist = (np.arange(3 * 256, dtype=np.uint64) * np.uint64(rank + 1))
    d = ctx.to_device(hist)
    lib.check(dll.bt_comm_allreduce_hist(h, d.ptr, len(hist)))
    ctx.sync()
    got = d.download(np.uint64, len(hist))
    assert np.array_equal(got, np.arange(3 * 256, dtype=np.uint64) * np.uint64(world * (world + 1) // 2))
    # gather: rank r sends 10 + 3r words of value r
    local = np.full(10 + 3 * rank, rank, np.uint32)
    dl = ctx.to_device(local)
    total = sum(10 + 3 * q for q in range(world))
    out = ctx.buffer(4 * total)
    offs = np.zeros(world + 1, np.uint64)
    lib.check(dll.bt_comm_gather_summaries(h, dl.ptr, len(local), out.ptr, total, offs.ctypes.data))
    ctx.sync()
    assert list(offs) == [sum(10 + 3 * q for q in range(i)) for i in range(world + 1)]
    if rank == 0:
        g = out.download(np.uint32, total)
        assert np.array_equal(g, np.concatenate([np.full(10 + 3 * q, q, np.uint32) for q in range(world)]))
    # all-to-all: rank r sends (r + q + 1) * 18 bytes of value 16 * r + q to rank q
    send_sizes = np.array([(rank + q + 1) * 18 for q in range(world)], np.uint64)
    send = np.concatenate([np.full(int(send_sizes[q]), 16 * rank + q, np.uint8) for q in range(world)])
    ds = ctx.to_device(send)
    recv_total = sum((q + rank + 1) * 18 for q in range(world))
    dr = ctx.buffer(recv_total)
    recv_sizes = np.zeros(world, np.uint64)
    lib.check(dll.bt_comm_alltoallv_matches(h, ds.ptr, send_sizes.ctypes.data, dr.ptr, recv_total, recv_sizes.ctypes.data))
    ctx.sync()
    assert list(recv_sizes) == [(q + rank + 1) * 18 for q in range(world)]
    assert np.array_equal(dr.download(np.uint8, recv_total), np.concatenate([np.full((q + rank + 1) * 18, 16 * q + rank, np.uint8) for q in range(world)]))
    # too small a receive buffer is an error, not a truncation — and an error on EVERY rank, also when only ONE rank's buffer is short
    # (the capacities travel with the sizes: nobody is left waiting in a send)
    assert dll.bt_comm_alltoallv_matches(h, ds.ptr, send_sizes.ctypes.data, dr.ptr, recv_total - (1 if rank == world - 1 else 0), recv_sizes.ctypes.data) != 0
    assert dll.bt_comm_gather_summaries(h, dl.ptr, len(local), out.ptr, total - (1 if rank == 0 else 0), offs.ctypes.data) != 0
    assert dll.bt_comm_alltoallv_matches(h, None, send_sizes.ctypes.data, dr.ptr, recv_total, recv_sizes.ctypes.data) != 0     # null send buffer with sizes
    # all-gather of variable-length byte strings: rank r contributes 7 + 5r bytes of value 100 + r
    mine = np.full(7 + 5 * rank, 100 + rank, np.uint8)
    dm = ctx.to_device(mine)
    ag_total = sum(7 + 5 * q for q in range(world))
    dg = ctx.buffer(ag_total)
    ag_offs = np.zeros(world + 1, np.uint64)
    lib.check(dll.bt_comm_allgatherv(h, dm.ptr, len(mine), dg.ptr, ag_total, ag_offs.ctypes.data))
    ctx.sync()
    assert list(ag_offs) == [sum(7 + 5 * q for q in range(i)) for i in range(world + 1)]
    assert np.array_equal(dg.download(np.uint8, ag_total), np.concatenate([np.full(7 + 5 * q, 100 + q, np.uint8) for q in range(world)]))
    assert dll.bt_comm_allgatherv(h, dm.ptr, len(mine), dg.ptr, ag_total - (1 if rank == 0 else 0), ag_offs.ctypes.data) != 0
    dm.free(), dg.free()
    lib.check(dll.bt_comm_destroy(h))
    for b in (d, dl, out, ds, dr):
        b.free()
    ctx.close()
    if queue is not None:
        queue.put(rank)


def test_one_rank_communicator(gpu_ctx):
    _, dll = _comm_lib()
    ident = (C.c_uint8 * 128)()
    assert dll.bt_comm_unique_id(ident) == 0
    _rank_body(0, 1, bytes(ident), None)


def test_two_ranks_when_two_gpus_are_visible():
    from bayestyper_amd import lib

    n = C.c_int()
    lib.bt_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip("one GPU visible: the two-rank exchange needs two")
    _, dll = _comm_lib()
    ident = (C.c_uint8 * 128)()
    assert dll.bt_comm_unique_id(ident) == 0
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_rank_body, args=(r, 2, bytes(ident), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1] and all(p.exitcode == 0 for p in procs)
