"""ctypes binding of libbtcomm.so (include/btcomm.h): the RCCL exchange steps of the sharded path, one rank per GPU.
Used by bench.py and the tests; the executable reaches the same library through host/Comm.cpp."""
import ctypes as C
import os

import numpy as np

from . import lib

_HERE = os.path.dirname(os.path.abspath(__file__))
ID_BYTES = 128
vp = C.c_void_p
_dll = None


def dll():
    global _dll
    if _dll is None:
        path = os.path.join(_HERE, "libbtcomm.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run bayestyper_amd/csrc/build.sh (or __graft_entry__.build())")
        d = C.CDLL(path)
        d.bt_comm_unique_id.argtypes = [vp]
        d.bt_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
        d.bt_comm_destroy.argtypes = [vp]
        d.bt_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        d.bt_comm_allreduce_hist.argtypes = [vp, vp, C.c_uint64]
        d.bt_comm_gather_summaries.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
        d.bt_comm_allgatherv.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
        d.bt_comm_alltoallv_matches.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]
        _dll = d
    return _dll


def unique_id():
    buf = (C.c_uint8 * ID_BYTES)()
    lib.check(dll().bt_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """one rank of a communicator on ctx's GPU and stream"""

    def __init__(self, ctx, ident, rank, world):
        self.ctx, self.rank, self.world = ctx, rank, world
        h = vp()
        lib.check(dll().bt_comm_init(ctx.h, (C.c_uint8 * ID_BYTES)(*ident), rank, world, C.byref(h)))
        self.h = h

    def allreduce(self, d_ptr, n):
        """in-place sum of n device uint64 counters over all ranks"""
        lib.check(dll().bt_comm_allreduce_hist(self.h, d_ptr, n))

    def gather_words(self, d_local_ptr, local_words, d_out_ptr, out_capacity_words):
        """-> word offsets of the ranks' parts (rank 0's d_out receives them in rank order)"""
        offs = np.zeros(self.world + 1, np.uint64)
        lib.check(dll().bt_comm_gather_summaries(self.h, d_local_ptr, local_words, d_out_ptr, out_capacity_words, offs.ctypes.data))
        return offs

    def allgather_bytes(self, d_local_ptr, local_bytes, d_out_ptr, out_capacity):
        offs = np.zeros(self.world + 1, np.uint64)
        lib.check(dll().bt_comm_allgatherv(self.h, d_local_ptr, local_bytes, d_out_ptr, out_capacity, offs.ctypes.data))
        return offs

    def barrier(self):
        d = self.ctx.to_device(np.ones(1, np.uint64))
        self.allreduce(d.ptr, 1)
        self.ctx.sync()
        d.free()

    def close(self):
        if self.h:
            dll().bt_comm_destroy(self.h)
            self.h = None
