"""ctypes binding of libbtgpu.so (include/btgpu.h).  No fallback: a missing library is an ImportError."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbtgpu.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
        "or bayestyper_amd/csrc/build.sh). There is no CPU fallback."
    )

_lib = C.CDLL(LIB_PATH)

u8p, u32p, u64p, i64p, f64p, f32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int64, C.c_double, C.c_float))
vp = C.c_void_p


def _sig(name, argtypes, restype=C.c_int):
    f = getattr(_lib, name)
    f.argtypes = argtypes
    f.restype = restype
    return f


bt_last_error = _sig("bt_last_error", [], C.c_char_p)
bt_version = _sig("bt_version", [])
bt_device_count = _sig("bt_device_count", [C.POINTER(C.c_int)])
bt_ctx_create = _sig("bt_ctx_create", [C.c_int, C.POINTER(vp)])
bt_ctx_destroy = _sig("bt_ctx_destroy", [vp])
bt_ctx_set_stream = _sig("bt_ctx_set_stream", [vp, vp])
bt_sync = _sig("bt_sync", [vp])
bt_ctx_info = _sig("bt_ctx_info", [vp, C.POINTER(C.c_int), u64p, u64p, C.c_char_p, C.c_size_t])
bt_malloc = _sig("bt_malloc", [vp, C.c_size_t, C.POINTER(vp)])
bt_free = _sig("bt_free", [vp, vp])
bt_memset = _sig("bt_memset", [vp, vp, C.c_int, C.c_size_t])
bt_memcpy_h2d = _sig("bt_memcpy_h2d", [vp, vp, vp, C.c_size_t])
bt_memcpy_d2h = _sig("bt_memcpy_d2h", [vp, vp, vp, C.c_size_t])
bt_timer_create = _sig("bt_timer_create", [vp, C.POINTER(vp)])
bt_timer_destroy = _sig("bt_timer_destroy", [vp])
bt_timer_start = _sig("bt_timer_start", [vp])
bt_timer_stop = _sig("bt_timer_stop", [vp])
bt_timer_elapsed_ms = _sig("bt_timer_elapsed_ms", [vp, f32p])
bt_kmers_from_sequence = _sig("bt_kmers_from_sequence", [vp, vp, C.c_uint64, C.c_uint32, vp, vp])
bt_nthash_batch = _sig("bt_nthash_batch", [vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, vp])
bt_bloom_create = _sig("bt_bloom_create", [vp, C.c_uint64, C.c_float, C.c_uint32, C.c_int, C.POINTER(vp)])
bt_bloom_load = _sig("bt_bloom_load", [vp, C.c_char_p, C.c_uint32, C.POINTER(vp)])
bt_bloom_save = _sig("bt_bloom_save", [vp, C.c_char_p])
bt_bloom_destroy = _sig("bt_bloom_destroy", [vp])
bt_bloom_info = _sig("bt_bloom_info", [vp, u64p, u64p, u32p, u32p, u64p])
bt_bloom_insert_batch = _sig("bt_bloom_insert_batch", [vp, vp, C.c_uint64])
bt_bloom_contains_batch = _sig("bt_bloom_contains_batch", [vp, vp, C.c_uint64, vp])
bt_bloom_read_bits = _sig("bt_bloom_read_bits", [vp, C.c_uint32, vp, C.c_uint64])
bt_bloom_clear = _sig("bt_bloom_clear", [vp])
bt_table_create = _sig("bt_table_create", [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(vp)])
bt_table_destroy = _sig("bt_table_destroy", [vp])
bt_table_status = _sig("bt_table_status", [vp, u64p, u64p, C.POINTER(C.c_int)])
bt_table_insert_batch = _sig("bt_table_insert_batch", [vp, vp, C.c_uint64, C.c_int])
bt_table_find_batch = _sig("bt_table_find_batch", [vp, vp, C.c_uint64, vp])
bt_table_read_slots = _sig("bt_table_read_slots", [vp, vp, C.c_uint64, vp, vp])
bt_table_export = _sig("bt_table_export", [vp, vp, vp, vp, C.c_uint64, u64p])
bt_table_count_intercluster = _sig("bt_table_count_intercluster", [vp, vp, vp, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32])
bt_table_classify_batch = _sig("bt_table_classify_batch", [vp, vp, vp, vp, C.c_uint64, vp])
bt_kmc_scan_create = _sig("bt_kmc_scan_create", [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.POINTER(vp)])
bt_kmc_scan_destroy = _sig("bt_kmc_scan_destroy", [vp])
bt_kmc_scan_run = _sig("bt_kmc_scan_run", [vp, vp, vp, C.c_uint32, vp, C.c_uint64, C.c_uint64, vp])
bt_kmc_scan_decode = _sig("bt_kmc_scan_decode", [vp, vp, C.c_uint64, C.c_uint64, vp, vp])



class BtError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise BtError(bt_last_error().decode())


def _np_ptr(a):
    return a.ctypes.data_as(vp)


class DeviceBuffer:
    """A device allocation owned by a Ctx (bt_malloc / bt_free)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = vp()
        check(bt_malloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(bt_memcpy_h2d(self.ctx.h, self.ptr, _np_ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(bt_memcpy_d2h(self.ctx.h, _np_ptr(out), self.ptr, out.nbytes))
        return out

    def zero(self):
        check(bt_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr:
            bt_free(self.ctx.h, self.ptr)
            self.ptr = None


class Ctx:
    def __init__(self, device=0):
        h = vp()
        check(bt_ctx_create(device, C.byref(h)))
        self.h = h.value

    def sync(self):
        check(bt_sync(self.h))

    def set_stream(self, stream_ptr):
        check(bt_ctx_set_stream(self.h, stream_ptr))

    def info(self):
        cu, tot, free = C.c_int(), C.c_uint64(), C.c_uint64()
        arch = C.create_string_buffer(64)
        check(bt_ctx_info(self.h, C.byref(cu), C.byref(tot), C.byref(free), arch, 64))
        return {"num_cu": cu.value, "hbm_total": tot.value, "hbm_free": free.value, "arch": arch.value.decode()}

    def buffer(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.nbytes).upload(arr)

    def close(self):
        if self.h:
            bt_ctx_destroy(self.h)
            self.h = None


class Timer:
    def __init__(self, ctx):
        h = vp()
        check(bt_timer_create(ctx.h, C.byref(h)))
        self.h = h.value

    def start(self):
        check(bt_timer_start(self.h))

    def stop(self):
        check(bt_timer_stop(self.h))

    def elapsed_ms(self):
        ms = C.c_float()
        check(bt_timer_elapsed_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        bt_timer_destroy(self.h)


class Bloom:
    """KmerBloom<k> (threaded=False) or ThreadedKmerBloom<k> (threaded=True) in HBM."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def create(cls, ctx, num_kmers, fpr, k, threaded):
        h = vp()
        check(bt_bloom_create(ctx.h, num_kmers, fpr, k, int(threaded), C.byref(h)))
        return cls(ctx, h.value)

    @classmethod
    def load(cls, ctx, prefix, k):
        h = vp()
        check(bt_bloom_load(ctx.h, prefix.encode(), k, C.byref(h)))
        return cls(ctx, h.value)

    def save(self, prefix):
        check(bt_bloom_save(self.h, prefix.encode()))

    def info(self):
        nk, nb, db = C.c_uint64(), C.c_uint64(), C.c_uint64()
        nh, ns = C.c_uint32(), C.c_uint32()
        check(bt_bloom_info(self.h, C.byref(nk), C.byref(nb), C.byref(nh), C.byref(ns), C.byref(db)))
        return {"num_kmers": nk.value, "num_bits": nb.value, "num_hashes": nh.value, "num_sub": ns.value, "device_bytes": db.value}

    def insert(self, packed):
        """packed: (n, 2) uint64 host array of canonical k-mers"""
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        d = self.ctx.to_device(packed)
        check(bt_bloom_insert_batch(self.h, d.ptr, len(packed)))
        self.ctx.sync()
        d.free()

    def contains(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        n = len(packed)
        d = self.ctx.to_device(packed)
        o = self.ctx.buffer(max(n, 1))
        check(bt_bloom_contains_batch(self.h, d.ptr, n, o.ptr))
        self.ctx.sync()
        out = o.download(np.uint8, n)
        d.free()
        o.free()
        return out

    def bits(self, sub=0):
        nbytes = (self.info()["num_bits"] + 7) // 8
        out = np.empty(nbytes, dtype=np.uint8)
        check(bt_bloom_read_bits(self.h, sub, _np_ptr(out), nbytes))
        return out

    def close(self):
        if self.h:
            bt_bloom_destroy(self.h)
            self.h = None


class Table:
    """ObservedKmerCountsHash<N> in HBM."""

    def __init__(self, ctx, expected, num_samples, k):
        self.ctx, self.num_samples, self.k = ctx, num_samples, k
        h = vp()
        check(bt_table_create(ctx.h, expected, num_samples, k, C.byref(h)))
        self.h = h.value

    def status(self):
        nk, cap, ov = C.c_uint64(), C.c_uint64(), C.c_int()
        check(bt_table_status(self.h, C.byref(nk), C.byref(cap), C.byref(ov)))
        return {"num_keys": nk.value, "capacity": cap.value, "overflowed": bool(ov.value)}

    def insert(self, packed, mark_parameter=False):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        d = self.ctx.to_device(packed)
        check(bt_table_insert_batch(self.h, d.ptr, len(packed), int(mark_parameter)))
        self.ctx.sync()
        d.free()

    def find(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        n = len(packed)
        d = self.ctx.to_device(packed)
        o = self.ctx.buffer(8 * max(n, 1))
        check(bt_table_find_batch(self.h, d.ptr, n, o.ptr))
        self.ctx.sync()
        out = o.download(np.int64, n)
        d.free()
        o.free()
        return out

    def count_intercluster(self, bloom, seq_bytes, is_decoy, female_ploidy, male_ploidy):
        arr = np.frombuffer(seq_bytes, dtype=np.uint8)
        d = self.ctx.to_device(arr)
        check(bt_table_count_intercluster(self.h, bloom.h, d.ptr, len(arr), int(is_decoy), female_ploidy, male_ploidy))
        self.ctx.sync()
        d.free()

    def classify(self, mg_bloom, packed, mult):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        mult = np.ascontiguousarray(mult, dtype=np.uint8)
        n = len(packed)
        d, m = self.ctx.to_device(packed), self.ctx.to_device(mult)
        o = self.ctx.buffer(max(n, 1))
        check(bt_table_classify_batch(self.h, mg_bloom.h, d.ptr, m.ptr, n, o.ptr))
        self.ctx.sync()
        out = o.download(np.uint8, n)
        for b in (d, m, o):
            b.free()
        return out

    def export(self):
        """-> (kmers (n,2) u64, counts (n,S) u8, meta (n,4) u8) sorted by (hi, lo)"""
        st = self.status()
        n = st["num_keys"]
        kmers = np.zeros((max(n, 1), 2), dtype=np.uint64)
        counts = np.zeros((max(n, 1), self.num_samples), dtype=np.uint8)
        meta = np.zeros((max(n, 1), 4), dtype=np.uint8)
        w = C.c_uint64()
        check(bt_table_export(self.h, _np_ptr(kmers), _np_ptr(counts), _np_ptr(meta), max(n, 1), C.byref(w)))
        n = w.value
        return kmers[:n], counts[:n], meta[:n]

    def close(self):
        if self.h:
            bt_table_destroy(self.h)
            self.h = None


class KmcScan:
    def __init__(self, ctx, k, p, counter_size, total, lut):
        self.ctx = ctx
        self.k, self.p, self.counter_size, self.total = k, p, counter_size, total
        self.rec_size = (k - p) // 4 + counter_size
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        h = vp()
        check(bt_kmc_scan_create(ctx.h, k, p, counter_size, total, _np_ptr(lut), C.byref(h)))
        self.h = h.value

    def run(self, bloom, table, sample_idx, d_records_ptr, first_record, n, d_hits_ptr=None):
        check(bt_kmc_scan_run(self.h, bloom.h, table.h, sample_idx, d_records_ptr, first_record, n, d_hits_ptr))

    def decode(self, payload, first_record, n):
        d = self.ctx.to_device(np.frombuffer(payload, dtype=np.uint8))
        ok = self.ctx.buffer(16 * max(n, 1))
        oc = self.ctx.buffer(4 * max(n, 1))
        check(bt_kmc_scan_decode(self.h, d.ptr, first_record, n, ok.ptr, oc.ptr))
        self.ctx.sync()
        kmers = ok.download(np.uint64, 2 * n).reshape(n, 2)
        counts = oc.download(np.uint32, n)
        for b in (d, ok, oc):
            b.free()
        return kmers, counts

    def close(self):
        if self.h:
            bt_kmc_scan_destroy(self.h)
            self.h = None
