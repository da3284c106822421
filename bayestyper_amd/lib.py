"""ctypes binding of libbtgpu.so (include/btgpu.h).  No fallback: a missing library is an ImportError."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BTGPU_LIB", os.path.join(_HERE, "libbtgpu.so"))   # BTGPU_LIB: alternative build of the same library (tuning experiments)

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
bt_ctx_use_default_stream = _sig("bt_ctx_use_default_stream", [vp])
bt_sync = _sig("bt_sync", [vp])
bt_ctx_info = _sig("bt_ctx_info", [vp, C.POINTER(C.c_int), u64p, u64p, C.c_char_p, C.c_size_t])
bt_malloc = _sig("bt_malloc", [vp, C.c_size_t, C.POINTER(vp)])
bt_free = _sig("bt_free", [vp, vp])
bt_memset = _sig("bt_memset", [vp, vp, C.c_int, C.c_size_t])
bt_memcpy_h2d = _sig("bt_memcpy_h2d", [vp, vp, vp, C.c_size_t])
bt_memcpy_d2h = _sig("bt_memcpy_d2h", [vp, vp, vp, C.c_size_t])
bt_memcpy_d2d = _sig("bt_memcpy_d2d", [vp, vp, vp, C.c_size_t])
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
bt_table_clear = _sig("bt_table_clear", [vp])
bt_table_reserve = _sig("bt_table_reserve", [vp, C.c_uint64])
bt_table_insert_batch = _sig("bt_table_insert_batch", [vp, vp, C.c_uint64, C.c_int])
bt_table_find_batch = _sig("bt_table_find_batch", [vp, vp, C.c_uint64, vp])
bt_table_read_slots = _sig("bt_table_read_slots", [vp, vp, C.c_uint64, vp, vp])
bt_table_export = _sig("bt_table_export", [vp, vp, vp, vp, C.c_uint64, u64p])
bt_find_paths_create = _sig("bt_find_paths_create", [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)])
bt_find_paths_destroy = _sig("bt_find_paths_destroy", [vp])
bt_find_paths_sample = _sig("bt_find_paths_sample", [vp, vp, vp])
bt_find_paths_sizes = _sig("bt_find_paths_sizes", [vp, vp, u64p])
bt_find_paths_fetch = _sig("bt_find_paths_fetch", [vp, vp])
bt_paths_create = _sig("bt_paths_create", [vp, vp, C.c_uint32, C.POINTER(vp), u64p])
bt_paths_destroy = _sig("bt_paths_destroy", [vp])
bt_paths_count_kmers = _sig("bt_paths_count_kmers", [vp, vp])
bt_paths_count_multigroup = _sig("bt_paths_count_multigroup", [vp, vp, vp, vp, u64p])
bt_paths_classify = _sig("bt_paths_classify", [vp, vp, vp, vp, vp])
bt_paths_candidates = _sig("bt_paths_candidates", [vp, vp, vp])
bt_paths_candidates_fetch = _sig("bt_paths_candidates_fetch", [vp, vp])
bt_table_count_parameter_kmers = _sig("bt_table_count_parameter_kmers", [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_float])
bt_table_kmer_stats = _sig("bt_table_kmer_stats", [vp, vp, vp, vp, vp, vp, vp])
bt_table_count_intercluster = _sig("bt_table_count_intercluster", [vp, vp, vp, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32])
bt_table_classify_batch = _sig("bt_table_classify_batch", [vp, vp, vp, vp, C.c_uint64, vp])
bt_kmc_scan_create = _sig("bt_kmc_scan_create", [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.POINTER(vp)])
bt_kmc_scan_create_bins = _sig("bt_kmc_scan_create_bins", [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.c_uint64, C.POINTER(vp)])
bt_kmc_scan_make_bloom = _sig("bt_kmc_scan_make_bloom", [vp, vp, vp, C.c_uint64, C.c_uint64])
bt_kmc_scan_destroy = _sig("bt_kmc_scan_destroy", [vp])
bt_kmc_scan_set_count_range = _sig("bt_kmc_scan_set_count_range", [vp, C.c_uint32, C.c_uint64])
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

    def clone(self):
        """bt_ctx_clone: a second context on the same GPU whose stream runs concurrently with this one's"""
        h = vp()
        check(bt_ctx_clone(self.h, C.byref(h)))
        c = Ctx.__new__(Ctx)
        c.h = h.value
        return c

    def sync(self):
        check(bt_sync(self.h))

    def set_stream(self, stream_ptr):
        """hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream); 0 = the device's default stream"""
        if not stream_ptr:
            check(bt_ctx_use_default_stream(self.h))
        else:
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

    def clear(self):
        check(bt_table_clear(self.h))

    def reserve(self, expected):
        check(bt_table_reserve(self.h, expected))

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

    def count_parameter_kmers(self, bloom, seq_bytes, starts, lens, decoy, seeds, fraction):
        """bt_table_count_parameter_kmers over regions of one sequence (uploaded here)"""
        d = self.ctx.to_device(np.frombuffer(seq_bytes, dtype=np.uint8))
        a = [np.ascontiguousarray(starts, np.uint64), np.ascontiguousarray(lens, np.uint64), np.ascontiguousarray(decoy, np.uint8), np.ascontiguousarray(seeds, np.uint32)]
        check(bt_table_count_parameter_kmers(self.h, bloom.h, d.ptr, len(a[0]), *[_np_ptr(x) for x in a], float(fraction)))
        self.ctx.sync()
        d.free()

    def kmer_stats(self, gender):
        """bt_table_kmer_stats -> (class_counts[7], dict of exact integer moments n / nonzero / sum / sumsq, each [S, 256])"""
        g = np.ascontiguousarray(gender, dtype=np.uint8)
        cls = np.zeros(7, np.uint64)
        mom = {k: np.zeros((self.num_samples, 256), np.uint64) for k in ("n", "nonzero", "sum", "sumsq")}
        check(bt_table_kmer_stats(self.h, _np_ptr(g), _np_ptr(cls), _np_ptr(mom["n"]), _np_ptr(mom["nonzero"]), _np_ptr(mom["sum"]), _np_ptr(mom["sumsq"])))
        return cls, mom

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


CAND_FIELDS = [("kmer_off", np.uint32), ("hap_kmer_mult", np.uint8), ("kmer_key", np.uint64), ("kmer_has_counts", np.uint8), ("kmer_counts", np.uint8),
               ("kmer_ic_mult", np.uint8), ("kv_off", np.uint32), ("kv_var", np.uint16), ("kv_bits", np.uint32), ("unique_off", np.uint32),
               ("unique_idx", np.uint32), ("multi_off", np.uint32), ("multi_idx", np.uint32), ("hap_allele", np.uint16), ("hapnest_off", np.uint32),
               ("hapnest_idx", np.uint32), ("nestdep_off", np.uint32), ("nestdep_cluster", np.uint32), ("nestdep_var_off", np.uint32), ("nestdep_var", np.uint16)]


class _CandOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in CAND_FIELDS]


class _CandSizes(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rows", "mult_bytes", "nnz", "kv_words", "num_unique", "num_multi", "hap_allele", "num_haplotypes", "hapnest",
                                          "nestdep", "nestdep_var")]


class Paths:
    """Path k-mer enumeration over flattened variant-cluster graphs (bt_paths_*; input: synth_graphs.flatten-style dict)"""

    def __init__(self, ctx, flat, k):
        from . import synth_graphs

        self.ctx, self.C, self.k = ctx, flat["num_clusters"], k
        batch, self._keep = synth_graphs.to_ctypes(flat)
        h, n = vp(), C.c_uint64()
        check(bt_paths_create(ctx.h, C.byref(batch), k, C.byref(h), C.byref(n)))
        self.h, self.num_windows = h.value, n.value

    def count_kmers(self, bloom):
        check(bt_paths_count_kmers(self.h, bloom.h))

    def count_multigroup(self, cluster_group, bloom, table):
        cg = np.ascontiguousarray(cluster_group, np.uint32)
        n = C.c_uint64()
        check(bt_paths_count_multigroup(self.h, _np_ptr(cg), bloom.h, table.h, C.byref(n)))
        return n.value

    def classify(self, table, mg_bloom):
        n = np.zeros(self.C, np.uint32)
        ex = np.zeros(self.C, np.uint8)
        check(bt_paths_classify(self.h, table.h, mg_bloom.h, _np_ptr(n), _np_ptr(ex)))
        return n, ex

    def candidates(self, table):
        sz = _CandSizes()
        check(bt_paths_candidates(self.h, table.h, C.byref(sz)))
        S = table.num_samples
        n = {"kmer_off": self.C + 1, "hap_kmer_mult": sz.mult_bytes, "kmer_key": sz.rows * 2, "kmer_has_counts": sz.rows, "kmer_counts": sz.rows * S,
             "kmer_ic_mult": sz.rows * 2, "kv_off": sz.rows + 1, "kv_var": sz.nnz, "kv_bits": sz.kv_words, "unique_off": self.C + 1, "unique_idx": sz.num_unique,
             "multi_off": self.C + 1, "multi_idx": sz.num_multi, "hap_allele": sz.hap_allele, "hapnest_off": sz.num_haplotypes + 1, "hapnest_idx": sz.hapnest,
             "nestdep_off": self.C + 1, "nestdep_cluster": sz.nestdep, "nestdep_var_off": sz.nestdep + 1, "nestdep_var": sz.nestdep_var}
        arrs = {name: np.zeros(max(int(n[name]), 1), dt) for name, dt in CAND_FIELDS}
        out = _CandOut()
        for name, _ in CAND_FIELDS:
            setattr(out, name, arrs[name].ctypes.data)
        check(bt_paths_candidates_fetch(self.h, C.byref(out)))
        return {name: arrs[name][: int(n[name])] for name, _ in CAND_FIELDS}

    def close(self):
        if self.h:
            bt_paths_destroy(self.h)
            self.h = None


class FindPaths:
    """Per-sample best-path search over flattened graphs (bt_find_paths_*)"""

    def __init__(self, ctx, flat, k, max_haps, num_samples):
        from . import synth_graphs

        self.ctx, self.C = ctx, flat["num_clusters"]
        self.nv = (flat["vertex_off"][1:] - flat["vertex_off"][:-1]).astype(np.int64)
        batch, self._keep = synth_graphs.to_ctypes(flat)
        h = vp()
        check(bt_find_paths_create(ctx.h, C.byref(batch), k, max_haps, num_samples, C.byref(h)))
        self.h = h.value

    def sample(self, bloom, seeds):
        sd = np.ascontiguousarray(seeds, np.uint32)
        check(bt_find_paths_sample(self.h, bloom.h, _np_ptr(sd)))

    def best_paths(self):
        n = np.zeros(self.C, np.uint32)
        tot = C.c_uint64()
        check(bt_find_paths_sizes(self.h, _np_ptr(n), C.byref(tot)))
        out = np.zeros(max(tot.value, 1), np.uint8)
        check(bt_find_paths_fetch(self.h, _np_ptr(out)))
        res, at = [], 0
        for c in range(self.C):
            m = int(n[c]) * int(self.nv[c])
            res.append(out[at:at + m].reshape(int(n[c]), int(self.nv[c])).copy())
            at += m
        return res

    def close(self):
        if self.h:
            bt_find_paths_destroy(self.h)
            self.h = None


class KmcScan:
    def __init__(self, ctx, k, p, counter_size, total, lut):
        self.ctx = ctx
        self.k, self.p, self.counter_size, self.total = k, p, counter_size, total
        self.rec_size = (k - p) // 4 + counter_size
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        h = vp()
        check(bt_kmc_scan_create_bins(ctx.h, k, p, counter_size, total, _np_ptr(lut), len(lut), C.byref(h)))   # len: 4^p + 1, or bins * 4^p + 1 (KMC2)
        self.h = h.value

    def set_count_range(self, min_count, max_count):
        check(bt_kmc_scan_set_count_range(self.h, min_count, max_count))

    def make_bloom(self, bloom, d_records_ptr, first_record, n):
        check(bt_kmc_scan_make_bloom(self.h, bloom.h, d_records_ptr, first_record, n))

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


# ---------------------------------------------------------------------------------------------------------------
# Gibbs genotyping (bt_gibbs_*)
# ---------------------------------------------------------------------------------------------------------------
bt_gibbs_create = _sig("bt_gibbs_create", [vp, vp, vp, C.POINTER(vp)])
bt_gibbs_destroy = _sig("bt_gibbs_destroy", [vp])
bt_gibbs_set_lut = _sig("bt_gibbs_set_lut", [vp, vp, vp])
bt_gibbs_set_noise_lut = _sig("bt_gibbs_set_noise_lut", [vp, vp])
bt_gibbs_init_chain = _sig("bt_gibbs_init_chain", [vp, C.c_uint32])
bt_gibbs_sweep = _sig("bt_gibbs_sweep", [vp, C.c_uint32, C.c_int])
bt_gibbs_run = _sig("bt_gibbs_run", [vp])
bt_gibbs_noise_counts = _sig("bt_gibbs_noise_counts", [vp, vp, C.c_int])
bt_gibbs_noise_iteration = _sig("bt_gibbs_noise_iteration", [vp, vp, C.c_int, vp])
bt_gibbs_noise_chain_begin = _sig("bt_gibbs_noise_chain_begin", [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)])
bt_gibbs_noise_chain_step = _sig("bt_gibbs_noise_chain_step", [vp, vp, vp])
bt_gibbs_noise_chain_end = _sig("bt_gibbs_noise_chain_end", [vp])
bt_ctx_clone = _sig("bt_ctx_clone", [vp, C.POINTER(vp)])
bt_gibbs_reset_groups = _sig("bt_gibbs_reset_groups", [vp])
bt_gibbs_result_sizes = _sig("bt_gibbs_result_sizes", [vp, u64p, u64p])
bt_gibbs_result_fetch = _sig("bt_gibbs_result_fetch", [vp] * 7)
bt_gibbs_result_words = _sig("bt_gibbs_result_words", [vp, C.POINTER(vp), u64p])
bt_gibbs_trace_enable = _sig("bt_gibbs_trace_enable", [vp, C.c_uint32])
bt_gibbs_trace_fetch = _sig("bt_gibbs_trace_fetch", [vp, vp, C.c_uint64, u64p])
bt_gibbs_posterior_summary = _sig("bt_gibbs_posterior_summary", [vp, vp])
bt_gibbs_device_bytes = _sig("bt_gibbs_device_bytes", [vp, u64p])
bt_diag_uset_replay = _sig("bt_diag_uset_replay", [C.c_uint32, vp, vp, C.c_uint64, vp, u32p])
bt_diag_rng = _sig("bt_diag_rng", [C.c_uint32, C.c_int, vp, vp, C.c_uint64, vp])
bt_diag_kmer_set_order = _sig("bt_diag_kmer_set_order", [vp, C.c_uint32, C.c_uint64, C.c_uint, vp, vp])


bt_noise_model_create = _sig("bt_noise_model_create", [vp, C.c_uint32, vp, C.POINTER(vp)])
bt_noise_model_destroy = _sig("bt_noise_model_destroy", [vp])
bt_noise_model_set_rng = _sig("bt_noise_model_set_rng", [vp, vp])
bt_noise_model_get_rng = _sig("bt_noise_model_get_rng", [vp, vp])
NOISE_REDUCE = C.CFUNCTYPE(C.c_int, vp, vp, C.c_uint64)
bt_gibbs_noise_chain = _sig("bt_gibbs_noise_chain", [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp])


class NoiseRng(C.Structure):   # include/btgpu.h: bt_noise_rng
    _fields_ = [("mt", C.c_uint32 * 624), ("mt_pos", C.c_uint32), ("saved_available", C.c_uint32), ("saved", C.c_double)]


class NoiseModel:
    """the noise half of the count model on the device (bt_noise_model_*): priors [(shape, scale)] per sample + the run's generator"""

    def __init__(self, ctx, priors):
        self.ctx, self.S = ctx, len(priors)
        pr = np.ascontiguousarray(np.asarray(priors, np.float32).reshape(-1))
        h = vp()
        check(bt_noise_model_create(ctx.h, self.S, _np_ptr(pr), C.byref(h)))
        self.h = h.value

    def set_generator(self, words626, saved):
        """words626 / saved as count_model.CountDistribution.export_generator() returns them"""
        r = NoiseRng()
        C.memmove(r.mt, np.ascontiguousarray(words626[:624], np.uint32).ctypes.data, 624 * 4)
        r.mt_pos, r.saved_available, r.saved = int(words626[624]), int(words626[625]), float(saved)
        check(bt_noise_model_set_rng(self.h, C.addressof(r)))

    def get_generator(self):
        r = NoiseRng()
        check(bt_noise_model_get_rng(self.h, C.addressof(r)))
        w = np.zeros(626, np.uint32)
        w[:624] = np.frombuffer(r.mt, np.uint32)
        w[624], w[625] = r.mt_pos, r.saved_available
        return w, r.saved

    def chain(self, gibbs, num_iterations, first_collect, reduce=None):
        """bt_gibbs_noise_chain -> rates [num_iterations, S]; reduce(d_hist_ptr, n) must only enqueue work on the context's stream"""
        rates = np.zeros(num_iterations * self.S)
        cb = None
        if reduce is not None:
            def hook(_user, d_hist, n):
                try:
                    reduce(d_hist, n)
                    return 0
                except Exception:   # an exception must not unwind through the library's frames
                    return 1
            cb = NOISE_REDUCE(hook)
        check(bt_gibbs_noise_chain(gibbs.h if gibbs is not None else None, self.h, num_iterations, first_collect, C.cast(cb, vp) if cb is not None else None, None, _np_ptr(rates)))
        return rates.reshape(num_iterations, self.S)

    def close(self):
        if self.h:
            bt_noise_model_destroy(self.h)
            self.h = None


def parse_result_words(w):
    """one launch's word string (bt_gibbs_result_words) -> (the dictionary Gibbs.results() returns, words consumed)"""
    w = np.ascontiguousarray(w, np.uint32)
    Cn, nd, nc, S = (int(x) for x in w[:4])
    sizes = w[4:4 + 2 * Cn].reshape(Cn, 2).astype(np.uint64)
    at = 4 + 2 * Cn
    keys = w[at:at + nd]
    at += nd
    freq = w[at:at + nd * S].reshape(nd, S)
    at = (at + nd * S + 1) & ~1
    stats = w[at:at + nc * 24].copy().view(np.float64).reshape(nc, 3, 4)
    at += nc * 24
    dip_off = np.concatenate([[0], np.cumsum(sizes[:, 0])]).astype(np.uint64)
    cell_off = np.concatenate([[0], np.cumsum(sizes[:, 1])]).astype(np.uint64)
    return {"dip_off": dip_off, "h1": (keys & 0xFFFF).astype(np.uint16), "h2": (keys >> 16).astype(np.uint16), "freq": freq.copy(), "cell_off": cell_off, "stats": stats}, at


class Gibbs:
    """A batch of variant-cluster groups on one GPU (bt_gibbs_*).  `flat` is a dict as produced by bayestyper_amd.synth."""

    def __init__(self, ctx, flat, lut_g, lut_n, **kw):
        from . import synth

        self.ctx, self.flat = ctx, flat
        self.S, self.C, self.G = flat["S"], flat["num_clusters"], flat["num_groups"]
        self.params, self.batch, self._keep = synth.to_ctypes(flat, **kw)
        h = vp()
        check(bt_gibbs_create(ctx.h, C.addressof(self.params), C.addressof(self.batch), C.byref(h)))
        self.h = h.value
        if lut_g is not None:
            self.set_lut(lut_g, lut_n)

    def set_lut(self, lut_g, lut_n):
        lut_g, lut_n = np.ascontiguousarray(lut_g, np.float64), np.ascontiguousarray(lut_n, np.float64)
        check(bt_gibbs_set_lut(self.h, _np_ptr(lut_g), _np_ptr(lut_n)))

    def set_noise_lut(self, lut_n):
        lut_n = np.ascontiguousarray(lut_n, np.float64)
        check(bt_gibbs_set_noise_lut(self.h, _np_ptr(lut_n)))

    def run(self):
        check(bt_gibbs_run(self.h))

    def init_chain(self, c):
        check(bt_gibbs_init_chain(self.h, c))

    def sweep(self, n, collect):
        check(bt_gibbs_sweep(self.h, n, int(collect)))

    def noise_counts(self, zero_first=True):
        d = self.ctx.buffer(self.S * 256 * 8)
        check(bt_gibbs_noise_counts(self.h, d.ptr, int(zero_first)))
        self.ctx.sync()
        out = d.download(np.uint64, self.S * 256)
        d.free()
        return out

    def noise_iteration(self, lut_n, collect):
        """bt_gibbs_noise_iteration: (the table for this iteration's sweep or None;) one sweep; noise counts [S*256] + clearGenotyperCache"""
        if lut_n is not None:
            lut_n = np.ascontiguousarray(lut_n, np.float64)
        h = np.zeros(self.S * 256, np.uint64)
        check(bt_gibbs_noise_iteration(self.h, _np_ptr(lut_n) if lut_n is not None else None, int(collect), _np_ptr(h)))
        return h

    def noise_chain_begin(self, num_iterations, first_collect):
        """bt_gibbs_noise_chain_begin -> True when the chain runs as one resident launch (then noise_chain_step per iteration, noise_chain_end)"""
        r = C.c_int(0)
        check(bt_gibbs_noise_chain_begin(self.h, num_iterations, first_collect, C.byref(r)))
        return bool(r.value)

    def noise_chain_step(self, lut_n):
        if lut_n is not None:
            lut_n = np.ascontiguousarray(lut_n, np.float64)
        h = np.zeros(self.S * 256, np.uint64)
        check(bt_gibbs_noise_chain_step(self.h, _np_ptr(lut_n) if lut_n is not None else None, _np_ptr(h)))
        return h

    def noise_chain_end(self):
        check(bt_gibbs_noise_chain_end(self.h))

    def reset_groups(self):
        check(bt_gibbs_reset_groups(self.h))

    def posterior_summary(self):
        """bt_gibbs_posterior_summary -> uint32 [C, S, 2] on the host (the device buffer is what a multi-GPU run gathers)"""
        buf = DeviceBuffer(self.ctx, self.C * self.S * 2 * 4)
        check(bt_gibbs_posterior_summary(self.h, buf.ptr))
        self.ctx.sync()
        out = buf.download(np.uint32, self.C * self.S * 2).reshape(self.C, self.S, 2)
        buf.free()
        return out

    def device_bytes(self):
        b = C.c_uint64()
        check(bt_gibbs_device_bytes(self.h, C.byref(b)))
        return b.value

    def trace_enable(self, n):
        check(bt_gibbs_trace_enable(self.h, n))
        self._trace_n = n

    def trace(self):
        """-> list per group of arrays [sweeps][vertex][S]"""
        goff = self.flat["group_cluster_off"]
        nv = (goff[1:] - goff[:-1]).astype(np.int64)
        total = int((nv * self._trace_n * self.S).sum())
        buf = np.zeros(max(total, 1), np.uint32)
        n0 = C.c_uint64()
        check(bt_gibbs_trace_fetch(self.h, _np_ptr(buf), len(buf), C.byref(n0)))
        out, off = [], 0
        for g in range(self.G):
            w = int(nv[g]) * self._trace_n * self.S
            out.append(buf[off:off + w].reshape(self._trace_n, int(nv[g]), self.S))
            off += w
        return out

    def results(self):
        self.ctx.sync()
        nd, nc = C.c_uint64(), C.c_uint64()
        check(bt_gibbs_result_sizes(self.h, C.byref(nd), C.byref(nc)))
        nd, nc = nd.value, nc.value
        dip_off = np.zeros(self.C + 1, np.uint64)
        cell_off = np.zeros(self.C + 1, np.uint64)
        h1, h2 = np.zeros(max(nd, 1), np.uint16), np.zeros(max(nd, 1), np.uint16)
        freq = np.zeros(max(nd, 1) * self.S, np.uint32)
        stats = np.zeros(max(nc, 1) * 12, np.float64)
        check(bt_gibbs_result_fetch(self.h, _np_ptr(dip_off), _np_ptr(h1), _np_ptr(h2), _np_ptr(freq), _np_ptr(cell_off), _np_ptr(stats)))
        return {"dip_off": dip_off, "h1": h1[:nd], "h2": h2[:nd], "freq": freq[: nd * self.S].reshape(nd, self.S), "cell_off": cell_off,
                "stats": stats[: nc * 12].reshape(nc, 3, 4)}

    def result_words(self):
        """the results as one word string in DEVICE memory (bt_gibbs_result_words): (device pointer, words); the sampler owns the buffer"""
        p, n = vp(), C.c_uint64()
        check(bt_gibbs_result_words(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def result_words_host(self):
        p, n = self.result_words()
        out = np.zeros(n, np.uint32)
        check(bt_memcpy_d2h(self.ctx.h, _np_ptr(out), p, out.nbytes))
        return out

    def close(self):
        if self.h:
            bt_gibbs_destroy(self.h)
            self.h = None
