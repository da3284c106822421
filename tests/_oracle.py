"""Test-side loader for the oracle (oracle/liboracle.so) and the compiled reference TUs (oracle/_ref/libbtref.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libbtref.so")
vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(vp)


def ensure_built():
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.startswith("oracle_") and f.endswith(".cpp")]
    stale = (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so")])


def ascii_kmers(kmers):
    """list[str] | (n,k) S1 array -> contiguous uint8 array of n*k chars"""
    if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint8:
        return np.ascontiguousarray(kmers)
    return np.frombuffer("".join(kmers).encode(), dtype=np.uint8).copy()


class Oracle:
    def __init__(self, path):
        self.l = L = C.CDLL(path)
        L.orc_ntp64.restype = C.c_uint64
        L.orc_ntp64.argtypes = [C.c_char_p, C.c_uint]
        L.orc_ntp64_seed.restype = C.c_uint64
        L.orc_ntp64_seed.argtypes = [C.c_char_p, C.c_uint, C.c_uint]
        L.orc_ntp64_batch.argtypes = [vp, C.c_uint64, C.c_uint, C.c_int, C.c_uint, vp]
        L.orc_pack_batch.argtypes = [vp, C.c_uint64, C.c_uint, vp]
        L.orc_unpack_batch.argtypes = [vp, C.c_uint64, C.c_uint, vp]
        L.orc_bloom_sizing.argtypes = [C.c_uint64, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint)]
        L.orc_bloom_new.restype = vp
        L.orc_bloom_new.argtypes = [C.c_uint64, C.c_float, C.c_uint, C.c_int]
        L.orc_bloom_load.restype = vp
        L.orc_bloom_load.argtypes = [C.c_char_p, C.c_uint]
        L.orc_bloom_save.argtypes = [vp, C.c_char_p]
        L.orc_bloom_free.argtypes = [vp]
        L.orc_bloom_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.orc_bloom_insert.argtypes = [vp, vp, C.c_uint64]
        L.orc_bloom_contains.argtypes = [vp, vp, C.c_uint64, vp]
        L.orc_bloom_bits.argtypes = [vp, C.c_uint, vp]
        L.orc_bloom_route.restype = C.c_uint
        L.orc_bloom_route.argtypes = [vp, C.c_char_p]
        L.orc_kmers_from_sequence.argtypes = [vp, C.c_uint64, C.c_uint, vp, vp]
        L.orc_table_new.restype = vp
        L.orc_table_new.argtypes = [C.c_uint, C.c_uint]
        L.orc_table_free.argtypes = [vp]
        L.orc_table_size.restype = C.c_uint64
        L.orc_table_size.argtypes = [vp]
        L.orc_table_insert.argtypes = [vp, vp, C.c_uint64, C.c_int]
        L.orc_table_count_intercluster.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_uint, C.c_uint]
        L.orc_table_classify.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]
        L.orc_table_count_parameter_kmers.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_uint, C.c_float]
        L.orc_table_export.restype = C.c_uint64
        L.orc_table_export.argtypes = [vp, vp, vp, vp]
        L.orc_graphs_new.restype = vp
        L.orc_graphs_new.argtypes = [C.c_uint, C.c_uint32] + [vp] * 15
        L.orc_graphs_free.argtypes = [vp]
        L.orc_graphs_set_edges.argtypes = [vp, vp, vp]
        L.orc_find_new.restype = vp
        L.orc_find_new.argtypes = [vp]
        L.orc_find_free.argtypes = [vp]
        L.orc_find_sample_paths.argtypes = [vp, vp, vp, vp, C.c_uint32]
        L.orc_find_sizes.argtypes = [vp, vp]
        L.orc_find_fetch.argtypes = [vp, vp]
        L.orc_paths_count_kmers.restype = C.c_uint64
        L.orc_paths_count_kmers.argtypes = [vp, vp]
        L.orc_paths_classify.argtypes = [vp, vp, vp, vp, vp]
        L.orc_paths_count_multigroup.restype = C.c_uint64
        L.orc_paths_count_multigroup.argtypes = [vp, vp, vp, vp]
        L.orc_paths_candidates.restype = vp
        L.orc_paths_candidates.argtypes = [vp, vp, vp]
        L.orc_paths_candidates_fetch.argtypes = [vp] * 21
        L.orc_table_kmer_stats.restype = None
        L.orc_table_kmer_stats.argtypes = [vp, vp, vp, vp]
        L.orc_kmc_write.argtypes = [C.c_char_p, vp, vp, C.c_uint64, C.c_uint, C.c_uint, C.c_uint]
        L.orc_kmc_open.restype = vp
        L.orc_kmc_open.argtypes = [C.c_char_p]
        L.orc_kmc_free.argtypes = [vp]
        L.orc_kmc_info.argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint64)]
        L.orc_kmc_lut.argtypes = [vp, vp]
        L.orc_kmc_payload_size.restype = C.c_uint64
        L.orc_kmc_payload_size.argtypes = [vp]
        L.orc_kmc_payload.argtypes = [vp, vp]
        L.orc_kmc_list.argtypes = [vp, vp, vp]
        L.orc_parse_sample_kmers.restype = C.c_uint64
        L.orc_parse_sample_kmers.argtypes = [vp, vp, vp, C.c_uint, C.c_uint64, C.c_uint64]
        L.orc_match_only.restype = C.c_uint64
        L.orc_match_only.argtypes = [vp, vp, C.c_uint64, C.c_uint64]

    # ---- hashing / packing ----
    def ntp64(self, kmers, k, seed=None):
        a = ascii_kmers(kmers)
        n = len(a) // k
        out = np.empty(n, dtype=np.uint64)
        self.l.orc_ntp64_batch(_ptr(a), n, k, 0 if seed is None else 1, 0 if seed is None else seed, _ptr(out))
        return out

    def pack(self, kmers, k):
        a = ascii_kmers(kmers)
        n = len(a) // k
        out = np.zeros((n, 2), dtype=np.uint64)
        self.l.orc_pack_batch(_ptr(a), n, k, _ptr(out))
        return out

    def unpack(self, packed, k):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        out = np.empty(len(packed) * k, dtype=np.uint8)
        self.l.orc_unpack_batch(_ptr(packed), len(packed), k, _ptr(out))
        return out

    def bloom_sizing(self, n, fpr):
        b, h = C.c_uint64(), C.c_uint()
        self.l.orc_bloom_sizing(n, fpr, C.byref(b), C.byref(h))
        return b.value, h.value

    def kmers_from_sequence(self, seq_bytes, k):
        a = np.frombuffer(seq_bytes, dtype=np.uint8).copy()
        kmers = np.zeros((len(a), 2), dtype=np.uint64)
        valid = np.zeros(len(a), dtype=np.uint8)
        self.l.orc_kmers_from_sequence(_ptr(a), len(a), k, _ptr(kmers), _ptr(valid))
        return kmers, valid

    # ---- KMC ----
    def kmc2_write(self, prefix, kmers_ascii, counts, k, p, counter_size=1, nbins=4):
        """KMC2 ("0x200") layout: signature bins with their own prefix tables"""
        a = ascii_kmers(kmers_ascii)
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        self.l.orc_kmc2_write.argtypes = [C.c_char_p, vp, vp, C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
        rc = self.l.orc_kmc2_write(prefix.encode(), _ptr(a), _ptr(c), len(c), k, p, counter_size, nbins)
        assert rc == 0, f"orc_kmc2_write failed ({rc})"

    def kmc_write(self, prefix, kmers_ascii, counts, k, p, counter_size=1):
        a = ascii_kmers(kmers_ascii)
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        rc = self.l.orc_kmc_write(prefix.encode(), _ptr(a), _ptr(c), len(c), k, p, counter_size)
        assert rc == 0, rc


class OrcBloom:
    def __init__(self, orc, num_kmers=None, fpr=None, k=55, threaded=False, handle=None):
        self.o, self.k = orc, k
        self.h = handle if handle is not None else orc.l.orc_bloom_new(num_kmers, fpr, k, int(threaded))

    @classmethod
    def load(cls, orc, prefix, k):
        h = orc.l.orc_bloom_load(prefix.encode(), k)
        assert h, "orc_bloom_load failed"
        return cls(orc, k=k, handle=h)

    def save(self, prefix):
        assert self.o.l.orc_bloom_save(self.h, prefix.encode()) == 0

    def info(self):
        nk, nb, nh, ns = C.c_uint64(), C.c_uint64(), C.c_uint(), C.c_uint()
        self.o.l.orc_bloom_info(self.h, C.byref(nk), C.byref(nb), C.byref(nh), C.byref(ns))
        return {"num_kmers": nk.value, "num_bits": nb.value, "num_hashes": nh.value, "num_sub": ns.value}

    def insert(self, kmers_ascii):
        a = ascii_kmers(kmers_ascii)
        self.o.l.orc_bloom_insert(self.h, _ptr(a), len(a) // self.k)

    def contains(self, kmers_ascii):
        a = ascii_kmers(kmers_ascii)
        n = len(a) // self.k
        out = np.zeros(n, dtype=np.uint8)
        self.o.l.orc_bloom_contains(self.h, _ptr(a), n, _ptr(out))
        return out

    def bits(self, sub=0):
        out = np.zeros((self.info()["num_bits"] + 7) // 8, dtype=np.uint8)
        self.o.l.orc_bloom_bits(self.h, sub, _ptr(out))
        return out

    def route(self, kmer):
        return self.o.l.orc_bloom_route(self.h, kmer.encode() if isinstance(kmer, str) else bytes(kmer))

    def close(self):
        if self.h:
            self.o.l.orc_bloom_free(self.h)
            self.h = None


class OrcTable:
    def __init__(self, orc, num_samples, k):
        self.o, self.k, self.S = orc, k, num_samples
        self.h = orc.l.orc_table_new(num_samples, k)

    def insert(self, kmers_ascii, mark_parameter=False):
        a = ascii_kmers(kmers_ascii)
        self.o.l.orc_table_insert(self.h, _ptr(a), len(a) // self.k, int(mark_parameter))

    def count_intercluster(self, bloom, seq_bytes, is_decoy, fp, mp):
        a = np.frombuffer(seq_bytes, dtype=np.uint8).copy()
        self.o.l.orc_table_count_intercluster(self.h, bloom.h, _ptr(a), len(a), int(is_decoy), fp, mp)

    def count_parameter_kmers(self, bloom, seq_bytes, is_decoy, seed, fraction):
        a = np.frombuffer(seq_bytes, dtype=np.uint8).copy()
        self.o.l.orc_table_count_parameter_kmers(self.h, bloom.h, _ptr(a), len(a), int(is_decoy), int(seed), float(fraction))

    def classify(self, mg_bloom, kmers_ascii, mult):
        a = ascii_kmers(kmers_ascii)
        m = np.ascontiguousarray(mult, dtype=np.uint8)
        out = np.zeros(len(m), dtype=np.uint8)
        self.o.l.orc_table_classify(self.h, mg_bloom.h, _ptr(a), _ptr(m), len(m), _ptr(out))
        return out

    def parse_sample_kmers(self, bloom, kmc, sample_idx, first=0, n=None):
        n = kmc.total - first if n is None else n
        return self.o.l.orc_parse_sample_kmers(self.h, bloom.h, kmc.h, sample_idx, first, n)

    def kmer_stats(self, gender):
        """calculateKmerStats -> (class_counts[7], stats[S,256,4] = count, fraction, mean, M2 of the running Welford update)"""
        g = np.ascontiguousarray(gender, dtype=np.uint8)
        cls = np.zeros(7, np.uint64)
        st = np.zeros((self.S, 256, 4), np.float64)
        self.o.l.orc_table_kmer_stats(self.h, _ptr(g), _ptr(cls), _ptr(st))
        return cls, st

    def export(self):
        """-> (packed kmers (n,2) u64, counts (n,S), meta (n,4)) sorted by ASCII k-mer"""
        n = self.o.l.orc_table_size(self.h)
        km = np.zeros(max(n, 1) * self.k, dtype=np.uint8)
        counts = np.zeros((max(n, 1), self.S), dtype=np.uint8)
        meta = np.zeros((max(n, 1), 4), dtype=np.uint8)
        n2 = self.o.l.orc_table_export(self.h, _ptr(km), _ptr(counts), _ptr(meta))
        assert n2 == n
        return self.o.pack(km[: n * self.k], self.k), counts[:n], meta[:n]

    def close(self):
        if self.h:
            self.o.l.orc_table_free(self.h)
            self.h = None


CAND_FIELDS = [("kmer_off", np.uint32), ("hap_kmer_mult", np.uint8), ("kmer_key", np.uint64), ("kmer_has_counts", np.uint8), ("kmer_counts", np.uint8),
               ("kmer_ic_mult", np.uint8), ("kv_off", np.uint32), ("kv_var", np.uint16), ("kv_bits", np.uint32), ("unique_off", np.uint32),
               ("unique_idx", np.uint32), ("multi_off", np.uint32), ("multi_idx", np.uint32), ("hap_allele", np.uint16), ("hapnest_off", np.uint32),
               ("hapnest_idx", np.uint32), ("nestdep_off", np.uint32), ("nestdep_cluster", np.uint32), ("nestdep_var_off", np.uint32), ("nestdep_var", np.uint16)]


def candidates_arrays(sizes, C_, S):
    """host arrays of bt_paths_candidates_out for the sizes (rows, mult_bytes, nnz, kv_words, num_unique, num_multi, hap_allele,
    num_haplotypes, hapnest, nestdep, nestdep_var)"""
    rows, mult, nnz, kvw, nu, nm, ha, nh, hn, nd, ndv = [int(x) for x in sizes]
    n = {"kmer_off": C_ + 1, "hap_kmer_mult": mult, "kmer_key": rows * 2, "kmer_has_counts": rows, "kmer_counts": rows * S, "kmer_ic_mult": rows * 2,
         "kv_off": rows + 1, "kv_var": nnz, "kv_bits": kvw, "unique_off": C_ + 1, "unique_idx": nu, "multi_off": C_ + 1, "multi_idx": nm,
         "hap_allele": ha, "hapnest_off": nh + 1, "hapnest_idx": hn, "nestdep_off": C_ + 1, "nestdep_cluster": nd, "nestdep_var_off": nd + 1,
         "nestdep_var": ndv}
    return {name: np.zeros(max(n[name], 1), dt) for name, dt in CAND_FIELDS}, n


class OrcGraphs:
    """flattened variant-cluster graphs (bayestyper_amd.synth_graphs.flatten) for the path-enumeration oracle"""

    def __init__(self, orc, flat, k):
        from bayestyper_amd import synth_graphs

        self.o, self.f, self.k = orc, flat, k
        names = [n for n in synth_graphs.FIELDS if n not in ("in_off", "in_src")]
        self.keep = [np.ascontiguousarray(flat[n]) if np.asarray(flat[n]).size else np.zeros(1, np.asarray(flat[n]).dtype) for n in names]
        self.h = orc.l.orc_graphs_new(k, flat["num_clusters"], *[_ptr(a) for a in self.keep])
        self.edges = [np.ascontiguousarray(flat["in_off"], np.uint32), np.ascontiguousarray(np.concatenate([flat["in_src"], [0]]), np.uint32)]
        orc.l.orc_graphs_set_edges(self.h, _ptr(self.edges[0]), _ptr(self.edges[1]))
        self.find = None

    def find_sample_paths(self, bloom, seeds, max_haps):
        """one sample's findSamplePaths + addPathIndices for every cluster; returns the accumulated best paths (list of (P, |V|) arrays)"""
        if self.find is None:
            self.find = self.o.l.orc_find_new(self.h)
        sd = np.ascontiguousarray(seeds, np.uint32)
        self.o.l.orc_find_sample_paths(self.find, self.h, bloom.h, _ptr(sd), max_haps)
        n = np.zeros(self.f["num_clusters"], np.uint32)
        self.o.l.orc_find_sizes(self.find, _ptr(n))
        nv = (self.f["vertex_off"][1:] - self.f["vertex_off"][:-1]).astype(np.int64)
        out = np.zeros(max(int((n.astype(np.int64) * nv).sum()), 1), np.uint8)
        self.o.l.orc_find_fetch(self.find, _ptr(out))
        res, at = [], 0
        for c in range(self.f["num_clusters"]):
            res.append(out[at:at + int(n[c]) * int(nv[c])].reshape(int(n[c]), int(nv[c])).copy())
            at += int(n[c]) * int(nv[c])
        return res

    def count_kmers(self, bloom=None):
        return self.o.l.orc_paths_count_kmers(self.h, bloom.h if bloom is not None else None)

    def count_multigroup(self, cluster_group, bloom, table):
        cg = np.ascontiguousarray(cluster_group, np.uint32)
        return self.o.l.orc_paths_count_multigroup(self.h, _ptr(cg), bloom.h, table.h)

    def classify(self, table, mg_bloom):
        n = np.zeros(self.f["num_clusters"], np.uint32)
        ex = np.zeros(self.f["num_clusters"], np.uint8)
        self.o.l.orc_paths_classify(self.h, table.h, mg_bloom.h, _ptr(n), _ptr(ex))
        return n, ex

    def candidates(self, table):
        sizes = np.zeros(11, np.uint64)
        h = self.o.l.orc_paths_candidates(self.h, table.h, _ptr(sizes))
        arrs, n = candidates_arrays(sizes, self.f["num_clusters"], table.S)
        self.o.l.orc_paths_candidates_fetch(h, *[_ptr(arrs[name]) for name, _ in CAND_FIELDS])
        return {name: arrs[name][: n[name]] for name, _ in CAND_FIELDS}

    def close(self):
        if self.find:
            self.o.l.orc_find_free(self.find)
            self.find = None
        if self.h:
            self.o.l.orc_graphs_free(self.h)
            self.h = None


class OrcKmc:
    def __init__(self, orc, prefix):
        self.o = orc
        self.h = orc.l.orc_kmc_open(prefix.encode())
        assert self.h, f"cannot open KMC db {prefix}"
        k, p, cs, tot = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint64()
        orc.l.orc_kmc_info(self.h, C.byref(k), C.byref(p), C.byref(cs), C.byref(tot))
        self.k, self.p, self.counter_size, self.total = k.value, p.value, cs.value, tot.value
        self.rec_size = (self.k - self.p) // 4 + self.counter_size

    def lut(self):
        self.o.l.orc_kmc_lut_entries.restype = C.c_uint64
        self.o.l.orc_kmc_lut_entries.argtypes = [vp]
        out = np.zeros(self.o.l.orc_kmc_lut_entries(self.h), dtype=np.uint64)   # 4^p + 1 (KMC1) or bins * 4^p + 1 (KMC2)
        self.o.l.orc_kmc_lut(self.h, _ptr(out))
        return out

    def payload(self):
        n = self.o.l.orc_kmc_payload_size(self.h)
        out = np.zeros(n, dtype=np.uint8)
        self.o.l.orc_kmc_payload(self.h, _ptr(out))
        return out

    def list(self):
        km = np.zeros(self.total * self.k, dtype=np.uint8)
        counts = np.zeros(self.total, dtype=np.uint32)
        self.o.l.orc_kmc_list(self.h, _ptr(km), _ptr(counts))
        return km, counts

    def close(self):
        if self.h:
            self.o.l.orc_kmc_free(self.h)
            self.h = None


_ORACLE = None


def load_oracle():
    global _ORACLE
    if _ORACLE is None:
        ensure_built()
        _ORACLE = Oracle(ORACLE_SO)
    return _ORACLE


class Ref:
    """ctypes view of oracle/_ref/libbtref.so (the reference's own code, k fixed at compile time)."""

    def __init__(self, path):
        self.l = L = C.CDLL(path)
        L.ref_kmer_size.restype = C.c_uint
        self.k = L.ref_kmer_size()
        L.ref_ntp64.restype = C.c_uint64
        L.ref_ntp64.argtypes = [C.c_char_p]
        L.ref_ntp64_seed.restype = C.c_uint64
        L.ref_ntp64_seed.argtypes = [C.c_char_p, C.c_uint]
        L.ref_bloom_sizing.argtypes = [C.c_uint64, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint)]
        for n in ("ref_kmerbloom_new", "ref_tbloom_new"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [C.c_uint64, C.c_float]
        L.ref_kmerbloom_load.restype = vp
        L.ref_kmerbloom_load.argtypes = [C.c_char_p]
        L.ref_kmerbloom_free.argtypes = [vp]
        L.ref_tbloom_free.argtypes = [vp]
        L.ref_kmerbloom_save.argtypes = [vp, C.c_char_p]
        for n in ("ref_kmerbloom_add", "ref_tbloom_add"):
            getattr(L, n).argtypes = [vp, vp, C.c_uint64]
        for n in ("ref_kmerbloom_lookup", "ref_tbloom_lookup", "ref_kmerbloom_lookup_packed"):
            getattr(L, n).argtypes = [vp, vp, C.c_uint64, vp]
        L.ref_kmers_from_sequence.argtypes = [vp, C.c_uint64, vp, vp]
        L.ref_kmc_total.restype = C.c_int64
        L.ref_kmc_total.argtypes = [C.c_char_p] + [C.POINTER(C.c_uint)] * 4
        L.ref_kmc_list.restype = C.c_int64
        L.ref_kmc_list.argtypes = [C.c_char_p, vp, vp, C.c_uint64]
        L.ref_kc_new.restype = vp
        L.ref_kc_free.argtypes = [vp]
        L.ref_kc_add_intercluster.argtypes = [vp, C.c_int, C.c_uint, C.c_uint]
        L.ref_kc_add_cluster.argtypes = [vp, C.c_uint, C.c_int]
        L.ref_kc_add_sample_count.argtypes = [vp, C.c_uint, C.c_uint]
        L.ref_kc_get.argtypes = [vp, vp, vp, vp]
        L.ref_nb_moments.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_nb_logpmf.restype = C.c_double
        L.ref_nb_logpmf.argtypes = [C.c_double, C.c_double, C.c_uint, C.c_uint]
        L.ref_log_addition.restype = C.c_double
        L.ref_log_addition.argtypes = [C.c_double, C.c_double]
        L.ref_double_compare.argtypes = [C.c_double, C.c_double]
        L.ref_logdiscrete_draws.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, vp]
        L.ref_discrete_draws.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, vp]
        L.ref_kmerstats.argtypes = [vp, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ref_sparsity_cover.restype = C.c_uint
        L.ref_sparsity_cover.argtypes = [vp, C.c_uint, C.c_uint, vp, C.c_uint, vp]


_REF = False


def load_ref():
    global _REF
    if _REF is False:
        _REF = Ref(REF_SO) if os.path.exists(REF_SO) else None
    return _REF


def random_kmers(rng, n, k):
    """n random ASCII k-mers as a contiguous uint8 array (n*k)"""
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * k)].copy()


def canonical_ascii(orc, kmers_u8, k):
    """canonical form of each ASCII k-mer (via the oracle's sliding window on each k-mer)"""
    n = len(kmers_u8) // k
    out = np.empty_like(kmers_u8)
    for i in range(n):
        km, valid = orc.kmers_from_sequence(kmers_u8[i * k:(i + 1) * k].tobytes(), k)
        out[i * k:(i + 1) * k] = orc.unpack(km[k - 1:k], k)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Gibbs oracle (oracle/oracle_gibbs.cpp)
# ---------------------------------------------------------------------------------------------------------------
def _gibbs_sigs(L):
    if getattr(L, "_gibbs_sigs_done", False):
        return
    d, u, u64 = C.c_double, C.c_uint, C.c_uint64
    L.orc_build_luts.argtypes = [u, vp, vp, vp, vp, vp]
    L.orc_build_noise_lut.argtypes = [u, vp, vp]
    L.orc_nb_moments.argtypes = [d, d, C.POINTER(d), C.POINTER(d)]
    L.orc_nb_logpmf.restype = d
    L.orc_nb_logpmf.argtypes = [d, d, u, u]
    L.orc_log_addition.restype = d
    L.orc_log_addition.argtypes = [d, d]
    L.orc_double_compare.argtypes = [d, d]
    L.orc_logdiscrete_draws.argtypes = [vp, u, u, u, vp]
    L.orc_discrete_draws.argtypes = [vp, u, u, u, vp]
    L.orc_kmerstats.argtypes = [vp, u, C.POINTER(u), C.POINTER(d), C.POINTER(d), C.POINTER(d)]
    L.orc_sparsity_cover.restype = u
    L.orc_sparsity_cover.argtypes = [vp, u, u, vp, u, vp]
    L.orc_rng.argtypes = [u, C.c_int, vp, vp, u64, vp]
    L.orc_uset_replay.argtypes = [u, vp, vp, u64, vp, C.POINTER(C.c_uint32)]
    L.orc_gibbs_create.restype = vp
    L.orc_gibbs_create.argtypes = [vp, vp, vp, vp]
    L.orc_gibbs_free.argtypes = [vp]
    L.orc_gibbs_set_noise_lut.argtypes = [vp, vp]
    L.orc_gibbs_trace_enable.argtypes = [vp, C.c_uint32]
    L.orc_gibbs_run.argtypes = [vp, u]
    L.orc_gibbs_init_chain.argtypes = [vp, C.c_uint32]
    L.orc_gibbs_sweep.argtypes = [vp, C.c_uint32, C.c_int]
    L.orc_gibbs_noise_counts.argtypes = [vp, vp, C.c_int]
    L.orc_gibbs_reset_groups.argtypes = [vp]
    L.orc_estimate_noise.restype = u64
    L.orc_estimate_noise.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, vp, u64, vp, u64, vp]
    L.orc_estimate_noise_and_genotypes.restype = u64
    L.orc_estimate_noise_and_genotypes.argtypes = [vp, C.c_float, C.c_float, vp, u64]
    L.orc_estimate_noise_and_genotypes_mt.restype = u64
    L.orc_estimate_noise_and_genotypes_mt.argtypes = [vp, C.c_float, C.c_float, vp, u64, C.c_uint]
    L.orc_gibbs_result_sizes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.orc_gibbs_result_fetch.argtypes = [vp] * 7
    L.orc_gibbs_trace_fetch.restype = u64
    L.orc_gibbs_trace_fetch.argtypes = [vp, C.c_uint32, vp, u64]
    L._gibbs_sigs_done = True


def build_luts(orc, S, mean=15.0, var=30.0, noise_rate=0.05):
    """CountDistribution LUTs for S samples with identical NB(mean,var) per copy and the given noise rate(s)"""
    _gibbs_sigs(orc.l)
    p, size = C.c_double(), C.c_double()
    orc.l.orc_nb_moments(mean, var, C.byref(p), C.byref(size))
    ps = np.full(S, p.value)
    sz = np.full(S, size.value)
    nr = np.full(S, noise_rate, dtype=np.float64) if np.isscalar(noise_rate) else np.asarray(noise_rate, np.float64)
    g = np.zeros(S * 65536, np.float64)
    n = np.zeros(S * 256, np.float64)
    orc.l.orc_build_luts(S, _ptr(ps), _ptr(sz), _ptr(nr), _ptr(g), _ptr(n))
    return g, n


class OrcGibbs:
    def __init__(self, orc, flat, lut_g, lut_n, **kw):
        from bayestyper_amd import synth

        _gibbs_sigs(orc.l)
        self.o, self.flat = orc, flat
        self.S, self.C = flat["S"], flat["num_clusters"]
        self.params, self.batch, self._keep = synth.to_ctypes(flat, **kw)
        self._keep += [lut_g, lut_n]
        self.h = orc.l.orc_gibbs_create(C.addressof(self.params), C.addressof(self.batch), _ptr(lut_g), _ptr(lut_n))

    def run(self, threads=1):
        self.o.l.orc_gibbs_run(self.h, threads)

    def trace_enable(self, n):
        self.o.l.orc_gibbs_trace_enable(self.h, n)

    def trace(self, group, n_vertices, max_sweeps):
        buf = np.zeros(max_sweeps * n_vertices * self.S, np.uint32)
        n = self.o.l.orc_gibbs_trace_fetch(self.h, group, _ptr(buf), len(buf))
        return buf[:n].reshape(-1, n_vertices, self.S)

    def init_chain(self, c):
        self.o.l.orc_gibbs_init_chain(self.h, c)

    def sweep(self, n, collect):
        self.o.l.orc_gibbs_sweep(self.h, n, int(collect))

    def noise_counts(self, zero_first=True):
        hist = np.zeros(self.S * 256, np.uint64)
        self.o.l.orc_gibbs_noise_counts(self.h, _ptr(hist), int(zero_first))
        return hist

    def set_noise_lut(self, lut_n):
        self._keep.append(lut_n)
        self.o.l.orc_gibbs_set_noise_lut(self.h, _ptr(lut_n))

    def reset_groups(self):
        self.o.l.orc_gibbs_reset_groups(self.h)

    def results(self):
        nd, nc = C.c_uint64(), C.c_uint64()
        self.o.l.orc_gibbs_result_sizes(self.h, C.byref(nd), C.byref(nc))
        nd, nc = nd.value, nc.value
        dip_off = np.zeros(self.C + 1, np.uint64)
        cell_off = np.zeros(self.C + 1, np.uint64)
        h1, h2 = np.zeros(max(nd, 1), np.uint16), np.zeros(max(nd, 1), np.uint16)
        freq = np.zeros(max(nd, 1) * self.S, np.uint32)
        stats = np.zeros(max(nc, 1) * 12, np.float64)
        self.o.l.orc_gibbs_result_fetch(self.h, _ptr(dip_off), _ptr(h1), _ptr(h2), _ptr(freq), _ptr(cell_off), _ptr(stats))
        return {"dip_off": dip_off, "h1": h1[:nd], "h2": h2[:nd], "freq": freq[: nd * self.S].reshape(nd, self.S), "cell_off": cell_off,
                "stats": stats[: nc * 12].reshape(nc, 3, 4)}

    # the noise drivers, restated whole in the oracle (InferenceEngine.cpp:135-276, 384-472); the object must have been
    # created with noise_seeding=1 over the WHOLE unit
    def estimate_noise(self, prior=(1.0, 0.01), variants_batch_size=100000):
        """-> (trace rows [(chain, iteration, rates...)], per-chain selected group indices, final mean rates)"""
        rows = self.params.num_chains * (self.params.burn_in + self.params.num_iterations + 1) + 1
        trace = np.zeros(rows * (2 + self.S))
        sel = np.zeros(self.params.num_chains * (1 + self.batch.num_groups) + 1, np.uint32)
        final = np.zeros(self.S)
        n = self.o.l.orc_estimate_noise(self.h, prior[0], prior[1], variants_batch_size, _ptr(trace), len(trace), _ptr(sel), len(sel), _ptr(final))
        assert n == len(trace)
        chains, p = [], 0
        for _ in range(self.params.num_chains):
            k = int(sel[p])
            chains.append(sel[p + 1:p + 1 + k].copy())
            p += 1 + k
        return trace.reshape(rows, 2 + self.S), chains, final

    def estimate_noise_and_genotypes(self, prior=(1.0, 0.01), threads=1):
        rows = self.params.num_chains * (self.params.burn_in + self.params.num_iterations + 1)
        trace = np.zeros(rows * (2 + self.S))
        n = self.o.l.orc_estimate_noise_and_genotypes_mt(self.h, prior[0], prior[1], _ptr(trace), len(trace), int(threads))
        assert n == len(trace)
        return trace.reshape(rows, 2 + self.S)

    def close(self):
        if self.h:
            self.o.l.orc_gibbs_free(self.h)
            self.h = None
