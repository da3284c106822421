"""The cluster stage's front end (bayestyper_amd/host/VariantFileParser.{hpp,cpp}): candidate VCF + genome -> units of
variant-cluster groups and the intercluster regions — VariantFileParser / VariantClusterGroup of the reference
(src/bayesTyper/VariantFileParser.cpp:185-1235, VariantClusterGroup.cpp:47-107, main.cpp:214-247)."""
import ctypes as C

import numpy as np

from . import dll

vp = C.c_void_p
dll.bth_cluster_stage_new.restype = vp
dll.bth_cluster_stage_new.argtypes = [C.c_uint, C.c_uint32, C.c_float]
dll.bth_cluster_stage_free.argtypes = [vp]
dll.bth_cluster_stage_add_sequence.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_ulonglong, C.c_int, C.c_char_p, C.c_uint]
dll.bth_cluster_stage_set_variants.argtypes = [vp, C.c_char_p, C.c_ulonglong, C.c_char_p, C.c_uint]
dll.bth_cluster_stage_next_unit.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_uint]
dll.bth_cluster_stage_dump.restype = C.c_ulonglong
dll.bth_cluster_stage_dump.argtypes = [vp, C.c_int, C.c_char_p, C.c_ulonglong]
dll.bth_cluster_stage_sort_regions.argtypes = [vp]
dll.bth_cluster_stage_unit_sizes.argtypes = [vp, vp, C.c_uint]
dll.bth_cluster_stage_graph.restype = vp
dll.bth_cluster_stage_graph.argtypes = [vp, C.c_uint32, C.c_uint32]


class ClusterStage:
    def __init__(self, k=55, max_allele_length=500000, copy_number_variant_threshold=0.5):
        self.k = k
        self.h = dll.bth_cluster_stage_new(k, max_allele_length, copy_number_variant_threshold)
        self._err = C.create_string_buffer(1024)

    def _check(self, rc):
        if rc < 0:
            raise ValueError(self._err.value.decode())
        return rc

    def add_sequence(self, name, sequence, is_decoy=False):
        s = sequence.encode() if isinstance(sequence, str) else bytes(sequence)
        self._check(dll.bth_cluster_stage_add_sequence(self.h, name.encode(), s, len(s), int(is_decoy), self._err, len(self._err)))

    def set_variants(self, vcf_text=None, path=None):
        if path is not None:
            self._check(dll.bth_cluster_stage_set_variants(self.h, path.encode(), 0, self._err, len(self._err)))
        else:
            t = vcf_text.encode() if isinstance(vcf_text, str) else bytes(vcf_text)
            self._check(dll.bth_cluster_stage_set_variants(self.h, t, len(t), self._err, len(self._err)))

    def next_unit(self, min_unit_variants):
        """parses the next unit; True when the variant file is exhausted"""
        return self._check(dll.bth_cluster_stage_next_unit(self.h, min_unit_variants, self._err, len(self._err))) == 1

    def _dump(self, what):
        n = dll.bth_cluster_stage_dump(self.h, what, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        dll.bth_cluster_stage_dump(self.h, what, buf, n)
        return buf.raw[:n].decode()

    def unit_text(self):
        return self._dump(0)

    def sort_regions(self):
        """sortInterclusterRegions (once, after the last unit): longest first, the order countInterclusterKmers walks them in"""
        dll.bth_cluster_stage_sort_regions(self.h)

    def regions_text(self):
        return self._dump(1)

    def counters_text(self):
        return self._dump(3)

    def unit_sizes(self):
        """clusters per group of the last unit"""
        n = np.zeros(1, np.uint32)
        dll.bth_cluster_stage_unit_sizes(self.h, n.ctypes.data, 1)
        out = np.zeros(1 + int(n[0]), np.uint32)
        dll.bth_cluster_stage_unit_sizes(self.h, out.ctypes.data, len(out))
        return out[1:]

    def graph(self, group, vertex):
        """handle of the VariantClusterGraph of one cluster of the last unit (read with the bth_graph_* accessors)"""
        g = dll.bth_cluster_stage_graph(self.h, group, vertex)
        if not g:
            raise ValueError("bth_cluster_stage_graph failed")
        return g

    def close(self):
        if self.h:
            dll.bth_cluster_stage_free(self.h)
            self.h = None


dll.bth_graph_free.argtypes = [vp]
dll.bth_graph_sizes.argtypes = [vp, vp]
dll.bth_graph_fetch.argtypes = [vp] * 12


def fetch_graph(h, num_variants, free=True):
    """flat arrays of a VariantClusterGraph handle: vertex sequences (2-bit codes), (variant, allele) per vertex, flags (1 disconnected,
    2 first nucleotides redundant), nested cluster index, reference_variant_indices, edges, alleles / dependency flag per variant"""
    p = lambda a: a.ctypes.data_as(vp)   # noqa: E731
    sizes = np.zeros(4, np.uint64)
    dll.bth_graph_sizes(h, p(sizes))
    nv, ne, nnt, nref = [int(x) for x in sizes]
    out = {"seq_off": np.zeros(nv + 1, np.uint64), "seq": np.zeros(max(nnt, 1), np.uint8), "var": np.zeros(nv, np.uint16), "allele": np.zeros(nv, np.uint16),
           "flags": np.zeros(nv, np.uint8), "nested": np.zeros(nv, np.uint32), "refvar_off": np.zeros(nv + 1, np.uint32), "refvar": np.zeros(max(nref, 1), np.uint16),
           "edges": np.zeros(max(2 * ne, 1), np.uint32), "num_alleles": np.zeros(num_variants, np.uint16), "dep": np.zeros(num_variants, np.uint8)}
    dll.bth_graph_fetch(h, *[p(out[k]) for k in ("seq_off", "seq", "var", "allele", "flags", "nested", "refvar_off", "refvar", "edges", "num_alleles", "dep")])
    if free:
        dll.bth_graph_free(h)
    out["seq"], out["refvar"], out["edges"] = out["seq"][:nnt], out["refvar"][:nref], out["edges"][: 2 * ne].reshape(-1, 2)
    return out


dll.bth_writer_new.restype = vp
dll.bth_writer_new.argtypes = [vp, C.c_char_p]
dll.bth_writer_free.argtypes = [vp]
dll.bth_writer_add_cluster.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint, C.c_uint, vp, C.c_ulonglong, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, C.c_char_p, C.c_uint]
dll.bth_writer_text.restype = C.c_ulonglong
dll.bth_writer_text.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ulonglong]
dll.bth_writer_finalise.restype = C.c_longlong
dll.bth_writer_finalise.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint]


class GenotypeWriter:
    """bthost::GenotypeWriter over the clusters of a ClusterStage's current unit (GenotypeWriter.cpp:57-556)"""

    def __init__(self, stage, sample_names):
        self.stage = stage
        self.h = dll.bth_writer_new(stage.h, "\t".join(sample_names).encode())
        self._err = C.create_string_buffer(1024)

    def add_cluster(self, group, vertex, S, H, hap_allele, h1, h2, freq, stats, ploidy, min_fraction, min_gpp=0.99, min_kmers=1.0):
        """the sampler's results of one cluster (arrays as returned by lib.Gibbs.results(), restricted to the cluster)"""
        arrs = [np.ascontiguousarray(hap_allele, np.uint16), np.ascontiguousarray(h1, np.uint16), np.ascontiguousarray(h2, np.uint16),
                np.ascontiguousarray(freq, np.uint32).reshape(-1), np.ascontiguousarray(stats, np.float64).reshape(-1), np.ascontiguousarray(ploidy, np.uint8),
                np.ascontiguousarray(min_fraction, np.float32)]
        n_dip = len(arrs[1])
        for a in arrs:
            if a.size == 0:
                a.resize(1, refcheck=False)
        p = lambda a: a.ctypes.data_as(vp)   # noqa: E731
        rc = dll.bth_writer_add_cluster(self.h, group, vertex, S, H, p(arrs[0]), n_dip, p(arrs[1]), p(arrs[2]), p(arrs[3]), p(arrs[4]), p(arrs[5]), min_gpp, min_kmers, p(arrs[6]),
                                        self._err, len(self._err))
        if rc != 0:
            raise ValueError(self._err.value.decode())

    def text(self, genome_filename="genome.fa", graph_options_header="", genotype_options_header=""):
        args = [genome_filename.encode(), graph_options_header.encode(), genotype_options_header.encode()]
        n = dll.bth_writer_text(self.h, *args, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        dll.bth_writer_text(self.h, *args, buf, n)
        return buf.raw[:n].decode()

    def finalise(self, output_prefix, gzip_output=False, genome_filename="genome.fa", graph_options_header="", genotype_options_header=""):
        n = dll.bth_writer_finalise(self.h, output_prefix.encode(), int(gzip_output), genome_filename.encode(), graph_options_header.encode(), genotype_options_header.encode(),
                                    self._err, len(self._err))
        if n < 0:
            raise ValueError(self._err.value.decode())
        return int(n)

    def close(self):
        if self.h:
            dll.bth_writer_free(self.h)
            self.h = None
