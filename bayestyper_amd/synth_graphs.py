"""Synthetic variant-cluster graphs for the path-enumeration stage (test / bench input tooling).

`build_graph` lays a cluster's variants out as the vertex list VariantClusterGraph's constructor produces
(src/bayesTyper/VariantClusterGraph.cpp:62-377: k-1 left flank, one vertex per alternative allele, reference segments split
at allele ends and at nested-cluster cut points, reference_variant_indices of the still-open variants) for clusters without N
runs; `random_paths` draws source-to-sink walks as best paths; `flatten` produces the arrays of bt_paths_batch
(include/btgpu.h).  Nothing here is used by the product path.
"""
import ctypes as C

import numpy as np

NONE16 = 0xFFFF
NONE32 = 0xFFFFFFFF


class Graph:
    def __init__(self):
        self.seq, self.var, self.allele, self.refvars, self.nested, self.disconnected, self.redundant = [], [], [], [], [], [], []
        self.out = []            # adjacency (vertex -> successors)
        self.edges = []          # (source, target) in insertion order (boost::add_edge order: in_edges() iterates in this order)
        self.num_alleles, self.has_dep = [], []

    def new_vertex(self):
        for lst, v in ((self.seq, np.zeros(0, np.uint8)), (self.var, NONE16), (self.allele, NONE16), (self.refvars, []), (self.nested, NONE32),
                       (self.disconnected, False), (self.redundant, False), (self.out, [])):
            lst.append(v)
        return len(self.seq) - 1

    def edge(self, a, b):
        self.out[a].append(b)
        self.edges.append((a, b))

    def init_vertex(self, v, seq, va, refvars, nested, redundant):
        self.seq[v], self.var[v], self.allele[v] = np.asarray(seq, np.uint8), va[0], va[1]
        self.refvars[v], self.nested[v], self.redundant[v] = list(refvars), nested, redundant
        self.disconnected[v] = nested != NONE32

    def add_vertices(self, cur, seqs, va, ref_set, nested_idx, redundant):   # VariantClusterGraph.cpp:290-316
        assert len(seqs) == len(nested_idx) + 1
        refvars = [r for r in sorted(ref_set) if r != va[0]]
        self.init_vertex(cur, seqs[0], va, refvars, NONE32, redundant)
        for i in range(1, len(seqs)):
            prev, cur = cur, self.new_vertex()
            self.edge(prev, cur)
            self.init_vertex(cur, seqs[i], va, refvars, nested_idx[i - 1], False)
        return cur


def build_graph(chrom, variants, k, contained=()):
    """chrom: uint8 2-bit codes; variants: sorted list of dict(pos, alts=[(ref_len, alt_codes)], has_dependency, num_redundant);
    contained: sorted list of (left_flank, right_flank, cluster_idx) nested clusters."""
    g = Graph()
    contained = list(contained)
    added = {}       # position -> ([vertices], [variant indices])   (std::map: iterate in key order)
    ref_set = set()
    cur = g.new_vertex()
    p0 = variants[0]["pos"]
    cur = g.add_vertices(cur, [chrom[p0 - (k - 1):p0]], (NONE16, NONE16), ref_set, [], False)
    prev = cur
    added[p0] = ([cur], [])
    for vi, var in enumerate(variants):
        g.num_alleles.append(len(var["alts"]) + 1 + (1 if var.get("has_dependency") else 0))
        g.has_dep.append(1 if var.get("has_dependency") else 0)
        redundant = var.get("num_redundant", 0) > 0
        max_ref = 0
        for ai, (ref_len, alt) in enumerate(var["alts"]):
            max_ref = max(max_ref, ref_len)
            nxt = g.new_vertex()
            g.edge(cur, nxt)
            nxt = g.add_vertices(nxt, [alt], (vi, ai + 1), ref_set, [], redundant)
            added.setdefault(var["pos"] + ref_len, ([], []))[0].append(nxt)
        added[var["pos"] + max_ref][1].append(vi)
        ref_set.add(vi)
        last_variant = vi + 1 == len(variants)
        next_position = None if last_variant else variants[vi + 1]["pos"]
        more = True
        while more:
            cur_position = min(added)
            next_vertices, var_list = added.pop(cur_position)
            for v in var_list:
                ref_set.discard(v)
            if not added:
                more = False
                cur_last = cur_position + k - 1 if last_variant else next_position
            else:
                cur_last = min(added)
                if not last_variant and cur_last > next_position:
                    more = False
                    cur_last = next_position
            seqs, nested_idx, prev_contained = [], [], None
            while contained and contained[0][0] < cur_last:
                lf, rf, cidx = contained.pop(0)
                assert cur_position <= lf and rf <= cur_last - k
                if prev_contained is not None:
                    nested_idx.append(prev_contained)
                seqs.append(chrom[cur_position:lf])
                prev_contained = cidx
                cur_position = rf + 1
            if prev_contained is not None:
                nested_idx.append(prev_contained)
            seqs.append(chrom[cur_position:cur_last])
            cur = g.new_vertex()
            is_ref = False
            for v in next_vertices:
                if v == prev:
                    is_ref = True
                    assert len(next_vertices) == 1
                g.edge(v, cur)
            if is_ref:
                cur = g.add_vertices(cur, seqs, (vi, 0), ref_set, nested_idx, redundant)
            else:
                cur = g.add_vertices(cur, seqs, (NONE16, NONE16), ref_set, nested_idx, False)
            added.setdefault(cur_last, ([], []))[0].append(cur)
        prev = cur
    g.sink = cur
    return g


def random_paths(g, rng, num_paths):
    """distinct source-to-sink walks as rows of a (P, |V|) 0/1 matrix (best_paths_indices)"""
    nv, seen, rows = len(g.seq), set(), []
    for _ in range(num_paths * 20):
        v, row = 0, np.zeros(nv, np.uint8)
        row[0] = 1
        while g.out[v]:
            v = g.out[v][int(rng.integers(len(g.out[v])))]
            row[v] = 1
        key = row.tobytes()
        if key not in seen:
            seen.add(key)
            rows.append(row)
        if len(rows) == num_paths:
            break
    return np.stack(rows)


def random_cluster(rng, k, num_variants, max_paths, chrom_len=None, nested_cluster=None, kinds=("snv", "ins", "del", "multi")):
    """one cluster on its own random chromosome: variants 0..120 nt apart (close enough for k-mers to span several)"""
    chrom_len = chrom_len or (2 * k + 200 * num_variants + 400 + 6 * k)
    chrom = rng.integers(0, 4, chrom_len).astype(np.uint8)
    variants, pos = [], k + 10
    contained = []
    for vi in range(num_variants):
        kind = kinds[int(rng.integers(len(kinds)))]
        if kind == "snv":
            alts, red = [(1, np.array([(chrom[pos] + 1 + rng.integers(3)) % 4], np.uint8))], 0
        elif kind == "ins":
            n = int(rng.integers(1, 30))
            alts, red = [(1, np.concatenate([[chrom[pos]], rng.integers(0, 4, n)]).astype(np.uint8))], 1
        elif kind == "del":
            n = int(rng.integers(2, 40))
            alts, red = [(n, np.array([chrom[pos]], np.uint8))], 1
        else:
            n = int(rng.integers(2, 12))
            alts = [(1, np.array([(chrom[pos] + 1) % 4], np.uint8)), (n, np.array([chrom[pos]], np.uint8)),
                    (1, np.concatenate([[chrom[pos]], rng.integers(0, 4, 5)]).astype(np.uint8))]
            red = 0   # mixed allele types: no common redundant prefix
        nest_here = nested_cluster is not None and vi == (num_variants - 1) // 2 and vi + 1 < num_variants
        if nest_here and nested_cluster % 2 == 0:
            # a long deletion whose REFERENCE allele contains the nested cluster: the (variant, 0) vertex is split at the cut and the
            # variant enters nested_variant_cluster_dependency
            n = 3 * k + 60
            alts, red = [(n, np.array([chrom[pos]], np.uint8))], 1
            contained.append((pos + 5, pos + 25, nested_cluster))
            nest_here = False
        variants.append({"pos": pos, "alts": alts, "has_dependency": False, "num_redundant": red})
        max_ref = max(a[0] for a in alts)
        gap = int(rng.integers(0, 120))
        if nest_here:
            gap = 3 * k + 40        # room for a contained cluster region followed by >= k reference nucleotides
            lf = pos + max_ref + 5
            contained.append((lf, lf + 20, nested_cluster))
            variants[-1]["has_dependency"] = False
        pos += max_ref + gap
    assert pos + k < chrom_len
    g = build_graph(chrom, variants, k, contained)
    g.chrom, g.variants, g.contained = chrom, variants, list(contained)   # kept for cross-checks against the C++ host constructor
    if contained:
        # the variants whose reference allele spans the cut get a missing allele (has_dependency), as VariantFileParser marks them
        for v in range(len(g.seq)):
            if g.nested[v] != NONE32:
                for r in g.refvars[v] + ([g.var[v]] if g.var[v] != NONE16 else []):
                    if not g.has_dep[r]:
                        g.has_dep[r] = 1
                        g.num_alleles[r] += 1
    g.paths = random_paths(g, rng, max_paths)
    return g


class PathsBatch(C.Structure):
    _fields_ = [("num_clusters", C.c_uint32)] + [(n, C.c_void_p) for n in (
        "vertex_off", "num_paths", "seq_off", "seq", "vertex_variant", "vertex_allele", "vertex_flags", "vertex_nested", "refvar_off", "refvar",
        "path_off", "path_vertices", "var_off", "var_num_alleles", "var_has_dependency", "in_off", "in_src")]


def flatten(graphs):
    """list of Graph (with .paths) -> dict of numpy arrays named like bt_paths_batch's fields"""
    f = {"num_clusters": len(graphs)}
    vertex_off, num_paths, seq_off, seqs, vvar, vall, vfl, vnest, roff, rv, poff, pv, voff, vna, vdep = [0], [], [0], [], [], [], [], [], [0], [], [0], [], [0], [], []
    in_off, in_src = [0], []
    for g in graphs:
        nv = len(g.seq)
        ins = [[] for _ in range(nv)]
        for a, b in g.edges:
            ins[b].append(a)
        for v in range(nv):
            in_src.extend(ins[v])
            in_off.append(len(in_src))
        if getattr(g, "paths", None) is None:
            g.paths = np.zeros((0, nv), np.uint8)
        vertex_off.append(vertex_off[-1] + nv)
        num_paths.append(g.paths.shape[0])
        for v in range(nv):
            seqs.append(g.seq[v])
            seq_off.append(seq_off[-1] + len(g.seq[v]))
            vvar.append(g.var[v])
            vall.append(g.allele[v])
            vfl.append((1 if g.disconnected[v] else 0) | (2 if g.redundant[v] else 0))
            vnest.append(g.nested[v])
            rv.extend(g.refvars[v])
            roff.append(len(rv))
        pv.append(g.paths.reshape(-1))
        poff.append(poff[-1] + g.paths.size)
        vna.extend(g.num_alleles)
        vdep.extend(g.has_dep)
        voff.append(len(vna))
    f["vertex_off"] = np.asarray(vertex_off, np.uint32)
    f["num_paths"] = np.asarray(num_paths, np.uint32)
    f["seq_off"] = np.asarray(seq_off, np.uint64)
    f["seq"] = np.ascontiguousarray(np.concatenate(seqs).astype(np.uint8)) if seqs else np.zeros(0, np.uint8)
    f["vertex_variant"] = np.asarray(vvar, np.uint16)
    f["vertex_allele"] = np.asarray(vall, np.uint16)
    f["vertex_flags"] = np.asarray(vfl, np.uint8)
    f["vertex_nested"] = np.asarray(vnest, np.uint32)
    f["refvar_off"] = np.asarray(roff, np.uint32)
    f["refvar"] = np.asarray(rv, np.uint16)
    f["path_off"] = np.asarray(poff, np.uint64)
    f["path_vertices"] = np.ascontiguousarray(np.concatenate(pv).astype(np.uint8)) if pv else np.zeros(0, np.uint8)
    f["var_off"] = np.asarray(voff, np.uint32)
    f["var_num_alleles"] = np.asarray(vna, np.uint16)
    f["var_has_dependency"] = np.asarray(vdep, np.uint8)
    f["in_off"] = np.asarray(in_off, np.uint32)
    f["in_src"] = np.asarray(in_src, np.uint32)
    return f


FIELDS = [n for n, _ in PathsBatch._fields_[1:]]


def to_ctypes(f):
    """-> (PathsBatch, keepalive list)"""
    b = PathsBatch()
    b.num_clusters = f["num_clusters"]
    keep = []
    for n in FIELDS:
        a = np.ascontiguousarray(f[n])
        if a.size == 0:
            a = np.zeros(1, a.dtype)
        keep.append(a)
        setattr(b, n, a.ctypes.data)
    return b, keep


def gibbs_batch_from_candidates(cand, f, groups, S, gender=None, ploidy=None, cluster_ids=None, sources=None, out_edges=None):
    """The hand-over between the two halves of the path: getHaplotypeCandidates' bundle (bt_paths_candidates_fetch / the oracle's,
    `cand`) for the graphs `f`, plus the group structure (`groups`: list of lists of cluster indices; `sources`: per group the local
    vertex ids of its source vertices, default every cluster; `out_edges`: per cluster the local vertex ids of its nested clusters,
    default none — VariantClusterGroup.cpp:47-107), -> the dict layout of bt_gibbs_batch (synth.flatten's).  Multicluster k-mers of a
    group share one record: kmer_shared numbers the distinct keys among the group's multicluster rows."""
    order = [c for g in groups for c in g]
    assert sorted(order) == list(range(f["num_clusters"])) and order == sorted(order), "groups must partition the clusters in order"
    G = len(groups)
    gender = np.zeros(S, np.uint8) if gender is None else np.asarray(gender, np.uint8)
    ploidy = np.full((G, S), 2, np.uint8) if ploidy is None else np.asarray(ploidy, np.uint8).reshape(G, S)
    R = int(cand["kmer_off"][-1])
    shared = np.full(R, -1, np.int32)
    num_shared = []
    for g in groups:
        keys = {}
        for c in g:
            r0 = int(cand["kmer_off"][c])
            for r in cand["multi_idx"][cand["multi_off"][c]:cand["multi_off"][c + 1]]:
                key = (int(cand["kmer_key"][2 * (r0 + r)]), int(cand["kmer_key"][2 * (r0 + r) + 1]))
                shared[r0 + r] = keys.setdefault(key, len(keys))
        num_shared.append(len(keys))
    H = f["num_paths"].astype(np.uint32)
    V = (f["var_off"][1:] - f["var_off"][:-1]).astype(np.uint32)
    goff = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.uint32)
    out = {
        "S": S, "gender": gender, "num_groups": G, "num_clusters": f["num_clusters"],
        "group_index": np.arange(G, dtype=np.uint32), "group_cluster_off": goff, "group_ploidy": np.ascontiguousarray(ploidy.reshape(-1)),
        "group_source_off": goff.copy() if sources is None else np.concatenate([[0], np.cumsum([len(x) for x in sources])]).astype(np.uint32),
        "group_sources": np.concatenate([np.arange(len(g), dtype=np.uint32) for g in groups]) if sources is None else np.concatenate([np.asarray(x, np.uint32) for x in sources]),
        "group_num_shared": np.asarray(num_shared, np.uint32),
        "cluster_idx": np.arange(f["num_clusters"], dtype=np.uint32) if cluster_ids is None else np.asarray(cluster_ids, np.uint32),
        "edge_off": np.zeros(f["num_clusters"] + 1, np.uint32) if out_edges is None else np.concatenate([[0], np.cumsum([len(x) for x in out_edges])]).astype(np.uint32),
        "edges": np.zeros(0, np.uint32) if out_edges is None else np.concatenate([np.asarray(x, np.uint32) for x in out_edges] + [np.zeros(0, np.uint32)]),
        "num_haplotypes": H, "num_variants": V, "kmer_shared": shared,
        "var_num_alleles": f["var_num_alleles"], "var_has_dependency": f["var_has_dependency"],
    }
    for name in ("kmer_off", "hap_kmer_mult", "kmer_has_counts", "kmer_counts", "kmer_ic_mult", "kv_off", "kv_var", "kv_bits", "unique_off", "unique_idx",
                 "multi_off", "multi_idx", "hap_allele", "hapnest_off", "hapnest_idx", "nestdep_off", "nestdep_cluster", "nestdep_var_off", "nestdep_var"):
        out[name] = cand[name]
    return out
