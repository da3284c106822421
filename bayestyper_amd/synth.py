"""Synthetic variant-cluster-group batches for the Gibbs path (BASELINE.md §3 / SURVEY §8d shapes).

Workload generation only (numpy): builds the flattened VariantClusterHaplotypes bundles that
``bt_gibbs_create`` (include/btgpu.h) consumes.  Shapes:

  A "SNV"              1 cluster/group, V=1, H=2,  K=110
  B "multi-SNV/indel"  1 cluster/group, V=4, H=10, K=440
  C "SV/nested"        3 clusters/group: root V=6,H=32 + two nested children of shape B, multicluster k-mers
  D "joint"            1 cluster/group, V=8, H=256, K=4000

k-mer counts: per-copy NB(mean 15, var 30); noise Poisson(0.05); truth diplotypes from Dirichlet(1) haplotype
frequencies (seeded).
"""
import ctypes as C

import numpy as np

NOHAP = 0xFFFF


class ClusterSpec:
    """One cluster's tensor bundle in Python form (see VariantClusterHaplotypes.hpp:52-109)."""

    def __init__(self, H, V):
        self.H, self.V = H, V
        self.M = np.zeros((0, H), np.uint8)
        self.has_counts = np.zeros(0, np.uint8)
        self.ic = np.zeros((0, 2), np.uint8)
        self.shared = np.zeros(0, np.int32)          # -1 or shared record index within the group
        self.kv = []                                   # per k-mer: list of (variant, bool[H])
        self.hap_allele = np.zeros((H, V), np.uint16)
        self.hap_nested = [[] for _ in range(H)]
        self.var_num_alleles = np.full(V, 2, np.uint16)
        self.var_has_dep = np.zeros(V, np.uint8)
        self.nestdep = {}                              # child variant_cluster_idx -> [variant idx desc]

    @property
    def K(self):
        return len(self.M)

    def add_kmers(self, rows, kv, has_counts=1, ic=(0, 0), shared=None):
        n = len(rows)
        self.M = np.concatenate([self.M, np.asarray(rows, np.uint8).reshape(n, self.H)])
        self.has_counts = np.concatenate([self.has_counts, np.full(n, has_counts, np.uint8)])
        self.ic = np.concatenate([self.ic, np.tile(np.asarray(ic, np.uint8), (n, 1))])
        self.shared = np.concatenate([self.shared, np.full(n, -1, np.int32) if shared is None else np.asarray(shared, np.int32)])
        self.kv.extend(kv)


def make_cluster(rng, V, H, kmers_per_allele, flank_kmers=0, ic_kmers=0, has_dependency=False):
    """biallelic variants; haplotype 0 is all-reference, the others are distinct random allele combinations"""
    c = ClusterSpec(H, V)
    if has_dependency:
        c.var_has_dep[:] = 1
        c.var_num_alleles[:] = 3
    combos = {tuple([0] * V)}
    haps = [tuple([0] * V)]
    # single-alt haplotypes first so every allele is covered, then random combinations
    for v in range(V):
        if len(haps) < H:
            t = tuple(1 if i == v else 0 for i in range(V))
            if t not in combos:
                combos.add(t)
                haps.append(t)
    while len(haps) < H:
        t = tuple(int(x) for x in rng.integers(0, 2, V))
        if t not in combos or 2 ** V <= len(combos):
            combos.add(t)
            haps.append(t)
    c.hap_allele[:, :] = np.asarray(haps, np.uint16)
    for v in range(V):
        for a in (0, 1):
            carriers = c.hap_allele[:, v] == a
            rows = np.tile(carriers.astype(np.uint8), (kmers_per_allele, 1))
            c.add_kmers(rows, [[(v, carriers.copy())] for _ in range(kmers_per_allele)])
    if flank_kmers:   # k-mers on every haplotype, overlapping no variant allele specifically
        c.add_kmers(np.ones((flank_kmers, H), np.uint8), [[] for _ in range(flank_kmers)])
    if ic_kmers:      # allele k-mers that also occur once elsewhere in the genome (intercluster multiplicity 2 = diploid)
        for j in range(ic_kmers):
            v = int(rng.integers(0, V))
            carriers = c.hap_allele[:, v] == 1
            c.add_kmers(carriers.astype(np.uint8)[None, :], [[(v, carriers.copy())]], ic=(2, 2))
    return c


class GroupSpec:
    def __init__(self, clusters, cluster_idx, edges=None, sources=None, num_shared=0):
        self.clusters = clusters
        self.cluster_idx = list(cluster_idx)
        self.edges = edges if edges is not None else [[] for _ in clusters]
        self.sources = sources if sources is not None else list(range(len(clusters)))
        self.num_shared = num_shared


def group_shape_A(rng, cid=0):
    return GroupSpec([make_cluster(rng, 1, 2, 55)], [cid])


def group_shape_B(rng, cid=0):
    return GroupSpec([make_cluster(rng, 4, 10, 55)], [cid])


def group_shape_D(rng, cid=0):
    return GroupSpec([make_cluster(rng, 8, 256, 250)], [cid])


def group_shape_C(rng, cid=0, root_H=32, root_kpa=500, child_kpa=55, shared_frac=0.05):
    """root (V=6, variant 0 = a deletion that removes both children) + two nested children; ~5 % multicluster k-mers"""
    root = make_cluster(rng, 6, root_H, root_kpa)
    kids = [make_cluster(rng, 4, 10, child_kpa, has_dependency=True) for _ in range(2)]
    cids = [cid, cid + 1, cid + 2]
    for h in range(root.H):
        if root.hap_allele[h, 0] == 0:           # haplotypes without the deletion run through the nested regions
            root.hap_nested[h] = sorted(cids[1:])
    for ch in cids[1:]:
        root.nestdep[ch] = [0]
    # multicluster k-mers: present in the root (on non-deletion haplotypes) and in one child (on all its haplotypes)
    n_sh = 0
    for ci, kid in enumerate(kids):
        n = max(1, int(shared_frac * kid.K))
        ids = np.arange(n_sh, n_sh + n, dtype=np.int32)
        carriers = root.hap_allele[:, 0] == 0
        root.add_kmers(np.tile(carriers.astype(np.uint8), (n, 1)), [[(0, carriers.copy())] for _ in range(n)], shared=ids)
        kid.add_kmers(np.ones((n, kid.H), np.uint8), [[] for _ in range(n)], shared=ids)
        n_sh += n
    return GroupSpec([root] + kids, cids, edges=[[1, 2], [], []], sources=[0], num_shared=n_sh)


SHAPES = {"A": group_shape_A, "B": group_shape_B, "C": group_shape_C, "D": group_shape_D}


def _truth_counts(rng, grp, S, ploidy, gender, mean=15.0, var=30.0, noise=0.05, unobserved_frac=0.02):
    """per cluster: counts (K,S) u8 + has_counts, consistent with a random truth diplotype per sample"""
    p = mean / var
    size = mean * mean / (var - mean)
    out = []
    shared_counts = {}
    root_dip = None
    for ci, c in enumerate(grp.clusters):
        freq = rng.dirichlet(np.ones(c.H))
        counts = np.zeros((c.K, S), np.uint8)
        for s in range(S):
            pl = int(ploidy[s])
            if root_dip is not None and ci > 0:   # nested child: copies = root haplotypes that run through it
                pl = sum(1 for h in root_dip[s] if h != NOHAP and grp.cluster_idx[ci] in grp.clusters[0].hap_nested[h])
            hs = list(rng.choice(c.H, size=pl, p=freq)) + [NOHAP] * (2 - pl)
            if ci == 0:
                root_dip = root_dip or {}
                root_dip[s] = hs
            m = np.zeros(c.K, np.int64)
            for h in hs:
                if h != NOHAP:
                    m += c.M[:, h]
            m += c.ic[:, int(gender[s])]
            lam = np.where(m > 0, 0, noise)
            cnt = rng.poisson(lam)
            pos = m > 0
            cnt[pos] = rng.negative_binomial(size * m[pos], p)
            counts[:, s] = np.minimum(cnt, 255)
        out.append(counts)
    # multicluster k-mers are ONE k-mer: all rows that share a record carry the same observed counts
    for ci, c in enumerate(grp.clusters):
        for k in np.nonzero(c.shared >= 0)[0]:
            j = int(c.shared[k])
            if j in shared_counts:
                out[ci][k] = shared_counts[j]
            else:
                shared_counts[j] = np.maximum(out[ci][k], 1)   # observed (count > 0) so the coupling is exercised
                out[ci][k] = shared_counts[j]
    has = []
    for ci, c in enumerate(grp.clusters):
        hc = c.has_counts.copy()
        # k-mers never seen in any sample and with no intercluster copy are absent from the count table
        absent = (out[ci].sum(axis=1) == 0) & (c.ic.sum(axis=1) == 0) & (c.shared < 0)
        hc[absent & (rng.random(c.K) < 0.9)] = 0
        has.append(hc)
    return out, has


def flatten(groups, S, rng, ploidy=None, gender=None, group_index=None):
    """list[GroupSpec] -> dict of numpy arrays laid out as bt_gibbs_batch"""
    G = len(groups)
    gender = np.zeros(S, np.uint8) if gender is None else np.asarray(gender, np.uint8)
    ploidy = np.full((G, S), 2, np.uint8) if ploidy is None else np.asarray(ploidy, np.uint8).reshape(G, S)
    b = {k: [] for k in ("cluster_idx", "edges", "num_haplotypes", "num_variants", "hap_kmer_mult", "kmer_has_counts", "kmer_counts", "kmer_ic_mult",
                         "kmer_shared", "kv_var", "kv_bits", "unique_idx", "multi_idx", "hap_allele", "hapnest_idx", "var_num_alleles",
                         "var_has_dependency", "nestdep_cluster", "nestdep_var", "group_sources")}
    offs = {k: [0] for k in ("group_cluster_off", "group_source_off", "edge_off", "kmer_off", "kv_off", "unique_off", "multi_off", "hapnest_off",
                             "nestdep_off", "nestdep_var_off")}
    num_shared = []
    for gi, grp in enumerate(groups):
        counts, has = _truth_counts(rng, grp, S, ploidy[gi], gender)
        num_shared.append(grp.num_shared)
        b["group_sources"].extend(grp.sources)
        offs["group_source_off"].append(len(b["group_sources"]))
        for ci, c in enumerate(grp.clusters):
            b["cluster_idx"].append(grp.cluster_idx[ci])
            b["edges"].extend(grp.edges[ci])
            offs["edge_off"].append(len(b["edges"]))
            b["num_haplotypes"].append(c.H)
            b["num_variants"].append(c.V)
            b["hap_kmer_mult"].append(c.M.reshape(-1))
            b["kmer_has_counts"].append(has[ci])
            b["kmer_counts"].append(counts[ci].reshape(-1))
            b["kmer_ic_mult"].append(c.ic.reshape(-1))
            b["kmer_shared"].append(c.shared)
            HW = (c.H + 31) // 32
            for k in range(c.K):
                for (v, bits) in c.kv[k]:
                    b["kv_var"].append(v)
                    w = np.zeros(HW, np.uint32)
                    idx = np.nonzero(bits)[0]
                    np.bitwise_or.at(w, idx // 32, (np.uint32(1) << (idx % 32).astype(np.uint32)))
                    b["kv_bits"].append(w)
                offs["kv_off"].append(len(b["kv_var"]))
            offs["kmer_off"].append(offs["kmer_off"][-1] + c.K)
            uniq = np.nonzero(c.shared < 0)[0].astype(np.uint32)
            multi = np.nonzero(c.shared >= 0)[0].astype(np.uint32)
            b["unique_idx"].append(uniq)
            offs["unique_off"].append(offs["unique_off"][-1] + len(uniq))
            b["multi_idx"].append(multi)
            offs["multi_off"].append(offs["multi_off"][-1] + len(multi))
            b["hap_allele"].append(c.hap_allele.reshape(-1))
            for h in range(c.H):
                b["hapnest_idx"].extend(c.hap_nested[h])
                offs["hapnest_off"].append(len(b["hapnest_idx"]))
            b["var_num_alleles"].append(c.var_num_alleles)
            b["var_has_dependency"].append(c.var_has_dep)
            for child, vs in c.nestdep.items():
                b["nestdep_cluster"].append(child)
                b["nestdep_var"].extend(vs)
                offs["nestdep_var_off"].append(len(b["nestdep_var"]))
            offs["nestdep_off"].append(len(b["nestdep_cluster"]))
        offs["group_cluster_off"].append(len(b["cluster_idx"]))

    def cat(lst, dt):
        if len(lst) and isinstance(lst[0], np.ndarray):
            return np.ascontiguousarray(np.concatenate(lst).astype(dt)) if len(lst) else np.zeros(0, dt)
        return np.ascontiguousarray(np.asarray(lst, dtype=dt))

    out = {
        "S": S, "gender": gender,
        "num_groups": G, "num_clusters": len(b["cluster_idx"]),
        "group_index": np.arange(G, dtype=np.uint32) if group_index is None else np.asarray(group_index, np.uint32),
        "group_cluster_off": cat(offs["group_cluster_off"], np.uint32),
        "group_ploidy": np.ascontiguousarray(ploidy.reshape(-1)),
        "group_source_off": cat(offs["group_source_off"], np.uint32),
        "group_sources": cat(b["group_sources"], np.uint32),
        "group_num_shared": cat(num_shared, np.uint32),
        "cluster_idx": cat(b["cluster_idx"], np.uint32),
        "edge_off": cat(offs["edge_off"], np.uint32),
        "edges": cat(b["edges"], np.uint32),
        "num_haplotypes": cat(b["num_haplotypes"], np.uint32),
        "num_variants": cat(b["num_variants"], np.uint32),
        "kmer_off": cat(offs["kmer_off"], np.uint32),
        "hap_kmer_mult": cat(b["hap_kmer_mult"], np.uint8),
        "kmer_has_counts": cat(b["kmer_has_counts"], np.uint8),
        "kmer_counts": cat(b["kmer_counts"], np.uint8),
        "kmer_ic_mult": cat(b["kmer_ic_mult"], np.uint8),
        "kmer_shared": cat(b["kmer_shared"], np.int32),
        "kv_off": cat(offs["kv_off"], np.uint32),
        "kv_var": cat(b["kv_var"], np.uint16),
        "kv_bits": cat(b["kv_bits"], np.uint32) if b["kv_bits"] else np.zeros(0, np.uint32),
        "unique_off": cat(offs["unique_off"], np.uint32),
        "unique_idx": cat(b["unique_idx"], np.uint32),
        "multi_off": cat(offs["multi_off"], np.uint32),
        "multi_idx": cat(b["multi_idx"], np.uint32),
        "hap_allele": cat(b["hap_allele"], np.uint16),
        "hapnest_off": cat(offs["hapnest_off"], np.uint32),
        "hapnest_idx": cat(b["hapnest_idx"], np.uint32),
        "var_num_alleles": cat(b["var_num_alleles"], np.uint16),
        "var_has_dependency": cat(b["var_has_dependency"], np.uint8),
        "nestdep_off": cat(offs["nestdep_off"], np.uint32),
        "nestdep_cluster": cat(b["nestdep_cluster"], np.uint32),
        "nestdep_var_off": cat(offs["nestdep_var_off"], np.uint32),
        "nestdep_var": cat(b["nestdep_var"], np.uint16),
    }
    return out


_OFFSET_OF = {   # offset array -> the arrays it indexes (used by replicate())
    "group_cluster_off": "cluster", "group_source_off": "group_sources", "edge_off": "edges", "kmer_off": "row", "kv_off": "kv_var",
    "unique_off": "unique_idx", "multi_off": "multi_idx", "hapnest_off": "hapnest_idx", "nestdep_off": "nestdep_cluster", "nestdep_var_off": "nestdep_var",
}


def replicate(flat, N, rng, mean=15.0, var=30.0, noise=0.05):
    """Vectorised: N copies of a single-template flat batch (any number of groups in the template) with fresh random counts
    for single-cluster groups (shapes A, B, D).  Group indices become 0..N*G-1."""
    G0, C0, S = flat["num_groups"], flat["num_clusters"], flat["S"]
    out = dict(flat)
    out["num_groups"], out["num_clusters"] = G0 * N, C0 * N
    out["group_index"] = np.arange(G0 * N, dtype=np.uint32)
    per_item = ("group_ploidy", "group_sources", "group_num_shared", "cluster_idx", "edges", "num_haplotypes", "num_variants", "hap_kmer_mult",
                "kmer_has_counts", "kmer_counts", "kmer_ic_mult", "kmer_shared", "kv_var", "kv_bits", "unique_idx", "multi_idx", "hap_allele",
                "hapnest_idx", "var_num_alleles", "var_has_dependency", "nestdep_cluster", "nestdep_var")
    for k in per_item:
        out[k] = np.tile(flat[k], N)
    for k in _OFFSET_OF:
        o = flat[k].astype(np.uint64)
        step = o[-1]
        body = (o[None, :-1] + (np.arange(N, dtype=np.uint64) * step)[:, None]).reshape(-1)
        out[k] = np.concatenate([body, [step * N]]).astype(np.uint32)
    # fresh counts for single-cluster groups
    if C0 == G0:
        p, size = mean / var, mean * mean / (var - mean)
        counts = out["kmer_counts"].reshape(N, -1, S).copy()
        for c in range(C0):
            H = int(flat["num_haplotypes"][c])
            r0, r1 = int(flat["kmer_off"][c]), int(flat["kmer_off"][c + 1])
            m0 = int(sum(int(flat["num_haplotypes"][i]) * (int(flat["kmer_off"][i + 1]) - int(flat["kmer_off"][i])) for i in range(c)))
            M = flat["hap_kmer_mult"][m0:m0 + (r1 - r0) * H].reshape(r1 - r0, H).astype(np.int64)
            ic = flat["kmer_ic_mult"][2 * r0:2 * r1].reshape(-1, 2).astype(np.int64)
            freq = rng.dirichlet(np.ones(H), size=N)                      # (N,H)
            cum = np.cumsum(freq, axis=1)
            u = rng.random((N, S, 2))
            hs = (u[..., None] > cum[:, None, None, :]).sum(-1).clip(0, H - 1)   # (N,S,2)
            m = M[:, hs[..., 0]] + M[:, hs[..., 1]]                       # (K,N,S)
            m = np.transpose(m, (1, 0, 2)) + ic[None, :, flat["gender"].astype(np.int64)]   # (N,K,S)
            cnt = rng.poisson(np.where(m > 0, 0.0, noise))
            pos = m > 0
            cnt[pos] = rng.negative_binomial(size * m[pos], p)
            counts[:, r0:r1, :] = np.minimum(cnt, 255).astype(np.uint8)
        out["kmer_counts"] = np.ascontiguousarray(counts.reshape(-1))
    return out


def make_batch(shape, n_groups, S, seed, templates=1):
    """n_groups groups of one shape: `templates` distinct structures, replicated (vectorised) with fresh counts"""
    rng = np.random.default_rng(seed)
    if shape == "C" or n_groups <= 64:
        groups = []
        cid = 0
        for _ in range(n_groups):
            g = SHAPES[shape](rng, cid)
            cid += len(g.clusters)
            groups.append(g)
        return flatten(groups, S, rng)
    tmpl = flatten([SHAPES[shape](rng, i) for i in range(templates)], S, rng)
    reps = (n_groups + templates - 1) // templates
    return replicate(tmpl, reps, rng)


# --------------------------------------------------------------------------------------------------------------
# ctypes views of bt_gibbs_params / bt_gibbs_batch (include/btgpu.h)
# --------------------------------------------------------------------------------------------------------------
class GibbsParams(C.Structure):
    _fields_ = [("num_samples", C.c_uint32), ("seed", C.c_uint32), ("num_chains", C.c_uint32), ("burn_in", C.c_uint32),
                ("num_iterations", C.c_uint32), ("kmer_subsampling_rate", C.c_float), ("max_haplotype_variant_kmers", C.c_uint32),
                ("noise_seeding", C.c_uint32), ("gender", C.c_void_p)]


_BATCH_PTRS = ["group_index", "group_cluster_off", "group_ploidy", "group_source_off", "group_sources", "group_num_shared", "cluster_idx", "edge_off",
               "edges", "num_haplotypes", "num_variants", "kmer_off", "hap_kmer_mult", "kmer_has_counts", "kmer_counts", "kmer_ic_mult", "kmer_shared",
               "kv_off", "kv_var", "kv_bits", "unique_off", "unique_idx", "multi_off", "multi_idx", "hap_allele", "hapnest_off", "hapnest_idx",
               "var_num_alleles", "var_has_dependency", "nestdep_off", "nestdep_cluster", "nestdep_var_off", "nestdep_var"]


class GibbsBatch(C.Structure):
    _fields_ = [("num_groups", C.c_uint32), ("num_clusters", C.c_uint32)] + [(n, C.c_void_p) for n in _BATCH_PTRS]


def to_ctypes(flat, seed=42, chains=20, burn=100, iters=250, rate=0.1, max_hvk=500, noise_seeding=0):
    """-> (GibbsParams, GibbsBatch, keepalive list)"""
    keep = []
    p = GibbsParams()
    p.num_samples, p.seed, p.num_chains, p.burn_in, p.num_iterations = flat["S"], seed, chains, burn, iters
    p.kmer_subsampling_rate, p.max_haplotype_variant_kmers, p.noise_seeding = rate, max_hvk, noise_seeding
    g = np.ascontiguousarray(flat["gender"], np.uint8)
    keep.append(g)
    p.gender = g.ctypes.data
    b = GibbsBatch()
    b.num_groups, b.num_clusters = flat["num_groups"], flat["num_clusters"]
    for n in _BATCH_PTRS:
        a = flat[n]
        if a.size == 0:
            a = np.zeros(1, a.dtype)
        a = np.ascontiguousarray(a)
        keep.append(a)
        setattr(b, n, a.ctypes.data)
    return p, b, keep


_PER_ITEM = ("group_ploidy", "group_sources", "group_num_shared", "cluster_idx", "edges", "num_haplotypes", "num_variants", "hap_kmer_mult",
             "kmer_has_counts", "kmer_counts", "kmer_ic_mult", "kmer_shared", "kv_var", "kv_bits", "unique_idx", "multi_idx", "hap_allele",
             "hapnest_idx", "var_num_alleles", "var_has_dependency", "nestdep_cluster", "nestdep_var")


def concat(flats):
    """concatenate flat batches (same S, same gender) into one; group_index is renumbered 0..G-1"""
    out = dict(flats[0])
    for k in _PER_ITEM:
        out[k] = np.ascontiguousarray(np.concatenate([f[k] for f in flats]))
    for k in _OFFSET_OF:
        parts, base = [], np.uint64(0)
        for f in flats:
            o = f[k].astype(np.uint64)
            parts.append(o[:-1] + base)
            base = base + o[-1]
        out[k] = np.concatenate(parts + [np.asarray([base], np.uint64)]).astype(np.uint32)
    out["num_groups"] = int(sum(f["num_groups"] for f in flats))
    out["num_clusters"] = int(sum(f["num_clusters"] for f in flats))
    out["group_index"] = np.arange(out["num_groups"], dtype=np.uint32)
    return out


def make_mixture(n_groups, S, seed, fractions=None, templates=4):
    """WGS-like mixture of group shapes, ordered by number of variants descending like the reference
    sorts groups (main.cpp:247).  Shape D (H=256) needs >= 8 samples' worth of haplotype candidates and is left out
    below that (--max-number-of-sample-haplotypes 32 caps H at 32*S)."""
    if fractions is None:
        fractions = {"A": 0.90, "B": 0.08, "C": 0.015, "D": 0.005} if S >= 8 else {"A": 0.90, "B": 0.085, "C": 0.015}
    rng = np.random.default_rng(seed)
    flats, counts = [], {}
    for shape in ("D", "C", "B", "A"):
        if shape not in fractions:
            continue
        n = int(round(n_groups * fractions[shape]))
        if n == 0:
            continue
        counts[shape] = n
        t = min(templates, n)
        if shape == "C":
            cid = 0
            groups = []
            for _ in range(t):
                g = SHAPES[shape](rng, cid)
                cid += 3
                groups.append(g)
            tmpl = flatten(groups, S, rng)
        else:
            tmpl = flatten([SHAPES[shape](rng, i) for i in range(t)], S, rng)
        reps = (n + t - 1) // t
        flats.append(replicate(tmpl, reps, rng))
    out = concat(flats)
    out["mixture"] = {k: int(v) for k, v in counts.items()}
    return out


def algorithmic_bytes_per_chain(flat, subsampling_rate=0.1):
    """HBM floor of the Gibbs path per (cluster, chain) = inputs once + state in/out once:
    K*H (M) + K*(S+4) (counts, flags, ic) + K_sub*4 (subset) + 2*(H*13 + S*4) (state in/out) + 2*2*2496 (two mt19937 in/out)"""
    S = flat["S"]
    H = flat["num_haplotypes"].astype(np.float64)
    K = (flat["kmer_off"][1:].astype(np.float64) - flat["kmer_off"][:-1].astype(np.float64))
    return float((K * H + K * (S + 4) + subsampling_rate * K * 4 + 2 * (H * 13 + S * 4) + 2 * 2 * 2496).sum())
