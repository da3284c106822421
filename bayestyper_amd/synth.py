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
    """biallelic variants; haplotype 0 is all-reference, the others are distinct random allele combinations.
    kmers_per_allele: one number, or a [V][2] table (k-mers of the reference / alternative allele of every variant)"""
    c = ClusterSpec(H, V)
    kpa = np.broadcast_to(np.asarray(kmers_per_allele, np.int64).reshape(-1, 2) if np.ndim(kmers_per_allele) else np.full((V, 2), int(kmers_per_allele)), (V, 2))
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
            n_ka = int(kpa[v, a])
            rows = np.tile(carriers.astype(np.uint8), (n_ka, 1))
            c.add_kmers(rows, [[(v, carriers.copy())] for _ in range(n_ka)])
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


def group_shape_C(rng, cid=0, root_H=32, root_kpa=500, child_kpa=55, shared_frac=0.05, root_V=6, kids=None):
    """root (variant 0 = a deletion that removes the children) + nested children; ~5 % multicluster k-mers.
    kids: list of (V, H, kmers_per_allele) of the children (default: two of shape B)"""
    root = make_cluster(rng, root_V, root_H, root_kpa)
    kid_dims = kids if kids is not None else [(4, 10, child_kpa)] * 2
    kids = [make_cluster(rng, v, h, kpa, has_dependency=True) for v, h, kpa in kid_dims]
    cids = [cid + i for i in range(1 + len(kids))]
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
    return GroupSpec([root] + kids, cids, edges=[list(range(1, 1 + len(kids)))] + [[] for _ in kids], sources=[0], num_shared=n_sh)


def _allele_kmers(rng, k=55, sv=False):
    """(reference, alternative) k-mer counts of one candidate variant: an SNV has k per allele; an insertion / deletion of
    L nucleotides k-1 on the short and k-1+L on the long allele (SV: hundreds to thousands of nucleotides)"""
    u = rng.random()
    if not sv and u < 0.7:
        return (k, k)
    L = int(rng.integers(200, 3000)) if sv else int(min(50, rng.geometric(0.2)))
    return (k - 1, k - 1 + L) if rng.random() < 0.5 else (k - 1 + L, k - 1)


def hetero_group(rng, shape, S, cid=0):
    """One group of a shape class with its dimensions (variants, haplotype candidates, k-mers per allele, flank / intercluster k-mers,
    nested children) drawn from the class's distribution — real clusters of a class differ, which is what pads tiles and diverges lanes."""
    if shape == "A":     # one biallelic SNV / indel
        return GroupSpec([make_cluster(rng, 1, 2, [_allele_kmers(rng)], flank_kmers=int(rng.integers(0, 4)) if rng.random() < 0.2 else 0,
                                       ic_kmers=int(rng.integers(1, 4)) if rng.random() < 0.05 else 0)], [cid])
    if shape == "B":     # a few SNVs / indels within k-1 of each other
        V = int(rng.integers(2, 7))
        H = int(min(2 ** V, max(V + 1, rng.integers(4, 17))))   # every allele is on some candidate (they come from paths)
        return GroupSpec([make_cluster(rng, V, H, [_allele_kmers(rng) for _ in range(V)], flank_kmers=int(rng.integers(0, 40)),
                                       ic_kmers=int(rng.integers(0, 3)))], [cid])
    if shape == "C":     # an SV region: a large root cluster (a deletion over nested small clusters + neighbours)
        V = int(rng.integers(3, 9))
        H = int(min(2 ** V, max(V + 1, rng.integers(8, 33))))
        kpa = [_allele_kmers(rng, sv=True)] + [_allele_kmers(rng, sv=rng.random() < 0.3) for _ in range(V - 1)]
        kids = []
        for _ in range(int(rng.integers(1, 4))):
            v = int(rng.integers(2, 6))
            kids.append((v, int(min(2 ** v, max(v + 1, rng.integers(4, 13)))), [_allele_kmers(rng) for _ in range(v)]))
        return group_shape_C(rng, cid, root_H=H, root_kpa=kpa, root_V=V, kids=kids)
    if shape == "D":     # the many-candidate tail: up to --max-number-of-sample-haplotypes (32) candidates per sample
        Hmax = min(256, 32 * S)
        H = int(rng.integers(max(16, Hmax // 2), Hmax + 1))
        V = int(rng.integers(6, 11))
        return GroupSpec([make_cluster(rng, V, H, [(int(rng.integers(100, 301)),) * 2 for _ in range(V)])], [cid])
    raise ValueError(shape)


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
    """Vectorised: N copies of a template batch (any number of groups, nested groups included), every copy with its own truth
    diplotypes and fresh counts.  Group indices become 0..N*G-1 (copy-major)."""
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
    p, size = mean / var, mean * mean / (var - mean)
    gender = flat["gender"].astype(np.int64)
    counts = out["kmer_counts"].reshape(N, -1, S).copy()
    mult_off = np.concatenate([[0], np.cumsum(flat["num_haplotypes"].astype(np.int64) * (flat["kmer_off"][1:].astype(np.int64) - flat["kmer_off"][:-1].astype(np.int64)))])
    hap_base = np.concatenate([[0], np.cumsum(flat["num_haplotypes"].astype(np.int64))])
    for g in range(G0):
        c0, c1 = int(flat["group_cluster_off"][g]), int(flat["group_cluster_off"][g + 1])
        base_pl = flat["group_ploidy"][g * S:(g + 1) * S].astype(np.int64)
        dips = {}                                  # cluster -> (haplotypes (N,S,2), valid (N,S,2))
        parent = {}
        for c in range(c0, c1):
            for e in flat["edges"][int(flat["edge_off"][c]):int(flat["edge_off"][c + 1])]:
                parent[c0 + int(e)] = c
        for c in range(c0, c1):                    # parents precede their children in the templates
            H = int(flat["num_haplotypes"][c])
            r0, r1 = int(flat["kmer_off"][c]), int(flat["kmer_off"][c + 1])
            M = flat["hap_kmer_mult"][mult_off[c]:mult_off[c + 1]].reshape(r1 - r0, H).astype(np.int64)
            ic = flat["kmer_ic_mult"][2 * r0:2 * r1].reshape(-1, 2).astype(np.int64)
            if c in parent:                        # copies = parent haplotypes that run through this cluster
                pc = parent[c]
                ph, pv = dips[pc]
                through = np.zeros(int(flat["num_haplotypes"][pc]), bool)
                cid = int(flat["cluster_idx"][c])
                for h in range(len(through)):
                    a, b = int(flat["hapnest_off"][hap_base[pc] + h]), int(flat["hapnest_off"][hap_base[pc] + h + 1])
                    through[h] = cid in flat["hapnest_idx"][a:b]
                pl = (pv & through[ph]).sum(-1)    # (N,S)
            else:
                pl = np.broadcast_to(base_pl, (N, S))
            freq = rng.dirichlet(np.ones(H), size=N)
            cum = np.cumsum(freq, axis=1)
            u = rng.random((N, S, 2))
            hs = (u[..., None] > cum[:, None, None, :]).sum(-1).clip(0, H - 1)   # (N,S,2)
            valid = np.arange(2)[None, None, :] < pl[..., None]
            dips[c] = (hs, valid)
            m = M[:, hs[..., 0]] * valid[None, ..., 0] + M[:, hs[..., 1]] * valid[None, ..., 1]     # (K,N,S)
            m = np.transpose(m, (1, 0, 2)) + ic[None, :, gender]                                     # (N,K,S)
            cnt = rng.poisson(np.where(m > 0, 0.0, noise))
            pos = m > 0
            cnt[pos] = rng.negative_binomial(size * m[pos], p)
            cnt[:, flat["kmer_has_counts"][r0:r1] == 0, :] = 0
            counts[:, r0:r1, :] = np.minimum(cnt, 255).astype(np.uint8)
        # multicluster k-mers are ONE k-mer: every row that shares a record carries the same (observed) counts
        first = {}
        for c in range(c0, c1):
            r0, r1 = int(flat["kmer_off"][c]), int(flat["kmer_off"][c + 1])
            sh = flat["kmer_shared"][r0:r1]
            for k in np.nonzero(sh >= 0)[0]:
                j = int(sh[k])
                if j in first:
                    counts[:, r0 + k, :] = counts[:, first[j], :]
                else:
                    counts[:, r0 + k, :] = np.maximum(counts[:, r0 + k, :], 1)
                    first[j] = r0 + k
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


def make_hetero_batch(shape, n_groups, S, seed):
    """n_groups groups of one shape class, every one with its own dimensions (hetero_group) — built group by group (tests)"""
    rng = np.random.default_rng(seed)
    groups, cid = [], 0
    for _ in range(n_groups):
        g = hetero_group(rng, shape, S, cid)
        cid += len(g.clusters)
        groups.append(g)
    return flatten(groups, S, rng)


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


_PTR_DTYPES = {"group_index": np.uint32, "group_cluster_off": np.uint32, "group_ploidy": np.uint8, "group_source_off": np.uint32, "group_sources": np.uint32,
               "group_num_shared": np.uint32, "cluster_idx": np.uint32, "edge_off": np.uint32, "edges": np.uint32, "num_haplotypes": np.uint32, "num_variants": np.uint32,
               "kmer_off": np.uint32, "hap_kmer_mult": np.uint8, "kmer_has_counts": np.uint8, "kmer_counts": np.uint8, "kmer_ic_mult": np.uint8, "kmer_shared": np.int32,
               "kv_off": np.uint32, "kv_var": np.uint16, "kv_bits": np.uint32, "unique_off": np.uint32, "unique_idx": np.uint32, "multi_off": np.uint32, "multi_idx": np.uint32,
               "hap_allele": np.uint16, "hapnest_off": np.uint32, "hapnest_idx": np.uint32, "var_num_alleles": np.uint16, "var_has_dependency": np.uint8,
               "nestdep_off": np.uint32, "nestdep_cluster": np.uint32, "nestdep_var_off": np.uint32, "nestdep_var": np.uint16}


def from_ctypes(batch, S, gender):
    """the inverse of to_ctypes: a GibbsBatch (include/btgpu.h: bt_gibbs_batch, plain pointers) -> a flat dict holding COPIES of its arrays"""
    def view(name, n):
        dt = np.dtype(_PTR_DTYPES[name])
        if n == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (int(n) * dt.itemsize)).from_address(getattr(batch, name))
        return np.frombuffer(buf, dt, int(n)).copy()

    G, Cn = int(batch.num_groups), int(batch.num_clusters)
    f = {"S": int(S), "gender": np.asarray(gender, np.uint8).copy(), "num_groups": G, "num_clusters": Cn}
    for name, n in (("group_index", G), ("group_cluster_off", G + 1), ("group_ploidy", G * S), ("group_source_off", G + 1), ("group_num_shared", G), ("cluster_idx", Cn),
                    ("edge_off", Cn + 1), ("num_haplotypes", Cn), ("num_variants", Cn), ("kmer_off", Cn + 1), ("unique_off", Cn + 1), ("multi_off", Cn + 1), ("nestdep_off", Cn + 1)):
        f[name] = view(name, n)
    H, V = f["num_haplotypes"].astype(np.int64), f["num_variants"].astype(np.int64)
    K = f["kmer_off"].astype(np.int64)
    R = int(K[-1]) if Cn else 0
    f["group_sources"] = view("group_sources", int(f["group_source_off"][-1]))
    f["edges"] = view("edges", int(f["edge_off"][-1]))
    f["hap_kmer_mult"] = view("hap_kmer_mult", int(((K[1:] - K[:-1]) * H).sum()))
    for name, n in (("kmer_has_counts", R), ("kmer_counts", R * S), ("kmer_ic_mult", R * 2), ("kmer_shared", R), ("kv_off", R + 1)):
        f[name] = view(name, n)
    kvo = f["kv_off"].astype(np.int64)
    f["kv_var"] = view("kv_var", int(kvo[-1]))
    f["kv_bits"] = view("kv_bits", int(((kvo[K[1:]] - kvo[K[:-1]]) * ((H + 31) // 32)).sum()))
    f["unique_idx"] = view("unique_idx", int(f["unique_off"][-1]))
    f["multi_idx"] = view("multi_idx", int(f["multi_off"][-1]))
    f["hap_allele"] = view("hap_allele", int((H * V).sum()))
    f["hapnest_off"] = view("hapnest_off", int(H.sum()) + 1)
    f["hapnest_idx"] = view("hapnest_idx", int(f["hapnest_off"][-1]))
    f["var_num_alleles"] = view("var_num_alleles", int(V.sum()))
    f["var_has_dependency"] = view("var_has_dependency", int(V.sum()))
    ND = int(f["nestdep_off"][-1])
    f["nestdep_cluster"] = view("nestdep_cluster", ND)
    f["nestdep_var_off"] = view("nestdep_var_off", ND + 1)
    f["nestdep_var"] = view("nestdep_var", int(f["nestdep_var_off"][-1]))
    return f


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


MIXTURE = {"A": 0.90, "B": 0.08, "C": 0.015, "D": 0.005}     # BASELINE.md §3 (est. WGS-like)
TEMPLATES = {"A": 384, "B": 384, "C": 96, "D": 32}              # distinct structures per shape class in a mixture batch


def make_mixture(n_groups, S, seed, fractions=None, templates=None, hetero=True):
    """WGS-like mixture of group shapes, ordered by size descending like the reference sorts groups (main.cpp:247).
    hetero: the dimensions of every structure (variants, haplotype candidates, k-mers, nested children) are drawn per template from
    the class's distribution (hetero_group); each of the `templates[shape]` structures is instantiated n / templates times, every
    instance with its own truth genotypes and counts.  Shape D, the many-candidate tail, has up to 32 x S candidates (the
    --max-number-of-sample-haplotypes cap), at most 256.  hetero=False: the fixed textbook shapes of SURVEY §8d."""
    fractions = dict(MIXTURE) if fractions is None else fractions
    tcount = dict(TEMPLATES)
    if isinstance(templates, int):
        tcount = {k: templates for k in tcount}
    elif templates:
        tcount.update(templates)
    rng = np.random.default_rng(seed)
    flats, counts = [], {}
    for shape in ("D", "C", "B", "A"):
        n = int(round(n_groups * fractions.get(shape, 0)))
        if n == 0:
            continue
        t = min(tcount[shape], n)
        counts[shape] = (n + t - 1) // t * t   # whole instances of every structure
        groups, cid = [], 0
        for _ in range(t):
            g = hetero_group(rng, shape, S, cid) if hetero else SHAPES[shape](rng, cid)
            cid += len(g.clusters)
            groups.append(g)
        tmpl = flatten(groups, S, rng)
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
