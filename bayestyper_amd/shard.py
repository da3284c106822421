"""Sharding of a genotyping batch over ranks (one process per GPU) and the gather of posterior summaries to rank 0.

BayesTyper parallelises inference over variant-cluster groups with a thread pool (InferenceEngine.cpp:335-382: groups are handed
to threads in chunks, every group is independent given the k-mer table and the count model).  Across GPUs the same independence
is used: every rank takes a subset of the groups, keeps each group's GLOBAL index (the per-group PRNG seeds are derived from
it, VariantClusterGroup.cpp:179-182), runs the whole schedule locally and only the compact per-(cluster, sample) posterior
summary travels: one gather to rank 0.  There is no collective inside the sampler.

Only the default genotyping mode is collective-free.  The noise-estimation drivers (InferenceEngine.cpp:135-276) add the
per-iteration noise-count histograms of all groups: `allreduce_noise_counts` is that one exchange.
"""
import numpy as np

# offset array -> (what one entry of the offset array describes, payload arrays sliced by it with their per-item width)
_GROUP_LEVEL = ("group_cluster_off", "group_source_off")


def group_cost(flat):
    """Relative cost of one group's Gibbs schedule: per sweep every cluster evaluates its diplotype candidates
    (H*(H+1)/2 log-sum steps) and, on cache epochs, sums over its k-mer subset.  Used for balancing only."""
    H = flat["num_haplotypes"].astype(np.float64)
    K = (flat["kmer_off"][1:] - flat["kmer_off"][:-1]).astype(np.float64)
    per_cluster = H * (H + 1) / 2 + 8.0 + K / 16.0
    csum = np.concatenate([[0.0], np.cumsum(per_cluster)])
    off = flat["group_cluster_off"].astype(np.int64)
    return csum[off[1:]] - csum[off[:-1]]


def assign_groups(costs, world):
    """Longest-processing-time assignment of groups to `world` ranks: groups sorted by cost (descending, stable) are dealt in a
    serpentine order.  Returns a list of sorted index arrays, one per rank; deterministic, identical on every rank."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    n = len(order)
    pos = np.arange(n)
    rnd, col = pos // world, pos % world
    owner = np.where(rnd % 2 == 0, col, world - 1 - col)
    return [np.sort(order[owner == r]) for r in range(world)]


def _slice_concat(arr, starts, ends, width=1):
    if len(starts) == 0:
        return arr[:0].copy()
    return np.ascontiguousarray(np.concatenate([arr[int(a) * width:int(b) * width] for a, b in zip(starts, ends)]))


def _rebuild_off(lengths):
    return np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint32)


def take_groups(flat, ids):
    """The flattened batch restricted to the groups `ids` (in that order).  `group_index` keeps the original values, so seeds —
    and therefore every sampled value — are those of the unsharded run."""
    ids = np.asarray(ids, dtype=np.int64)
    S = flat["S"]
    gco = flat["group_cluster_off"].astype(np.int64)
    clusters = np.concatenate([np.arange(gco[g], gco[g + 1]) for g in ids]) if len(ids) else np.zeros(0, np.int64)
    out = {"S": S, "gender": flat["gender"], "num_groups": len(ids), "num_clusters": len(clusters)}
    if "mixture" in flat:
        out["mixture"] = flat["mixture"]
    out["group_index"] = flat["group_index"][ids].astype(np.uint32)
    out["group_cluster_off"] = _rebuild_off(gco[ids + 1] - gco[ids])
    out["group_ploidy"] = np.ascontiguousarray(flat["group_ploidy"].reshape(-1, S)[ids].reshape(-1))
    gso = flat["group_source_off"].astype(np.int64)
    out["group_source_off"] = _rebuild_off(gso[ids + 1] - gso[ids])
    out["group_sources"] = _slice_concat(flat["group_sources"], gso[ids], gso[ids + 1])
    out["group_num_shared"] = flat["group_num_shared"][ids].astype(np.uint32)

    c = clusters
    H = flat["num_haplotypes"].astype(np.int64)
    V = flat["num_variants"].astype(np.int64)
    ko = flat["kmer_off"].astype(np.int64)
    K = ko[1:] - ko[:-1]
    kvo = flat["kv_off"].astype(np.int64)
    out["cluster_idx"] = flat["cluster_idx"][c].astype(np.uint32)
    out["num_haplotypes"] = flat["num_haplotypes"][c].astype(np.uint32)
    out["num_variants"] = flat["num_variants"][c].astype(np.uint32)
    eo = flat["edge_off"].astype(np.int64)
    out["edge_off"] = _rebuild_off(eo[c + 1] - eo[c])
    out["edges"] = _slice_concat(flat["edges"], eo[c], eo[c + 1])
    out["kmer_off"] = _rebuild_off(K[c])
    # per cluster blocks whose start is a running sum over ALL clusters of the source batch
    mult_start = np.concatenate([[0], np.cumsum(K * H)])
    hw = (H + 31) // 32
    nnz = kvo[ko[1:]] - kvo[ko[:-1]]
    kvb_start = np.concatenate([[0], np.cumsum(nnz * hw)])
    hv_start = np.concatenate([[0], np.cumsum(H * V)])
    hap_start = np.concatenate([[0], np.cumsum(H)])
    var_start = np.concatenate([[0], np.cumsum(V)])
    out["hap_kmer_mult"] = _slice_concat(flat["hap_kmer_mult"], mult_start[c], mult_start[c + 1])
    out["kmer_has_counts"] = _slice_concat(flat["kmer_has_counts"], ko[c], ko[c + 1])
    out["kmer_counts"] = _slice_concat(flat["kmer_counts"], ko[c], ko[c + 1], S)
    out["kmer_ic_mult"] = _slice_concat(flat["kmer_ic_mult"], ko[c], ko[c + 1], 2)
    out["kmer_shared"] = _slice_concat(flat["kmer_shared"], ko[c], ko[c + 1])
    # kv_off is per k-mer row: lengths of the kept rows, in order
    rows = np.concatenate([np.arange(ko[i], ko[i + 1]) for i in c]) if len(c) else np.zeros(0, np.int64)
    out["kv_off"] = _rebuild_off(kvo[rows + 1] - kvo[rows])
    out["kv_var"] = _slice_concat(flat["kv_var"], kvo[ko[c]], kvo[ko[c + 1]])
    out["kv_bits"] = _slice_concat(flat["kv_bits"], kvb_start[c], kvb_start[c + 1])
    uo, mo = flat["unique_off"].astype(np.int64), flat["multi_off"].astype(np.int64)
    out["unique_off"] = _rebuild_off(uo[c + 1] - uo[c])
    out["unique_idx"] = _slice_concat(flat["unique_idx"], uo[c], uo[c + 1])
    out["multi_off"] = _rebuild_off(mo[c + 1] - mo[c])
    out["multi_idx"] = _slice_concat(flat["multi_idx"], mo[c], mo[c + 1])
    out["hap_allele"] = _slice_concat(flat["hap_allele"], hv_start[c], hv_start[c + 1])
    hno = flat["hapnest_off"].astype(np.int64)
    haps = np.concatenate([np.arange(hap_start[i], hap_start[i + 1]) for i in c]) if len(c) else np.zeros(0, np.int64)
    out["hapnest_off"] = _rebuild_off(hno[haps + 1] - hno[haps])
    out["hapnest_idx"] = _slice_concat(flat["hapnest_idx"], hno[hap_start[c]], hno[hap_start[c + 1]])
    out["var_num_alleles"] = _slice_concat(flat["var_num_alleles"], var_start[c], var_start[c + 1])
    out["var_has_dependency"] = _slice_concat(flat["var_has_dependency"], var_start[c], var_start[c + 1])
    ndo, ndvo = flat["nestdep_off"].astype(np.int64), flat["nestdep_var_off"].astype(np.int64)
    out["nestdep_off"] = _rebuild_off(ndo[c + 1] - ndo[c])
    out["nestdep_cluster"] = _slice_concat(flat["nestdep_cluster"], ndo[c], ndo[c + 1])
    deps = np.concatenate([np.arange(ndo[i], ndo[i + 1]) for i in c]) if len(c) else np.zeros(0, np.int64)
    out["nestdep_var_off"] = _rebuild_off(ndvo[deps + 1] - ndvo[deps]) if len(deps) else np.zeros(1, np.uint32)
    out["nestdep_var"] = _slice_concat(flat["nestdep_var"], ndvo[ndo[c]], ndvo[ndo[c + 1]])
    return out


def cluster_ids_of(flat, ids):
    """positions (in the unsharded batch) of the clusters of groups `ids`, in shard order"""
    gco = flat["group_cluster_off"].astype(np.int64)
    ids = np.asarray(ids, dtype=np.int64)
    return np.concatenate([np.arange(gco[g], gco[g + 1]) for g in ids]) if len(ids) else np.zeros(0, np.int64)


def summary_from_results(res, num_clusters, S):
    """Host-side statement of bt_gibbs_posterior_summary (include/btgpu.h): per (cluster, sample) the most frequently sampled
    diplotype as h1 | h2 << 16 (ties: smallest (h1, h2)) and its sampling frequency; 0xFFFFFFFF / 0 when nothing was sampled."""
    out = np.zeros((num_clusters, S, 2), np.uint32)
    out[:, :, 0] = 0xFFFFFFFF
    off = res["dip_off"].astype(np.int64)
    for c in range(num_clusters):
        a, b = off[c], off[c + 1]
        if a == b:
            continue
        h1, h2 = res["h1"][a:b].astype(np.uint32), res["h2"][a:b].astype(np.uint32)
        order = np.lexsort((h2, h1))
        for s in range(S):
            f = res["freq"][a:b, s][order]
            j = int(np.argmax(f))   # first maximum in (h1, h2) order
            if f[j] > 0:
                e = order[j]
                out[c, s, 0] = h1[e] | (h2[e] << np.uint32(16))
                out[c, s, 1] = f[j]
    return out


def gather_summaries(local, cluster_ids, num_clusters_total, rank, world, device=None):
    """Gather per-rank summaries (uint32/int32 array [C_local, S, 2], torch tensor or numpy) to rank 0 and scatter them into the
    unsharded cluster order.  Uses the initialised default process group (nccl == RCCL on GPU tensors, gloo on CPU tensors):
    ranks hold different cluster counts, so sizes are exchanged first and the payload is padded to the maximum.
    Returns the [num_clusters_total, S, 2] array on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local).view(np.int32))
    if device is not None:
        t = t.to(device)
    S = t.shape[1]
    ids = torch.as_tensor(np.asarray(cluster_ids, dtype=np.int64), device=t.device)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    if world == 1:
        full = torch.zeros((num_clusters_total, S, 2), dtype=t.dtype, device=t.device)
        full[ids] = t
        return full.cpu().numpy().view(np.uint32)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(x.item()) for x in sizes))
    pad_t = torch.zeros((cap, S, 2), dtype=t.dtype, device=t.device)
    pad_t[: t.shape[0]] = t
    pad_i = torch.zeros(cap, dtype=torch.int64, device=t.device)
    pad_i[: t.shape[0]] = ids
    got_t = [torch.zeros_like(pad_t) for _ in range(world)] if rank == 0 else None
    got_i = [torch.zeros_like(pad_i) for _ in range(world)] if rank == 0 else None
    dist.gather(pad_t, got_t, dst=0)
    dist.gather(pad_i, got_i, dst=0)
    if rank != 0:
        return None
    full = torch.zeros((num_clusters_total, S, 2), dtype=t.dtype, device=t.device)
    for r in range(world):
        m = int(sizes[r].item())
        full[got_i[r][:m]] = got_t[r][:m]
    return full.cpu().numpy().view(np.uint32)


def allreduce_noise_counts(hist):
    """Sum of the per-rank noise-count histograms (estimateNoise: InferenceEngine.cpp:232-262 adds the counts of all groups before
    CountDistribution::sampleNoiseParameters).  `hist`: torch tensor, reduced in place."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


def hist_reducer(device=None):
    """callable(np.uint64[S*256]) -> histogram summed over all ranks, for host.inference_engine.InferenceEngine(reduce_hist=...).
    The counters travel as int64 (all-reduce has no uint64; a count never approaches 2^63)."""
    import torch

    def reduce(hist):
        t = torch.from_numpy(np.ascontiguousarray(hist).view(np.int64).copy())
        if device is not None:
            t = t.to(device)
        allreduce_noise_counts(t)
        return t.cpu().numpy().view(np.uint64)

    return reduce
