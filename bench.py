#!/usr/bin/env python3
"""bench.py — BayesTyper hot path on MI355X: variant-cluster Gibbs iterations/sec + k-mer matches/sec.

One "step" = one pass of the hot path over one batch of synthetic input held in HBM:
  (1) k-mer matching (KmerCounter::parseSampleKmers, KmerCounter.cpp:388-524): the count table is emptied and the KMC record
      stream of EVERY sample is scanned into it (decode -> path-Bloom test -> count-table update).  Sample 0's scan inserts the
      matched k-mers (the cold path), the others find them and add their counts — the weighting of a real S-sample run;
  (2) the default-mode Gibbs schedule (20 chains x (100 burn-in + 250 collected) sweeps,
      InferenceEngine::estimateGenotypesCallback) for every variant-cluster group of the batch;
  (3) the launch's collected samples in the product's result layout (bt_gibbs_result_fetch: every cluster's diplotype sampling frequencies ordered by
      (h1, h2) + its allele k-mer statistics, packed on the device, one copy per array to the host), gathered to rank 0 when --gpus > 1 (RCCL,
      libbtcomm.so) — what `bayesTyper genotype` does per launch before it turns the samples into genotypes (the `results` sub-record times that host step).

Workload at N=1 = BASELINE.json configs[2], the largest single-GPU configuration ("GRCh38 whole genome, CEU trio (3 samples),
SNV+indel+SV merged candidates"): S=3 and one launch-sized slice of the unit — 600 000 variant-cluster groups (80-140 GB of sampler
state in HBM; the narrow tiles of the expensive groups run in rounds next to the two-haplotype tiles) in the WGS-like
mixture of BASELINE.md §3 (90 % biallelic SNV/indel groups, 8 % multi-variant clusters, 1.5 % nested SV groups, 0.5 % many-candidate
clusters with up to 32 x S haplotype candidates), every structure with its own dimensions (bayestyper_amd/synth.py: hetero_group) and
every group with its own truth genotypes and counts; a whole genome (5-15 x 10^6 groups) is a sequence of such launches.  KMC: a
2x10^8-record stream per sample (13-byte records, k=55, p=7), 2 % path-k-mer hit rate against a fpr-1e-4 ThreadedKmerBloom of
5x10^7 path k-mers (SURVEY 8d's stream: 10^9 records per sample).  --samples 10 gives the north star's 10-sample mixture.
Several GPUs (--gpus N, one rank per GPU), default = weak scaling: ONE unit of N x --groups groups — N launch-sized blocks, block r generated and sampled by
rank r, every group with its unit-wide index (from which its seeds derive, so the samples do not depend on the placement) — and a KMC stream per rank; the
ranks' results travel to rank 0 through the product's own exchange library (libbtcomm.so: RCCL over xGMI; torch.distributed/gloo only passes the communicator
id).  Per-rank launch times and the gather time are reported separately (config.per_rank).  A real unit is 5-15 x 10^6 groups, i.e. several launches per GPU:
per-GPU work stays launch-sized as N grows.  Placement independence is checked on a sample: rank 0 runs its first 4 096 groups again as a batch of their own
and compares the posterior summaries (config.subset_equals_batch).  --scaling strong shards ONE --groups batch over the ranks instead (LPT on a cost proxy,
checked against the unsharded batch: config.sharded_equals_unsharded).

Prints ONE JSON line (rank 0).  `value` = variant-cluster Gibbs iterations (cluster-sweeps) per second over the whole job; k-mer
matches/sec is reported beside it.  `roofline` describes the dominant kernel (the Gibbs sweep kernel); `roofline_kmer_match` the KMC
scan kernel (average over the S launches of a step: one inserting, S-1 finding).  `cpu_baseline` times the oracle (a scalar
restatement of the reference's algorithm, parity-pinned in tests/) on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
K = 55
KMC_P = 7
REC = 13                # (55-7)/4 suffix bytes + 1 counter byte
KMER_MATCH_BYTES_PER_RECORD = 15.9   # 13 B record + E[probes] + hit * table update, pure-algorithmic floor


_HASH_SETS = {
    None: ("bt_gibbs", "bt_rng_device", "bt_table", "bt_bloom", "bt_kmer_device", "bt_internal", "bt_ctx"),
    "gibbs": ("bt_gibbs", "bt_rng_device", "bt_internal", "bt_ctx"),
    "kmc": ("bt_table", "bt_bloom", "bt_kmer_device", "bt_internal", "bt_ctx"),
}


def source_hash(part=None):
    """hash of the device sources of the kernels a step runs (Gibbs sampler, count table + KMC scan, Bloom filter, their shared headers — not the
    graph-stage kernels of bt_paths.hip / bt_find_paths.hip, which a step does not launch): profiles/ summaries made from the same sources carry the same hash.
    part = "gibbs" / "kmc": the sources of the Gibbs launch / of the KMC scan alone (a summary of one of them stays valid when only the other's sources change)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    names = _HASH_SETS[part]
    for f in sorted(f for f in glob.glob(os.path.join(ROOT, "bayestyper_amd", "csrc", "*.h*")) if os.path.basename(f).startswith(names)):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _matches(d, part):
    """a profiles/ summary was made from this tree's sources of `part` (or, older summaries, from exactly this tree's sources)"""
    key = "source_hash_" + part
    return d.get(key) == source_hash(part) if key in d else d.get("source_hash") == source_hash()


def committed_traffic(kind):
    """HBM bytes per launch of the Gibbs / KMC-scan kernels from the committed PMC passes of this command (profiles/*_traffic.json,
    written by tools/pmc_traffic.py on the GPU box), but only when they were measured on the sources this library was built from"""
    import glob

    part = "kmc" if kind.startswith("kmc") else "gibbs"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if kind in d and _matches(d, part):
            return d[kind], os.path.relpath(f, ROOT)
    return None, None


def committed_issue_profile(S=None, groups=None):
    """the VALU instruction counts of one Gibbs schedule of the default command (profiles/*_issue.json, tools/issue_profile.py from single-schedule SQ counter
    passes), when measured on the sources this library was built from"""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_issue.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if "gibbs" in d and _matches(d, "gibbs") and (S is None or (d.get("S") == S and abs(d.get("groups", 0) - groups) <= 0.02 * groups)):
            return d, os.path.relpath(f, ROOT)
    return None, None


def rank_batch(groups, S, rank, world, scaling, templates=None):
    """this rank's groups of the job.  weak (default): the unit is `world` launch-sized blocks of `groups` groups, block r generated (seed 1000 + r) and sampled
    by rank r; a group's unit-wide index is r * groups_of_a_block + its index in the block.  strong: ONE batch of `groups` groups dealt to the ranks by
    longest-processing-time on the cost proxy (shard.assign_groups, what `BT_GPUS=N bayesTyper genotype` does with a unit).
    -> (flat batch of this rank, clusters of the whole job, the unit (strong only, else None), this rank's group ids in the unit (strong only))"""
    from bayestyper_amd import shard, synth

    if scaling == "strong" and world > 1:
        unit = synth.make_mixture(groups, S, seed=1000, templates=templates)
        my_ids = shard.assign_groups(shard.group_cost(unit), world)[rank]
        flat = shard.take_groups(unit, my_ids)
        flat["mixture"] = unit["mixture"]
        return flat, unit["num_clusters"], unit, my_ids
    flat = synth.make_mixture(groups, S, seed=1000 + rank, templates=templates)
    flat["group_index"] = (flat["group_index"].astype(np.uint64) + rank * flat["num_groups"]).astype(np.uint32)   # unit-wide group index -> seeds
    return flat, flat["num_clusters"] * world, None, None


def sharded_precheck(ctx, comm, dist, torch, dev, rank, world, S):
    """Before anything is timed on N > 1 GPUs: a 2 000-group unit of the mixture is genotyped (short schedule) SHARDED over the ranks the way
    `BT_GPUS=N bayesTyper genotype` deals a unit (shard.assign_groups), every rank's posterior summaries are gathered to rank 0 through the product's
    exchange step (bt_comm_gather_summaries over RCCL), and rank 0 compares them with its own UNSHARDED run of the unit.  Every rank learns the outcome; a
    difference ends the run on all of them — a scaling run can then not report a number for a wrong result (the first N > 1 RCCL execution this code ever gets
    is the driver's scaling run).  -> True"""
    from bayestyper_amd import lib, shard, synth
    from bayestyper_amd.host import count_model

    unit = synth.make_mixture(2000, S, seed=77)
    parts = shard.assign_groups(shard.group_cost(unit), world)
    mine = shard.take_groups(unit, parts[rank])
    lg, ln = count_model.build_luts(S, mean=15.0, var=30.0, noise_rate=0.05)
    kw = dict(seed=42, chains=2, burn=10, iters=20)

    def summary(flat):
        g = lib.Gibbs(ctx, flat, lg, ln, **kw)
        g.run()
        d = torch.zeros(max(1, flat["num_clusters"] * S * 2), dtype=torch.int32, device=dev)
        ctx.sync()
        torch.cuda.synchronize()
        lib.check(lib.bt_gibbs_posterior_summary(g.h, d.data_ptr()))
        ctx.sync()
        g.close()
        return d

    d_mine = summary(mine)
    Cm, Ct = mine["num_clusters"], unit["num_clusters"]
    c_all = torch.zeros(world, dtype=torch.int64, device=dev)
    c_all[rank] = Cm
    torch.cuda.synchronize()
    comm.allreduce(c_all.data_ptr(), world)
    ctx.sync()
    c_all = [int(x) for x in c_all.tolist()]
    ok = sum(c_all) == Ct
    d_all = torch.zeros(Ct * S * 2 if rank == 0 else 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    comm.gather_words(d_mine.data_ptr(), Cm * S * 2, d_all.data_ptr(), d_all.numel())
    ctx.sync()
    if rank == 0 and ok:
        whole = torch.zeros(Ct, S * 2, dtype=torch.int32, device=dev)
        at = 0
        for r in range(world):
            ids_r = torch.from_numpy(np.ascontiguousarray(shard.cluster_ids_of(unit, parts[r])).astype(np.int64)).to(dev)
            whole[ids_r] = d_all[at * S * 2: (at + c_all[r]) * S * 2].view(c_all[r], S * 2)
            at += c_all[r]
        ref = summary(unit)
        ok = bool(torch.equal(ref.view(Ct, S * 2), whole)) and int(ref.abs().sum().item()) > 0
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
    dist.broadcast(flag, src=0)
    if not int(flag.item()):
        raise RuntimeError("bench: the sharded pre-check failed — the summaries gathered from %d ranks differ from rank 0's unsharded run of the same unit" % world)
    return True


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--groups", type=int, default=600_000, help="variant-cluster groups per GPU")
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="weak (default): a unit of N x --groups groups, one launch-sized block per rank; strong: one --groups batch sharded over the ranks")
    ap.add_argument("--no-verify", action="store_true", help="strong scaling: skip the check of the gathered summaries against the unsharded batch (run by rank 0 after the timed region)")
    ap.add_argument("--records", type=int, default=1_000_000_000, help="KMC records per sample per step (whole job under strong scaling: every rank scans its share)")
    ap.add_argument("--path-kmers", type=int, default=50_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="target duration of the all-cores CPU leg (the sample is sized from a short probe)")
    ap.add_argument("--no-extra", action="store_true", help="skip the sub-records (ten-sample mixture, joint noise genotyping, C4-sized sub-filters)")
    ap.add_argument("--unique-structures", action="store_true", help="every multi-variant / nested / many-candidate group gets a structure of its own (slow to generate)")
    ap.add_argument("--hit-rate", type=float, default=0.02)
    ap.add_argument("--cpu-groups-per-core", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-paths", action="store_true", help="skip the graph-stage throughput figures")
    ap.add_argument("--path-clusters", type=int, default=50_000)
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-resident (PCIe-inclusive) scan figure")
    ap.add_argument("--pcie-records", type=int, default=100_000_000)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from bayestyper_amd import lib, synth
    from bayestyper_amd.host import count_model

    ctx = lib.Ctx(local_rank)
    # (the context keeps its own non-blocking stream; torch's tensors are only memory here.  Wherever a torch operation and a library call touch the same
    # buffer one after the other, both_sync() orders them.  On torch's current stream — the legacy default stream — the launch classes of a Gibbs schedule
    # did not overlap: the launch took 4.75 s instead of 3.8 s.)

    def both_sync():
        ctx.sync()
        torch.cuda.synchronize()
    S = args.samples
    comm = None
    precheck_ok = None
    if world > 1:
        # the exchange steps are the product's (libbtcomm.so: RCCL on the context's stream); the gloo group only carries the communicator id
        from bayestyper_amd import comm as btcomm

        dist.init_process_group(backend="gloo")
        ident = [btcomm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        comm = btcomm.Comm(ctx, ident[0], rank, world)
        precheck_ok = sharded_precheck(ctx, comm, dist, torch, dev, rank, world, S)

    # ------------------------------------------------------------------ Gibbs batch (this rank's groups)
    from bayestyper_amd import shard

    strong = args.scaling == "strong" and world > 1
    verify = strong_verify = args.scaling == "strong" and world > 1 and not args.no_verify
    mix_templates = {"B": 10 ** 9, "C": 10 ** 9, "D": 10 ** 9} if args.unique_structures else None   # (capped at the class's group count)
    flat, C_total, unit, my_ids = rank_batch(args.groups, S, rank, world, args.scaling, mix_templates)
    if strong:
        my_clusters = shard.cluster_ids_of(unit, my_ids)
        if not (verify and rank == 0):
            unit = None
    G, C = flat["num_groups"], flat["num_clusters"]
    lut_g, lut_n = count_model.build_luts(S, mean=15.0, var=30.0, noise_rate=0.05)
    gibbs = lib.Gibbs(ctx, flat, lut_g, lut_n, seed=42)
    sweeps_per_group = gibbs.params.num_chains * (gibbs.params.burn_in + gibbs.params.num_iterations)
    gibbs_chains, gibbs_device_bytes = gibbs.params.num_chains, gibbs.device_bytes()
    cluster_sweeps_per_step = C_total * sweeps_per_group          # whole job
    d_summary = torch.zeros(C * S * 2, dtype=torch.int32, device=dev)
    both_sync()
    if world > 1:   # variable-length gather (bt_comm_gather_summaries): rank 0 receives the ranks' parts in rank order
        c_all = torch.zeros(world, dtype=torch.int64, device=dev)
        c_all[rank] = C
        both_sync()
        comm.allreduce(c_all.data_ptr(), world)
        both_sync()
        c_all = [int(x) for x in c_all.tolist()]
        d_gathered = torch.zeros(sum(c_all) * S * 2 if rank == 0 else 2, dtype=torch.int32, device=dev)   # (strong scaling's check: posterior summaries)
    state = {"res": None, "d_words": None, "d_all": None}


    # ------------------------------------------------------------------ KMC stream + path Bloom + count table (in HBM)
    # under strong scaling the job's stream of args.records records per sample is split over the ranks (every rank scans its share against the
    # replicated path filter, as the executable's ranks do with their record ranges); weak scaling gives every rank a stream of its own
    R = args.records // world if strong else args.records
    gen = torch.Generator(device=dev)
    gen.manual_seed(4 + rank)
    records = torch.randint(0, 256, (R * REC + 16,), dtype=torch.uint8, device=dev, generator=gen)
    records.view(-1)[REC - 1: R * REC: REC] = torch.randint(1, 200, (R,), dtype=torch.uint8, device=dev, generator=gen)   # counts 1..199
    lut = (np.arange(4 ** KMC_P + 1, dtype=np.float64) * (R / 4 ** KMC_P)).astype(np.uint64)
    lut[-1] = R
    both_sync()
    scan = lib.KmcScan(ctx, K, KMC_P, 1, R, lut)
    bloom = lib.Bloom.create(ctx, args.path_kmers + 1_000_000, 1e-4, K, threaded=True)       # main.cpp:517
    n_hit = int(R * args.hit_rate)
    # members that are in the database: decode the records on the device (a chunk at a time), insert a strided subset into the path Bloom
    stride = max(1, R // max(n_hit, 1))
    CH = 50_000_000 // stride * stride or stride
    kmers = torch.zeros((min(CH, R), 2), dtype=torch.int64, device=dev)
    cnts = torch.zeros(min(CH, R), dtype=torch.int32, device=dev)
    inserted = 0
    for a in range(0, R, CH):
        m = min(CH, R - a)
        lib.check(lib.bt_kmc_scan_decode(scan.h, records.data_ptr() + a * REC, a, m, kmers.data_ptr(), cnts.data_ptr()))
        both_sync()
        members = kmers[:m:stride][: max(0, n_hit - inserted)].contiguous()
        both_sync()
        if members.shape[0]:
            lib.check(lib.bt_bloom_insert_batch(bloom.h, members.data_ptr(), members.shape[0]))
        inserted += members.shape[0]
        both_sync()
    absent = torch.randint(-(2 ** 62), 2 ** 62, (max(args.path_kmers - n_hit, 1), 2), dtype=torch.int64, device=dev, generator=gen)
    absent[:, 1] &= (1 << 46) - 1            # 55-mers: 110 bits
    both_sync()
    lib.check(lib.bt_bloom_insert_batch(bloom.h, absent.data_ptr(), absent.shape[0]))
    both_sync()
    del kmers, cnts, absent, members
    table = lib.Table(ctx, max(int(n_hit * 1.5), 1024), S, K)
    d_hits = torch.zeros(1, dtype=torch.int64, device=dev)
    both_sync()
    t_gibbs, t_kmc = lib.Timer(ctx), [lib.Timer(ctx) for _ in range(S)]

    def step(i, timed):
        # (1) k-mer matching: a fresh table per step, one scan per sample (column s of the count table)
        table.clear()
        for smp in range(S):
            if timed:
                t_kmc[smp].start()
            scan.run(bloom, table, smp, records.data_ptr(), 0, R, d_hits.data_ptr())
            if timed:
                t_kmc[smp].stop()
        # (2) Gibbs: the whole default schedule for every group of the batch
        if timed:
            t_gibbs.start()
        gibbs.run()
        if timed:
            t_gibbs.stop()
        # (3) the launch's results: one rank -> host in the product's layout; several ranks -> packed into one word string on the device
        # (bt_gibbs_result_words), gathered from there to rank 0 (RCCL), which copies the gathered string to pinned host memory once — what
        # the executable's ranks do (host/main.cpp: DeviceWords + gatherResults); no rank's results visit its own host
        if timed:
            ctx.sync()   # (the launch is asynchronous: wait for it here so that the fetch is timed by itself; results() would wait anyway)
        tf = time.perf_counter()
        res = None
        if world == 1:
            res = gibbs.results()
        else:
            d_mine, n_mine = gibbs.result_words()
        fetch_s = time.perf_counter() - tf
        gather_s = 0.0
        if world > 1:
            tg = time.perf_counter()
            n_all = torch.zeros(world, dtype=torch.int64, device=dev)
            n_all[rank] = n_mine
            both_sync()
            comm.allreduce(n_all.data_ptr(), world)
            both_sync()
            total = int(n_all.sum().item())
            if state["d_all"] is None or state["d_all"].numel() < (total if rank == 0 else 2):
                state["d_all"] = torch.zeros(total if rank == 0 else 2, dtype=torch.int32, device=dev)
            both_sync()
            comm.gather_words(d_mine, n_mine, state["d_all"].data_ptr(), state["d_all"].numel())
            ctx.sync()
            if rank == 0:   # rank 0 holds every rank's results on the host, as the executable's rank 0 does (pinned staging, allocated once)
                if state.get("h_all") is None or state["h_all"].numel() < total:
                    state["h_all"] = torch.empty(int(total * 1.25), dtype=torch.int32, pin_memory=True)
                state["h_all"][:total].copy_(state["d_all"][:total], non_blocking=True)
            both_sync()
            gather_s = time.perf_counter() - tg
            state["gather_words"] = (n_mine, total)
            if strong:
                lib.check(lib.bt_gibbs_posterior_summary(gibbs.h, d_summary.data_ptr()))
                comm.gather_words(d_summary.data_ptr(), C * S * 2, d_gathered.data_ptr(), d_gathered.numel())
        state["res"] = res
        if timed:
            return [t.elapsed_ms() for t in t_kmc], t_gibbs.elapsed_ms(), fetch_s * 1e3, gather_s * 1e3
        return None

    def barrier():
        if world > 1:
            comm.barrier()
        both_sync()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    kmc_ms, gibbs_ms, fetch_ms, gather_ms = [], [], [], []
    for i in range(args.steps):
        a, b, f_ms, g_ms = step(args.warmup + i, True)
        kmc_ms.append(a)
        gibbs_ms.append(b)
        fetch_ms.append(f_ms)
        gather_ms.append(g_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gw = state.get("gather_words", (0, 0))
        props = torch.cuda.get_device_properties(dev)
        # what identifies the GPU beyond its index in this process (ranks that are each shown ONE device all call it cuda:0): uuid or PCI bus id when torch has them
        ident = None
        for attr in ("uuid", "pci_bus_id"):
            v = getattr(props, attr, None)
            if v not in (None, "", 0):
                ident = f"{attr} {v}"
                break
        if ident is None:
            try:
                ident = "pci " + torch.cuda.get_device_properties(dev).name + " / " + str(torch.cuda.mem_get_info(dev)[1]) + " B / HIP_VISIBLE_DEVICES=" + os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "all"))
            except Exception:
                ident = "unknown"
        mine = [C, float(np.mean(gibbs_ms)), float(np.mean([sum(x) for x in kmc_ms])), float(np.mean(fetch_ms)), float(np.mean(gather_ms)), int(gw[0]) * 4, int(gw[1]) * 4,
                int(dev.index), "%s, %d CUs, %s" % (props.name, props.multi_processor_count, ident), os.getpid()]
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        per_rank = [{"rank": r, "clusters": int(x[0]), "gibbs_launch_ms": x[1], "kmc_scans_ms": x[2], "result_pack_ms": x[3], "gather_ms": x[4],
                     "result_string_bytes": int(x[5]), "gathered_bytes_all_ranks": int(x[6]), "device": x[7], "device_name": x[8], "pid": x[9]} for r, x in enumerate(rows)]
        # two ranks on ONE GPU would halve each other: fatal — but only on positive evidence (the same uuid / PCI id twice), never because the identity is unknown
        known = [x[8] for x in rows if ("uuid " in x[8] or "pci_bus_id " in x[8])]
        if len(known) == world and len(set(known)) != world:
            raise RuntimeError("bench: two ranks of the job ran on the same GPU: %s" % known)
    hits = int(d_hits.item())
    st = table.status()
    if st["overflowed"]:
        raise RuntimeError("bench: the count table overflowed")

    # ------------------------------------------------------------------ strong scaling self-check: gathered summaries == the unsharded run
    verified = None
    if strong:
        if rank == 0:   # the assignment is deterministic: rank 0 knows every rank's clusters
            whole = torch.zeros(C_total * S * 2, dtype=torch.int32, device=dev).view(C_total, S * 2)
            parts = shard.assign_groups(shard.group_cost(unit), world) if verify else None
            at = 0
            for r in range(world):
                n = c_all[r]
                if verify:
                    ids_r = torch.from_numpy(np.ascontiguousarray(shard.cluster_ids_of(unit, parts[r])).astype(np.int64)).to(dev)
                    whole[ids_r] = d_gathered[at * S * 2: (at + n) * S * 2].view(n, S * 2)
                at += n
            if verify:
                gibbs.close()
                g_all = lib.Gibbs(ctx, unit, lut_g, lut_n, seed=42)
                for _ in range(args.warmup + args.steps):   # (a further bt_gibbs_run continues every group's chains: as many schedules as the shards have run)
                    g_all.run()
                ref_summary = torch.zeros(C_total * S * 2, dtype=torch.int32, device=dev)
                both_sync()
                lib.check(lib.bt_gibbs_posterior_summary(g_all.h, ref_summary.data_ptr()))
                both_sync()
                verified = bool(torch.equal(ref_summary.view(C_total, S * 2), whole))
                g_all.close()
                if not verified:
                    raise RuntimeError("bench: the gathered summaries of the sharded run differ from the unsharded run")

    # ------------------------------------------------------------------ placement independence on a sample (rank 0): the first groups of the batch as a batch
    # of their own (other tiles, other launch classes, other lanes) give the same posterior summaries — what makes a sharded unit equal the unsharded one
    subset_ok = None
    if rank == 0 and not strong and not args.no_verify:
        n_sub = min(4096, G)
        sub = shard.take_groups(flat, np.arange(n_sub))
        lib.check(lib.bt_gibbs_posterior_summary(gibbs.h, d_summary.data_ptr()))
        both_sync()
        g_sub = lib.Gibbs(ctx, sub, lut_g, lut_n, seed=42)
        for _ in range(args.warmup + args.steps):   # (a further bt_gibbs_run continues every group's chains: as many schedules as the batch has run)
            g_sub.run()
        d_sub = torch.zeros(sub["num_clusters"] * S * 2, dtype=torch.int32, device=dev)
        both_sync()
        lib.check(lib.bt_gibbs_posterior_summary(g_sub.h, d_sub.data_ptr()))
        both_sync()
        subset_ok = bool(torch.equal(d_sub, d_summary[: d_sub.numel()]))
        g_sub.close()
        del d_sub
        if not subset_ok:
            raise RuntimeError("bench: a group's samples depend on the batch it is launched in")

    # ------------------------------------------------------------------ the host step that follows a launch (rank 0, outside the timed region):
    # VariantClusterGenotyper::getGenotypes + the formatted output columns of every variant of every cluster (bayesTyper genotype -p: host threads)
    results_rec = None
    if rank == 0 and world == 1 and state["res"] is not None:
        from bayestyper_amd.host import genotypes as hgt

        mf = hgt.min_fraction_observed_kmers([15.0] * S)
        from bayestyper_amd import hostinfo as _hi

        cores_ = _hi.baseline_threads(_hi.host_facts())   # (host threads as `bayesTyper genotype -p` would be given on this box: two per core of the container's quota)
        tg_ = time.perf_counter()
        nbytes = hgt.batch_output_columns(flat, state["res"], mf, cores_)
        collect_all = time.perf_counter() - tg_
        n1 = min(G, 20_000)
        sub1 = shard.take_groups(flat, np.arange(n1))
        c1 = sub1["num_clusters"]
        r1 = {"dip_off": state["res"]["dip_off"][: c1 + 1], "h1": state["res"]["h1"], "h2": state["res"]["h2"], "freq": state["res"]["freq"],
              "cell_off": state["res"]["cell_off"][: c1 + 1], "stats": state["res"]["stats"]}
        tg_ = time.perf_counter()
        hgt.batch_output_columns(sub1, r1, mf, 1)
        collect_one = time.perf_counter() - tg_
        results_rec = {"clusters": C, "result_fetch_ms": float(np.mean(fetch_ms)),
                       "result_bytes": int(sum(state["res"][k].nbytes for k in ("dip_off", "h1", "h2", "freq", "cell_off", "stats"))),
                       "get_genotypes_s": collect_all, "get_genotypes_threads": cores_, "get_genotypes_clusters_per_sec": C / collect_all,
                       "get_genotypes_one_thread_clusters_per_sec": c1 / collect_one, "formatted_bytes": nbytes,
                       "note": "result_fetch_ms is inside the timed step (bt_gibbs_result_sizes + bt_gibbs_result_fetch: count, pack on the device, one copy per array); "
                               "get_genotypes_s = bthost::getGenotypes + the formatted VCF columns of every variant of the batch on the host threads `bayesTyper genotype -p` uses"}
    state["res"] = None

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1 only): the oracle on host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle

        orc = _oracle.load_oracle()
        from bayestyper_amd import hostinfo, shard

        # what the box gives this process: the cgroup quota, not os.cpu_count(), is the number of cores a CPU leg can use (round 5 reported "256 cores" on a
        # box whose container had a 16-core quota and ran 256 threads on it: 9x one thread where 32 threads give 16.7x, tools/cpu_scaling.py)
        facts = hostinfo.host_facts()
        cores = hostinfo.baseline_threads(facts)   # threads of the all-cores legs
        og_lut_g, og_lut_n = _oracle.build_luts(orc, S)

        def cpu_run(cf, threads):
            og = _oracle.OrcGibbs(orc, cf, og_lut_g, og_lut_n, seed=42)
            tc = time.perf_counter()
            og.run(threads)
            dt = time.perf_counter() - tc
            og.close()
            return dt

        def cpu_leg(n_groups, threads, seed):
            cf = synth.make_mixture(n_groups, S, seed=seed)
            return cf, cpu_run(cf, threads)

        # all cores: a short probe sizes the sample for ~args.cpu_seconds of work (the tail of a small sample — a few long nested groups per
        # thread — would otherwise dominate); one core: a sample of its own, a few seconds
        probe, probe_s = cpu_leg(max(2048, 16 * cores), cores, 998)
        rate = probe["num_clusters"] * sweeps_per_group / probe_s
        n_cpu = int(max(4096, min(args.groups, args.cpu_seconds * rate / sweeps_per_group * probe["num_groups"] / probe["num_clusters"])))
        cflat, cpu_s = cpu_leg(n_cpu, cores, 999)
        cpu_sweeps = cflat["num_clusters"] * sweeps_per_group
        # one core: a down-scaled copy of the SAME batch — every k-th group of the all-cores sample (its groups are ordered by class, so the class
        # mix is the all-cores leg's; round 4 generated a small mixture of its own, whose rounding gave it another mix)
        stride = max(1, int(round(cflat["num_groups"] / max(256, n_cpu / cores * 0.4))))
        one_flat = shard.take_groups(cflat, np.arange(0, cflat["num_groups"], stride))
        one_s = cpu_run(one_flat, 1)
        # thread scaling on every k-th group of that sample (k chosen per thread count so that a point takes a few seconds)
        scaling = {"1": {"groups": int(one_flat["num_groups"]), "s": one_s, "value": one_flat["num_clusters"] * sweeps_per_group / one_s},
                   str(cores): {"groups": int(cflat["num_groups"]), "s": cpu_s, "value": cpu_sweeps / cpu_s}}
        for nthr in (8, 32, 64, facts["affinity"]):
            if str(nthr) in scaling or nthr > facts["affinity"]:
                continue
            per_thread_s = 3.0
            want = max(256, int(rate * min(nthr, facts["effective_cores"]) / max(1.0, min(cores, facts["effective_cores"])) * per_thread_s / sweeps_per_group))
            k = max(1, cflat["num_groups"] // want)
            sf = shard.take_groups(cflat, np.arange(0, cflat["num_groups"], k)) if k > 1 else cflat
            ds = cpu_run(sf, nthr)
            scaling[str(nthr)] = {"groups": int(sf["num_groups"]), "s": ds, "value": sf["num_clusters"] * sweeps_per_group / ds}
        for v_ in scaling.values():
            v_["over_one_thread"] = v_["value"] / scaling["1"]["value"]
        cpu_one = one_flat["num_clusters"] * sweeps_per_group / one_s
        # k-mer matching: decode -> Bloom lookup for a bounded slice of an equivalent database — the reference's shape (ONE producer thread
        # decoding the KMC records, KmerCounter.cpp:469-505) and the best-effort shape (every core decodes its own record range)
        kcpu = kcpu_par = None
        try:
            import tempfile

            rng = np.random.default_rng(1)
            n_db = 400_000
            km = _oracle.random_kmers(rng, n_db, K).reshape(n_db, K)
            km = np.unique(km, axis=0)
            with tempfile.TemporaryDirectory() as td:
                pref = os.path.join(td, "db")
                orc.kmc_write(pref, np.ascontiguousarray(km).reshape(-1), np.ones(len(km), np.uint32), K, KMC_P, 1)
                db = _oracle.OrcKmc(orc, pref)
                ob = _oracle.OrcBloom(orc, 2_000_000, 1e-4, K, threaded=True)
                ob.insert(np.ascontiguousarray(km[:: 50]).reshape(-1))
                tk = time.perf_counter()
                orc.l.orc_match_only(ob.h, db.h, 0, db.total)
                kcpu = db.total / (time.perf_counter() - tk)
                import threading

                nthr = min(cores, 64)
                per = db.total // nthr
                thr = [threading.Thread(target=orc.l.orc_match_only, args=(ob.h, db.h, i * per, per)) for i in range(nthr)]   # (ctypes releases the GIL)
                tk = time.perf_counter()
                for t_ in thr:
                    t_.start()
                for t_ in thr:
                    t_.join()
                kcpu_par = per * nthr / (time.perf_counter() - tk)
                ob.close()
                db.close()
        except Exception:   # the k-mer CPU leg is informative only
            pass
        cpu = {"value": cpu_sweeps / cpu_s, "unit": "cluster-sweeps/s", "cores": facts["effective_cores"], "threads": cores, "kind": "port",
               "host": facts, "scaling": dict(sorted(scaling.items(), key=lambda kv: int(kv[0]))),
               "sample": f"{cflat['num_groups']} groups of the same shape mixture ({cflat['mixture']}), S={S}, full 20x350 schedule, "
                         f"{cpu_s:.1f} s on {cores} threads (oracle/oracle_gibbs.cpp; threads pull groups from a shared queue, largest first, as InferenceEngine.cpp:335-382)",
               "one_core": {"value": cpu_one, "sample": f"every {stride}th group of the all-cores sample ({one_flat['num_groups']} groups, the same class mix), {one_s:.1f} s on 1 thread"},
               "allcores_over_one_core": cpu_sweeps / cpu_s / cpu_one,
               # what the port is measured against: the reference's own objects, one thread, in the build container at survey time (SURVEY.md section 6:
               # 2.8e5 cluster-sweeps/s for a biallelic SNV cluster at S = 10, 3.5e4 for a 4-SNV cluster with ten candidates) against the oracle's 2.0e5 / 3.4e4
               # there (docs/HISTORY.md section 2), 0.71 / 0.97 of it; since round 6 an estimated 0.94 / 1.06 (below)
               "vs_reference_probe": {"A_S10": 0.95, "B_S10": 1.1, "A_S10_before_round6": 0.71, "B_S10_before_round6": 0.97,
                                      "note": "oracle / reference single-thread cluster-sweeps/s.  The survey-time probe (build container) gave 0.71 / 0.97; round 6 found the port's "
                                              "differences from the reference's build on this path — sampleDiplotype's sampler and candidate list and sampleDiplotypes' list of non-zero "
                                              "haplotypes were not reserved (DiscreteSampler.cpp:39, VariantClusterGenotyper.cpp:672-673,709-713) and the port was built -O2 where the "
                                              "reference's CMakeLists uses -O3 — and fixed them: 2.60e5 -> 3.4 - 3.6e5 (A) and 5.65e4 -> 6.3 - 6.9e4 (B) cluster-sweeps/s back to back in the "
                                              "build container (a shared machine: +- 5 %), i.e. x 1.3 - 1.4 / x 1.1 - 1.2 on the probe's ratios (the reference itself cannot be re-run here).  The port is not the reference; a speed-up over it is a reported "
                                              "baseline, not a target"},
               "kmer_matches_per_sec_single_producer": kcpu, "kmer_matches_per_sec_parallel_decode": kcpu_par,
               "kmer_match_note": "400 000-record database: one thread decoding and probing (the reference's single producer is the serial stage) / "
                                  "every core its own record range (up to 64 threads)"}

    # ------------------------------------------------------------------ graph stages (rank 0, outside the timed region): best-path search,
    # path k-mer enumeration, classification, haplotype candidates on synthetic SNV/indel clusters — reported beside the headline metric
    paths = None
    if rank == 0 and world == 1 and not args.no_paths:   # like the CPU baseline: single-GPU runs only (other ranks would idle at the teardown)
        from bayestyper_amd import synth_graphs

        prng = np.random.default_rng(11)
        n_cl = args.path_clusters
        gs = [synth_graphs.random_cluster(prng, K, int(prng.integers(1, 4)), int(prng.integers(2, 5)), kinds=("snv", "snv", "snv", "ins", "del")) for _ in range(n_cl)]
        fg = synth_graphs.flatten(gs)

        def timed(fn):
            both_sync()
            t = time.perf_counter()
            r = fn()
            ctx.sync()
            return r, time.perf_counter() - t

        gp, t_create = timed(lambda: lib.Paths(ctx, fg, K))
        W = gp.num_windows
        pb = lib.Bloom.create(ctx, W + 1_000_000, 1e-4, K, threaded=True)
        _, t_bloom = timed(lambda: gp.count_kmers(pb))
        ptab = lib.Table(ctx, max(W // 2, 1024), S, K)
        mgb = lib.Bloom.create(ctx, 1000, 1e-4, K, threaded=False)
        _, t_cls = timed(lambda: gp.classify(ptab, mgb))
        cand, t_cand = timed(lambda: gp.candidates(ptab))
        for g_ in gs:
            g_.paths = None
        fg2 = synth_graphs.flatten(gs)
        gf, _ = timed(lambda: lib.FindPaths(ctx, fg2, K, 32, 1))
        _, t_find = timed(lambda: gf.sample(bloom, np.arange(n_cl, dtype=np.uint32) + 7))   # the bench's path Bloom stands in for a sample filter
        # the cluster stage's multigroup pass (one cluster per group) in the reference's single-thread order, into a fresh filter
        pb2 = lib.Bloom.create(ctx, W + 1_000_000, 1e-4, K, threaded=True)
        mgt = lib.Table(ctx, max(W // 8, 1024), 1, K)
        _, t_mg = timed(lambda: gp.count_multigroup(np.arange(n_cl, dtype=np.uint32), pb2, mgt))
        pb2.close(), mgt.close()
        paths = {"clusters": n_cl, "kmer_windows": int(W), "rows": int(cand["kmer_off"][-1]),
                 "enumerate_windows_per_sec": W / t_create, "bloom_insert_windows_per_sec": W / t_bloom, "classify_windows_per_sec": W / t_cls,
                 "candidates_windows_per_sec": W / t_cand, "find_sample_paths_clusters_per_sec": n_cl / t_find, "multigroup_windows_per_sec": W / t_mg,
                 "note": "host wall-clock around each C-ABI call (includes the host-side assembly of the candidates stage)"}
        for x in (gp, pb, ptab, mgb, gf):
            x.close()

    # ------------------------------------------------------------------ PCIe-inclusive k-mer matching (rank 0, N=1, outside the timed region):
    # the same scan with the records in HOST memory, streamed by bt_kmc_scan_run_host (pinned staging, copy stream overlapping the scan)
    pcie = None
    if rank == 0 and world == 1 and not args.no_pcie:
        import ctypes as ct

        n_host = min(R, args.pcie_records)
        h_rec = records[: n_host * REC + 16].cpu().numpy()
        hh = ct.c_uint64()
        fn = lib._lib.bt_kmc_scan_run_host
        fn.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_void_p, ct.c_uint32, ct.c_void_p, ct.c_uint64, ct.c_uint64, ct.c_uint64, ct.c_void_p]
        best = None
        for _ in range(2):   # the first pass also faults the pinned staging buffers in
            tp = time.perf_counter()
            lib.check(fn(scan.h, bloom.h, table.h, 0, h_rec.ctypes.data, 0, n_host, 0, ct.byref(hh)))
            dt = time.perf_counter() - tp
            best = dt if best is None else min(best, dt)
        pcie = {"records": int(n_host), "records_per_sec": n_host / best, "host_gbytes_per_sec": n_host * REC / best / 1e9,
                "note": "bt_kmc_scan_run_host from pageable host memory (host copy -> pinned staging -> H2D -> scan, overlapped); never part of `value`"}
        del h_rec

    # ------------------------------------------------------------------ sub-records (rank 0, N=1, outside the timed region)
    extra = {}
    if rank == 0 and world == 1 and not args.no_extra:
        # (1) the north star's ten-sample mixture (BASELINE configs[3] per GPU): 100 000 groups, full default schedule, beside the oracle on all cores
        gibbs.close()
        S10 = 10
        f10 = synth.make_mixture(100_000, S10, seed=1010)
        lg10, ln10 = count_model.build_luts(S10, mean=15.0, var=30.0, noise_rate=0.05)
        f10x = synth.concat([f10] * 4)   # the throughput regime: 4 x the batch, every copy with group indices (hence seeds, hence chains) of its own
        f10x["group_index"] = np.arange(f10x["num_groups"], dtype=np.uint32)
        f10x["mixture"] = {k: 4 * v for k, v in f10["mixture"].items()}
        g10 = lib.Gibbs(ctx, f10x, lg10, ln10, seed=42)
        t10 = lib.Timer(ctx)
        ms10 = []
        for _ in range(2):
            t10.start()
            g10.run()
            t10.stop()
            ms10.append(t10.elapsed_ms())
        g10_bytes = g10.device_bytes()
        g10.close()
        rec10 = {"workload": "BASELINE configs[3] shape on one GPU: %d groups of the mixture (%s; 4 copies of 100 352 generated groups, each copy with its own group "
                             "indices, i.e. its own chains), S=10, 20 chains x (100+250) sweeps" % (f10x["num_groups"], f10x["mixture"]),
                 "groups": int(f10x["num_groups"]), "ms_per_schedule": min(ms10), "cluster_sweeps_per_sec": f10x["num_clusters"] * sweeps_per_group / (min(ms10) * 1e-3),
                 "device_bytes": g10_bytes}
        del f10x
        if not args.no_cpu_baseline:
            c10 = synth.make_mixture(max(2048, 64 * cores), S10, seed=1011)
            og = _oracle.OrcGibbs(orc, c10, *_oracle.build_luts(orc, S10), seed=42)
            tc = time.perf_counter()
            og.run(cores)
            dt = time.perf_counter() - tc
            og.close()
            rec10["cpu_allcores_cluster_sweeps_per_sec"] = c10["num_clusters"] * sweeps_per_group / dt
            rec10["cpu_sample"] = f"{c10['num_groups']} groups, {dt:.1f} s on {cores} threads ({facts['effective_cores']:.0f} effective cores)"
            rec10["gpu_over_cpu_allcores"] = rec10["cluster_sweeps_per_sec"] / rec10["cpu_allcores_cluster_sweeps_per_sec"]
        issue10, issue10_src = committed_issue_profile(S10, int(rec10["groups"]))
        if issue10:
            rec10["issue_frac"] = issue10["gibbs"]["valu_issue_cycles_per_schedule"] / (1024 * 2.4e9 * rec10["ms_per_schedule"] * 1e-3)
            rec10["valu_insts_per_cluster_sweep"] = issue10["gibbs"]["valu_insts_per_cluster_sweep"]
            rec10["resident_waves_per_simd"] = issue10["gibbs"]["wave_quad_cycles"] * 4 / (2.4e9 * rec10["ms_per_schedule"] * 1e-3) / 1024
            rec10["issue_source"] = issue10_src
        extra["samples10"] = rec10
        # (1b) BASELINE configs[4]: --noise-genotyping (estimateNoiseAndGenotypes) at 30 samples through the C++ InferenceEngine the executable
        # ships, beside the default mode on the same batch.  The drivers iterate on the host (bt_gibbs_noise_iteration: one synchronisation per iteration, the
        # rates drawn by libstdc++'s own gamma distribution — bit for bit the reference's; BT_NOISE_ON_DEVICE=1 runs a chain without host round trips).
        from bayestyper_amd.host.inference_engine import InferenceEngine

        def noise_pair(flat, S_, chains, label):
            """default mode and --noise-genotyping on the same batch through the engine: wall-clock of the whole driver call"""
            flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
            cd = count_model.CountDistribution(S_, prior=(1.0, 0.01), seed=42)
            for s_ in range(S_):
                cd.set_genomic(s_, 15.0, 30.0)
            eng = InferenceEngine(ctx, 42, chains=chains)
            tn = time.perf_counter()
            r = eng.estimate_genotypes(flat, cd)
            t_default = time.perf_counter() - tn
            r.close()
            tn = time.perf_counter()
            r, _rows = eng.estimate_noise_and_genotypes(flat, cd)
            t_noise = time.perf_counter() - tn
            r.close()
            sw = flat["num_clusters"] * chains * 350
            return cd, {"workload": label, "default_mode_cluster_sweeps_per_sec": sw / t_default, "default_mode_s": t_default,
                        "noise_genotyping_cluster_sweeps_per_sec": sw / t_noise, "noise_genotyping_s": t_noise, "noise_over_default_time": t_noise / t_default,
                        "iterations_per_sec": chains * 350 / t_noise}

        S30 = 30
        f30 = synth.make_mixture(2_000, S30, seed=3030)
        chains30 = 1   # (one of the twenty chains: the rate per sweep is what is reported; a full schedule of this leg takes minutes)
        cd30, rec30 = noise_pair(f30, S30, chains30, "BASELINE configs[4] shape on one GPU: 2 000 groups of the mixture (%s), S=30 (up to 256 haplotype candidates), %d chain x "
                                 "(100+250) iterations, wall-clock of the whole driver call (sampler construction and result fetch included)" % (f30["mixture"], chains30))
        rec30["note"] = ("in this mode every genotyper's caches are cleared every iteration (InferenceEngine.cpp:92), so every sweep recomputes its per-(sample, diplotype) "
                         "sums over the k-mer subset, and an iteration lasts as long as its slowest group (a 256-candidate cluster at 30 samples) — on the CPU as well: "
                         "cpu_allcores_* is the oracle's estimateNoiseAndGenotypes on the same batch")
        if not args.no_cpu_baseline:
            it_cpu = (2, 3)
            og = _oracle.OrcGibbs(orc, f30, *cd30.tables(), noise_seeding=1, seed=42, chains=1, burn=it_cpu[0], iters=it_cpu[1])
            tc = time.perf_counter()
            og.estimate_noise_and_genotypes(threads=cores)
            dt = time.perf_counter() - tc
            og.close()
            rec30["cpu_allcores_cluster_sweeps_per_sec"] = f30["num_clusters"] * sum(it_cpu) / dt
            rec30["cpu_sample"] = f"the same batch, {sum(it_cpu)} iterations, {dt:.1f} s on {cores} threads (groups of an iteration dealt to the threads)"
            rec30["gpu_over_cpu_allcores"] = rec30["noise_genotyping_cluster_sweeps_per_sec"] / rec30["cpu_allcores_cluster_sweeps_per_sec"]
        cd30.close()
        extra["noise_genotyping"] = rec30
        # ... and in the throughput regime: the ten-sample batch above (100 000 groups fill the GPU, so an iteration is no longer one group's latency)
        cd10, rec10n = noise_pair(f10, S10, 1, "the samples10 batch (100 000 groups, S=10) in --noise-genotyping mode beside the default mode, 1 chain x (100+250) iterations")
        cd10.close()
        extra["noise_genotyping_samples10"] = rec10n
        # (1c) the noise driver of the default mode (estimateNoise, InferenceEngine.cpp:135-276: 20 chains x 350 iterations on a random 100 000-variant subset of the
        # unit's single-cluster groups, every iteration = sweep + noise counts + one exact host draw per sample): iterations per second at chr20 size.  A chain is
        # ONE resident launch (gibbs_chain_kernel + the pinned mailbox); the next chain's sampler is built from the device-resident unit by a helper thread.
        def estimate_noise_leg(S_, chains_):
            flat_ = synth.make_mixture(180_000, S_, seed=2020 + S_, fractions={"A": 0.92, "B": 0.08})
            flat_["group_index"] = np.arange(flat_["num_groups"], dtype=np.uint32)
            cd_ = count_model.CountDistribution(S_, prior=(1.0, 0.01), seed=42)
            for s_ in range(S_):
                cd_.set_genomic(s_, 15.0, 30.0)
            eng_ = InferenceEngine(ctx, 42, chains=chains_)
            tn = time.perf_counter()
            eng_.estimate_noise(cd_, flat_)
            dt_ = time.perf_counter() - tn
            cd_.close()
            return {"groups": int(flat_["num_groups"]), "mixture": flat_["mixture"], "chains": chains_, "iterations": chains_ * 350, "seconds": dt_,
                    "iterations_per_sec": chains_ * 350 / dt_, "ms_per_iteration": dt_ / (chains_ * 350) * 1e3}

        extra["estimate_noise"] = {"workload": "estimateNoise on a chr20-sized unit (180 000 groups: two-haplotype and ten-candidate single clusters; a random 100 000-variant subset "
                                               "per chain), wall-clock of the whole driver call: unit upload, sampler constructions, 350 iterations per chain",
                                   "S1": estimate_noise_leg(1, 20), "S3": estimate_noise_leg(3, 20), "S10": estimate_noise_leg(10, 6)}
        # (2) k-mer matching against sub-filters of BASELINE configs[3] size: a ThreadedKmerBloom of 10^9 path k-mers (~37 KB per sub-filter,
        # 2.4 GB) — the LDS-staged probe at the other end of its range
        big = lib.Bloom.create(ctx, 1_000_000_000, 1e-4, K, threaded=True)
        for a in range(0, 400_000_000, 100_000_000):   # fill a part of it (bit density decides the probe depth of non-members)
            fill = torch.randint(-(2 ** 62), 2 ** 62, (100_000_000, 2), dtype=torch.int64, device=dev, generator=gen)
            fill[:, 1] &= (1 << 46) - 1
            both_sync()
            lib.check(lib.bt_bloom_insert_batch(big.h, fill.data_ptr(), fill.shape[0]))
            both_sync()
            del fill
        n_big = min(R, 200_000_000)
        table.clear()
        tb = lib.Timer(ctx)
        msb = []
        for _ in range(3):
            tb.start()
            scan.run(big, table, 0, records.data_ptr(), 0, n_big, d_hits.data_ptr())
            tb.stop()
            msb.append(tb.elapsed_ms())
        extra["kmer_match_c4_subfilters"] = {"records": n_big, "path_filter": "ThreadedKmerBloom(10^9 k-mers, fpr 1e-4): %d bytes per sub-filter, 40 %% filled" % (big.info()["num_bits"] // 8),
                                             "ms": min(msb), "records_per_sec": n_big / (min(msb) * 1e-3)}
        big.close()

    if rank == 0:
        ms_per_step = elapsed * 1000.0 / args.steps
        total_cluster_sweeps = cluster_sweeps_per_step * args.steps          # whole job (all ranks)
        gibbs_avg_ms = float(np.mean(gibbs_ms))
        kmc = np.asarray(kmc_ms, np.float64)                                 # [steps][S] launch times
        kmc_avg_ms = float(kmc.mean())                                       # average launch of kmc_scan_kernel in the timed region
        kmc_step_ms = float(kmc.sum(axis=1).mean())                          # the S scans of a step
        gibbs_bytes = synth.algorithmic_bytes_per_chain(flat) * gibbs_chains                   # one launch = all chains of all clusters
        gibbs_gbs = gibbs_bytes / (gibbs_avg_ms * 1e-3) / 1e9
        kmc_gbs = R * KMER_MATCH_BYTES_PER_RECORD / (kmc_avg_ms * 1e-3) / 1e9
        gibbs_traffic, gibbs_traffic_src = committed_traffic("gibbs_bytes_per_schedule")
        kmc_traffic, kmc_traffic_src = committed_traffic("kmc_bytes_per_scan")
        if (args.groups, S, args.records) != (600_000, 3, 1_000_000_000) or world != 1:   # the committed passes are of the default command
            gibbs_traffic = kmc_traffic = gibbs_traffic_src = kmc_traffic_src = None
        issue, issue_src = committed_issue_profile(S, G)
        if world != 1:
            issue = issue_src = None
        SIMDS, CLOCK = 1024, 2.4e9
        shape_note = ("BASELINE configs[2] WGS trio" if S == 3 else "BASELINE configs[3] 10-sample mixture" if S == 10 else "mixture") + \
            ": one launch-sized slice of the unit, %d groups/GPU (%s; heterogeneous structures), S=%d, 20 chains x (100+250) sweeps; k-mer matching: %d scans/step of a " \
            "%d-record KMC stream (13 B, k=55, p=7) into an emptied count table, %d path k-mers, hit rate %.3f, ThreadedKmerBloom fpr 1e-4" % (
                G, flat["mixture"], S, S, R, args.path_kmers, args.hit_rate)
        out = {
            "metric": "variant-cluster Gibbs iterations/sec + k-mer matches/sec at 1/2/4/8 MI355X",
            "value": total_cluster_sweeps / elapsed,
            "unit": "variant-cluster Gibbs iterations (cluster-sweeps)/s",
            "kmer_matches_per_sec": S * R * world / (kmc_step_ms * 1e-3),
            "kmer_hits_per_sec": hits / (args.steps + args.warmup) * world / (kmc_step_ms * 1e-3),
            "gibbs_kernel_cluster_sweeps_per_sec": cluster_sweeps_per_step / (gibbs_avg_ms * 1e-3),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64 log-probabilities over u8 k-mer counts (Gibbs); u64/u8 integer (k-mer matching)",
            "data": "synthetic",
            "config": {"workload": shape_note, "groups_per_gpu": G, "clusters_per_gpu": C, "clusters_total": C_total, "samples": S, "kmc_records_per_gpu_per_sample": R,
                       "sharding": ("one batch sharded over the ranks (LPT on a cost proxy); " if strong else "one unit of N launch-sized blocks, block r on rank r (unit-wide group indices); ") +
                                   "KMC streams per rank; gather of every rank's results (diplotype sampling frequencies + allele k-mer statistics) to rank 0",
                       "sharded_equals_unsharded": verified if verified is not None else precheck_ok,
                       "sharded_precheck": precheck_ok, "subset_equals_batch": subset_ok, "per_rank": per_rank,
                       "result_fetch_ms": float(np.mean(fetch_ms)), "gather_ms": float(np.mean(gather_ms)) if world > 1 else None},
            "results": results_rec,
            "roofline": {"kernel": "gibbs_hot_kernel + gibbs_simple_kernel (the concurrent launches of one schedule; gibbs_kernel for tiles that do not keep every vertex in LDS)", "bound": "hbm", "achieved": gibbs_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gibbs_gbs / HBM_PEAK_GBS, "traffic": gibbs_traffic, "traffic_source": gibbs_traffic_src, "algorithmic_bytes": gibbs_bytes, "avg_launch_ms": gibbs_avg_ms,
                         "limiter": "VALU issue + latency of a sequential sampler (no dense contraction: MFMA busy cycles 0); the HBM fraction is tiny by construction",
                         "issue_frac": issue["gibbs"]["valu_issue_cycles_per_schedule"] / (SIMDS * CLOCK * gibbs_avg_ms * 1e-3) if issue else None,
                         "valu_insts_per_cluster_sweep": issue["gibbs"]["valu_insts_per_cluster_sweep"] if issue else None,
                         # wavefront-cycles of the schedule (a count: the same whether its launches ran one after the other, as under the counter passes, or
                         # together) over the duration of the concurrent launches measured here
                         "resident_waves_per_simd": issue["gibbs"]["wave_quad_cycles"] * 4 / (CLOCK * gibbs_avg_ms * 1e-3) / SIMDS if issue else None,
                         "waves_per_simd_by_registers": {"gibbs_simple_kernel": 3, "gibbs_hot_kernel": 2},
                         "issue_source": issue_src,
                         "issue_note": "issue_frac = (VALU instructions of one schedule, priced 2 cycles for 32-bit, 4 for f64 add/mul/fma and 64-bit integer, 16 / 8 for "
                                       "f64 / f32 transcendental: MI355X_MICROARCH.md) / (1024 SIMDs x 2.4 GHz x the launch time measured here); counts from single-schedule "
                                       "rocprofv3 --pmc passes of the same batch (tools/sq_counters.sh, tools/issue_profile.py), used only when made from the same sources",
                         "note": "latency/issue-bound sequential sampler: the HBM floor (inputs + state once per chain, SURVEY 8d) is tiny by construction; "
                                 "avg_launch_ms spans the sampling launches of one schedule; traffic = FETCH_SIZE + WRITE_SIZE of those launches from the committed PMC "
                                 "passes of this command (separate rocprofv3 --pmc runs, KiB -> bytes), filled in only when they were made from the same device sources "
                                 "(source_hash), else null"},
            "roofline_kmer_match": {"kernel": "one KMC scan = kmc_partition_kernel + kmc_probe_bucket_kernel + kmc_apply_kernel per 2^26-record chunk", "bound": "hbm", "achieved": kmc_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": kmc_gbs / HBM_PEAK_GBS, "traffic": kmc_traffic, "traffic_source": kmc_traffic_src, "avg_launch_ms": kmc_avg_ms, "launches_per_step": S,
                                    "insert_launch_ms": float(kmc[:, 0].mean()), "find_launch_ms": float(kmc[:, 1:].mean()) if S > 1 else None,
                                    "bytes_per_record": KMER_MATCH_BYTES_PER_RECORD, "bloom_hits_per_scan": hits // ((args.steps + args.warmup) * S), "table_keys": st["num_keys"]},
            "cpu_baseline": cpu,
            "graph_stages": paths,
            "kmer_match_from_host_memory": pcie,
            "gibbs_device_bytes": gibbs_device_bytes,
            "source_hash": source_hash(), "source_hash_gibbs": source_hash("gibbs"), "source_hash_kmc": source_hash("kmc"),
        }
        out.update(extra)
        if cpu:
            out["gpu_over_cpu_allcores"] = out["gibbs_kernel_cluster_sweeps_per_sec"] / cpu["value"]
        print(json.dumps(out))
    if world > 1:
        comm.barrier()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
