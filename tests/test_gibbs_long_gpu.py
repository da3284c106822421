"""GPU parity at the schedule bench.py times: the reference's default 20 chains x (100 burn-in + 250 collected) sweeps
(main.cpp:389-391) on the multi-variant (B), nested-SV (C) and many-candidate (D) shape classes, every group with its own
dimensions, at S = 3 (BASELINE configs[2]) and S = 10 (configs[3]) — EVERY sweep's diplotypes of every sample traced against the
oracle (InferenceEngine.cpp:278-333, VariantClusterHaplotypes.cpp:235-372), then the sampling frequencies and allele k-mer statistics.

The second half forces the state machines whose rare branches only long chains reach: with counts thinned to ~1.5x coverage the
posteriors are diffuse, so a sample changes its diplotype many times per chain (more runs than the 8-entry run log EV_CAP holds,
more distinct diplotypes than the 4 kept k-mer-stats caches KSC_WAYS, nested parents that flip under their children: the deferred
nested statistics A_PENDNEST and the version-gated nested info A_NVER).  The tests assert on the ORACLE's traces that those
conditions were actually reached."""
import numpy as np
import pytest

import _oracle
from test_gibbs_gpu import assert_noise_rows, assert_parity, run_both

pytestmark = pytest.mark.gpu
FULL = dict(chains=20, burn=100, iters=250)
EV_CAP, KSC_WAYS = 32, 4          # bt_gibbs_tile.hpp (mirrored here only to assert that the tests go past them)


def oracle_threads():
    import os
    return max(8, min(64, (os.cpu_count() or 8) // 2))


def run_full(gpu_ctx, oracle, flat, lut=None, trace_all=True, **kw):
    from bayestyper_amd import lib

    S = flat["S"]
    lut_g, lut_n = _oracle.build_luts(oracle, S) if lut is None else lut
    sweeps = kw["chains"] * (kw["burn"] + kw["iters"])
    og = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, **kw)
    gg = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, **kw)
    if trace_all:
        og.trace_enable(sweeps)
        gg.trace_enable(sweeps)
    gg.run()                      # asynchronous on the context's stream: the oracle runs meanwhile
    og.run(oracle_threads())
    gpu_ctx.sync()
    ro, rg = og.results(), gg.results()
    traces = []
    if trace_all:
        goff = flat["group_cluster_off"]
        tg = gg.trace()
        for g in range(flat["num_groups"]):
            to = og.trace(g, int(goff[g + 1] - goff[g]), sweeps)
            assert len(to) == sweeps
            if not np.array_equal(to, tg[g]):
                bad = int(np.argwhere((to != tg[g]).any(axis=(1, 2)))[0, 0])
                raise AssertionError(f"group {g}: diplotype trace diverges at sweep {bad} (chain {bad // (kw['burn'] + kw['iters'])})")
            traces.append(to)
    og.close()
    gg.close()
    exact = assert_parity(flat, ro, rg, kw["chains"] * kw["iters"])
    assert exact == flat["num_clusters"]
    return traces


def flip_statistics(traces, burn, iters, chains):
    """from per-group traces [sweep][vertex][S]: per (group, vertex, sample, chain) the number of diplotype runs among the collected
    sweeps and the number of distinct diplotypes -> (max runs, max distinct, number of root-vertex changes over all sweeps)"""
    per = burn + iters
    max_runs = max_distinct = root_changes = 0
    for t in traces:
        for c in range(chains):
            col = t[c * per + burn:(c + 1) * per]                       # collected sweeps of this chain
            runs = 1 + (col[1:] != col[:-1]).sum(axis=0)                 # [vertex][S]
            max_runs = max(max_runs, int(runs.max()))
            for v in range(col.shape[1]):
                for s in range(col.shape[2]):
                    max_distinct = max(max_distinct, len(np.unique(col[:, v, s])))
        root_changes += int((t[1:, 0] != t[:-1, 0]).sum())
    return max_runs, max_distinct, root_changes


@pytest.mark.parametrize("shape,n,S", [("B", 24, 3), ("C", 8, 3), ("D", 3, 3), ("B", 24, 10), ("C", 8, 10), ("D", 3, 10)])
def test_full_default_schedule_shapes_BCD(gpu_ctx, oracle, shape, n, S):
    from bayestyper_amd import synth

    flat = synth.make_hetero_batch(shape, n, S, seed=900 + 7 * S + ord(shape))
    if shape == "C":
        assert int(flat["multi_off"][-1]) > 0 and flat["num_clusters"] >= 2 * n     # nested groups with multicluster k-mers
    run_full(gpu_ctx, oracle, flat, seed=61, **FULL)


def thin(flat, rng, keep):
    """counts thinned binomially to `keep` of the coverage (multicluster rows keep ONE count per shared k-mer record)"""
    out = dict(flat)
    S = flat["S"]
    cnt = flat["kmer_counts"].reshape(-1, S).copy()
    new = rng.binomial(cnt, keep).astype(np.uint8)
    shared = flat["kmer_shared"]
    goff, koff = flat["group_cluster_off"], flat["kmer_off"]
    for g in range(flat["num_groups"]):
        first = {}
        for c in range(goff[g], goff[g + 1]):
            for k in range(koff[c], koff[c + 1]):
                j = int(shared[k])
                if j >= 0:
                    if j in first:
                        new[k] = new[first[j]]
                    else:
                        new[k] = np.maximum(new[k], 1)
                        first[j] = k
    out["kmer_counts"] = np.ascontiguousarray(new.reshape(-1))
    return out


@pytest.mark.parametrize("shape,n,S", [("A", 64, 3), ("B", 24, 3), ("C", 8, 3), ("D", 3, 3), ("A", 70, 10), ("C", 6, 10), ("B", 20, 10)])
def test_low_coverage_forces_the_rare_branches(gpu_ctx, oracle, shape, n, S):
    """diffuse posteriors: run logs overflow (> EV_CAP runs per chain), more than KSC_WAYS distinct diplotypes per chain, nested
    parents flipping; the full default schedule, every sweep traced"""
    from bayestyper_amd import synth

    rng = np.random.default_rng(77 + S)
    keep = 0.01 if shape == "D" else 0.1      # the many-candidate clusters have hundreds of k-mers per allele
    flat = thin(synth.make_hetero_batch(shape, n, S, seed=1300 + 3 * S + ord(shape)), rng, keep)
    lut = _oracle.build_luts(oracle, S, mean=15.0 * keep, var=30.0 * keep, noise_rate=0.05)
    traces = run_full(gpu_ctx, oracle, flat, lut=lut, seed=88, **FULL)
    runs, distinct, root_changes = flip_statistics(traces, FULL["burn"], FULL["iters"], FULL["chains"])
    assert runs > 2 * EV_CAP, f"only {runs} runs per chain: the run log never overflowed"
    if shape != "A":
        assert distinct > KSC_WAYS, f"only {distinct} distinct diplotypes per chain"
    if shape == "C":
        assert root_changes > 50 * n, "the nested parents hardly ever changed their diplotype"


def test_mixed_tile_classes_full_schedule_S3(gpu_ctx, oracle):
    """all launch classes of the bench's mixture in ONE batch at the full schedule (narrow tiles share pool blocks, the classes run
    concurrently): 96 A + 12 B + 3 C + 1 D groups"""
    from bayestyper_amd import synth

    S = 3
    rng = np.random.default_rng(5)
    parts = []
    for shape, n in (("D", 1), ("C", 3), ("B", 12), ("A", 96)):
        f = synth.make_hetero_batch(shape, n, S, seed=1700 + ord(shape))
        parts.append(f)
    flat = synth.concat(parts)
    run_full(gpu_ctx, oracle, flat, seed=101, trace_all=True, **FULL)


def _noise_genotyping_case(gpu_ctx, oracle, flat, kw):
    """estimateNoiseAndGenotypes through the C++ InferenceEngine (the class the executable ships, bound by host/inference_engine.py) against the
    oracle's restatement of InferenceEngine.cpp:384-472: every sampled noise rate of every iteration of every chain, then the collected samples"""
    from bayestyper_amd.host import count_model
    from bayestyper_amd.host.inference_engine import InferenceEngine

    S = flat["S"]
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)

    def cd():
        d = count_model.CountDistribution(S, prior=(1.0, 0.01), seed=kw["seed"])
        for s in range(S):
            d.set_genomic(s, 15.0, 30.0)
        return d

    cd_o, cd_g = cd(), cd()
    og = _oracle.OrcGibbs(oracle, flat, *cd_o.tables(), noise_seeding=1, **kw)
    want = og.estimate_noise_and_genotypes()
    ro = og.results()
    og.close()
    eng = InferenceEngine(gpu_ctx, kw["seed"], burn=kw["burn"], samples=kw["iters"], chains=kw["chains"])
    gg, got = eng.estimate_noise_and_genotypes(flat, cd_g)
    rg = gg.results()
    gg.close()
    assert got.shape == want.shape == (kw["chains"] * (1 + kw["burn"] + kw["iters"]), 2 + S)
    assert_noise_rows(got, want)
    exact = assert_parity(flat, ro, rg, kw["chains"] * kw["iters"])
    assert exact == flat["num_clusters"]


def test_noise_genotyping_ten_samples_sv_rich_cpp_engine(gpu_ctx, oracle):
    """BASELINE configs[3]'s sample count in --noise-genotyping mode: nested SV groups with multicluster k-mers, multi-variant clusters, a
    many-candidate cluster and two-haplotype clusters in one unit, 3 x (20 + 50) iterations (caches cleared every iteration)"""
    from bayestyper_amd import synth

    S = 10
    flat = synth.concat([synth.make_hetero_batch("D", 1, S, seed=61), synth.make_hetero_batch("C", 5, S, seed=62), synth.make_hetero_batch("B", 10, S, seed=63),
                         synth.make_hetero_batch("A", 40, S, seed=64)])
    assert int(flat["multi_off"][-1]) > 0
    _noise_genotyping_case(gpu_ctx, oracle, flat, dict(seed=2468, chains=3, burn=20, iters=50))


def test_config_C5_thirty_samples_joint_cpp_engine(gpu_ctx, oracle):
    """BASELINE configs[4]: --noise-genotyping, 30 samples, --max-number-of-sample-haplotypes 32 (256 merged candidates per cluster) next to
    small clusters, 2 x (10 + 20) iterations through the C++ estimateNoiseAndGenotypes"""
    from bayestyper_amd import synth

    S = 30
    flat = synth.concat([synth.make_batch("D", 2, S, seed=51), synth.make_hetero_batch("B", 3, S, seed=52), synth.make_hetero_batch("A", 11, S, seed=53)])
    assert int(flat["num_haplotypes"].max()) == 256
    _noise_genotyping_case(gpu_ctx, oracle, flat, dict(seed=4321, chains=2, burn=10, iters=20))
