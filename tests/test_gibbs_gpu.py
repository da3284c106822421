"""GPU parity tests for the Gibbs genotyping path (through the C ABI) against the oracle on identical inputs and seed.
Bar (BASELINE.json north_star): genotype posteriors within 1e-4.  The draw streams are reproduced exactly, so the integer
diplotype sampling frequencies are expected to be identical; the tests assert the 1e-4 bar and report exact equality."""
import os

import numpy as np
import pytest

import _oracle

pytestmark = pytest.mark.gpu
TOL = 1e-4   # posterior tolerance stated by north_star


def run_both(gpu_ctx, oracle, flat, trace=0, flat_gpu=None, **kw):
    """the oracle on `flat`, the GPU on `flat_gpu` (default: the same batch)"""
    from bayestyper_amd import lib

    S = flat["S"]
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    og = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, **kw)
    gg = lib.Gibbs(gpu_ctx, flat if flat_gpu is None else flat_gpu, lut_g, lut_n, **kw)
    if trace:
        og.trace_enable(trace)
        gg.trace_enable(trace)
    og.run(8)
    gg.run()
    gpu_ctx.sync()
    ro, rg = og.results(), gg.results()
    # the same results as the word string a rank hands to the gather (bt_gibbs_result_words, packed on the device)
    rw, used = lib.parse_result_words(gg.result_words_host())
    assert used == gg.result_words()[1] and set(rw) == set(rg)
    for key in rg:
        assert np.array_equal(rw[key], rg[key], equal_nan=(key == "stats")), key
    tr = None
    if trace:
        goff = flat["group_cluster_off"]
        tg = gg.trace()
        tr = [(og.trace(g, int(goff[g + 1] - goff[g]), trace), tg[g]) for g in range(flat["num_groups"])]
    og.close()
    gg.close()
    return ro, rg, tr


RATE_RTOL = 1e-12   # BT_NOISE_ON_DEVICE=1 only: noise rates drawn on the device (bt_gibbs_noise_chain: ocml log / pow / sqrt) against libstdc++'s draws through glibc: the last bits may differ


def assert_noise_rows(got, want):
    """rows of the noise parameter file: exact (the default: the drivers iterate on the host, the draws are libstdc++'s own); with BT_NOISE_ON_DEVICE the rates within RATE_RTOL"""
    import os

    assert got.shape == want.shape
    assert np.array_equal(got[:, :2], want[:, :2])
    if not os.environ.get("BT_NOISE_ON_DEVICE"):
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got[:, 2:], want[:, 2:], rtol=RATE_RTOL, atol=0), f"noise rates differ by {np.abs(got[:, 2:] / want[:, 2:] - 1).max()} (relative)"


def posteriors(r, c, S):
    """{(h1,h2): freq/total} per sample for cluster c"""
    e0, e1 = int(r["dip_off"][c]), int(r["dip_off"][c + 1])
    tot = r["freq"][e0:e1].sum(axis=0).astype(np.float64)
    return {(int(r["h1"][e]), int(r["h2"][e])): r["freq"][e] / np.maximum(tot, 1) for e in range(e0, e1)}, tot


def assert_parity(flat, ro, rg, n_collect):
    S = flat["S"]
    exact = 0
    for c in range(flat["num_clusters"]):
        po, to = posteriors(ro, c, S)
        pg, tg = posteriors(rg, c, S)
        assert (to == n_collect).all() and (tg == n_collect).all()
        keys = set(po) | set(pg)
        worst = max(np.abs(po.get(k, np.zeros(S)) - pg.get(k, np.zeros(S))).max() for k in keys)
        assert worst <= TOL, f"cluster {c}: diplotype posterior differs by {worst}"
        exact += int(worst == 0)
    # allele k-mer statistics (NAK/FAK/MAC inputs): counts exact, means to 1e-9 relative
    so, sg = ro["stats"], rg["stats"]
    assert so.shape == sg.shape
    assert np.array_equal(so[:, :, 0], sg[:, :, 0])
    assert np.allclose(so[:, :, 1:3], sg[:, :, 1:3], rtol=1e-9, atol=1e-12)
    return exact


@pytest.mark.parametrize("shape,n,S", [("A", 64, 3), ("A", 50, 10), ("B", 16, 3), ("A", 30, 1)])
def test_single_cluster_groups(gpu_ctx, oracle, shape, n, S):
    from bayestyper_amd import synth

    flat = synth.make_batch(shape, n, S, seed=1000 + S)
    kw = dict(seed=42, chains=4, burn=20, iters=50)
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=40, **kw)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}: diplotype trace diverges at sweep {np.argwhere((to != tg[:len(to)]).any(axis=(1, 2)))[:1]}"
    exact = assert_parity(flat, ro, rg, 4 * 50)
    assert exact == flat["num_clusters"]


@pytest.mark.parametrize("S", [30, 4])
def test_joint_shape_many_haplotypes(gpu_ctx, oracle, S):
    """shape D of SURVEY 8(d): 256 haplotype candidates (32 per sample x 8 merged), up to the maximum of 30 samples (main.cpp:72): the
    per-(sample, diplotype) tables exceed the dense-table budget, so the tag-checked direct-mapped cache is the one in use; sparse
    frequency sampler over hundreds of zero-count haplotypes, hash-set emulation with rehashes up to 541 buckets"""
    from bayestyper_amd import synth

    flat = synth.make_batch("D", 2, S, seed=2000 + S)
    assert int(flat["num_haplotypes"].max()) == 256
    kw = dict(seed=9, chains=2, burn=3, iters=6) if S == 30 else dict(seed=9, chains=2, burn=6, iters=12)
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=9, **kw)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}"
    exact = assert_parity(flat, ro, rg, 2 * kw["iters"])
    assert exact == flat["num_clusters"]


@pytest.mark.parametrize("shape,n,kw", [("B", 12, dict(chains=2, burn=10, iters=20)), ("C", 3, dict(chains=2, burn=6, iters=12)), ("D", 2, dict(chains=2, burn=3, iters=6))])
def test_config_C4_ten_samples_shapes_BCD(gpu_ctx, oracle, shape, n, kw):
    """BASELINE configs[3] (WGS, 10 samples): the multi-variant, nested-SV and many-candidate shapes at S=10, every group with its own
    dimensions (synth.hetero_group: ragged tiles), per-sweep diplotype traces and sampling frequencies against the oracle"""
    from bayestyper_amd import synth

    S = 10
    flat = synth.make_hetero_batch(shape, n, S, seed=400 + ord(shape))
    if shape == "D":
        assert int(flat["num_haplotypes"].min()) >= 128
    sweeps = kw["burn"] + kw["iters"]
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=sweeps, seed=17, **kw)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}"
    exact = assert_parity(flat, ro, rg, kw["chains"] * kw["iters"])
    assert exact == flat["num_clusters"]


def test_config_C3_trio_mixture_the_bench_workload(gpu_ctx, oracle):
    """BASELINE configs[2] (WGS trio, the workload bench.py times): a 700-group batch of the bench's own mixture generator at S=3 — two-haplotype
    clusters (the kernel's simple path, its own launch class), multi-variant clusters, nested SV groups and many-candidate clusters, every
    structure with its own dimensions — through several launch classes at once.  Traces of the first sweeps and the sampling frequencies /
    allele statistics of a 3 x (10 + 30) schedule against the oracle."""
    from bayestyper_amd import synth

    S = 3
    flat = synth.make_mixture(700, S, seed=303, templates={"A": 48, "B": 24, "C": 6, "D": 3})
    mix = flat["mixture"]
    assert mix["A"] >= 500 and mix["B"] >= 40 and mix["C"] >= 8 and mix["D"] >= 2
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=12, seed=23, chains=3, burn=10, iters=30)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}"
    exact = assert_parity(flat, ro, rg, 3 * 30)
    assert exact == flat["num_clusters"]


def test_config_C5_noise_genotyping_joint_30_samples(gpu_ctx, oracle):
    """BASELINE configs[4]: --noise-genotyping (estimateNoiseAndGenotypes, InferenceEngine.cpp:384-472) on the joint shape — 30 samples,
    --max-number-of-sample-haplotypes 32 => 256 merged haplotype candidates per cluster — together with small clusters in the same
    unit: every sampled noise rate of every iteration and the collected genotype samples against the oracle's driver."""
    from bayestyper_amd import synth
    from bayestyper_amd.host import count_model
    from bayestyper_amd.host.inference_engine import InferenceEngine

    S = 30
    flat = synth.concat([synth.make_batch("D", 2, S, seed=51), synth.make_hetero_batch("B", 3, S, seed=52), synth.make_hetero_batch("A", 11, S, seed=53)])
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    assert int(flat["num_haplotypes"].max()) == 256
    kw = dict(seed=4321, chains=2, burn=2, iters=3)

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
    assert got.shape == (kw["chains"] * (1 + kw["burn"] + kw["iters"]), 2 + S)
    assert_noise_rows(got, want)
    exact = assert_parity(flat, ro, rg, kw["chains"] * kw["iters"])
    assert exact == flat["num_clusters"]


def test_default_schedule_shape_A(gpu_ctx, oracle):
    """the reference's default schedule: 20 chains x (100 burn-in + 250 collected) (main.cpp:389-391)"""
    from bayestyper_amd import synth

    flat = synth.make_batch("A", 24, 10, seed=77)
    ro, rg, _ = run_both(gpu_ctx, oracle, flat)
    assert_parity(flat, ro, rg, 20 * 250)


def test_nested_groups_with_multicluster_kmers(gpu_ctx, oracle):
    from bayestyper_amd import synth

    flat = synth.make_batch("C", 6, 3, seed=5)
    kw = dict(seed=7, chains=3, burn=15, iters=40)
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=30, **kw)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}"
    assert_parity(flat, ro, rg, 3 * 40)


def test_ploidy_gender_and_ragged_inputs(gpu_ctx, oracle):
    """haploid / absent chromosomes (male X/Y), male intercluster multiplicities, a group with a single k-mer-less sample"""
    from bayestyper_amd import synth

    rng = np.random.default_rng(3)
    groups = [synth.group_shape_A(rng, i) for i in range(12)] + [synth.GroupSpec([synth.make_cluster(rng, 2, 4, 20, ic_kmers=6)], [100 + i]) for i in range(6)]
    S = 4
    ploidy = np.full((len(groups), S), 2, np.uint8)
    ploidy[::3, 1] = 1
    ploidy[1::3, 2] = 0
    ploidy[5] = 0
    flat = synth.flatten(groups, S, rng, ploidy=ploidy, gender=[0, 1, 1, 0])
    kw = dict(seed=99, chains=3, burn=10, iters=30)
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=20, **kw)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}"
    assert_parity(flat, ro, rg, 3 * 30)


def test_stepwise_driving_and_noise_counts(gpu_ctx, oracle):
    """init_chain / sweep / noise_counts driven step by step as the noise drivers do (InferenceEngine.cpp:60-98)"""
    from bayestyper_amd import lib, synth

    flat = synth.make_batch("A", 40, 3, seed=11)
    S = 3
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    kw = dict(seed=5, chains=2, burn=5, iters=10, noise_seeding=1)
    og = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, **kw)
    gg = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, **kw)
    for chain in range(2):
        og.init_chain(chain)
        gg.init_chain(chain)
        for it in range(8):
            og.sweep(1, it >= 4)
            gg.sweep(1, it >= 4)
            ho, hg = og.noise_counts(), gg.noise_counts()
            assert np.array_equal(ho, hg) and ho.sum() > 0
            if it == 3:   # a noise-rate update in the middle of the chain
                _, ln2 = _oracle.build_luts(oracle, S, noise_rate=0.2)
                og.set_noise_lut(ln2)
                gg.set_noise_lut(ln2)
        if chain == 0:    # estimateNoise deletes the genotypers after every chain
            og.reset_groups()
            gg.reset_groups()
    ro, rg = og.results(), gg.results()
    assert_parity(flat, ro, rg, 4)
    og.close(), gg.close()


def test_noise_drivers_match_oracle(gpu_ctx, oracle, tmp_path):
    """InferenceEngine.estimate_noise / estimate_noise_and_genotypes on the GPU (host/inference_engine.py over bt_gibbs_sweep,
    bt_gibbs_noise_counts, bt_gibbs_set_noise_lut) against the oracle's restatement of InferenceEngine.cpp:135-276 and :384-472:
    every sampled noise rate of every iteration of every chain (they are functions of the integer histograms and of the run's
    generator, so equality is exact), the groups each chain selected, and the collected genotype samples."""
    from bayestyper_amd import synth
    from bayestyper_amd.host import count_model
    from bayestyper_amd.host.inference_engine import InferenceEngine

    S = 3
    parts = [synth.make_batch("A", 60, S, seed=21, templates=6), synth.make_batch("C", 3, S, seed=22), synth.make_batch("B", 9, S, seed=23, templates=3)]
    flat = synth.concat(parts)
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    kw = dict(seed=1234, chains=3, burn=6, iters=9)

    def cd():
        d = count_model.CountDistribution(S, prior=(1.0, 0.01), seed=kw["seed"])
        for s in range(S):
            d.set_genomic(s, 15.0, 30.0)
        return d

    eng = InferenceEngine(gpu_ctx, kw["seed"], burn=kw["burn"], samples=kw["iters"], chains=kw["chains"])
    # estimateNoise: a variant budget that selects a different subset of the single-cluster groups in every chain
    cd_o, cd_g = cd(), cd()
    og = _oracle.OrcGibbs(oracle, flat, *cd_o.tables(), noise_seeding=1, **kw)
    want, chains, final = og.estimate_noise(variants_batch_size=25)
    og.close()
    assert len({tuple(c) for c in chains}) == kw["chains"] and all(0 < len(c) < 69 for c in chains)
    got = eng.estimate_noise(cd_g, flat, output_prefix=str(tmp_path / "noise"), variants_batch_size=25)
    assert_noise_rows(got, want)
    assert np.allclose(cd_g.noise_rates(), final, rtol=RATE_RTOL, atol=0)
    assert len(open(tmp_path / "noise.txt").read().split("\n")) == len(want) + 2
    # estimateNoiseAndGenotypes over all groups (nested groups included)
    cd_o, cd_g = cd(), cd()
    og = _oracle.OrcGibbs(oracle, flat, *cd_o.tables(), noise_seeding=1, **kw)
    want = og.estimate_noise_and_genotypes()
    ro = og.results()
    og.close()
    gg, got = eng.estimate_noise_and_genotypes(flat, cd_g)
    rg = gg.results()
    gg.close()
    assert_noise_rows(got, want)
    exact = assert_parity(flat, ro, rg, kw["chains"] * kw["iters"])
    assert exact == flat["num_clusters"]


@pytest.mark.parametrize("S,parts,mode", [(3, (("A", 150, 6), ("B", 40, 3), ("C", 6, 1)), "hybrid"), (3, (("A", 150, 6), ("B", 40, 3), ("C", 6, 1), ("D", 3, 1)), "all"),
                                          (3, (("A", 150, 6), ("B", 40, 3), ("C", 6, 1), ("D", 3, 1)), "lds_cap"), (1, (("A", 200, 8), ("B", 20, 2)), "hybrid"),
                                          (10, (("A", 70, 4), ("B", 12, 2)), "hybrid")])
def test_resident_noise_chain_equals_launch_per_iteration(gpu_ctx, oracle, monkeypatch, S, parts, mode):
    """bt_gibbs_noise_chain_begin / _step / _end (a chain of a noise driver as ONE resident launch: sampler state kept in registers / LDS across the
    iterations, histogram and table exchanged through pinned memory) against bt_gibbs_noise_iteration (a sweep + a tally launch and a synchronisation per
    iteration) on two samplers over the same batch — two-haplotype tiles (gibbs_simple_kernel), LDS-resident multi-allelic and many-candidate clusters
    (gibbs_hot_kernel) and nested groups (gibbs_kernel, hot arrays swapped per visit) — with another noise table every iteration, collecting from the
    fourth iteration on, two chains with a group reset between them: the histogram of every iteration, every sampled diplotype of every sweep and the
    collected results are identical.  mode "hybrid" (the default): a batch with large dense tables (the many-candidate clusters) runs iteration 0 of a chain as
    ordinary launches — the whole GPU fills the tables the first sweep asks for — and the resident launch from iteration 1 on; "all": resident from
    iteration 0 on, the tables filled by the tiles' own lanes; "lds_cap": the chain keeps the hot arrays of every tile but the two-haplotype ones in HBM (what
    it does to the few tiles whose LDS block would keep the chain's workgroups from being resident together)."""
    from bayestyper_amd import lib, synth

    if mode != "hybrid":
        monkeypatch.setenv("BT_NOISE_CHAIN_WIDE", "1")   # (the many-candidate clusters' large tables: refilled by their own lanes inside the resident launch)
    if mode == "lds_cap":
        monkeypatch.setenv("BT_NOISE_CHAIN_LDS_CAP", "1")
    flat = synth.concat([synth.make_batch(sh, n, S, seed=31 + i, templates=t) for i, (sh, n, t) in enumerate(parts)])
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    n_it, first_collect = 9, 3
    tables = [_oracle.build_luts(oracle, S, noise_rate=0.02 + 0.03 * i)[1] for i in range(n_it)]
    kw = dict(seed=77, chains=2, burn=first_collect, iters=n_it - first_collect, noise_seeding=1)
    ga, gb = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, **kw), lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, **kw)
    ga.trace_enable(2 * n_it)
    gb.trace_enable(2 * n_it)
    for chain in range(2):
        # (one after the other: both samplers enqueue on the context's stream, and a resident launch stays on it until its chain ends)
        ga.set_noise_lut(lut_n)
        ga.init_chain(chain)
        want = [ga.noise_iteration(tables[it] if it else None, it >= first_collect) for it in range(n_it)]
        gb.set_noise_lut(lut_n)
        gb.init_chain(chain)
        assert gb.noise_chain_begin(n_it, first_collect), "the batch should fit the GPU as one resident launch"
        for it in range(n_it):
            hb = gb.noise_chain_step(tables[it] if it else None)
            assert hb.sum() > 0 and np.array_equal(want[it], hb), (chain, it)
        gb.noise_chain_end()
        if chain == 0:
            ga.reset_groups()
            gb.reset_groups()
    gpu_ctx.sync()
    for g, (ta, tb) in enumerate(zip(ga.trace(), gb.trace())):
        assert np.array_equal(ta, tb), f"group {g}"
    ra, rb = ga.results(), gb.results()
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), k
    assert ra["freq"].sum() > 0
    # a chain given up half way: the launch stops at its next exchange and the sampler is usable again
    gb.reset_groups()
    gb.set_noise_lut(lut_n)
    gb.init_chain(0)
    assert gb.noise_chain_begin(50, 50)
    gb.noise_chain_step(None)
    gb.noise_chain_end()
    gb.reset_groups()
    gb.init_chain(1)
    gb.sweep(2, False)
    gpu_ctx.sync()
    ga.close(), gb.close()


def test_noise_drivers_launch_per_iteration_path(gpu_ctx, oracle, tmp_path, monkeypatch):
    """BT_NOISE_CHAIN_OFF=1: the noise drivers iterate with a launch + synchronisation per iteration (what batches that cannot be resident take): same rows"""
    monkeypatch.setenv("BT_NOISE_CHAIN_OFF", "1")
    test_noise_drivers_match_oracle(gpu_ctx, oracle, tmp_path)


def test_noise_drivers_device_chain_opt_in(gpu_ctx, oracle, tmp_path, monkeypatch):
    """BT_NOISE_ON_DEVICE=1: a driver's whole chain runs on the device (bt_gibbs_noise_chain, no host round trip per iteration; the rates drawn with ocml's
    log / pow / sqrt): the rates agree within RATE_RTOL, the collected genotype samples are identical.  (The default — the host loop — is exact.)"""
    monkeypatch.setenv("BT_NOISE_ON_DEVICE", "1")
    test_noise_drivers_match_oracle(gpu_ctx, oracle, tmp_path)


def test_unit_in_several_launches(gpu_ctx):
    """InferenceEngine.estimate_genotypes with max_groups_per_launch: a unit too large for one launch is run as consecutive launches;
    samples and posterior summaries equal those of the single launch (groups keep their unit-wide index)"""
    from bayestyper_amd import synth
    from bayestyper_amd.host import count_model
    from bayestyper_amd.host.inference_engine import InferenceEngine

    S = 2
    flat = synth.concat([synth.make_batch("A", 40, S, seed=31, templates=5), synth.make_batch("C", 3, S, seed=32), synth.make_batch("B", 10, S, seed=33, templates=2)])
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    cd = count_model.CountDistribution(S, seed=5)
    for s in range(S):
        cd.set_genomic(s, 15.0, 30.0)
    eng = InferenceEngine(gpu_ctx, 5, burn=10, samples=20, chains=2)
    whole = eng.estimate_genotypes(flat, cd)
    parts = eng.estimate_genotypes(flat, cd, max_groups_per_launch=17)
    a, b = whole.results(), parts.results()
    for k in ("dip_off", "h1", "h2", "freq", "cell_off", "stats"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(whole.posterior_summary(), parts.posterior_summary()) and len(parts.parts) == 4
    # ... and sized by the engine itself from the free HBM (bt_gibbs_state_bytes against bt_ctx_info; here the GPU is pretended small)
    import os

    os.environ["BT_GIBBS_FREE_BYTES"] = str(6 << 20)
    try:
        auto = eng.estimate_genotypes(flat, cd)
    finally:
        del os.environ["BT_GIBBS_FREE_BYTES"]
    c = auto.results()
    for k in ("dip_off", "h1", "h2", "freq", "cell_off", "stats"):
        assert np.array_equal(a[k], c[k]), k
    assert len(auto.parts) > 1 and sum(auto.parts) == flat["num_groups"]
    whole.close(), parts.close(), auto.close()


def test_sharded_run_equals_unsharded_and_summary_definition(gpu_ctx, oracle):
    """Multi-GPU path by construction: the groups of a batch split over two 'ranks' (run one after the other on this GPU), each
    keeping its global group indices, give exactly the unsharded posterior summaries; bt_gibbs_posterior_summary agrees with its
    host-side statement applied to the fetched results (and, through it, with the oracle)."""
    from bayestyper_amd import lib, shard, synth

    S = 2
    flat = synth.concat([synth.make_batch("A", 70, S, seed=5, templates=3), synth.make_batch("B", 9, S, seed=6, templates=2), synth.make_batch("C", 3, S, seed=7)])
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    kw = dict(seed=42, chains=2, burn=10, iters=30)

    def gpu_summary(f):
        g = lib.Gibbs(gpu_ctx, f, lut_g, lut_n, **kw)
        g.run()
        summ, res = g.posterior_summary(), g.results()
        g.close()
        return summ, res

    full, res = gpu_summary(flat)
    assert np.array_equal(full, shard.summary_from_results(res, flat["num_clusters"], S))
    og = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, **kw)
    og.run(8)
    assert np.array_equal(full, shard.summary_from_results(og.results(), flat["num_clusters"], S))
    og.close()
    parts = shard.assign_groups(shard.group_cost(flat), 2)
    merged = np.zeros_like(full)
    for ids in parts:
        summ, _ = gpu_summary(shard.take_groups(flat, ids))
        merged[shard.cluster_ids_of(flat, ids)] = summ
    assert np.array_equal(merged, full)


def test_graphs_to_genotypes_end_to_end(gpu_ctx, oracle):
    """The whole hot path on one small unit, GPU side only through the C ABI: graphs + best paths -> path k-mers -> KMC scan into the
    count table -> classifyPathKmers -> getHaplotypeCandidates -> Gibbs.  The oracle runs the same pipeline; the bundles and the
    diplotype sampling frequencies must agree."""
    import copy

    from _oracle import OrcBloom, OrcGraphs, OrcKmc, OrcTable
    from bayestyper_amd import lib, synth_graphs

    K, S = 55, 2
    rng = np.random.default_rng(77)
    gs = [synth_graphs.random_cluster(rng, K, int(rng.integers(1, 4)), int(rng.integers(2, 6)), kinds=("snv", "ins", "del")) for _ in range(10)]
    gs[1] = copy.deepcopy(gs[0])                      # a second cluster over the same sequence: multicluster k-mers inside one group
    gs[1].paths = synth_graphs.random_paths(gs[1], rng, 2)
    f = synth_graphs.flatten(gs)
    groups = [[0, 1]] + [[c] for c in range(2, len(gs))]
    og, gp = OrcGraphs(oracle, f, K), lib.Paths(gpu_ctx, f, K)
    ob, gb = OrcBloom(oracle, 100_000, 1e-3, K, threaded=True), lib.Bloom.create(gpu_ctx, 100_000, 1e-3, K, threaded=True)
    og.count_kmers(ob)
    gp.count_kmers(gb)
    # "reads": every path k-mer of the first path of each cluster gets NB-like counts in both samples -> KMC databases -> table
    nt = np.frombuffer(b"ACGT", np.uint8)
    hap_text = np.concatenate([np.concatenate([nt[g.seq[v]] for v in range(len(g.seq)) if g.paths[0, v]] + [np.frombuffer(b"N", np.uint8)]) for g in gs])
    km, va = oracle.kmers_from_sequence(hap_text.tobytes(), K)
    present = np.unique(km[va == 1], axis=0)
    ot, gt = OrcTable(oracle, S, K), lib.Table(gpu_ctx, 50_000, S, K)
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        for s in range(S):
            cnt = rng.poisson(15, len(present)).astype(np.uint32) + 1
            pref = os.path.join(td, f"s{s}")
            asc = oracle.unpack(present, K).reshape(-1, K)
            order = np.lexsort(asc.T[::-1])                 # KMC order = ascending ASCII order
            oracle.kmc_write(pref, np.ascontiguousarray(asc[order]).reshape(-1), cnt[order], K, 3, 1)
            db = OrcKmc(oracle, pref)
            ot.parse_sample_kmers(ob, db, s)
            sc = lib.KmcScan(gpu_ctx, db.k, db.p, db.counter_size, db.total, db.lut())
            buf = gpu_ctx.to_device(db.payload())
            sc.run(gb, gt, s, buf.ptr, 0, db.total)
            gpu_ctx.sync()
            sc.close(), buf.free(), db.close()
    omg, gmg = OrcBloom(oracle, 10, 1e-4, K), lib.Bloom.create(gpu_ctx, 10, 1e-4, K, threaded=False)
    n_o, ex_o = og.classify(ot, omg)
    n_g, ex_g = gp.classify(gt, gmg)
    assert np.array_equal(n_o, n_g) and np.array_equal(ex_o, ex_g)
    co, cg = og.candidates(ot), gp.candidates(gt)
    for name in co:
        assert np.array_equal(co[name], cg[name]), name
    assert len(co["multi_idx"]) > 0 and co["kmer_has_counts"].sum() > len(co["kmer_has_counts"]) // 3 and co["kmer_counts"].sum() > 1000
    # each side gets the batch assembled from ITS OWN bundle, by different code: the product's harness for the GPU, the test's own for the oracle
    from test_cli_gpu import _gibbs_batch as oracle_gibbs_batch

    flat_gpu = synth_graphs.gibbs_batch_from_candidates(cg, f, groups, S)
    flat = oracle_gibbs_batch(co, f, groups, S, np.full((len(groups), S), 2, np.uint8), [0] * S, list(range(len(gs))), [list(range(len(g))) for g in groups], [[] for _ in gs])
    for name in flat:
        assert np.array_equal(np.asarray(flat[name]), np.asarray(flat_gpu[name])), name
    kw = dict(seed=3, chains=2, burn=10, iters=30)
    ro, rg, tr = run_both(gpu_ctx, oracle, flat, trace=15, flat_gpu=flat_gpu, **kw)
    for g, (to, tg) in enumerate(tr):
        assert np.array_equal(to, tg[: len(to)]), f"group {g}"
    assert_parity(flat, ro, rg, 2 * 30)
    for x in (og, gp, ob, gb, ot, gt, omg, gmg):
        x.close()


def test_genotype_posteriors_gpp_gq_calls(gpu_ctx, oracle):
    """The contract in the reference's own output terms: per variant and sample GPP / APP within 1e-4 (they are expected identical),
    GQ, allele filters, genotype calls and AC / ACP equal — GPU sampler + C++ host getGenotypes against oracle sampler + oracle getGenotypes."""
    from bayestyper_amd import synth
    from bayestyper_amd.host import genotypes

    S = 3
    rng = np.random.default_rng(8)
    groups = [synth.group_shape_A(rng, i) for i in range(40)] + [synth.group_shape_B(rng, 100 + i) for i in range(8)] + [synth.group_shape_C(rng, 200 + 3 * i, root_H=8, root_kpa=60) for i in range(3)]
    ploidy = np.full((len(groups), S), 2, np.uint8)
    ploidy[::5, 1] = 1
    ploidy[7, 2] = 0
    flat = synth.flatten(groups, S, rng, ploidy=ploidy, gender=[0, 1, 1])
    ro, rg, _ = run_both(gpu_ctx, oracle, flat, seed=21, chains=4, burn=20, iters=60)
    mf = genotypes.min_fraction_observed_kmers([15.0] * S)
    goff = flat["group_cluster_off"]
    calls = 0
    for g in range(flat["num_groups"]):
        for c in range(goff[g], goff[g + 1]):
            a = genotypes.cluster_genotypes(flat, rg, c, ploidy[g], mf)                                            # product: GPU results + host layer
            b = genotypes.cluster_genotypes(flat, ro, c, ploidy[g], mf, fn=oracle.l.orc_cluster_genotypes)         # oracle end to end
            assert np.abs(a["gpp"] - b["gpp"]).max() <= TOL and np.abs(a["app"] - b["app"]).max() <= TOL
            for k in ("gq", "filters", "estimate", "total_count", "alt_counts", "non_covered"):
                assert np.array_equal(a[k], b[k]), (c, k)
            assert np.allclose(a["acp"], b["acp"], atol=TOL) and np.allclose(a["alt_freq"], b["alt_freq"], atol=TOL)
            calls += int((a["estimate"][:, :, 0] != 0xFFFF).sum())
            # and the text GenotypeWriter would emit for these variants (QUAL, FILTER, INFO statistics, sample columns)
            assert genotypes.cluster_output_columns(flat, rg, c, ploidy[g], mf) == genotypes.cluster_output_columns(flat, ro, c, ploidy[g], mf, fn=oracle.l.orc_cluster_output_columns)
    assert calls > 50


def test_scale_properties_of_a_mixture_batch(gpu_ctx):
    """Size-independent properties at a batch size the scalar oracle would need minutes for (20 000 groups of the bench's shape
    mixture, full 20 x 350 schedule): every (cluster, sample) collects exactly chains x samples draws, a second run of the same batch
    reproduces every frequency (the launch is deterministic), and the device-side posterior summary agrees with its host definition."""
    from bayestyper_amd import lib, shard, synth
    from bayestyper_amd.host import count_model

    S = 1
    flat = synth.make_mixture(20_000, S, seed=77)
    lut_g, lut_n = count_model.build_luts(S, mean=15.0, var=30.0, noise_rate=0.05)
    runs = []
    for _ in range(2):
        g = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, seed=42)
        g.run()
        runs.append((g.results(), g.posterior_summary()))
        g.close()
    (r0, s0), (r1, s1) = runs
    for k in ("dip_off", "h1", "h2", "freq"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(r0["stats"], r1["stats"]) and np.array_equal(s0, s1)
    total = np.add.reduceat(r0["freq"][:, 0].astype(np.int64), r0["dip_off"][:-1].astype(np.int64))
    assert (total == 20 * 250).all()                                   # Null-ploidy samples would still record (none, none)
    assert np.array_equal(s0, shard.summary_from_results(r0, flat["num_clusters"], S))
    assert (s0[:, :, 1] > 0).all() and (s0[:, :, 1] <= 20 * 250).all()


def test_wide_refill_and_stepwise_run_change_nothing(gpu_ctx, oracle, monkeypatch):
    """The whole-GPU table refill (nan_fill_kernel + ucache_prefill_kernel after a stepwise chain start and after every cache-clearing noise count),
    bt_gibbs_run chain by chain (BT_GIBBS_STEPWISE), the choice between gibbs_single_kernel, gibbs_hot_kernel and gibbs_kernel and the packing of several tiles
    into one workgroup's LDS only move work: every collected statistic is bit-identical to the single-launch schedule,
    and a noise drivers' loop gives the same noise counts and samples with the refill switched off."""
    from bayestyper_amd import lib, synth

    S = 10
    flat = synth.concat([synth.make_hetero_batch("D", 1, S, seed=71), synth.make_hetero_batch("C", 3, S, seed=72), synth.make_hetero_batch("B", 6, S, seed=73),
                         synth.make_hetero_batch("A", 20, S, seed=74)])
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    kw = dict(seed=97, chains=3, burn=15, iters=30)

    def default_run(env):
        for k in ("BT_GIBBS_STEPWISE", "BT_GIBBS_NO_PREFILL", "BT_GIBBS_NO_WIDE_FILL", "BT_GIBBS_NO_HOT_KERNEL", "BT_GIBBS_SINGLE_KERNEL", "BT_GIBBS_MAX_CLASSES", "BT_GIBBS_PACK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, **kw)
        g.run()
        r = g.results()
        g.close()
        return r

    def noise_loop(env):
        for k in ("BT_GIBBS_STEPWISE", "BT_GIBBS_NO_PREFILL", "BT_GIBBS_NO_WIDE_FILL", "BT_GIBBS_NOISE_GLOBAL_ATOMICS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, noise_seeding=1, **kw)
        hists = []
        for chain in range(2):
            g.init_chain(chain)
            for it in range(12):
                g.sweep(1, it >= 4)
                hists.append(g.noise_counts().copy())
        r = g.results()
        g.close()
        return hists, r

    base = default_run({})
    # (BT_GIBBS_NO_HOT_KERNEL: the sampling launches through gibbs_kernel's generic pointers instead of gibbs_hot_kernel's LDS accesses)
    # (BT_GIBBS_SINGLE_KERNEL: one-cluster tiles through gibbs_single_kernel instead of gibbs_hot_kernel; BT_GIBBS_PACK: several tiles per workgroup sharing a slab
    #  of LDS instead of one; BT_GIBBS_MAX_CLASSES: seven launch classes — what a process with eight hardware queues gets — instead of three)
    for env in ({"BT_GIBBS_STEPWISE": "1"}, {"BT_GIBBS_STEPWISE": "1", "BT_GIBBS_NO_PREFILL": "1"}, {"BT_GIBBS_NO_HOT_KERNEL": "1"}, {"BT_GIBBS_SINGLE_KERNEL": "1"},
                {"BT_GIBBS_SINGLE_KERNEL": "1", "BT_GIBBS_PACK": "auto"}, {"BT_GIBBS_PACK": "2,65536"}, {"BT_GIBBS_MAX_CLASSES": "7"}):
        got = default_run(env)
        for k in base:
            assert np.array_equal(base[k], got[k]), (env, k)
    h0, r0 = noise_loop({})
    # ... and with the noise counts tallied by the OP_NOISE branch of gibbs_kernel (what more than 152 samples get) instead of gibbs_noise_kernel
    for env in ({"BT_GIBBS_NO_PREFILL": "1"}, {"BT_GIBBS_NO_PREFILL": "1", "BT_GIBBS_NO_WIDE_FILL": "1"}, {"BT_GIBBS_NOISE_GLOBAL_ATOMICS": "1"}):
        h1, r1 = noise_loop(env)
        assert all(np.array_equal(a, b) for a, b in zip(h0, h1)), env
        for k in r0:
            assert np.array_equal(r0[k], r1[k]), (env, k)


def test_resident_chain_falls_back_when_its_workgroups_are_not_resident_together(tmp_path):
    """bt_gibbs_noise_chain_begin checks residency against the WHOLE GPU; a CU mask (or another process / rank on the GPU) takes slots that check
    cannot see.  The launch's roll call (bt_noise_chain.hpp: nc_begin) then fails before any sampler state is touched, and bt_gibbs_noise_chain_step runs the
    chain launch by launch: same histograms, same results, no 60 s stall, no failed run.  (A CU mask in the child's environment is not honoured on the test
    boxes, so the child's roll call is told to expect three workgroups more than are launched — BT_NOISE_CHAIN_TEST_ABSENT_WGS —, which is what a launch with
    three workgroups waiting for a slot looks like from the inside.)"""
    import json
    import subprocess
    import sys

    env = dict(os.environ)
    env.update({"BT_NOISE_CHAIN_TEST_ABSENT_WGS": "3", "BT_GIBBS_DEBUG": "1", "BT_NOISE_CHAIN_ROLLCALL_S": "1.0"})
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_rollcall_child.py"), "4000"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().split("\n")[-1])
    assert out["histograms_equal"] and out["results_equal"], out
    assert "not resident together" in p.stderr, p.stderr[-2000:]
    assert out["resident"] and not out["resident_again"], out
