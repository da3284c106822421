"""Host logic of the noise drivers (bayestyper_amd/host/inference_engine.py, host/InferenceEngine.cpp, host/CountDistribution.cpp)
against the oracle's whole-driver restatement of InferenceEngine::estimateNoise / estimateNoiseAndGenotypes.

No GPU here: the per-rank sampler handed to the engine is the ORACLE's (tests may use it), so what is under test is everything
the host adds around the sampler — the per-chain group selection, the noise-rate draws of the run's CountDistribution, the
iteration order, the file rows, and (two gloo ranks) the sharded run with its per-iteration histogram all-reduce — all of which
must reproduce the unsharded oracle driver bit for bit.  The GPU sampler is put through the same comparison in test_gibbs_gpu.py.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from bayestyper_amd import shard, synth  # noqa: E402

S = 2
KW = dict(seed=77, chains=3, burn=4, iters=6)


def _unit():
    parts = [synth.make_batch("A", 14, S, seed=5, templates=3), synth.make_batch("C", 2, S, seed=7), synth.make_batch("B", 5, S, seed=6, templates=2)]
    flat = synth.concat(parts)
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    return flat


def _count_distribution(seed):
    from bayestyper_amd.host import count_model

    cd = count_model.CountDistribution(S, prior=(1.0, 0.01), seed=seed)
    for s in range(S):
        cd.set_genomic(s, 15.0 + s, 30.0 + 2 * s)
    return cd


def _engine(orc, reduce_hist=None):
    import _oracle
    from bayestyper_amd.host.inference_engine import InferenceEngine

    sampler = lambda flat, lut_g, lut_n, **kw: _oracle.OrcGibbs(orc, flat, lut_g, lut_n, **kw)   # noqa: E731
    return InferenceEngine(None, KW["seed"], burn=KW["burn"], samples=KW["iters"], chains=KW["chains"], sampler=sampler, reduce_hist=reduce_hist)


def _oracle_unit(orc, flat, cd):
    import _oracle

    lut_g, lut_n = cd.tables()
    return _oracle.OrcGibbs(orc, flat, lut_g, lut_n, noise_seeding=1, **KW)


@pytest.fixture(scope="module")
def orc():
    import _oracle

    return _oracle.load_oracle()


def test_noise_group_selection(orc):
    from bayestyper_amd.host.inference_engine import NoiseGroupSelector, unit_group_shape

    flat = _unit()
    cd = _count_distribution(KW["seed"])
    og = _oracle_unit(orc, flat, cd)
    nclu, nvar = unit_group_shape(flat)
    assert (nclu == 1).sum() == 19 and (nclu == 3).sum() == 2
    for batch in (9, 100000):   # a binding and a non-binding variant budget
        _, chains, _ = og.estimate_noise(variants_batch_size=batch)
        sel = NoiseGroupSelector(nclu, nvar, KW["seed"], batch)
        for c in range(KW["chains"]):
            mine = sel.next_chain()
            assert np.array_equal(mine, chains[c])
            assert np.all(nclu[mine] == 1) and np.all(np.diff(mine.astype(np.int64)) > 0)
            if batch == 9:
                assert nvar[mine].sum() >= 9 > nvar[mine[:-1]].sum() or len(mine) == 19
        sel.close()
    og.close()


def test_estimate_noise_single_rank(orc, tmp_path):
    flat = _unit()
    cd_o, cd_h = _count_distribution(KW["seed"]), _count_distribution(KW["seed"])
    og = _oracle_unit(orc, flat, cd_o)
    want, _, final = og.estimate_noise(variants_batch_size=12)
    og.close()
    eng = _engine(orc)
    got = eng.estimate_noise(cd_h, flat, output_prefix=str(tmp_path / "unit_noise_parameters"), sample_names=["s0", "s1"], variants_batch_size=12)
    assert got.shape == want.shape == (KW["chains"] * (KW["burn"] + KW["iters"] + 1) + 1, 2 + S)
    assert np.array_equal(got, want)   # every rate of every iteration, bit for bit
    assert np.array_equal(cd_h.noise_rates(), final) and np.all(final > 0)
    assert eng.low_variant_warning is False
    lines = open(tmp_path / "unit_noise_parameters.txt").read().split("\n")
    assert lines[0] == "Chain\tIteration\ts0\ts1" and lines[-1] == "" and len(lines) == len(want) + 2
    assert lines[1] == "1\t0\t" + "\t".join("%g" % r for r in want[0, 2:])
    assert lines[-2] == "0\t0\t" + "\t".join("%g" % r for r in final)


def test_estimate_noise_and_genotypes_single_rank(orc):
    flat = _unit()
    cd_o, cd_h = _count_distribution(KW["seed"]), _count_distribution(KW["seed"])
    og = _oracle_unit(orc, flat, cd_o)
    want = og.estimate_noise_and_genotypes()
    res_o = og.results()
    g, got = _engine(orc).estimate_noise_and_genotypes(flat, cd_h)
    res_h = g.results()
    assert np.array_equal(got, want)
    for k in ("dip_off", "h1", "h2", "freq", "cell_off", "stats"):
        assert np.array_equal(res_o[k], res_h[k]), k
    assert res_h["freq"].sum() == flat["num_clusters"] * S * KW["chains"] * KW["iters"]
    og.close(), g.close()


def test_noise_parameter_row_format():
    from bayestyper_amd.host.inference_engine import noise_parameter_row

    rates = [0.05, 1.23456789e-7, 123456789.0, 1.0]
    assert noise_parameter_row(3, 17, rates) == "3\t17\t" + "\t".join("%g" % r for r in rates) + "\n"


# ---- two ranks ------------------------------------------------------------------------------------------------
def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    import _oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from bayestyper_amd.host.inference_engine import unit_group_shape

    orc = _oracle.load_oracle()
    flat = _unit()
    parts = shard.assign_groups(shard.group_cost(flat), world)
    mine = shard.take_groups(flat, parts[rank])
    eng = _engine(orc, reduce_hist=shard.hist_reducer())
    cd = _count_distribution(KW["seed"])
    trace_noise = eng.estimate_noise(cd, mine, unit_shape=unit_group_shape(flat), variants_batch_size=12)
    final = cd.noise_rates()
    cd2 = _count_distribution(KW["seed"])
    g, trace_both = eng.estimate_noise_and_genotypes(mine, cd2)
    local = shard.summary_from_results(g.results(), mine["num_clusters"], S)
    full = shard.gather_summaries(local, shard.cluster_ids_of(flat, parts[rank]), flat["num_clusters"], rank, world)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), trace_noise=trace_noise, final=final, trace_both=trace_both,
             full=full if full is not None else np.zeros(0))
    dist.destroy_process_group()


def test_two_ranks_reproduce_the_unsharded_drivers(orc):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    flat = _unit()
    og = _oracle_unit(orc, flat, _count_distribution(KW["seed"]))
    want_noise, _, want_final = og.estimate_noise(variants_batch_size=12)
    og.close()
    og = _oracle_unit(orc, flat, _count_distribution(KW["seed"]))
    want_both = og.estimate_noise_and_genotypes()
    want_full = shard.summary_from_results(og.results(), flat["num_clusters"], S)
    og.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r = [np.load(os.path.join(d, f"rank{k}.npz")) for k in range(2)]
        for k in range(2):   # every rank holds the same rates at every iteration
            assert np.array_equal(r[k]["trace_noise"], want_noise)
            assert np.array_equal(r[k]["final"], want_final)
            assert np.array_equal(r[k]["trace_both"], want_both)
        assert np.array_equal(r[0]["full"], want_full)


def test_default_mode_in_several_launches(orc):
    """estimate_genotypes over a unit cut into launches of a few groups each == one launch (groups are independent, seeds follow the
    unit-wide group index)"""
    flat = _unit()
    eng = _engine(orc)
    cd = _count_distribution(KW["seed"])
    whole = eng.estimate_genotypes(flat, cd)
    pieces = eng.estimate_genotypes(flat, cd, max_groups_per_launch=4)
    a, b = whole.results(), pieces.results()
    for k in ("dip_off", "h1", "h2", "freq", "cell_off", "stats"):
        assert np.array_equal(a[k], b[k]), k
    assert len(pieces.parts) == -(-flat["num_groups"] // 4)
    whole.close(), pieces.close()


def test_count_distribution_generator_travels():
    """the run's generator (std::mt19937 + the gamma distribution's saved normal variate) exported after some draws and imported into another
    CountDistribution continues the same stream of noise rates — what bt_gibbs_noise_chain relies on when a chain's draws move to the device"""
    rng = np.random.default_rng(3)
    a, b = _count_distribution(11), _count_distribution(99)
    for i in range(7):   # an odd number of normal variates leaves one saved in some of these draws
        a.sample_noise_parameters(rng.integers(0, 50, S * 256).astype(np.uint64))
    words, saved = a.export_generator()
    assert words[624] <= 624 and words[625] in (0, 1)
    b.import_generator(words, saved)
    for i in range(40):
        h = rng.integers(0, 50, S * 256).astype(np.uint64)
        a.sample_noise_parameters(h)
        b.sample_noise_parameters(h)
        assert np.array_equal(a.noise_rates(), b.noise_rates())
    w2, s2 = b.export_generator()
    w1, s1 = a.export_generator()
    assert np.array_equal(w1, w2) and s1 == s2
