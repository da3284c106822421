"""CPU tests pinning the Gibbs side of the oracle: (a) component by component against the reference's own Boost-free
translation units (oracle/_ref), (b) against the known answers captured from the compiled reference,
(c) internal invariants of the restated sampler on synthetic cluster batches.  The orchestration classes
(VariantClusterGenotyper, FrequencyDistribution, ...) include Boost headers and cannot be built here: parity unpinned."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _oracle
from _oracle import _ptr

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))


@pytest.fixture(scope="module")
def orc(oracle):
    _oracle._gibbs_sigs(oracle.l)
    return oracle


def test_known_answers(orc):
    nb = GOLD["nb_moments"]
    p, size = C.c_double(), C.c_double()
    orc.l.orc_nb_moments(nb["mean"], nb["var"], C.byref(p), C.byref(size))
    assert (p.value, size.value) == (nb["p"], nb["size"])
    for obs, scale, want in GOLD["nb_logpmf"]:
        assert orc.l.orc_nb_logpmf(p.value, size.value, obs, scale) == want
    ld = GOLD["logdiscrete"]
    w = np.asarray(ld["logw"], np.float64)
    out = np.zeros(len(ld["draws"]), np.uint32)
    orc.l.orc_logdiscrete_draws(_ptr(w), len(w), ld["seed"], len(out), _ptr(out))
    assert out.tolist() == ld["draws"]
    gs = GOLD["gamma_seq"]
    a = np.asarray([x[0] for x in gs["params"]], np.float64)
    b = np.asarray([x[1] for x in gs["params"]], np.float64)
    o = np.zeros(len(a))
    orc.l.orc_rng(gs["seed"], 2, _ptr(a), _ptr(b), len(a), _ptr(o))
    assert o.tolist() == gs["values"]
    m5 = GOLD["mt19937_5"]
    o = np.zeros(10)
    ten = np.asarray([10.0])
    orc.l.orc_rng(5, 5, _ptr(ten), None, 0, _ptr(o))
    assert o.astype(int).tolist() == m5["shuffle_0_9"]
    six = np.full(6, 6.0)
    o = np.zeros(6)
    orc.l.orc_rng(5, 3, _ptr(six), None, 6, _ptr(o))
    assert o.astype(int).tolist() == m5["uniform_int_0_6_x6"]
    rate = np.asarray([0.1])
    o = np.zeros(1000)
    orc.l.orc_rng(5, 4, _ptr(rate), None, 1000, _ptr(o))
    assert int(o.sum()) == m5["bernoulli_0.1f_hits_in_1000"]
    o = np.zeros(2)
    orc.l.orc_rng(5, 1, None, None, 2, _ptr(o))
    assert o.tolist() == m5["canonical_x2"]
    for key, n in (("insert_0_9", 10), ("insert_0_13", 14)):
        ops = np.zeros(n, np.uint8)
        vals = np.arange(n, dtype=np.uint32)
        order = np.zeros(n, np.uint32)
        cnt = C.c_uint32()
        orc.l.orc_uset_replay(n, _ptr(ops), _ptr(vals), n, _ptr(order), C.byref(cnt))
        assert order[: cnt.value].tolist() == GOLD["unordered_set_order"][key]


def test_components_vs_reference(orc, ref):
    rng = np.random.default_rng(21)
    # logAddition / doubleCompare
    for _ in range(2000):
        a, b = rng.normal(-50, 40), rng.normal(-50, 40)
        assert ref.l.ref_log_addition(a, b) == orc.l.orc_log_addition(a, b)
        c = a * (1 + rng.choice([0, 1e-16, 1e-14, 1e-12]))
        assert ref.l.ref_double_compare(a, c) == orc.l.orc_double_compare(a, c)
    # NB log-pmf over the LUT domain, incl. the moments conversion with the p cap at 0.99
    for mean, var in ((15, 30), (15, 15.0001), (40.5, 400), (3, 2)):
        p, s, p2, s2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        ref.l.ref_nb_moments(mean, var, C.byref(p), C.byref(s))
        orc.l.orc_nb_moments(mean, var, C.byref(p2), C.byref(s2))
        assert (p.value, s.value) == (p2.value, s2.value)
        for obs in (0, 1, 7, 100, 255, 300, 1000):
            for scale in (1, 2, 5, 64, 255):
                assert ref.l.ref_nb_logpmf(p.value, s.value, obs, scale) == orc.l.orc_nb_logpmf(p.value, s.value, obs, scale)
    # LogDiscreteSampler / DiscreteSampler draw streams
    for n in (1, 2, 3, 10, 55, 528):
        w = rng.normal(-30, 15, n)
        for seed in (1, 42, 12345):
            o1, o2 = np.zeros(200, np.uint32), np.zeros(200, np.uint32)
            ref.l.ref_logdiscrete_draws(_ptr(w), n, seed, 200, _ptr(o1))
            orc.l.orc_logdiscrete_draws(_ptr(w), n, seed, 200, _ptr(o2))
            assert np.array_equal(o1, o2)
            pw = rng.random(n) + 0.01
            ref.l.ref_discrete_draws(_ptr(pw), n, seed, 200, _ptr(o1))
            orc.l.orc_discrete_draws(_ptr(pw), n, seed, 200, _ptr(o2))
            assert np.array_equal(o1, o2)
    # KmerStats (Welford)
    for n in (1, 2, 17, 500):
        v = rng.choice([0.0, 0.5, 7.25, 15.0, 31.0], n)
        r, o = [], []
        for lib_, fn, acc in ((ref.l, "ref_kmerstats", r), (orc.l, "orc_kmerstats", o)):
            cnt, fr, me, va = C.c_uint(), C.c_double(), C.c_double(), C.c_double()
            getattr(lib_, fn)(_ptr(v), n, C.byref(cnt), C.byref(fr), C.byref(me), C.byref(va))
            acc.extend([cnt.value, fr.value, me.value, va.value])
        assert r == o
    # SparsityEstimator: randomised greedy cover incl. its draw stream
    for (rows, cols) in ((110, 2), (440, 10), (300, 32), (64, 7)):
        for seed in (3, 99):
            M = (rng.random((rows, cols)) < 0.3).astype(np.uint8) * rng.integers(1, 3, (rows, cols)).astype(np.uint8)
            M[M.sum(axis=1) == 0, 0] = 1
            mask = (rng.random(rows) < 0.7).astype(np.uint8)
            o1, o2 = np.zeros(cols, np.uint32), np.zeros(cols, np.uint32)
            n1 = ref.l.ref_sparsity_cover(_ptr(M), rows, cols, _ptr(mask), seed, _ptr(o1))
            n2 = orc.l.orc_sparsity_cover(_ptr(M), rows, cols, _ptr(mask), seed, _ptr(o2))
            assert n1 == n2 and np.array_equal(o1[:n1], o2[:n2])


def test_lut_properties(orc):
    g, n = _oracle.build_luts(orc, 2, noise_rate=[0.05, 0.5])
    g = g.reshape(2, 256, 256)
    n = n.reshape(2, 256)
    # each row is a proper distribution with the tail folded into count 255 (CountDistribution.cpp:285-306,314-347)
    assert np.allclose(np.exp(g[:, 1:, :]).sum(axis=2), 1.0, atol=1e-9)
    assert np.allclose(np.exp(n).sum(axis=1), 1.0, atol=1e-12)
    assert (g[:, 1:, :] <= 0).all() and (n <= 0).all()
    assert g[0, 0, 0] == 0 and np.isneginf(g[0, 0, 1:]).all()
    assert g[0, 1, 10] == GOLD["nb_logpmf"][0][2] and g[0, 2, 35] == GOLD["nb_logpmf"][1][2]


@pytest.mark.parametrize("shape,n,S", [("A", 12, 3), ("B", 4, 2), ("C", 2, 2)])
def test_sampler_invariants(orc, shape, n, S):
    """conservation: every collected sweep adds exactly one diplotype per (cluster, sample); determinism; thread-count independence"""
    from bayestyper_amd import synth

    flat = synth.make_batch(shape, n, S, seed=7)
    g, nz = _oracle.build_luts(orc, S)
    res = []
    for threads in (1, 3):
        og = _oracle.OrcGibbs(orc, flat, g, nz, chains=3, burn=10, iters=25)
        og.run(threads)
        res.append(og.results())
        og.close()
    r = res[0]
    for k in ("dip_off", "h1", "h2", "freq", "stats"):
        assert np.array_equal(r[k], res[1][k])
    for c in range(flat["num_clusters"]):
        e0, e1 = int(r["dip_off"][c]), int(r["dip_off"][c + 1])
        assert (r["freq"][e0:e1].sum(axis=0) == 3 * 25).all()
        assert (r["h1"][e0:e1] <= r["h2"][e0:e1]).all()
