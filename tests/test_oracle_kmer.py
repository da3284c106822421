"""CPU tests: the k-mer side of the oracle is pinned (a) against the known answers captured from the compiled
reference (tests/golden/known_answers.json) and (b) bit-for-bit against the reference's own
translation units compiled unmodified (oracle/_ref/libbtref.so; skipped when that library has not been built)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _oracle
from _oracle import OrcBloom, OrcKmc, OrcTable, _ptr

K = 55
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")))


def test_known_answers_nthash(oracle):
    km = GOLD["kmer"]
    assert len(km) == K
    assert oracle.l.orc_ntp64(km.encode(), K) == int(GOLD["ntp64"], 16)
    assert oracle.l.orc_ntp64_seed(km.encode(), K, 1029283129) == int(GOLD["ntp64_seed_1029283129"], 16)
    b = OrcBloom(oracle, 1000, 1e-4, K, threaded=True)
    assert b.route(km) == GOLD["threaded_bloom_route"]
    b.close()
    # the k-mer is its own canonical form
    kmers, valid = oracle.kmers_from_sequence(km.encode(), K)
    assert valid[K - 1] == 1 and valid[: K - 1].sum() == 0
    assert oracle.unpack(kmers[K - 1:K], K).tobytes().decode() == km


def test_known_answers_bloom_sizing(oracle):
    for n, fpr, bits, hashes in GOLD["bloom_sizing"]:
        assert oracle.bloom_sizing(n, fpr) == (bits, hashes)


def test_nthash_vs_reference(oracle, ref):
    assert ref.k == K
    rng = np.random.default_rng(1)
    km = _oracle.random_kmers(rng, 2000, K)
    h = oracle.ntp64(km, K)
    hs = oracle.ntp64(km, K, seed=1029283129)
    for i in range(0, 2000, 7):
        s = km[i * K:(i + 1) * K].tobytes()
        assert ref.l.ref_ntp64(s) == h[i]
        assert ref.l.ref_ntp64_seed(s, 1029283129) == hs[i]


@pytest.mark.parametrize("n,fpr", [(1000, 1e-4), (1000, 1e-3), (1, 1e-3), (123457, 1e-4), (50_000_000, 1e-4), (3_000_000_000, 1e-3)])
def test_bloom_sizing_vs_reference(oracle, ref, n, fpr):
    b, h = C.c_uint64(), C.c_uint()
    ref.l.ref_bloom_sizing(n, fpr, C.byref(b), C.byref(h))
    assert oracle.bloom_sizing(n, fpr) == (b.value, h.value)


@pytest.mark.parametrize("n,fpr", [(1000, 1e-4), (1000, 1e-3), (5000, 1e-2)])
def test_kmerbloom_bit_image_vs_reference(oracle, ref, tmp_path, n, fpr):
    """KmerBloom: identical .bloomMeta/.bloomData bytes and identical lookups incl. false positives"""
    rng = np.random.default_rng(2)
    members = _oracle.random_kmers(rng, n, K)
    probes = np.concatenate([members[: 200 * K], _oracle.random_kmers(rng, 20000, K)])
    rb = ref.l.ref_kmerbloom_new(n, fpr)
    ref.l.ref_kmerbloom_add(rb, _ptr(members), n)
    ref.l.ref_kmerbloom_save(rb, str(tmp_path / "ref").encode())
    ob = OrcBloom(oracle, n, fpr, K)
    ob.insert(members)
    ob.save(str(tmp_path / "orc"))
    for ext in (".bloomMeta", ".bloomData"):
        assert open(tmp_path / ("ref" + ext), "rb").read() == open(tmp_path / ("orc" + ext), "rb").read()
    hits_ref = np.zeros(len(probes) // K, dtype=np.uint8)
    ref.l.ref_kmerbloom_lookup(rb, _ptr(probes), len(hits_ref), _ptr(hits_ref))
    assert np.array_equal(hits_ref, ob.contains(probes))
    assert hits_ref[:200].all()
    # the bitset overloads of the reference agree with the packed representation the product uses
    packed = oracle.pack(probes[: 500 * K], K)
    hp = np.zeros(500, dtype=np.uint8)
    ref.l.ref_kmerbloom_lookup_packed(rb, _ptr(packed), 500, _ptr(hp))
    assert np.array_equal(hp, hits_ref[:500])
    # load round trip
    ob2 = OrcBloom.load(oracle, str(tmp_path / "ref"), K)
    assert np.array_equal(ob2.bits(), ob.bits())
    ref.l.ref_kmerbloom_free(rb)
    ob.close()
    ob2.close()


def test_threaded_bloom_vs_reference(oracle, ref):
    """ThreadedKmerBloom: 65 536 sub-filters; tiny per-filter size makes false positives common, so identical
    lookups pin routing + sizing + probe sequence"""
    rng = np.random.default_rng(3)
    n = 300_000
    members = _oracle.random_kmers(rng, n, K)
    probes = np.concatenate([members[: 500 * K], _oracle.random_kmers(rng, 60000, K)])
    rb = ref.l.ref_tbloom_new(n, 1e-2)
    ref.l.ref_tbloom_add(rb, _ptr(members), n)
    ob = OrcBloom(oracle, n, 1e-2, K, threaded=True)
    ob.insert(members)
    hr = np.zeros(len(probes) // K, dtype=np.uint8)
    ref.l.ref_tbloom_lookup(rb, _ptr(probes), len(hr), _ptr(hr))
    ho = ob.contains(probes)
    assert np.array_equal(hr, ho)
    assert hr[:500].all() and 0 < hr[500:].sum() < 0.2 * 60000
    ref.l.ref_tbloom_free(rb)
    ob.close()


def test_canonical_kmers_vs_reference(oracle, ref):
    rng = np.random.default_rng(4)
    seq = np.frombuffer(b"ACGTacgtNnRX", dtype=np.uint8)[rng.choice(12, size=6000, p=[0.22] * 4 + [0.025] * 4 + [0.005] * 4)].copy()
    seq[100:300] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200)]   # long clean stretch
    # a palindromic stretch (its own reverse complement) exercises the tie rule
    pal = b"ACGT" * 20
    seq[1000:1080] = np.frombuffer(pal, dtype=np.uint8)
    km_o, v_o = oracle.kmers_from_sequence(seq.tobytes(), K)
    km_r = np.zeros_like(km_o)
    v_r = np.zeros_like(v_o)
    ref.l.ref_kmers_from_sequence(_ptr(seq), len(seq), _ptr(km_r), _ptr(v_r))
    assert np.array_equal(v_o, v_r)
    assert np.array_equal(km_o, km_r)
    assert v_o.sum() > 100


def make_kmc(oracle, tmp_path, rng, n, p=7, counter_size=1, name="db"):
    km = _oracle.random_kmers(rng, n, K)
    km = _oracle.canonical_ascii(oracle, km, K).reshape(n, K)
    km = np.unique(km, axis=0)
    counts = np.minimum(1 + rng.geometric(0.5, size=len(km)), 255).astype(np.uint32)
    prefix = str(tmp_path / name)
    oracle.kmc_write(prefix, np.ascontiguousarray(km).reshape(-1), counts, K, p, counter_size)
    return prefix, np.ascontiguousarray(km).reshape(-1), counts


@pytest.mark.parametrize("p,cs", [(7, 1), (3, 2), (7, 4)])
def test_kmc_reader_vs_reference(oracle, ref, tmp_path, p, cs):
    """the reference's CKMCFile reads the database our writer produced, and our reader lists the same records"""
    rng = np.random.default_rng(5)
    prefix, km, counts = make_kmc(oracle, tmp_path, rng, 3000, p, cs)
    k, mode, c, pp = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
    total = ref.l.ref_kmc_total(prefix.encode(), C.byref(k), C.byref(mode), C.byref(c), C.byref(pp))
    assert (total, k.value, mode.value, c.value, pp.value) == (len(counts), K, 0, cs, p)
    rk = np.zeros(total * K, dtype=np.uint8)
    rc = np.zeros(total, dtype=np.uint32)
    assert ref.l.ref_kmc_list(prefix.encode(), _ptr(rk), _ptr(rc), total) == total
    assert np.array_equal(rk, km) and np.array_equal(rc, counts)
    db = OrcKmc(oracle, prefix)
    ok, oc = db.list()
    assert (db.k, db.p, db.counter_size, db.total) == (K, p, cs, total)
    assert np.array_equal(ok, km) and np.array_equal(oc, counts)
    db.close()


def test_kmercounts_arithmetic_vs_reference(oracle, ref):
    """saturating u8 semantics of KmerCounts / ObservedKmerCounts against the reference class, via the oracle table"""
    rng = np.random.default_rng(6)
    km = "ACGTACGTTGCAAGCTTAGCCATGGATCCGATTACAGGCTTAACGGTCATGCAAT"
    for trial in range(30):
        kc = ref.l.ref_kc_new()
        t = OrcTable(oracle, 30, K)
        bloom_all = OrcBloom(oracle, 10, 1e-3, K)
        bloom_all.insert([km])
        bloom_none = OrcBloom(oracle, 10, 1e-3, K)
        mg = int(rng.random() < 0.3)   # the reference asserts the multigroup flag is consistent per k-mer (KmerCounts.cpp:152)
        for _ in range(rng.integers(1, 300)):
            op = rng.integers(0, 3)
            if op == 0:
                decoy, fp, mp = int(rng.random() < 0.1), int(rng.integers(0, 3)), int(rng.integers(0, 3))
                ref.l.ref_kc_add_intercluster(kc, decoy, fp, mp)
                t.count_intercluster(bloom_all, km.encode(), decoy, fp, mp)
            elif op == 1:
                mult = int(rng.integers(1, 120))
                ref.l.ref_kc_add_cluster(kc, mult, mg)
                t.classify(bloom_all if mg else bloom_none, [km], [mult])
        meta, counts, ex = np.zeros(4, np.uint8), np.zeros(30, np.uint8), np.zeros(1, np.uint8)
        ref.l.ref_kc_get(kc, _ptr(meta), _ptr(counts), _ptr(ex))
        _, oc, om = t.export()
        if len(om):
            assert om[0, 0] == meta[0] and om[0, 2] == meta[2] and om[0, 3] == meta[3]
        else:
            assert meta[0] == 0
        ref.l.ref_kc_free(kc)
        for x in (t, bloom_all, bloom_none):
            x.close()
    # addSampleCount saturation
    kc = ref.l.ref_kc_new()
    tot = 0
    for c in [100, 100, 54, 1, 200]:
        ref.l.ref_kc_add_sample_count(kc, 3, c)
        tot = min(255, tot + c)
        meta, counts, ex = np.zeros(4, np.uint8), np.zeros(30, np.uint8), np.zeros(1, np.uint8)
        ref.l.ref_kc_get(kc, _ptr(meta), _ptr(counts), _ptr(ex))
        assert counts[3] == tot
    ref.l.ref_kc_free(kc)


def test_parse_sample_kmers_oracle_consistency(oracle, tmp_path):
    """KMC scan in the oracle: every db k-mer that is in the path set is counted; Bloom false positives also enter"""
    rng = np.random.default_rng(7)
    prefix, km, counts = make_kmc(oracle, tmp_path, rng, 20000)
    n = len(counts)
    path_idx = rng.choice(n, size=2000, replace=False)
    path = km.reshape(n, K)[path_idx]
    bloom = OrcBloom(oracle, 2000 + 1000, 1e-2, K, threaded=True)
    bloom.insert(np.ascontiguousarray(path).reshape(-1))
    t = OrcTable(oracle, 3, K)
    db = OrcKmc(oracle, prefix)
    hits = t.parse_sample_kmers(bloom, db, 1)
    pk, pc, pm = t.export()
    assert hits == len(pk) >= 2000
    want = dict(zip(map(bytes, oracle.pack(np.ascontiguousarray(path).reshape(-1), K)), counts[path_idx]))
    got = {bytes(k): c[1] for k, c in zip(pk, pc)}
    for k_, c_ in want.items():
        assert got[k_] == c_
    assert (pc[:, 0] == 0).all() and (pc[:, 2] == 0).all()
    for x in (t, bloom, db):
        x.close()


@pytest.mark.parametrize("nbins", [1, 5])
def test_kmc2_database_layout_vs_reference(oracle, ref, tmp_path, nbins):
    """KMC2 ("0x200") databases — one prefix table per signature bin: the reference's CKMCFile lists what the test writer wrote
    (pins the layout), and the oracle's reader lists the same records in the same order."""
    rng = np.random.default_rng(17)
    km = np.unique(_oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 3000, K), K).reshape(-1, K), axis=0)
    cnt = rng.integers(1, 250, len(km)).astype(np.uint32)
    prefix = str(tmp_path / "db2")
    oracle.kmc2_write(prefix, np.ascontiguousarray(km).reshape(-1), cnt, K, 3, 1, nbins)
    k, mode, c, pp = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
    total = ref.l.ref_kmc_total(prefix.encode(), C.byref(k), C.byref(mode), C.byref(c), C.byref(pp))
    assert total == len(km) and (k.value, mode.value, c.value, pp.value) == (K, 0, 1, 3)
    rk = np.zeros(total * K, np.uint8)
    rc = np.zeros(total, np.uint32)
    assert ref.l.ref_kmc_list(prefix.encode(), _ptr(rk), _ptr(rc), total) == total
    db = _oracle.OrcKmc(oracle, prefix)
    ok, oc = db.list()
    db.close()
    assert np.array_equal(ok.reshape(-1), rk) and np.array_equal(oc, rc)
    # same multiset as written (bins reorder the records)
    order = np.lexsort(rk.reshape(-1, K).T[::-1])
    assert np.array_equal(rk.reshape(-1, K)[order], km) and np.array_equal(rc[order], cnt)
