"""GPU parity tests (through the C ABI of libbtgpu.so) for the k-mer side of the hot path:
ntHash, canonical k-mers, KmerBloom / ThreadedKmerBloom, the count table and the KMC scan.
Bar: bit-exact against the oracle (which is itself pinned against the compiled reference, tests/test_oracle_kmer.py)."""
import os

import numpy as np
import pytest

import _oracle
from _oracle import OrcBloom, OrcKmc, OrcTable

pytestmark = pytest.mark.gpu
K = 55


def _sorted_export(kmers, counts, meta):
    order = np.lexsort((kmers[:, 0], kmers[:, 1]))
    return kmers[order], counts[order], meta[order]


def test_nthash_and_canonical(gpu_ctx, oracle):
    from bayestyper_amd import lib

    rng = np.random.default_rng(11)
    km = _oracle.random_kmers(rng, 50_000, K)
    packed = oracle.pack(km, K)
    d = gpu_ctx.to_device(packed)
    o = gpu_ctx.buffer(8 * len(packed))
    for seeded, seed in ((0, 0), (1, 1029283129), (1, 7)):
        lib.check(lib.bt_nthash_batch(gpu_ctx.h, d.ptr, len(packed), K, seeded, seed, o.ptr))
        gpu_ctx.sync()
        got = o.download(np.uint64, len(packed))
        want = oracle.ntp64(km, K, seed=seed if seeded else None)
        assert np.array_equal(got, want)
    d.free(), o.free()

    # canonical k-mers of a messy sequence (N runs, lower case, palindromes, ragged tail, empty)
    seq = np.frombuffer(b"ACGTacgtNnRX", dtype=np.uint8)[rng.choice(12, size=70_001, p=[0.22] * 4 + [0.025] * 4 + [0.005] * 4)].copy()
    seq[5000:5400] = np.frombuffer(b"ACGT" * 100, dtype=np.uint8)
    for n in (0, 1, K - 1, K, K + 1, 1023, 1024, 1025, len(seq)):
        s = seq[:n]
        want_k, want_v = oracle.kmers_from_sequence(s.tobytes(), K)
        ds = gpu_ctx.to_device(s if n else np.zeros(1, np.uint8))
        dk, dv = gpu_ctx.buffer(16 * max(n, 1)), gpu_ctx.buffer(max(n, 1))
        lib.check(lib.bt_kmers_from_sequence(gpu_ctx.h, ds.ptr, n, K, dk.ptr, dv.ptr))
        gpu_ctx.sync()
        got_k = dk.download(np.uint64, 2 * n).reshape(n, 2)
        got_v = dv.download(np.uint8, n)
        assert np.array_equal(got_v, want_v)
        assert np.array_equal(got_k[want_v == 1], want_k[want_v == 1])
        for b in (ds, dk, dv):
            b.free()


@pytest.mark.parametrize("k", [15, 31, 32, 33, 55, 64])
def test_other_kmer_sizes(gpu_ctx, oracle, k):
    """the reference fixes k at compile time (BT_KMER_SIZE); the library takes it at run time, k <= 64"""
    from bayestyper_amd import lib

    rng = np.random.default_rng(k)
    seq = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, size=5000, p=[0.2475] * 4 + [0.01])].copy()
    want_k, want_v = oracle.kmers_from_sequence(seq.tobytes(), k)
    ds = gpu_ctx.to_device(seq)
    dk, dv = gpu_ctx.buffer(16 * len(seq)), gpu_ctx.buffer(len(seq))
    lib.check(lib.bt_kmers_from_sequence(gpu_ctx.h, ds.ptr, len(seq), k, dk.ptr, dv.ptr))
    gpu_ctx.sync()
    got_k = dk.download(np.uint64, 2 * len(seq)).reshape(-1, 2)
    got_v = dv.download(np.uint8, len(seq))
    assert np.array_equal(got_v, want_v) and np.array_equal(got_k[want_v == 1], want_k[want_v == 1])
    valid = want_k[want_v == 1]
    o = gpu_ctx.buffer(8 * len(valid))
    dvk = gpu_ctx.to_device(valid)
    lib.check(lib.bt_nthash_batch(gpu_ctx.h, dvk.ptr, len(valid), k, 1, 99, o.ptr))
    gpu_ctx.sync()
    assert np.array_equal(o.download(np.uint64, len(valid)), oracle.ntp64(oracle.unpack(valid, k), k, seed=99))
    for b in (ds, dk, dv, o, dvk):
        b.free()


@pytest.mark.parametrize("n,fpr", [(1000, 1e-4), (1000, 1e-3), (1, 1e-3), (40_000, 1e-2)])
def test_kmerbloom_bit_exact(gpu_ctx, oracle, tmp_path, n, fpr):
    from bayestyper_amd import lib

    rng = np.random.default_rng(12)
    members = _oracle.random_kmers(rng, n, K)
    probes = np.concatenate([members[: min(n, 300) * K], _oracle.random_kmers(rng, 30_000, K)])
    ob = OrcBloom(oracle, n, fpr, K)
    ob.insert(members)
    gb = lib.Bloom.create(gpu_ctx, n, fpr, K, threaded=False)
    gi, oi = gb.info(), ob.info()
    assert (gi["num_kmers"], gi["num_bits"], gi["num_hashes"], gi["num_sub"]) == (oi["num_kmers"], oi["num_bits"], oi["num_hashes"], 1)
    gb.insert(oracle.pack(members, K))
    assert np.array_equal(gb.bits(), ob.bits())
    assert np.array_equal(gb.contains(oracle.pack(probes, K)), ob.contains(probes))
    # save -> byte-identical files; load (ours and the oracle's) -> identical filters
    gb.save(str(tmp_path / "gpu"))
    ob.save(str(tmp_path / "orc"))
    for ext in (".bloomMeta", ".bloomData"):
        assert open(tmp_path / ("gpu" + ext), "rb").read() == open(tmp_path / ("orc" + ext), "rb").read()
    gb2 = lib.Bloom.load(gpu_ctx, str(tmp_path / "orc"), K)
    assert np.array_equal(gb2.bits(), ob.bits())
    assert np.array_equal(gb2.contains(oracle.pack(probes, K)), ob.contains(probes))
    with pytest.raises(lib.BtError):
        lib.Bloom.load(gpu_ctx, str(tmp_path / "orc"), 31)   # k mismatch (the reference asserts)
    with pytest.raises(lib.BtError):
        lib.Bloom.load(gpu_ctx, str(tmp_path / "missing"), K)
    for x in (gb, gb2, ob):
        x.close()


def test_threaded_bloom_bit_exact(gpu_ctx, oracle):
    from bayestyper_amd import lib

    rng = np.random.default_rng(13)
    n = 400_000
    members = _oracle.random_kmers(rng, n, K)
    probes = np.concatenate([members[: 1000 * K], _oracle.random_kmers(rng, 200_000, K)])
    ob = OrcBloom(oracle, n, 1e-2, K, threaded=True)
    ob.insert(members)
    gb = lib.Bloom.create(gpu_ctx, n, 1e-2, K, threaded=True)
    gi, oi = gb.info(), ob.info()
    assert (gi["num_kmers"], gi["num_bits"], gi["num_hashes"], gi["num_sub"]) == (oi["num_kmers"], oi["num_bits"], oi["num_hashes"], 65536)
    gb.insert(oracle.pack(members, K))
    for sub in (0, 1, 39854, 65535):
        assert np.array_equal(gb.bits(sub), ob.bits(sub))
    hg, ho = gb.contains(oracle.pack(probes, K)), ob.contains(probes)
    assert np.array_equal(hg, ho)
    assert ho[:1000].all() and 0 < ho[1000:].sum() < 40_000   # false positives present and identical
    # idempotence: inserting again changes nothing
    gb.insert(oracle.pack(members[: 5000 * K], K))
    assert np.array_equal(gb.contains(oracle.pack(probes, K)), ho)
    gb.close(), ob.close()


@pytest.mark.parametrize("routed", ["0", "partitioned"])
def test_kmc_decode_and_scan(gpu_ctx, oracle, tmp_path, monkeypatch, routed):
    """parseSampleKmers: decode -> path-Bloom -> table add; table contents bit-exact incl. Bloom false positives.
    partitioned: the one-pass partition by the upper route bits + probe through the caches (what a scan of a ThreadedKmerBloom gets from 2^22 records),
    forced at this small size; routed = 0: the direct kernel"""
    from bayestyper_amd import lib
    from test_oracle_kmer import make_kmc

    monkeypatch.setenv("BT_KMC_ROUTED", "0" if routed == "0" else "1")

    rng = np.random.default_rng(14)
    S = 3
    dbs = []
    for s, (p, cs) in enumerate([(7, 1), (3, 2), (7, 1)]):
        dbs.append(make_kmc(oracle, tmp_path, rng, 60_000 + 1000 * s, p, cs, name=f"s{s}"))
    # path k-mers: a slice of every sample's k-mers + k-mers in no sample
    path = np.concatenate([km.reshape(-1, K)[rng.choice(len(c), 4000, replace=False)] for _, km, c in dbs] + [_oracle.random_kmers(rng, 3000, K).reshape(-1, K)])
    path = np.unique(_oracle.canonical_ascii(oracle, np.ascontiguousarray(path).reshape(-1), K).reshape(-1, K), axis=0)
    path_flat = np.ascontiguousarray(path).reshape(-1)
    ob = OrcBloom(oracle, len(path), 1e-2, K, threaded=True)
    ob.insert(path_flat)
    gb = lib.Bloom.create(gpu_ctx, len(path), 1e-2, K, threaded=True)
    gb.insert(oracle.pack(path_flat, K))
    ot = OrcTable(oracle, S, K)
    gt = lib.Table(gpu_ctx, 40_000, S, K)
    # parameter k-mers first (main.cpp:543-584)
    param = _oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 500, K), K)
    ot.insert(param, mark_parameter=True)
    gt.insert(oracle.pack(param, K), mark_parameter=True)
    d_hits = gpu_ctx.buffer(8).zero()
    total_hits = 0
    for s, (prefix, km, counts) in enumerate(dbs):
        db = OrcKmc(oracle, prefix)
        payload = db.payload()
        scan = lib.KmcScan(gpu_ctx, db.k, db.p, db.counter_size, db.total, db.lut())
        # decode parity
        gk, gc = scan.decode(payload, 0, db.total)
        assert np.array_equal(gk, oracle.pack(km, K)) and np.array_equal(gc, counts)
        # scan in ragged chunks (chunk starts are record-aligned and 16-byte aligned)
        d_payload = gpu_ctx.to_device(payload)
        first = 0
        for chunk in (16 * 13, 16 * 1000, db.total):
            n = min(chunk, db.total - first)
            assert (first * db.rec_size) % 16 == 0
            scan.run(gb, gt, s, d_payload.ptr + first * db.rec_size, first, n, d_hits.ptr)
            first += n
        gpu_ctx.sync()
        total_hits += ot.parse_sample_kmers(ob, db, s)
        d_payload.free(), scan.close(), db.close()
    assert int(d_hits.download(np.uint64, 1)[0]) == total_hits
    assert not gt.status()["overflowed"]
    gk, gc, gm = _sorted_export(*gt.export())
    wk, wc, wm = _sorted_export(*ot.export())
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc)
    assert np.array_equal(gm, wm)
    # find: every exported key is found, random keys are not
    slots = gt.find(wk)
    assert (slots >= 0).all() and len(np.unique(slots)) == len(slots)
    assert (gt.find(oracle.pack(_oracle.random_kmers(rng, 1000, K), K)) == -1).all()
    for x in (gt, gb, ob, ot):
        x.close()


def test_table_duplicates_in_a_wavefront_clear_reserve_overflow(gpu_ctx, oracle):
    """addKmer under contention: every key 64 times in a row (the lanes of one wavefront race for one slot), colliding home slots in a
    small table; bt_table_reserve moves every record (keys, flags, counts) into a larger table; bt_table_clear empties it; a table
    that is too small reports the overflow instead of silently dropping records."""
    from bayestyper_amd import lib

    rng = np.random.default_rng(99)
    uniq = oracle.pack(_oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 3000, K), K), K)
    uniq = np.unique(uniq, axis=0)
    t = lib.Table(gpu_ctx, 2000, 2, K)                      # capacity 4096: load 0.73, long probe chains
    t.insert(np.repeat(uniq, 64, axis=0), mark_parameter=True)
    st = t.status()
    assert st["num_keys"] == len(uniq) and not st["overflowed"] and st["capacity"] == 4096
    k0, c0, m0 = _sorted_export(*t.export())
    assert np.array_equal(k0, _sorted_export(uniq, np.zeros((len(uniq), 2), np.uint8), np.zeros((len(uniq), 4), np.uint8))[0])
    slots = t.find(uniq)
    assert (slots >= 0).all() and len(np.unique(slots)) == len(uniq)
    # grow: same records, new slots
    t.reserve(100_000)
    st = t.status()
    assert st["capacity"] == 262144 and st["num_keys"] == len(uniq)
    k1, c1, m1 = _sorted_export(*t.export())
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1) and np.array_equal(m0, m1) and (m1[:, 0] & 0x20).all()
    assert (t.find(uniq) >= 0).all()
    t.clear()
    assert t.status()["num_keys"] == 0 and (t.find(uniq) == -1).all()
    t.insert(uniq[:10])
    assert t.status()["num_keys"] == 10
    t.close()
    # overflow is reported, and reserve refuses a table that already lost records
    small = lib.Table(gpu_ctx, 100, 1, K)                   # capacity 1024
    small.insert(uniq)
    st = small.status()
    assert st["overflowed"] and st["num_keys"] <= 1024
    with pytest.raises(RuntimeError):
        small.reserve(10_000)
    small.close()


@pytest.mark.parametrize("n_samples", [3, 10, 30])
def test_table_same_keys_from_every_xcd(gpu_ctx, n_samples):
    """addKmer when the SAME keys arrive from workgroups all over the chip at the same time (the key set repeated 32 times in one batch: the copies of a
    key sit megabytes apart, i.e. in workgroups on different XCDs with their own L2): a key is published with ordered write-through stores (key words
    acknowledged before READY is stored) and probed with agent-scope loads — no key may end up in two slots, none may be lost; repeated on fresh tables
    because the race is a matter of timing.  Three samples = 32-byte slots; ten = 48-byte slots, whose state and key words straddle 64 / 128 / 256-byte
    boundaries (different channels); thirty = 64-byte slots.  The key set holds the keys a stale (zero) key word could be mistaken for: the all-A k-mer
    (0, 0), keys with a zero high word (k-mers ending in 23 A's) and keys with a zero low word, next to keys that share the other word with them."""
    from bayestyper_amd import lib

    rng = np.random.default_rng(77 + n_samples)
    lo = rng.integers(1, 2 ** 62, 150_000, dtype=np.uint64)
    hi = rng.integers(1, 2 ** 46, 150_000, dtype=np.uint64)
    zero = np.zeros(20_000, dtype=np.uint64)
    keys = np.concatenate([np.stack([lo, hi], axis=1),
                           np.stack([lo[:20_000], zero], axis=1),            # (L, 0) next to (L, H)
                           np.stack([zero, hi[:20_000]], axis=1),            # (0, H) next to (L, H)
                           np.zeros((1, 2), dtype=np.uint64)])               # poly-A
    uniq = np.unique(keys, axis=0)
    batch = np.ascontiguousarray(np.tile(uniq, (32, 1)))
    want = uniq[np.lexsort((uniq[:, 0], uniq[:, 1]))]
    for rep in range(5):
        t = lib.Table(gpu_ctx, 260_000, n_samples, K)               # capacity 2^19: load 0.36
        t.insert(batch[rng.permutation(len(batch))] if rep else batch)
        st = t.status()
        assert st["num_keys"] == len(uniq) and not st["overflowed"], (rep, st)
        k, _, _ = t.export()
        assert np.array_equal(k[np.lexsort((k[:, 0], k[:, 1]))], want)
        slots = t.find(uniq)
        assert (slots >= 0).all() and len(np.unique(slots)) == len(uniq)
        t.close()


def test_kmc_counter_range_filter(gpu_ctx, oracle, tmp_path):
    """CKMCFile::ReadNextKmer skips records whose counter is outside the header's [min_count, max_count] (kmc_file.cpp:496-511):
    a scan with the range set equals a scan of a database that holds only the in-range records; same for makeBloom"""
    from bayestyper_amd import lib
    from test_oracle_kmer import make_kmc

    rng = np.random.default_rng(5)
    prefix, km, counts = make_kmc(oracle, tmp_path, rng, 30_000, 7, 1, name="full")
    lo, hi = 2, 6
    keep = (counts >= lo) & (counts <= hi)
    assert 0 < keep.sum() < len(counts)
    sub = str(tmp_path / "sub")
    oracle.kmc_write(sub, np.ascontiguousarray(km.reshape(-1, K)[keep]).reshape(-1), counts[keep].astype(np.uint32), K, 7, 1)
    gb = lib.Bloom.create(gpu_ctx, len(counts), 1e-3, K, threaded=True)
    gb.insert(oracle.pack(km, K))
    tables, blooms = [], []
    for pref, rng_ in ((prefix, (lo, hi)), (sub, None)):
        db = OrcKmc(oracle, pref)
        scan = lib.KmcScan(gpu_ctx, db.k, db.p, db.counter_size, db.total, db.lut())
        if rng_:
            scan.set_count_range(*rng_)
        t = lib.Table(gpu_ctx, 40_000, 1, K)
        d = gpu_ctx.to_device(db.payload())
        scan.run(gb, t, 0, d.ptr, 0, db.total)
        sb = lib.Bloom.create(gpu_ctx, 30_000, 1e-3, K, threaded=False)
        scan.make_bloom(sb, d.ptr, 0, db.total)
        gpu_ctx.sync()
        tables.append(_sorted_export(*t.export()))
        blooms.append(sb.bits(0))
        for x in (t, sb, scan, db):
            x.close()
        d.free()
    assert len(tables[0][0]) == keep.sum()
    for a, b in zip(tables[0], tables[1]):
        assert np.array_equal(a, b)
    assert np.array_equal(blooms[0], blooms[1])
    gb.close()


def test_intercluster_and_classify(gpu_ctx, oracle):
    """countInterclusterKmers + the table half of classifyPathKmers: flags and multiplicities bit-exact"""
    from bayestyper_amd import lib

    rng = np.random.default_rng(15)
    S = 2
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 120_000)].copy()
    genome[30_000:30_020] = ord("N")
    genome[50_000:58_000] = genome[10_000:18_000]          # a repeat -> multiplicities 2+
    genome[90_000:90_400] = np.frombuffer(b"A" * 400, dtype=np.uint8)   # poly-A -> one k-mer > 127 times
    regions = [(0, 40_000, 0, 2, 1), (40_000, 100_000, 0, 2, 2), (100_000, 120_000, 1, 0, 0)]   # (start, end, decoy, f, m)
    # path set = k-mers of some windows of the genome
    path_k, path_v = oracle.kmers_from_sequence(genome[9_000:19_000].tobytes(), K)
    pa_k, pa_v = oracle.kmers_from_sequence(genome[89_900:90_500].tobytes(), K)
    dk_k, dk_v = oracle.kmers_from_sequence(genome[100_500:101_000].tobytes(), K)
    path = np.unique(np.concatenate([path_k[path_v == 1], pa_k[pa_v == 1], dk_k[dk_v == 1]]), axis=0)
    path_ascii = oracle.unpack(path, K)
    ob = OrcBloom(oracle, len(path), 1e-3, K, threaded=True)
    ob.insert(path_ascii)
    gb = lib.Bloom.create(gpu_ctx, len(path), 1e-3, K, threaded=True)
    gb.insert(path)
    ot, gt = OrcTable(oracle, S, K), lib.Table(gpu_ctx, 30_000, S, K)
    for (a, b, decoy, f, m) in regions:
        ot.count_intercluster(ob, genome[a:b].tobytes(), decoy, f, m)
        gt.count_intercluster(gb, genome[a:b].tobytes(), decoy, f, m)
    gk, gc, gm = _sorted_export(*gt.export())
    wk, wc, wm = _sorted_export(*ot.export())
    assert np.array_equal(gk, wk) and np.array_equal(gm, wm)
    assert (wm[:, 0] & 0x10).any() and (wm[:, 0] & 0x08).any() and (wm[:, 2] >= 4).any()   # max-mult, decoy and repeats exercised
    # classify: path k-mers with multiplicities; a multigroup Bloom holding a subset; some k-mers twice (two clusters)
    mg_members = path_ascii.reshape(-1, K)[rng.choice(len(path), 50, replace=False)]
    omg = OrcBloom(oracle, 50, 1e-4, K)
    omg.insert(np.ascontiguousarray(mg_members).reshape(-1))
    gmg = lib.Bloom.create(gpu_ctx, 50, 1e-4, K, threaded=False)
    gmg.insert(oracle.pack(np.ascontiguousarray(mg_members).reshape(-1), K))
    extra = _oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 300, K), K)      # never seen: only enters when mult > 127
    cls_ascii = np.concatenate([path_ascii, path_ascii[: 100 * K], extra])
    mult = np.concatenate([rng.integers(1, 4, len(path)), rng.integers(1, 4, 100), rng.choice([1, 2, 128, 200], 300)]).astype(np.uint8)
    # order-dependence: the oracle processes in order; keep duplicates' multiplicities <= 127 so the result is order-free
    ex_o = ot.classify(omg, cls_ascii, mult)
    ex_g = gt.classify(gmg, oracle.pack(cls_ascii, K), mult)
    gk, gc, gm = _sorted_export(*gt.export())
    wk, wc, wm = _sorted_export(*ot.export())
    assert np.array_equal(gk, wk) and np.array_equal(gm, wm)
    # isExcluded is reported after each update; for duplicated k-mers compare only the final state
    uniq = np.ones(len(mult), bool)
    uniq[:100] = False
    uniq[len(path):len(path) + 100] = False
    assert np.array_equal(ex_o[uniq], ex_g[uniq])
    assert ((wm[:, 0] & 0x02) != 0).sum() == 100 and ((wm[:, 0] & 0x04) != 0).sum() >= 50
    for x in (gt, gb, gmg, ob, omg, ot):
        x.close()


def test_large_scan_properties(gpu_ctx, oracle):
    """size-independent properties at a size the scalar oracle would take minutes for: every inserted path k-mer is
    found again with its count, hit count >= members, a second identical scan saturates counts exactly"""
    from bayestyper_amd import lib

    rng = np.random.default_rng(16)
    n, p = 3_000_000, 7
    # synthetic sorted unique database built directly in packed KMC form: random suffix bytes under sorted prefixes
    prefixes = np.sort(rng.integers(0, 4 ** p, size=n))
    suffix = rng.integers(0, 256, size=(n, 12), dtype=np.uint8)
    counts = rng.integers(1, 201, size=n).astype(np.uint8)
    rec = np.concatenate([suffix, counts[:, None]], axis=1)
    lut = np.searchsorted(prefixes, np.arange(4 ** p + 1)).astype(np.uint64)
    scan = lib.KmcScan(gpu_ctx, K, p, 1, n, lut)
    gk, gc = scan.decode(rec.reshape(-1), 0, n)
    assert np.array_equal(gc, counts.astype(np.uint32))
    members = rng.choice(n, 100_000, replace=False)
    mk = np.unique(gk[members], axis=0)
    bloom = lib.Bloom.create(gpu_ctx, len(mk) + 1_000_000, 1e-4, K, threaded=True)
    bloom.insert(mk)
    table = lib.Table(gpu_ctx, 300_000, 1, K)
    d_rec = gpu_ctx.to_device(rec.reshape(-1))
    d_hits = gpu_ctx.buffer(8).zero()
    scan.run(bloom, table, 0, d_rec.ptr, 0, n, d_hits.ptr)
    gpu_ctx.sync()
    hits = int(d_hits.download(np.uint64, 1)[0])
    st = table.status()
    assert hits >= len(members) and not st["overflowed"]
    ek, ec, em = table.export()
    lookup = {bytes(k): int(c[0]) for k, c in zip(ek, ec)}
    # duplicates in the random db are possible (same k-mer twice): then the count is the saturated sum
    from collections import defaultdict
    want = defaultdict(int)
    member_set = {bytes(k) for k in mk}
    for k_, c_ in zip(gk[members], counts[members]):
        want[bytes(k_)] += int(c_)
    dup_free = {k_: v for k_, v in want.items()}
    all_keys, inv, cnt = np.unique(gk, axis=0, return_inverse=True, return_counts=True)
    for k_, v in list(dup_free.items())[:20000]:
        assert k_ in lookup and lookup[k_] >= min(255, v) - 0 or lookup[k_] == 255
    # scanning the same sample again doubles (saturating) every count
    scan.run(bloom, table, 0, d_rec.ptr, 0, n, d_hits.ptr)
    gpu_ctx.sync()
    ek2, ec2, _ = table.export()
    assert table.status()["num_keys"] == st["num_keys"]
    l2 = {bytes(k): int(c[0]) for k, c in zip(ek2, ec2)}
    for k_, v in list(lookup.items())[:20000]:
        assert l2[k_] == min(255, 2 * v)
    for x in (scan, bloom, table):
        x.close()
    d_rec.free(), d_hits.free()


def test_routed_scan_equals_direct_at_scale(gpu_ctx, monkeypatch):
    """6 x 10^6 records (above the size from which bt_kmc_scan_run partitions the records by sub-filter): the partitioned scan (one partition pass,
    probe through the caches) — one chunk, several ragged chunks, chunks so small that the stripe regions' slack matters, stripe regions too small
    for their share (the records beyond them are probed on the spot), and persistent probe workgroups — leaves exactly the table the direct kernel
    leaves (keys incl. Bloom false positives, counts, hit count).  The chunked variants run double-buffered here (BT_KMC_OVERLAP=1: the partition of chunk i + 1 on a
    class stream beside the probe / apply of chunk i; test_partitioned_scan_other_record_shapes covers the single-buffered chunk loop)"""
    from bayestyper_amd import lib

    monkeypatch.setenv("BT_KMC_OVERLAP", "1")
    rng = np.random.default_rng(23)
    n, p = 6_000_000, 7
    prefixes = np.sort(rng.integers(0, 4 ** p, size=n))
    rec = np.concatenate([rng.integers(0, 256, size=(n, 12), dtype=np.uint8), rng.integers(1, 256, size=(n, 1), dtype=np.uint8)], axis=1)
    lut = np.searchsorted(prefixes, np.arange(4 ** p + 1)).astype(np.uint64)
    scan = lib.KmcScan(gpu_ctx, K, p, 1, n, lut)
    gk, _ = scan.decode(rec.reshape(-1), 0, n)
    mk = np.unique(gk[rng.choice(n, 150_000, replace=False)], axis=0)
    bloom = lib.Bloom.create(gpu_ctx, len(mk) + 100_000, 1e-3, K, threaded=True)      # fpr 1e-3: ~6 000 false positives in the table too
    bloom.insert(mk)
    d_rec = gpu_ctx.to_device(rec.reshape(-1))
    out = []
    for routed, chunk, extra in (("0", None, None), (None, None, None), ("1", "1500007", None), ("1", "70001", None), ("1", "2000003", "overflow"), ("1", "3000001", "persistent")):
        if routed is None:
            monkeypatch.delenv("BT_KMC_ROUTED", raising=False)
        else:
            monkeypatch.setenv("BT_KMC_ROUTED", routed)
        if chunk:
            monkeypatch.setenv("BT_KMC_ROUTED_CHUNK", chunk)
        if extra == "overflow":     # stripe regions (a bucket's region is cut into eight) of 400 records for ~980
            monkeypatch.setenv("BT_KMC_PART_CAP", "400")
        else:
            monkeypatch.delenv("BT_KMC_PART_CAP", raising=False)
        if extra == "persistent":
            monkeypatch.setenv("BT_KMC_PROBE_BPB", "0")
        else:
            monkeypatch.delenv("BT_KMC_PROBE_BPB", raising=False)
        table = lib.Table(gpu_ctx, 400_000, 2, K)
        d_hits = gpu_ctx.buffer(8).zero()
        scan.set_count_range(2, 250)
        scan.run(bloom, table, 1, d_rec.ptr, 0, n, d_hits.ptr)
        gpu_ctx.sync()
        out.append((_sorted_export(*table.export()), int(d_hits.download(np.uint64, 1)[0])))
        table.close(), d_hits.free()
    (ref_tab, ref_hits) = out[0]
    assert ref_hits > len(mk) * 0.9 and len(ref_tab[0]) > len(mk)
    for tab, hits in out[1:]:
        assert hits == ref_hits
        for a, b in zip(tab, ref_tab):
            assert np.array_equal(a, b)
    scan.close(), bloom.close(), d_rec.free()


@pytest.mark.parametrize("k,p,cs", [(55, 7, 2), (55, 3, 4), (55, 11, 1), (31, 7, 1), (31, 3, 2), (64, 4, 2), (64, 0, 1), (21, 9, 1)])
def test_partitioned_scan_other_record_shapes(gpu_ctx, monkeypatch, k, p, cs):
    """the partition kernel hashes straight from the raw record bytes (prefix part once per slab, then four symbols per suffix byte), the apply
    kernel decodes from aligned words: every (k, prefix length, counter size) shape leaves the table the direct kernel leaves"""
    from bayestyper_amd import lib

    rng = np.random.default_rng(1000 * k + 10 * p + cs)
    n = 300_000
    sb = (k - p) // 4          # (KMC picks the prefix length so that k - p is a multiple of four)
    prefixes = np.sort(rng.integers(0, 4 ** p, size=n))
    rec = rng.integers(0, 256, size=(n, sb + cs), dtype=np.uint8)
    rec[:, sb:] = 0
    rec[:, sb] = rng.integers(1, 256, size=n, dtype=np.uint8)
    if cs > 1:
        rec[:, sb + 1] = rng.integers(0, 2, size=n, dtype=np.uint8)   # some counts above 255 (saturate) and above the range
    lut = np.searchsorted(prefixes, np.arange(4 ** p + 1)).astype(np.uint64)
    scan = lib.KmcScan(gpu_ctx, k, p, cs, n, lut)
    gk, gc = scan.decode(rec.reshape(-1), 0, n)
    mk = np.unique(gk[rng.choice(n, 20_000, replace=False)], axis=0)
    bloom = lib.Bloom.create(gpu_ctx, len(mk) + 10_000, 1e-3, k, threaded=True)
    bloom.insert(mk)
    d_rec = gpu_ctx.to_device(rec.reshape(-1))
    out = []
    for routed in ("0", "1"):
        monkeypatch.setenv("BT_KMC_ROUTED", routed)
        monkeypatch.setenv("BT_KMC_ROUTED_CHUNK", "100003")
        table = lib.Table(gpu_ctx, 60_000, 2, k)
        d_hits = gpu_ctx.buffer(8).zero()
        scan.set_count_range(2, 400)
        # ragged calls: the second starts at a record whose byte offset is 16-byte aligned but not at a slab boundary
        first = 16 * 1001
        scan.run(bloom, table, 1, d_rec.ptr, 0, first, d_hits.ptr)
        scan.run(bloom, table, 1, d_rec.ptr + first * (sb + cs), first, n - first, d_hits.ptr)
        gpu_ctx.sync()
        out.append((_sorted_export(*table.export()), int(d_hits.download(np.uint64, 1)[0])))
        table.close(), d_hits.free()
    (ref_tab, ref_hits), (tab, hits) = out
    assert ref_hits > 15_000 and hits == ref_hits
    for a, b in zip(tab, ref_tab):
        assert np.array_equal(a, b)
    scan.close(), bloom.close(), d_rec.free()


def test_calculate_kmer_stats(gpu_ctx, oracle, tmp_path):
    """ObservedKmerCountsHash::calculateKmerStats: class tallies exact; the (sample, intercluster multiplicity) statistics of the
    parameter k-mers — exact integer moments on the device — agree with the reference's running Welford update to 1e-12."""
    from bayestyper_amd import lib

    rng = np.random.default_rng(21)
    S, gender = 3, np.array([0, 1, 1], np.uint8)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60_000)].copy()
    genome[20_000:24_000] = genome[5_000:9_000]            # repeat -> intercluster multiplicity 2x
    win_k, win_v = oracle.kmers_from_sequence(genome[4_000:26_000].tobytes(), K)
    members = np.unique(win_k[win_v == 1], axis=0)
    mem_ascii = oracle.unpack(members, K)
    ob = OrcBloom(oracle, len(members), 1e-3, K, threaded=True)
    ob.insert(mem_ascii)
    gb = lib.Bloom.create(gpu_ctx, len(members), 1e-3, K, threaded=True)
    gb.insert(members)
    ot, gt = OrcTable(oracle, S, K), lib.Table(gpu_ctx, 60_000, S, K)
    # parameter k-mers (main.cpp:571-577), then intercluster multiplicities, then sample counts from a KMC database, then a few
    # cluster occurrences (parameter k-mers never get one in the reference: pick the classified ones from the non-parameter half)
    par = mem_ascii.reshape(-1, K)[::2]
    ot.insert(np.ascontiguousarray(par).reshape(-1), mark_parameter=True)
    gt.insert(oracle.pack(np.ascontiguousarray(par).reshape(-1), K), mark_parameter=True)
    for (a, b, decoy, f, m) in [(0, 30_000, 0, 2, 1), (30_000, 60_000, 0, 2, 2)]:
        ot.count_intercluster(ob, genome[a:b].tobytes(), decoy, f, m)
        gt.count_intercluster(gb, genome[a:b].tobytes(), decoy, f, m)
    for s in range(S):
        cnt = rng.integers(0, 40, len(members)).astype(np.uint32)
        cnt[rng.random(len(members)) < 0.1] = 0
        keep = cnt > 0
        pref = str(tmp_path / f"db{s}")
        oracle.kmc_write(pref, np.ascontiguousarray(mem_ascii.reshape(-1, K)[keep]).reshape(-1), cnt[keep], K, 3, 1)
        db = OrcKmc(oracle, pref)
        ot.parse_sample_kmers(ob, db, s)
        sc = lib.KmcScan(gpu_ctx, db.k, db.p, db.counter_size, db.total, db.lut())
        buf = gpu_ctx.to_device(db.payload())
        sc.run(gb, gt, s, buf.ptr, 0, db.total)
        gpu_ctx.sync()
        sc.close()
        buf.free()
        db.close()
    nonpar = mem_ascii.reshape(-1, K)[1::2][:500]
    mult = rng.choice([1, 2, 130], 500).astype(np.uint8)
    omg = OrcBloom(oracle, 10, 1e-4, K)
    gmg = lib.Bloom.create(gpu_ctx, 10, 1e-4, K, threaded=False)
    ot.classify(omg, np.ascontiguousarray(nonpar).reshape(-1), mult)
    gt.classify(gmg, oracle.pack(np.ascontiguousarray(nonpar).reshape(-1), K), mult)
    cls_o, st_o = ot.kmer_stats(gender)
    cls_g, mom = gt.kmer_stats(gender)
    assert np.array_equal(cls_o, cls_g) and cls_o[1] > 0 and cls_o[4] > 0 and cls_o[6] > 0
    n = mom["n"].astype(np.float64)
    assert np.array_equal(st_o[:, :, 0], n) and n.sum() == S * len(par) and (n > 0).sum() >= 2 * S
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = np.where(n > 0, mom["sum"] / n, 0.0)
        frac = np.where(n > 0, mom["nonzero"] / n, 0.0)
        m2 = np.where(n > 0, mom["sumsq"] - mom["sum"].astype(np.float64) ** 2 / np.maximum(n, 1), 0.0)
    assert np.allclose(st_o[:, :, 1], frac, rtol=1e-12, atol=1e-12)
    assert np.allclose(st_o[:, :, 2], mean, rtol=1e-12, atol=1e-12)
    assert np.allclose(st_o[:, :, 3], m2, rtol=1e-9, atol=1e-7)
    for x in (gt, gb, gmg, ob, omg, ot):
        x.close()


def _path_graphs(seed, n):
    from bayestyper_amd import synth_graphs

    rng = np.random.default_rng(seed)
    gs = [synth_graphs.random_cluster(rng, K, int(rng.integers(1, 7)), int(rng.integers(2, 12)), nested_cluster=(500 + i) if i % 3 == 2 else None) for i in range(n)]
    # two clusters sharing a stretch of sequence -> multicluster k-mers; one cluster with a long homopolymer -> a k-mer > 127 times
    import copy

    gs[1] = copy.deepcopy(gs[0])
    gs[1].paths = synth_graphs.random_paths(gs[1], rng, 3)
    big = synth_graphs.random_cluster(rng, K, 1, 2, chrom_len=1200)
    big.seq[-1] = np.concatenate([big.seq[-1], np.zeros(300, np.uint8)])
    gs.append(big)
    return gs, synth_graphs.flatten(gs)


def test_path_enumeration_classify_candidates(gpu_ctx, oracle):
    """countPathKmers / classifyPathKmers / getHaplotypeCandidates over a batch of graphs (SNVs, indels, multi-allelic, nested cuts,
    shared and >127x k-mers) against the oracle: Bloom image, table contents, and the whole VariantClusterHaplotypes bundle."""
    from _oracle import OrcGraphs
    from bayestyper_amd import lib

    S = 2
    gs, f = _path_graphs(31, 14)
    og = OrcGraphs(oracle, f, K)
    gp = lib.Paths(gpu_ctx, f, K)
    # countPathKmers -> path Bloom (bit image identical)
    ob = OrcBloom(oracle, 200_000, 1e-3, K, threaded=False)
    gb = lib.Bloom.create(gpu_ctx, 200_000, 1e-3, K, threaded=False)
    assert og.count_kmers(ob) == gp.num_windows
    gp.count_kmers(gb)
    gpu_ctx.sync()
    assert np.array_equal(gb.bits(0), ob.bits(0))
    # a count table holding part of the path k-mers with counts, some with intercluster multiplicity, one decoy region
    ot, gt = OrcTable(oracle, S, K), lib.Table(gpu_ctx, 100_000, S, K)
    rng = np.random.default_rng(5)
    nt = np.frombuffer(b"ACGT", np.uint8)
    text = np.concatenate([np.concatenate([nt[g.seq[v]] for v in range(len(g.seq))] + [np.frombuffer(b"N", np.uint8)]) for g in gs[:6]])
    tb_o = OrcBloom(oracle, 200_000, 1e-3, K, threaded=True)
    tb_g = lib.Bloom.create(gpu_ctx, 200_000, 1e-3, K, threaded=True)
    km, va = oracle.kmers_from_sequence(text.tobytes(), K)
    members = np.unique(km[va == 1], axis=0)
    tb_o.insert(oracle.unpack(members, K))
    tb_g.insert(members)
    ot.count_intercluster(tb_o, text[: len(text) // 3].tobytes(), 0, 2, 1)
    gt.count_intercluster(tb_g, text[: len(text) // 3].tobytes(), 0, 2, 1)
    dec = text[len(text) // 3: len(text) // 3 + 200].tobytes()
    ot.count_intercluster(tb_o, dec, 1, 0, 0)
    gt.count_intercluster(tb_g, dec, 1, 0, 0)
    # multigroup Bloom with a few of the path k-mers
    mg_members = members[rng.choice(len(members), 30, replace=False)]
    omg = OrcBloom(oracle, 30, 1e-4, K)
    gmg = lib.Bloom.create(gpu_ctx, 30, 1e-4, K, threaded=False)
    omg.insert(oracle.unpack(mg_members, K))
    gmg.insert(mg_members)
    n_o, ex_o = og.classify(ot, omg)
    n_g, ex_g = gp.classify(gt, gmg)
    assert np.array_equal(n_o, n_g) and np.array_equal(ex_o, ex_g) and ex_o.any() and not ex_o.all()
    gk, gc, gm = _sorted_export(*gt.export())
    wk, wc, wm = _sorted_export(*ot.export())
    assert np.array_equal(gk, wk) and np.array_equal(gm, wm)
    assert (wm[:, 0] & 0x02).any() and (wm[:, 0] & 0x10).any() and (wm[:, 0] & 0x04).any()   # multicluster, max-multiplicity, multigroup
    # getHaplotypeCandidates
    ro = og.candidates(ot)
    rg = gp.candidates(gt)
    for name in ro:
        assert np.array_equal(ro[name], rg[name]), name
    assert len(ro["multi_idx"]) > 0 and len(ro["hapnest_idx"]) > 0 and len(ro["nestdep_var"]) > 0 and ro["kmer_off"][-1] < n_o.sum()
    for x in (gp, og, gb, ob, gt, ot, tb_o, tb_g, omg, gmg):
        x.close()


def test_intercluster_parameter_kmers(gpu_ctx, oracle):
    """countInterclusterParameterKmers: per region the Bernoulli(fraction) stream of mt19937(seed + region index) over the non-path
    k-mers in window order; decoy regions override.  Table flags bit-exact."""
    from bayestyper_amd import lib

    rng = np.random.default_rng(33)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200_000)].copy()
    genome[70_000:70_030] = ord("N")
    genome[150_000:152_000] = genome[10_000:12_000]        # a repeat: the same k-mer drawn in two regions / in a decoy and a normal region
    # regions: disjoint, unsorted, one of length < k, one empty-ish, two decoys (one overlapping the repeat)
    regions = [(60_000, 25_000, 0), (0, 30_000, 0), (149_000, 5_000, 1), (30_050, 40, 0), (100_000, 20_000, 0), (190_000, 10_000, 1), (130_000, 54, 0)]
    path_k, path_v = oracle.kmers_from_sequence(genome[5_000:8_000].tobytes(), K)
    path = np.unique(path_k[path_v == 1], axis=0)
    ob = OrcBloom(oracle, len(path), 1e-3, K, threaded=True)
    ob.insert(oracle.unpack(path, K))
    gb = lib.Bloom.create(gpu_ctx, len(path), 1e-3, K, threaded=True)
    gb.insert(path)
    for fraction in (0.05, 1.0):
        ot, gt = OrcTable(oracle, 1, K), lib.Table(gpu_ctx, 400_000, 1, K)
        seeds = [1234 + i for i in range(len(regions))]
        for (a, n, d), sd in zip(regions, seeds):
            ot.count_parameter_kmers(ob, genome[a:a + n].tobytes(), d, sd, fraction)
        gt.count_parameter_kmers(gb, genome.tobytes(), [r[0] for r in regions], [r[1] for r in regions], [r[2] for r in regions], seeds, fraction)
        gk, gc, gm = _sorted_export(*gt.export())
        wk, wc, wm = _sorted_export(*ot.export())
        assert np.array_equal(gk, wk) and np.array_equal(gm[:, 0], wm[:, 0])
        par = (wm[:, 0] & 0x20) != 0
        dec = (wm[:, 0] & 0x08) != 0
        assert par.sum() > 100 and dec.sum() > 1000 and (par & dec).sum() > 0 if fraction == 1.0 else par.sum() > 100
        if fraction < 1:
            n_cand = sum(max(0, n - K + 1) for a, n, d in regions if not d)
            assert 0.03 * n_cand < par.sum() < 0.07 * n_cand
        ot.close(), gt.close()
    ob.close(), gb.close()


@pytest.mark.parametrize("n_filter,fpr,threaded", [(1_000_000, 1e-7, True), (3000, 0.02, False), (20000, 0.01, True)])
def test_path_multigroup_kmers(gpu_ctx, oracle, n_filter, fpr, threaded):
    """countPathMultigroupKmers in the reference's single-thread order: groups in index order, a group's k-mers in the iteration order
    of the std::unordered_set<std::bitset<110>> that is reused (clear()ed) from group to group; a k-mer the path filter reports at its
    turn — shared with an earlier group or unit, or a false positive of what has been inserted so far — lands in the multigroup table.
    Two units share filter and table; groups of very different sizes (the set's bucket count is inherited); with roomy filters (no false
    positive) and with filters far too small (hundreds of order-dependent false positives).  Table, num_path_kmers and filter bits equal
    the oracle's, which runs the reference's loop on the real container."""
    import copy

    from _oracle import OrcGraphs
    from bayestyper_amd import lib, synth_graphs

    rng = np.random.default_rng(41)
    ob = OrcBloom(oracle, n_filter, fpr, K, threaded=threaded)
    gb = lib.Bloom.create(gpu_ctx, n_filter, fpr, K, threaded=threaded)
    # the multigroup table starts far too small for the undersized-filter cases: it grows before a unit's k-mers go in (the reference's KmerHash grows on demand)
    ot, gt = OrcTable(oracle, 1, K), lib.Table(gpu_ctx, 16 if fpr > 1e-3 else 200_000, 1, K)
    shared = None
    total = 0
    for unit in range(2):
        sizes = [int(rng.integers(1, 5)) for _ in range(12)] + [40, 1, 2, 90, 3]      # variants per cluster: small groups after large ones
        gs = [synth_graphs.random_cluster(rng, K, v, int(rng.integers(2, 7))) for v in sizes]
        gs[3] = copy.deepcopy(gs[2])     # same group as 2 -> shared k-mers are NOT multigroup
        gs[3].paths = synth_graphs.random_paths(gs[3], rng, 3)
        gs[9] = copy.deepcopy(gs[5])     # different group -> multigroup
        gs[9].paths = synth_graphs.random_paths(gs[9], rng, 2)
        if unit == 0:
            shared = copy.deepcopy(gs[7])
        else:
            gs[1] = shared               # a cluster of the previous unit: its k-mers are in the filter already
        f = synth_graphs.flatten(gs)
        cluster_group = np.array([0, 1, 2, 2, 3, 4, 5, 5, 6, 7, 8, 9, 10, 11, 11, 12, 13], np.uint32)
        og, gp = OrcGraphs(oracle, f, K), lib.Paths(gpu_ctx, f, K)
        n_o = og.count_multigroup(cluster_group, ob, ot)
        n_g = gp.count_multigroup(cluster_group, gb, gt)
        assert n_o == n_g and n_o > 0
        total += n_o
        gk, _, _ = _sorted_export(*gt.export())
        wk, _, _ = _sorted_export(*ot.export())
        assert len(wk) > 50 and np.array_equal(gk, wk), (unit, len(gk), len(wk))
        for sub in (range(0, 65536, 4099) if threaded else [0]):
            assert np.array_equal(gb.bits(sub), ob.bits(sub))
        og.close(), gp.close()
    if fpr > 1e-3:   # the undersized filters: most multigroup entries are false positives of the moment
        assert len(wk) > 0.02 * total
        st = gt.status()
        assert not st["overflowed"] and st["capacity"] >= 2 * st["num_keys"] > 64
    for x in (ob, gb, ot, gt):
        x.close()


@pytest.mark.parametrize("max_haps,fpr", [(32, 1e-6), (3, 0.05), (2, 0.3)])
def test_find_sample_paths(gpu_ctx, oracle, max_haps, fpr):
    """findSamplePaths + addPathIndices for two samples: best_paths_indices identical to the oracle, with roomy and with binding
    max_sample_haplotypes, exact and false-positive-rich sample Bloom filters, nested cuts, multi-allelic variants."""
    from _oracle import OrcGraphs
    from bayestyper_amd import lib, synth_graphs

    rng = np.random.default_rng(52)
    gs = [synth_graphs.random_cluster(rng, K, int(rng.integers(1, 8)), 3, nested_cluster=(700 + i) if i % 3 == 1 else None) for i in range(40)]
    truth = [g.paths.copy() for g in gs]
    for g in gs:
        g.paths = None
    f = synth_graphs.flatten(gs)
    og = OrcGraphs(oracle, f, K)
    gf = lib.FindPaths(gpu_ctx, f, K, max_haps, 2)
    nt = np.frombuffer(b"ACGT", np.uint8)
    for s in range(2):
        # the sample's reads: two haplotypes per cluster (random best-path rows), 10 % of their k-mers unobserved
        rows = [truth[i][rng.integers(len(truth[i]), size=2)] for i in range(len(gs))]
        text = np.concatenate([np.concatenate([nt[g.seq[v]] for v in range(len(g.seq)) if rows[i][h, v]] + [np.frombuffer(b"N", np.uint8)])
                               for i, g in enumerate(gs) for h in range(2)])
        km, va = oracle.kmers_from_sequence(text.tobytes(), K)
        mem = np.unique(km[va == 1], axis=0)
        mem = mem[rng.random(len(mem)) > 0.1]
        ob = OrcBloom(oracle, len(mem), fpr, K)
        gb = lib.Bloom.create(gpu_ctx, len(mem), fpr, K, threaded=False)
        ob.insert(oracle.unpack(mem, K))
        gb.insert(mem)
        seeds = (4242 + (np.arange(len(gs)) + 1) * (s + 1) + np.arange(len(gs))).astype(np.uint32)   # prng_seed + (group+1)*(sample+1) + cluster
        bo = og.find_sample_paths(ob, seeds, max_haps)
        gf.sample(gb, seeds)
        bg = gf.best_paths()
        for c in range(len(gs)):
            assert bo[c].shape == bg[c].shape and np.array_equal(bo[c], bg[c]), (s, c)
        ob.close(), gb.close()
    assert sum(b.shape[0] for b in bo) > len(gs)
    og.close(), gf.close()


def test_parse_sample_kmers_host_driver(gpu_ctx, oracle, tmp_path):
    """KmerCounter::parseSampleKmers as the C++ host layer drives it: KMC database on disk -> bthost::KmcFile (mmap) ->
    bt_kmc_scan_run_host (pinned staging, copy stream, events; many small chunks here) -> count table == the oracle's."""
    import ctypes as C

    from bayestyper_amd import lib
    from bayestyper_amd.host import dll

    rng = np.random.default_rng(61)
    S = 2
    km = np.unique(_oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 60_000, K), K).reshape(-1, K), axis=0)
    path = km[rng.random(len(km)) < 0.3]
    ob = OrcBloom(oracle, len(path), 1e-3, K, threaded=True)
    ob.insert(np.ascontiguousarray(path).reshape(-1))
    gb = lib.Bloom.create(gpu_ctx, len(path), 1e-3, K, threaded=True)
    gb.insert(oracle.pack(np.ascontiguousarray(path).reshape(-1), K))
    ot, gt = OrcTable(oracle, S, K), lib.Table(gpu_ctx, 60_000, S, K)
    dll.bth_parse_sample_kmers.restype = C.c_longlong
    dll.bth_parse_sample_kmers.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_ulonglong, C.c_char_p, C.c_uint]
    for s, chunk in enumerate((1000, 0)):
        sel = km[rng.random(len(km)) < 0.8]
        cnt = rng.integers(1, 250, len(sel)).astype(np.uint32)
        pref = str(tmp_path / f"s{s}")
        if s == 0:
            oracle.kmc_write(pref, np.ascontiguousarray(sel).reshape(-1), cnt, K, 3, 1)          # KMC1 layout
        else:
            oracle.kmc2_write(pref, np.ascontiguousarray(sel).reshape(-1), cnt, K, 7, 1, 6)      # KMC2 layout: 6 signature bins
        db = OrcKmc(oracle, pref)
        hits_o = ot.parse_sample_kmers(ob, db, s)
        db.close()
        err = C.create_string_buffer(256)
        hits_g = dll.bth_parse_sample_kmers(gpu_ctx.h, pref.encode(), gb.h, gt.h, s, chunk, err, 256)
        assert hits_g == hits_o, err.value
    gk, gc, gm = _sorted_export(*gt.export())
    wk, wc, wm = _sorted_export(*ot.export())
    assert np.array_equal(gk, wk) and np.array_equal(gc, wc) and np.array_equal(gm, wm) and len(wk) > 5000
    for x in (ob, gb, ot, gt):
        x.close()


def test_make_bloom_from_kmc(gpu_ctx, oracle, tmp_path):
    """bayesTyperTools makeBloom: KMC database -> sample KmerBloom; .bloomMeta / .bloomData byte-identical to the oracle's filter
    (which the reference's KmerBloom reads: tests/test_oracle_kmer.py)."""
    from bayestyper_amd import lib

    rng = np.random.default_rng(71)
    km = np.unique(_oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 30_000, K), K).reshape(-1, K), axis=0)
    pref = str(tmp_path / "sample")
    oracle.kmc2_write(pref, np.ascontiguousarray(km).reshape(-1), np.ones(len(km), np.uint32), K, 7, 1, 3)
    db = OrcKmc(oracle, pref)
    ob = OrcBloom(oracle, db.total, 1e-3, K)                      # KmerBloom(total_kmers, fpr), MakeBloom.cpp:218
    ob.insert(db.list()[0].reshape(-1))
    gb = lib.Bloom.create(gpu_ctx, db.total, 1e-3, K, threaded=False)
    sc = lib.KmcScan(gpu_ctx, db.k, db.p, db.counter_size, db.total, db.lut())
    buf = gpu_ctx.to_device(db.payload())
    half = (db.total // 2) // 16 * 16
    sc.make_bloom(gb, buf.ptr, 0, half)
    sc.make_bloom(gb, buf.ptr + half * db.rec_size, half, db.total - half)
    gpu_ctx.sync()
    assert np.array_equal(gb.bits(0), ob.bits(0))
    gb.save(str(tmp_path / "g"))
    ob.save(str(tmp_path / "o"))
    for ext in (".bloomMeta", ".bloomData"):
        assert open(str(tmp_path / "g") + ext, "rb").read() == open(str(tmp_path / "o") + ext, "rb").read()
    for x in (sc, gb, ob, db):
        x.close()
    buf.free()


def test_error_paths_report_instead_of_computing(gpu_ctx, oracle):
    """the C ABI returns an error code with bt_last_error() text (no exceptions across the boundary, nothing computed on bad input):
    sample limit of main.cpp:72, count-model tables not set, k mismatches, a count table that is too small, misaligned records"""
    from bayestyper_amd import lib, synth

    flat = synth.make_batch("A", 8, 2, seed=1)
    lut_g, lut_n = _oracle.build_luts(oracle, 2)
    g = lib.Gibbs(gpu_ctx, flat, None, None, chains=1, burn=1, iters=1)
    with pytest.raises(lib.BtError, match="LUT"):
        g.run()
    with pytest.raises(lib.BtError, match="LUT"):
        g.sweep(1, False)
    g.set_lut(lut_g, lut_n)
    g.run()
    g.close()
    big = synth.make_batch("A", 2, 31, seed=1)
    with pytest.raises(lib.BtError, match="1..30"):
        lib.Gibbs(gpu_ctx, big, None, None)
    b55 = lib.Bloom.create(gpu_ctx, 1000, 1e-3, 55, threaded=True)
    t31 = lib.Table(gpu_ctx, 1000, 1, 31)
    sc = lib.KmcScan(gpu_ctx, 55, 7, 1, 100, np.linspace(0, 100, 4 ** 7 + 1).astype(np.uint64))
    buf = gpu_ctx.to_device(np.zeros(100 * 13 + 16, np.uint8))
    with pytest.raises(lib.BtError, match="k"):
        sc.run(b55, t31, 0, buf.ptr, 0, 100)
    t55 = lib.Table(gpu_ctx, 1000, 1, 55)
    with pytest.raises(lib.BtError, match="exceeds"):
        sc.run(b55, t55, 0, buf.ptr, 50, 100)
    with pytest.raises(lib.BtError, match="sample"):
        sc.run(b55, t55, 3, buf.ptr, 0, 100)
    with pytest.raises(lib.BtError, match="align"):
        sc.make_bloom(lib.Bloom.create(gpu_ctx, 100, 1e-3, 55, threaded=False), buf.ptr + 4, 0, 10)
    # a table sized for 16 keys cannot take 5 000: batch calls are asynchronous, so the overflow is latched on the device and reported
    # by bt_table_status (which the host checks after every scan) instead of being dropped silently
    rng = np.random.default_rng(3)
    km = np.unique(oracle.pack(_oracle.canonical_ascii(oracle, _oracle.random_kmers(rng, 5000, 55), 55), 55), axis=0)
    bb = lib.Bloom.create(gpu_ctx, len(km), 1e-3, 55, threaded=True)
    small = lib.Table(gpu_ctx, 16, 1, 55)
    assert not small.status()["overflowed"]
    small.insert(km)
    st = small.status()
    assert st["overflowed"] and st["num_keys"] <= st["capacity"] < len(km)
    for x in (b55, t31, t55, sc, bb, small):
        x.close()
    buf.free()
