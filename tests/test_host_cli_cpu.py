"""CPU tests of the pieces around the two command lines (bayestyper_amd/host: Options, Sample, ChromosomePloidy, InferenceUnit,
KmerHashOrder, the `bayesTyper` executable's argument handling) — no GPU needed: the executable is only run up to the point where it
would create a GPU context.  HybridHash's iteration order and CountAllocation are pinned against the reference's own code
(oracle/_ref/libbtref.so)."""
import ctypes as C
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _oracle  # noqa: E402
import test_cluster_stage_cpu as T  # noqa: E402

K = 55
EXE = os.path.join(ROOT, "bayestyper_amd", "bayesTyper")
KMER = "ACGTACGTTGCAAGCTTAGCCATGGATCCGATTACAGGCTTAACGGTCATGCAAT"


def _dll():
    from bayestyper_amd.host import dll

    dll.bth_bitset_hash.restype = C.c_uint64
    dll.bth_bitset_hash.argtypes = [C.c_uint64, C.c_uint64, C.c_uint]
    dll.bth_hybrid_hash_order.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint64, C.c_void_p]
    dll.bth_count_allocation.argtypes = [C.c_ushort, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    dll.bth_unit_roundtrip.restype = C.c_ulonglong
    dll.bth_unit_roundtrip.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_ulonglong, C.c_char_p, C.c_uint]
    dll.bth_chromosome_ploidy.restype = C.c_ulonglong
    dll.bth_chromosome_ploidy.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ulonglong, C.c_char_p, C.c_uint]
    return dll


def _pack(oracle, ascii_u8):
    return np.ascontiguousarray(oracle.pack(ascii_u8, K), np.uint64)


def test_bitset_hash_known_answer_and_reference(oracle, ref):
    """std::hash<std::bitset<110>> % 4^12 of the SURVEY A.2 k-mer is 5318332 (captured from the compiled reference); our restatement of
    libstdc++'s _Hash_bytes agrees with the reference's HybridHash::rootHashIndex on random k-mers"""
    dll = _dll()
    lo, hi = (int(x) for x in _pack(oracle, np.frombuffer(KMER.encode(), np.uint8)).reshape(-1))
    assert dll.bth_bitset_hash(lo, hi, K) % 4 ** 12 == 5318332
    ref.l.ref_hybrid_hash_root.restype = C.c_uint64
    ref.l.ref_hybrid_hash_root.argtypes = [C.c_char_p, C.c_uint64]
    assert ref.l.ref_hybrid_hash_root(KMER.encode(), 4 ** 12) == 5318332
    rng = np.random.default_rng(3)
    km = _oracle.random_kmers(rng, 500, K)
    packed = _pack(oracle, km).reshape(-1, 2)
    for i in range(500):
        want = ref.l.ref_hybrid_hash_root(km[i * K:(i + 1) * K].tobytes(), 4 ** 12)
        assert dll.bth_bitset_hash(int(packed[i, 0]), int(packed[i, 1]), K) % 4 ** 12 == want


def test_multigroup_kmer_set_order_vs_reference_container(oracle, ref):
    """The order bt_paths_count_multigroup visits a group's k-mers in = the iteration order of the std::unordered_set<std::bitset<110>> the
    reference collects them in and clear()s between groups (KmerCounter.cpp:111-119).  The library's replay of that container (the same
    host/device code the kernel runs: bt_diag_kmer_set_order) against the real container filled through the reference's Nucleotide::ntToBit
    (oracle/_ref): a sequence of groups of very different sizes — growth through several rehashes inside a group, small groups that
    inherit a large bucket count, empty groups — ranks and bucket counts must agree group by group."""
    from bayestyper_amd import lib

    rng = np.random.default_rng(2024)
    sizes = [5, 1, 0, 40, 13, 14, 3, 700, 2, 60, 5000, 1, 17, 30000, 12, 100]
    groups = []
    for n in sizes:
        km = np.unique(_oracle.random_kmers(rng, n, K).reshape(-1, K), axis=0) if n else np.zeros((0, K), np.uint8)
        groups.append(km[rng.permutation(len(km))])
    off = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.uint64)
    flat = np.ascontiguousarray(np.concatenate(groups)).reshape(-1)
    want_order = np.zeros(int(off[-1]), np.uint32)
    want_buckets = np.zeros(len(groups), np.uint64)
    ref.l.ref_group_kmer_set_orders.restype = None
    ref.l.ref_group_kmer_set_orders.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    ref.l.ref_group_kmer_set_orders(flat.ctypes.data, off.ctypes.data, len(groups), want_order.ctypes.data, want_buckets.ctypes.data)
    buckets = 1
    for g, km in enumerate(groups):
        n = len(km)
        packed = _pack(oracle, np.ascontiguousarray(km).reshape(-1)) if n else np.zeros((0, 2), np.uint64)
        rank = np.zeros(max(n, 1), np.uint32)
        final = C.c_uint64(0)
        lib.check(lib.bt_diag_kmer_set_order(packed.ctypes.data if n else rank.ctypes.data, n, buckets, K, rank.ctypes.data, C.byref(final)))
        order = np.zeros(n, np.uint32)
        order[rank[:n]] = np.arange(n, dtype=np.uint32)          # order[j] = the k-mer visited j-th
        assert np.array_equal(order, want_order[int(off[g]):int(off[g + 1])]), f"group {g} ({n} k-mers, {buckets} buckets inherited)"
        assert final.value == int(want_buckets[g]), (g, final.value, int(want_buckets[g]))
        buckets = final.value
    assert buckets >= 30000


@pytest.mark.parametrize("root_size,n", [(4 ** 12, 20000), (64, 3000), (1, 300)])
def test_parameter_kmer_order_vs_reference(oracle, ref, root_size, n):
    """KmerHash::shuffle(seed) + iteration (what picks the <= 10^6 parameter k-mers, main.cpp:326-341): the reference's own HybridHash
    (sorted leaves, one mt19937 over all leaves) against bthost::hybridHashShuffledOrder — with the real 4^12 roots (few collisions) and
    with tiny root tables where every leaf holds many k-mers"""
    dll = _dll()
    rng = np.random.default_rng(n)
    km = np.unique(_oracle.random_kmers(rng, n, K).reshape(-1, K), axis=0)
    km = km[rng.permutation(len(km))]
    n = len(km)
    flat = np.ascontiguousarray(km).reshape(-1)
    ref.l.ref_hybrid_hash_order.restype = C.c_uint64
    ref.l.ref_hybrid_hash_order.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint, C.c_int, C.c_void_p]
    packed = _pack(oracle, flat)
    for seed, shuffle in ((42, 1), (7, 1), (0, 0)):
        want = np.zeros(n, np.uint32)
        assert ref.l.ref_hybrid_hash_order(flat.ctypes.data, n, root_size, seed, shuffle, want.ctypes.data) == n
        if shuffle:
            got = np.zeros(n, np.uint32)
            dll.bth_hybrid_hash_order(packed.ctypes.data, n, K, seed, root_size, got.ctypes.data)
            assert np.array_equal(got, want)
        else:   # unshuffled iteration: root order, then BitsetLess = the 110-bit value ascending
            roots = np.array([dll.bth_bitset_hash(int(a), int(b), K) % root_size for a, b in packed.reshape(-1, 2)], np.uint64)
            p = packed.reshape(-1, 2)
            assert np.array_equal(want, np.lexsort((p[:, 0], p[:, 1], roots)).astype(np.uint32))


def test_count_allocation_vs_reference(ref):
    dll = _dll()
    rng = np.random.default_rng(1)
    S = 5
    s1, c1 = rng.integers(0, S, 4000).astype(np.uint16), rng.integers(0, 256, 4000).astype(np.uint8)
    s2, c2 = rng.integers(0, S, 900).astype(np.uint16), rng.integers(0, 256, 900).astype(np.uint8)
    got, want = np.zeros(S * 256, np.uint64), np.zeros(S * 256, np.uint64)
    dll.bth_count_allocation(S, s1.ctypes.data, c1.ctypes.data, len(s1), s2.ctypes.data, c2.ctypes.data, len(s2), got.ctypes.data)
    ref.l.ref_count_allocation.argtypes = [C.c_ushort, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    ref.l.ref_count_allocation(S, s1.ctypes.data, c1.ctypes.data, len(s1), s2.ctypes.data, c2.ctypes.data, len(s2), want.ctypes.data)
    assert np.array_equal(got, want) and got.sum() == 4900
    expect = np.zeros((S, 256), np.uint64)
    np.add.at(expect, (np.concatenate([s1, s2]), np.concatenate([c1, c2])), 1)
    assert np.array_equal(got.reshape(S, 256), expect)


def _stage(genome, vcf):
    from bayestyper_amd.host.cluster_stage import ClusterStage

    st = ClusterStage(K)
    for g in genome:
        st.add_sequence(*g)
    st.set_variants(vcf_text=vcf)
    return st


def test_unit_file_round_trip(tmp_path):
    """variant_clusters.bin (own gzip'd layout): groups, clusters, variants, alleles, contained clusters, edges and best-path matrices
    survive the round trip; a file of another format is refused with a message"""
    from bayestyper_amd.host import dll as _  # noqa: F401

    dll = _dll()
    rng = np.random.default_rng(5)
    genome = [[f"chr{i + 1}", "".join(rng.choice(list("ACGT"), n)), False] for i, n in enumerate([30000, 20000])]
    st = _stage(genome, T.make_vcf(rng, genome, K, 40, False, extra_contig=False, sv_blocks=2))
    assert st.next_unit(10 ** 9)
    err = C.create_string_buffer(512)
    fn = str(tmp_path / "variant_clusters.bin").encode()
    n = dll.bth_unit_roundtrip(st.h, fn, 2, None, 0, err, len(err))
    assert n > 0, err.value
    buf = C.create_string_buffer(int(n) + 1)
    dll.bth_unit_roundtrip(st.h, fn, 2, buf, n, err, len(err))
    text = buf.raw[:n].decode()
    assert text.startswith("7 ##BayesTyperOptions=test\n123 45 1099511627776\n") and text.endswith("IDENTICAL\n")
    assert st.unit_text() in text
    with gzip.open(fn.decode()) as f:
        assert f.read(10) == b"BTAMDUNIT1"
    bad = tmp_path / "other.bin"
    with gzip.open(bad, "wb") as f:
        f.write(b"22 serialization::archive 15 0 0")
    r = subprocess.run([EXE, "genotype", "-v", str(bad), "-c", str(tmp_path), "-s", _samples(tmp_path), "-g", _genome(tmp_path, genome)], capture_output=True, text=True)
    assert r.returncode == 1 and "not a variant clusters file of this build" in r.stderr
    st.close()


def _samples(tmp_path, rows=("sample1\tF\t/none/a", "sample2\tMale\t/none/b")):
    p = tmp_path / "samples.tsv"
    p.write_text("\n".join(rows) + "\n")
    return str(p)


def _genome(tmp_path, genome):
    p = tmp_path / "genome.fa"
    with open(p, "w") as f:
        for name, seq, _ in genome:
            f.write(f">{name} description\n")
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70].lower() + "\n")
    return str(p)


def test_chromosome_ploidy(tmp_path):
    dll = _dll()
    genome = [["chr1", "ACGT" * 30, False], ["chrX", "ACGT" * 30, False], ["Y", "ACGT" * 30, False], ["decoy1", "ACGT" * 30, True]]
    st = _stage(genome, "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")

    def table(ploidy_file, genders):
        err = C.create_string_buffer(1024)
        n = dll.bth_chromosome_ploidy(st.h, ploidy_file.encode(), genders.encode(), None, 0, err, len(err))
        if n == 0:
            raise ValueError(err.value.decode())
        buf = C.create_string_buffer(int(n) + 1)
        dll.bth_chromosome_ploidy(st.h, ploidy_file.encode(), genders.encode(), buf, n, err, len(err))
        return buf.raw[:n].decode()

    # human defaults (ChromosomePloidy.cpp:42-91): X diploid / haploid, Y absent / haploid; decoys have no ploidy
    assert table("", "FMF") == "chr1\t2\t2\t222\nchrX\t2\t1\t212\nY\t0\t1\t010\n"
    f = tmp_path / "ploidy.txt"
    f.write_text("chr1\t2\t2\nchrX\t2\t2\nY\t1\t0\n")
    assert table(str(f), "MF") == "chr1\t2\t2\t22\nchrX\t2\t2\t22\nY\t1\t0\t01\n"
    f.write_text("chr1\t2\t2\nchrX\t3\t2\nY\t1\t0\n")
    with pytest.raises(ValueError, match="Female ploidy"):
        table(str(f), "F")
    f.write_text("chr1\t2\t2\nchrX\t2\t1\n")
    with pytest.raises(ValueError, match='Chromosome "Y" in reference genome does not appear'):
        table(str(f), "F")
    f.write_text("chr1\t2\t2\nchr1\t2\t1\n")
    with pytest.raises(ValueError, match="appear multiple times"):
        table(str(f), "F")
    st.close()


def test_command_lines_arguments(tmp_path):
    """bayesTyper <command>: usage, help (exit code 1 like main.cpp:146-150), required / unknown / malformed options, the samples file's
    error messages (Sample.cpp:38-67, main.cpp:163-192) — everything up to the first GPU call"""
    run = lambda *a: subprocess.run([EXE, *a], capture_output=True, text=True)   # noqa: E731
    r = run()
    assert r.returncode == 0 and "Usage: bayesTyper <command> [options]" in r.stdout and "You are using BayesTyper" in r.stdout
    r = run("frobnicate")
    assert r.returncode == 0 and "genotype\tgenotype variant clusters" in r.stdout
    for cmd, must in (("cluster", ["--variant-file", "--min-number-of-unit-variants", "--max-number-of-sample-haplotypes arg (=32)", "--copy-number-variant-threshold arg (=0.5)"]),
                      ("genotype", ["--variant-clusters-file", "--cluster-data-dir", "--gibbs-burn-in arg (=100)", "--number-of-gibbs-chains arg (=20)", "--noise-rate-prior arg (=1,0.01)",
                                    "--min-genotype-posterior arg (=0.99)", "--chromosome-ploidy-file", "--noise-genotyping", "-z [ --gzip-output ]"])):
        for args in ((cmd,), (cmd, "-h"), (cmd, "--help")):
            r = run(*args)
            assert r.returncode == 1 and all(m in r.stdout for m in must), (args, r.stdout)
    r = run("cluster", "-v", "x.vcf", "-s", "s.tsv")
    assert r.returncode == 1 and "the option '--genome-file' is required but missing" in r.stderr
    r = run("cluster", "-v", "x.vcf", "-s", "s.tsv", "-g", "g.fa", "--bogus", "1")
    assert r.returncode == 1 and "unrecognised option '--bogus'" in r.stderr
    r = run("genotype", "-v", "a", "-c", "b", "-s", "c", "-g", "d", "--gibbs-samples", "many")
    assert r.returncode == 1 and "is invalid" in r.stderr
    r = run("genotype", "-v", "a", "-c", "b", "-s", "c", "-g", "d", "--noise-rate-prior", "1")
    assert r.returncode == 1 and "should be two values (comma-seperated)" in r.stderr
    r = run("cluster", "-v", "x.vcf", "-s", str(tmp_path / "missing.tsv"), "-g", "g.fa")
    assert r.returncode == 1 and "ERROR: Unable to open file" in r.stderr
    r = run("cluster", "-v", "x.vcf", "-s", _samples(tmp_path, ["onlytwo\tF"]), "-g", "g.fa")
    assert r.returncode == 1 and "should contain three tab-seperated columns" in r.stderr
    r = run("cluster", "-v", "x.vcf", "-s", _samples(tmp_path, ["a\tX\tprefix"]), "-g", "g.fa")
    assert r.returncode == 1 and 'should be either "F" (Female) or "M" (Male)' in r.stderr
    r = run("cluster", "-v", "x.vcf", "-s", _samples(tmp_path, [f"s{i}\tF\tp" for i in range(31)]), "-g", "g.fa")
    assert r.returncode == 1 and "maximum number of samples supported by BayesTyper is currently 30" in r.stderr
    empty = tmp_path / "empty.tsv"
    empty.write_text("")
    r = run("cluster", "-v", "x.vcf", "-s", str(empty), "-g", "g.fa")
    assert r.returncode == 1 and "Samples file empty" in r.stderr
    r = run("cluster", "-v", "x.vcf", "-s", _samples(tmp_path), "-g", str(tmp_path / "nogenome.fa"))
    assert r.returncode == 1 and "Unable to open file" in r.stderr and "Parsed information for 2 sample(s)" in r.stdout


def test_gz_files_do_not_depend_on_the_thread_count(tmp_path):
    """writeGzFile (parameter_kmers.fa.gz of the cluster stage): a large content is compressed in pieces on -p threads but written as ONE gzip member — any
    gzip reader reads it whole — and the bytes are those of a one-thread run (the pieces are cut by size, not by thread)"""
    from bayestyper_amd.host import dll

    dll.bth_write_gz.argtypes = [C.c_char_p, C.c_char_p, C.c_ulonglong, C.c_uint]
    rng = np.random.default_rng(3)
    rec = b">k\n" + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 55)) + b"\n"
    content = b"".join(b">k%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 55)) + b"\n" for i in range(40_000)) * 8   # 19 MB: five pieces
    assert len(content) > 3 * (4 << 20) and rec
    files = []
    for threads in (1, 7):
        f = str(tmp_path / f"p{threads}.gz")
        assert dll.bth_write_gz(f.encode(), content, len(content), threads) == 0
        files.append(open(f, "rb").read())
    assert files[0] == files[1]
    assert gzip.decompress(files[0]) == content            # one member: the standard library reads it in one go
    assert files[0].count(b"\x1f\x8b\x08") >= 1 and gzip.GzipFile(fileobj=__import__("io").BytesIO(files[0])).read() == content
    small = b"chr1\t10\t20\n" * 100
    f = str(tmp_path / "small.gz")
    assert dll.bth_write_gz(f.encode(), small, len(small), 4) == 0 and gzip.open(f).read() == small


def test_parameter_kmer_lines_parse_the_same_on_any_thread_count():
    """parseKmerLines (genotype's reading of parameter_kmers.fa.gz; Nucleotide::ntToBit packing: symbol i in bits 2i, 2i+1 of lo / hi): the regular layout is
    cut among the -p threads, any other layout goes line by line; a malformed line is an error either way"""
    from bayestyper_amd.host import dll

    dll.bth_parse_kmer_lines.argtypes = [C.c_char_p, C.c_ulonglong, C.c_ulonglong, C.c_uint, C.c_uint, C.c_void_p, C.c_ulonglong, C.POINTER(C.c_ulonglong), C.c_char_p, C.c_uint]
    rng = np.random.default_rng(11)
    k, n = 55, 20_001
    sym = rng.integers(0, 4, (n, k), dtype=np.uint8)
    lines = [bytes(np.frombuffer(b"ACGT", np.uint8)[row]) for row in sym]
    w = sym.astype(np.uint64)
    lo = (w[:, :32] << (2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
    hi = (w[:, 32:] << (2 * np.arange(k - 32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
    want = np.stack([lo, hi], axis=1).reshape(-1)

    def parse(text, begin, threads):
        out = np.zeros(2 * n + 2, np.uint64)
        got, err = C.c_ulonglong(0), C.create_string_buffer(400)
        rc = dll.bth_parse_kmer_lines(text, len(text), begin, k, threads, out.ctypes.data, len(out), C.byref(got), err, 400)
        return rc, out[: 2 * got.value], err.value.decode()

    head = b">k55\n"
    body = b"\n".join(lines) + b"\n"
    for text in (head + body, head + body[:-1]):            # with and without the last newline
        for threads in (1, 5, 64):
            rc, got, err = parse(text, len(head), threads)
            assert rc == 0 and np.array_equal(got, want), err
    rc, got, _ = parse(head, len(head), 4)
    assert rc == 0 and len(got) == 0                          # no k-mers at all
    bad = head + body[:56 * 100] + b"ACGN" + body[56 * 100 + 4:]
    rc, _, err = parse(bad, len(head), 8)
    assert rc == 1 and "malformed kmer line" in err
    ragged = head + body[:56 * 3] + b"ACGT\n" + body[56 * 3:]   # a short line: the line-by-line path reports it
    rc, _, err = parse(ragged, len(head), 8)
    assert rc == 1 and "malformed kmer line: ACGT" in err
