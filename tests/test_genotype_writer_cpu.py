"""bthost::GenotypeWriter (host/GenotypeWriter.cpp) against the oracle: the genotype-derived columns from oracle_gibbs.cpp
(orc_cluster_output_columns), the line / header assembly from oracle/oracle_writer.py.  The clusters come from the host's
VariantFileParser on a synthetic VCF; the sampler results are fabricated (any consistent arrays exercise the writer)."""
import ctypes as C
import gzip
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

import _oracle  # noqa: E402
import oracle_writer  # noqa: E402
import test_cluster_stage_cpu as T  # noqa: E402


def fabricate_results(rng, S, H, num_alleles, ploidy):
    """random but well-formed sampler output of one cluster: haplotype allele indices, sampled diplotypes with frequencies, allele k-mer statistics"""
    V = len(num_alleles)
    hap_allele = np.zeros((H, V), np.uint16)
    for v, na in enumerate(num_alleles):
        hap_allele[:, v] = rng.integers(0, na, H)
    pairs = sorted({(int(min(a, b)), int(max(a, b))) for a, b in rng.integers(0, H, (6, 2))})
    h1 = np.array([p[0] for p in pairs], np.uint16)
    h2 = np.array([p[1] for p in pairs], np.uint16)
    freq = np.zeros((len(pairs), S), np.uint32)
    for s in range(S):
        if ploidy[s] == 0:
            continue
        w = rng.random(len(pairs)) ** 6    # one diplotype usually dominates -> confident calls
        freq[:, s] = rng.multinomial(500, w / w.sum())
    A = int(sum(num_alleles))
    stats = np.zeros((S, A, 3, 4))
    for s in range(S):
        for a in range(A):
            if rng.random() < 0.85:
                n = int(rng.integers(1, 500))
                stats[s, a, 0] = [n, 1.0, rng.uniform(0, 40), 1.0]                     # count_stats: mean = NAK
                stats[s, a, 1] = [n, 1.0, rng.uniform(0, 1), 0.1]                      # fraction_stats: mean = FAK
                stats[s, a, 2] = [n, 1.0, rng.uniform(0, 30), 2.0]                     # mean_stats: mean = MAC
    return hap_allele, h1, h2, freq, stats


def test_vcf_matches_oracle(oracle, tmp_path):
    from bayestyper_amd.host.cluster_stage import ClusterStage, GenotypeWriter
    from bayestyper_amd.host import genotypes

    k, S = 15, 3
    rng = np.random.default_rng(77)
    genome = T.random_genome(rng, [9000, 4000, 1500], (2,))
    vcf = T.make_vcf(rng, genome, k, 40, True, extra_contig=False, sv_blocks=6)
    st = ClusterStage(k)
    for g in genome:
        st.add_sequence(*g)
    st.set_variants(vcf_text=vcf)
    assert st.next_unit(10 ** 9)
    units, *_ = T.parse_dump("UNIT 1\n" + st.unit_text())
    names = ["s_one", "s_two", "s_three"]
    w = GenotypeWriter(st, names)
    mf = genotypes.min_fraction_observed_kmers([15.0] * S)
    fn = oracle.l.orc_cluster_output_columns
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_uint, C.c_uint, C.c_uint] + [C.c_void_p] * 3 + [C.c_ulonglong] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_char_p, C.c_ulonglong]
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    seqs = {name: seq for name, seq, _ in genome}
    lines = {}
    order = [(gi, vi) for gi, g in enumerate(units[0]) for vi in range(len(g["vertices"]))]
    rng.shuffle(order)   # clusters arrive in any order (worker threads in the reference); the writer sorts
    seen_dep = seen_anc = False
    for gi, vi in order:
        g, v = units[0][gi], units[0][gi]["vertices"][vi]
        ploidy = np.array([2, 1 if gi % 3 == 0 else 2, 0 if gi % 5 == 0 else 2], np.uint8)
        num_alleles = [1 + len(alts) + dep for (_, _, dep, alts) in v["vars"]]
        H = int(rng.integers(2, 7))
        hap_allele, h1, h2, freq, stats = fabricate_results(rng, S, H, num_alleles, ploidy)
        w.add_cluster(gi, vi, S, H, hap_allele, h1, h2, freq, stats, ploidy, mf)
        # oracle: genotype columns (C++) + line assembly (Python)
        vna, vdep = np.array(num_alleles, np.uint16), np.array([dep for (_, _, dep, _) in v["vars"]], np.uint8)
        args = [S, H, len(num_alleles), p(hap_allele), p(vna), p(vdep), len(h1), p(h1), p(h2), p(np.ascontiguousarray(freq).reshape(-1)), p(np.ascontiguousarray(stats).reshape(-1)), p(ploidy),
                0.99, 1.0, p(np.ascontiguousarray(mf, np.float32))]
        n = fn(*args, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        fn(*args, buf, n)
        cols = buf.raw[:n].decode().split("\n")[:-1]
        infos = [(pos, vid, dep, alts) for (pos, vid, dep, alts) in v["vars"]]
        start = infos[0][0] + 1
        end = max(pos + 1 + max(rl for rl, _ in alts) - 1 for pos, _, _, alts in infos)
        vcr = "%s:%d-%d" % (v["chrom"], start, end)
        for (pos, vid, dep, alts), col, aco in zip(infos, cols, v["aco"]):
            full_alts = [(rl, seq, a) for (rl, seq), a in zip(alts, aco)]
            lines.setdefault(v["chrom"], []).append((pos, oracle_writer.vcf_line(v["chrom"], seqs[v["chrom"]], pos, vid, bool(dep), full_alts, col, len(infos), vcr, len(g["vertices"]),
                                                                                g["region"], H)))
            seen_dep |= bool(dep)
            seen_anc |= ";ANC=" in col
    want = oracle_writer.vcf_header("ref.fa", genome, "##graph=x\n", "##genotype=y\n", names)
    for name, _, _ in genome:
        for _, line in sorted(lines.get(name, [])):
            want += line
    got = w.text("ref.fa", "##graph=x\n", "##genotype=y\n")
    assert got == want
    assert seen_dep and seen_anc and "##contig=<ID=chr3" not in got and got.count("\n") == want.count("\n") > 60
    # files
    n = w.finalise(str(tmp_path / "out"), False, "ref.fa", "##graph=x\n", "##genotype=y\n")
    assert n == sum(len(x) for x in lines.values()) and open(tmp_path / "out.vcf").read() == want
    w.finalise(str(tmp_path / "outz"), True, "ref.fa", "##graph=x\n", "##genotype=y\n")
    assert gzip.open(tmp_path / "outz.vcf.gz", "rt").read() == want
    w.close(), st.close()
