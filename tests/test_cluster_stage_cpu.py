"""Cluster-stage front end on the host (bayestyper_amd/host/VariantFileParser.cpp: VCF -> variant clusters -> groups, intercluster
regions) against the oracle's restatement of VariantFileParser.cpp:185-1160 / VariantClusterGroup.cpp:47-107 (oracle_cluster.cpp),
plus properties that hold independently of either implementation.

Parity here is "unpinned" (the reference TU needs Boost, absent in this image: no reference-generated fixture can exist); what
the comparison pins is that the product and a statement-by-statement restatement agree on every cluster index, vertex order,
edge order, flank, contained cluster, variant record, region and counter."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import _oracle  # noqa: E402


@pytest.fixture(scope="module")
def orc():
    o = _oracle.load_oracle()
    o.l.orc_cluster_stage.restype = C.c_ulonglong
    o.l.orc_cluster_stage.argtypes = [C.c_char_p, C.c_ulonglong, C.c_uint, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_ulonglong), C.c_char_p, C.c_uint, C.c_uint,
                                      C.c_float, C.c_uint, C.c_char_p, C.c_ulonglong]
    return o


def oracle_text(orc, vcf, genome, k, min_unit_variants, max_allele_length=500000, thr=0.5):
    names = (C.c_char_p * len(genome))(*[g[0].encode() for g in genome])
    seqs = (C.c_char_p * len(genome))(*[g[1].encode() for g in genome])
    lens = (C.c_ulonglong * len(genome))(*[len(g[1]) for g in genome])
    decoy = bytes(int(g[2]) for g in genome)
    v = vcf.encode()
    n = orc.l.orc_cluster_stage(v, len(v), len(genome), names, seqs, lens, decoy, k, max_allele_length, thr, min_unit_variants, None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    orc.l.orc_cluster_stage(v, len(v), len(genome), names, seqs, lens, decoy, k, max_allele_length, thr, min_unit_variants, buf, n)
    return buf.raw[:n].decode()


def run_all(stage, min_unit_variants):
    """every unit of the file, in the text layout of the oracle's orc_cluster_stage"""
    out, unit = [], 1
    while True:
        done = stage.next_unit(min_unit_variants)
        out.append(f"UNIT {unit}\n" + stage.unit_text())
        unit += 1
        if done:
            break
    text = "".join(out) + "REGIONS\n" + stage.regions_text()
    stage.sort_regions()
    return text + "SORTED\n" + stage.regions_text() + "COUNTERS\n" + stage.counters_text()


def host_text(vcf, genome, k, min_unit_variants, max_allele_length=500000, thr=0.5):
    from bayestyper_amd.host.cluster_stage import ClusterStage

    st = ClusterStage(k, max_allele_length, thr)
    for name, seq, dec in genome:
        st.add_sequence(name, seq, dec)
    st.set_variants(vcf_text=vcf)
    t = run_all(st, min_unit_variants)
    st.close()
    return t


NT = "ACGT"


def random_genome(rng, lengths, decoys=()):
    out = []
    for i, n in enumerate(lengths):
        s = "".join(rng.choice(list(NT), n))
        if i == 0:   # a lower-case stretch and a run of N
            s = s[:300] + s[300:360].lower() + s[360:700] + "N" * 25 + s[725:]
        out.append([f"chr{i + 1}", s, i in decoys])
    return out


def make_vcf(rng, genome, k, n_per_chrom, with_format, extra_contig=True, sv_blocks=0):
    """candidate variants of every flavour the parser distinguishes, sorted by contig and position"""
    header = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" + ("\tFORMAT\ts1" if with_format else "") + "\n"
    rows = []
    vid = 0
    for name, seq, _ in genome:
        L = len(seq)
        pos_used = set()
        recs = []
        # background: SNVs / small indels / multi-allelic / MNV, some in dense runs so that they cluster
        anchors = sorted(rng.choice(np.arange(2, L - 2), size=n_per_chrom, replace=False).tolist())
        for a in anchors:
            burst = int(rng.integers(1, 4)) if rng.random() < 0.3 else 1
            p = a
            for _ in range(burst):
                if p in pos_used or p >= L - 2:
                    break
                pos_used.add(p)
                ref_len = int(rng.choice([1, 1, 1, 2, 5]))
                ref = seq[p:p + ref_len]
                kind = rng.random()
                if kind < 0.5:     # SNV / substitution
                    alts = ["".join(NT[(NT.find(c.upper()) + 1 + int(rng.integers(0, 3))) % 4] if c.upper() in NT else "A" for c in ref)]
                elif kind < 0.7:   # insertion
                    alts = [ref + "".join(rng.choice(list(NT), int(rng.integers(1, 9))))]
                elif kind < 0.85 and ref_len > 1:   # deletion
                    alts = [ref[0]]
                else:              # multi-allelic mixture
                    alts = [ref[0] + "".join(rng.choice(list(NT), 3)), NT[(NT.find(ref[0].upper()) + 1) % 4] + ref[1:]]
                    alts = list(dict.fromkeys(alts))
                alts = [x for x in alts if x.upper() != ref.upper()] or [NT[(NT.find(ref[0].upper()) + 2) % 4] + ref[1:]]
                recs.append((p, ref, alts))
                p += int(rng.integers(1, k))
        # long deletions spanning later variants (nested clusters, '*' alleles)
        for _ in range(max(1, n_per_chrom // 12)):
            p = int(rng.integers(k, max(k + 1, L - 6 * k)))
            if p in pos_used:
                continue
            pos_used.add(p)
            dl = int(rng.integers(2 * k, 4 * k))
            recs.append((p, seq[p:p + dl], [seq[p]]))
        # "SV soup": overlapping long deletions with small variants inside -> nested clusters, clusters touched by a later allele from
        # two sides (merge sets), merges of merge sets
        slots = list(range(2 * k, max(2 * k + 1, L - 20 * k), 20 * k))
        rng.shuffle(slots)
        for b0 in slots[:sv_blocks]:
            def put(p, ref, alts):
                if p not in pos_used and ref and all(c in NT for c in ref.upper()):
                    pos_used.add(p)
                    recs.append((p, ref, alts))
            if rng.random() < 0.5:   # three levels: a deletion inside a deletion with an SNV inside, and an allele leaving the inner one
                put(b0, seq[b0:b0 + 14 * k], [seq[b0]])
                put(b0 + 3 * k, seq[b0 + 3 * k:b0 + 9 * k], [seq[b0 + 3 * k]])
                put(b0 + 5 * k, seq[b0 + 5 * k], [NT[(NT.find(seq[b0 + 5 * k].upper()) + 1) % 4]])
                if rng.random() < 0.6:
                    q = b0 + 5 * k + int(rng.integers(1, k))
                    put(q, seq[q:q + 5 * k], [seq[q]])
            else:
                for _ in range(int(rng.integers(2, 5))):
                    p = b0 + int(rng.integers(0, 8 * k))
                    dl = int(rng.integers(k, 6 * k))
                    put(p, seq[p:p + dl], [seq[p]] if rng.random() < 0.7 else [seq[p], seq[p:p + dl // 2]])
            for _ in range(int(rng.integers(2, 7))):
                p = b0 + int(rng.integers(0, 14 * k))
                put(p, seq[p], [NT[(NT.find(seq[p].upper()) + 1) % 4] if seq[p].upper() in NT else "A"])
        # tandem duplication insertion (copy-number context extends the group end)
        for _ in range(2):
            p = int(rng.integers(3 * k, max(3 * k + 1, L - 8 * k)))
            if p in pos_used:
                continue
            pos_used.add(p)
            unit = seq[p + 1:p + 1 + 2 * k]
            recs.append((p, seq[p], [seq[p] + unit]))
        # chromosome ends, a reference mismatch
        recs.append((3, seq[3], [NT[(NT.find(seq[3].upper()) + 1) % 4]]))
        recs.append((L - 5, seq[L - 5], [NT[(NT.find(seq[L - 5].upper()) + 1) % 4]]))
        q = int(rng.integers(k, L - k))
        if q not in pos_used and seq[q].upper() in NT:
            pos_used.add(q)
            recs.append((q, NT[(NT.find(seq[q].upper()) + 1) % 4], [NT[(NT.find(seq[q].upper()) + 2) % 4]]))
        seen = set()
        open_end = -1
        for p, ref, alts in sorted(recs):
            if p in seen or not ref or any(c.upper() not in NT for c in ref):
                continue
            seen.add(p)
            alts = [a for a in dict.fromkeys(alts) if a.upper() != ref.upper()]
            if not alts:
                continue
            alt_field = ",".join(alts)
            if open_end >= p and rng.random() < 0.7:   # inside an earlier allele: the missing allele may be spelled out
                alt_field += ",*"
            open_end = max(open_end, p + len(ref) - 1)
            info = "." if rng.random() < 0.5 else "AN=2;ACO=" + ",".join(rng.choice(["cs1", "cs2:cs3", "."], len(alts))) + ";X=1"
            if alt_field.endswith(",*") and info != ".":
                info = "AN=2"
            rows.append(f"{name}\t{p + 1}\tv{vid}\t{ref}\t{alt_field}\t.\t.\t{info}" + ("\tGT\t0/1" if with_format else ""))
            vid += 1
        if extra_contig and name == genome[0][0]:
            rows.append("chrUn\t10\tvu\tA\tC\t.\t.\t." + ("\tGT\t0/1" if with_format else ""))
    return header + "\n".join(rows) + "\n"


def parse_dump(text):
    """-> units: list of groups: {region, nvar, sources, vertices: [{cluster_idx, left, right, edges, contained, vars: [(pos, id, dep, alts)]}]}"""
    units, cur = [], None
    section = None
    regions, sorted_regions, counters = [], [], ""
    for line in text.split("\n"):
        if line.startswith("UNIT "):
            cur = []
            units.append(cur)
            section = "unit"
        elif line in ("REGIONS", "SORTED", "COUNTERS"):
            section = line
        elif section == "unit" and line.startswith("GROUP "):
            m = re.match(r"GROUP (\d+) region=(\S+) nvar=(\d+) sources=(\S*)", line)
            cur.append({"region": m.group(2), "nvar": int(m.group(3)), "sources": [int(x) for x in m.group(4).split(",") if x], "vertices": []})
        elif section == "unit" and line.startswith(" VERTEX "):
            m = re.match(r" VERTEX (\d+) cluster_idx=(\d+) chrom=(\S+) left=(\d+) right=(\d+) edges=(\S*) contained=(\S*)", line)
            cur[-1]["vertices"].append({"cluster_idx": int(m.group(2)), "chrom": m.group(3), "left": int(m.group(4)), "right": int(m.group(5)),
                                        "edges": [int(x) for x in m.group(6).split(",") if x],
                                        "contained": [tuple(int(y) for y in x.split(":")) for x in m.group(7).split(";") if x], "vars": []})
        elif section == "unit" and line.startswith("  VAR "):
            m = re.match(r"  VAR pos=(\d+) id=(\S+) dep=(\d) type=(\d) red=(\d+) alts=(\S+)", line)
            alts = [(int(a.split(":")[0]), a.split(":")[1]) for a in m.group(6).split("|")]
            cur[-1]["vertices"][-1]["vars"].append((int(m.group(1)), m.group(2), int(m.group(3)), alts))
            cur[-1]["vertices"][-1].setdefault("red", []).append(int(m.group(5)))
            cur[-1]["vertices"][-1].setdefault("aco", []).append([a.split(":", 2)[2] for a in m.group(6).split("|")])
        elif section == "REGIONS" and line:
            c, d, s, e = line.split("\t")
            regions.append((c, int(d), int(s), int(e)))
        elif section == "SORTED" and line:
            c, d, s, e = line.split("\t")
            sorted_regions.append((c, int(d), int(s), int(e)))
        elif section == "COUNTERS" and line:
            counters = line
    return units, regions, sorted_regions, counters


CASES = [(11, 15, [4000, 2500, 1200], (2,), 40, True, 25, 0), (12, 15, [6000], (), 90, False, 10 ** 9, 3), (13, 55, [30000, 9000], (), 70, True, 40, 4),
         (14, 31, [12000, 5000, 800], (1,), 60, False, 1, 2), (15, 15, [20000], (), 20, True, 10 ** 9, 30), (16, 21, [40000, 30000], (), 10, False, 200, 40)]   # 30+: the chromosome is tiled with SV blocks


def structure_stats(units):
    groups = [g for u in units for g in u]
    return {"groups": len(groups), "multi": sum(len(g["vertices"]) > 1 for g in groups), "nested": sum(any(v["edges"] for v in g["vertices"]) for g in groups),
            "merged": sum(sorted(v["cluster_idx"] for v in g["vertices"]) != list(range(len(g["vertices"]))) for g in groups),
            "depth2": sum(any(any(g["vertices"][c]["edges"] for c in v["edges"]) for v in g["vertices"]) for g in groups),
            "max_vertices": max(len(g["vertices"]) for g in groups)}


@pytest.mark.parametrize("seed,k,lengths,decoys,n_per_chrom,with_format,min_unit,sv_blocks", CASES)
def test_host_parser_matches_oracle(orc, seed, k, lengths, decoys, n_per_chrom, with_format, min_unit, sv_blocks):
    rng = np.random.default_rng(seed)
    genome = random_genome(rng, lengths, decoys)
    genome.append(["chrQuiet", "".join(rng.choice(list(NT), 3 * k)), False])   # a chromosome without variants
    vcf = make_vcf(rng, genome[:-1], k, n_per_chrom, with_format, sv_blocks=sv_blocks)
    want = oracle_text(orc, vcf, genome, k, min_unit)
    got = host_text(vcf, genome, k, min_unit)
    assert "ERROR" not in want
    assert got == want
    units, regions, sorted_regions, counters = parse_dump(got)
    assert sum(len(u) for u in units) > 5
    if min_unit < 100:
        assert len(units) > 1
    if sv_blocks >= 30:   # the structures the SV soup is there for did occur
        st = structure_stats(units)
        assert st["nested"] >= 5 and st["merged"] >= 3 and st["depth2"] >= 1 and st["max_vertices"] >= 3, st


def test_properties_of_the_clustering(orc):
    """independent of either implementation: every accepted variant sits in exactly one cluster; clusters of different groups are
    at least k apart (after the copy-number extension, so only >= k is checked); a nested cluster lies strictly inside its parent
    and is listed among its contained clusters; the regions are exactly the complement of the accepted variants' reference spans."""
    k = 15
    rng = np.random.default_rng(5)
    genome = random_genome(rng, [5000, 3000], ())
    vcf = make_vcf(rng, genome, k, 80, True, extra_contig=False)
    text = host_text(vcf, genome, k, 30)
    units, regions, sorted_regions, counters = parse_dump(text)
    groups = [g for u in units for g in u]
    seen_pos = set()
    spans = {name: np.zeros(len(seq), bool) for name, seq, _ in genome}
    n_nested = 0
    for g in groups:
        assert g["nvar"] == sum(len(v["vars"]) for v in g["vertices"])
        assert sorted(g["sources"]) == [i for i, v in enumerate(g["vertices"]) if not any(i in w["edges"] for w in g["vertices"])]
        for vi, v in enumerate(g["vertices"]):
            assert v["left"] == v["vars"][0][0] and v["right"] >= v["vars"][-1][0]
            for pos, vid, dep, alts in v["vars"]:
                assert (v["chrom"], pos) not in seen_pos
                seen_pos.add((v["chrom"], pos))
                for ref_len, _ in alts:
                    spans[v["chrom"]][pos:pos + ref_len] = True
                    assert pos + ref_len - 1 <= v["right"]
            for child in v["edges"]:
                c = g["vertices"][child]
                assert v["left"] < c["left"] and c["right"] < v["right"]
                assert (c["cluster_idx"], c["left"], c["right"]) in v["contained"]
                n_nested += 1
    assert n_nested > 0
    # groups of a chromosome, ordered by position, do not come within a k-mer of each other
    by_chrom = {}
    for g in groups:
        chrom, rng_ = g["region"].split(":")
        a, b = (int(x) for x in rng_.split("-"))
        by_chrom.setdefault(chrom, []).append((a, b))
    for lst in by_chrom.values():
        lst.sort()
        for (a0, b0), (a1, b1) in zip(lst, lst[1:]):
            assert a1 - b0 >= k
    # intercluster regions (those of at least k nucleotides are listed) = complement of the accepted reference spans
    for name, seq, _ in genome:
        free = ~spans[name]
        edges = np.flatnonzero(np.diff(np.concatenate([[0], free.view(np.int8), [0]])))
        want = [(name, 0, int(s), int(e) - 1) for s, e in zip(edges[::2], edges[1::2]) if e - s >= k]
        assert [r for r in regions if r[0] == name] == want
    lens = [e - s for _, _, s, e in sorted_regions]
    assert lens == sorted(lens, reverse=True) and sorted(sorted_regions) == sorted(regions)
    m = re.search(r"total=(\d+) parsed=(\d+) clusters=(\d+) groups=(\d+)", counters)
    assert int(m.group(1)) == int(m.group(2)) == vcf.count("\n") - 2
    assert int(m.group(3)) == sum(len(g["vertices"]) for g in groups) and int(m.group(4)) == len(groups)


def test_errors_and_file_input(tmp_path, orc):
    import gzip

    from bayestyper_amd.host.cluster_stage import ClusterStage

    k = 15
    rng = np.random.default_rng(8)
    genome = random_genome(rng, [3000], ())
    vcf = make_vcf(rng, genome, k, 30, False, extra_contig=False)
    # .vcf and .vcf.gz files give the same result as the text
    want = host_text(vcf, genome, k, 10 ** 9)
    for fn, opener in (("c.vcf", open), ("c.vcf.gz", gzip.open)):
        with opener(tmp_path / fn, "wt") as f:
            f.write(vcf)
        st = ClusterStage(k)
        st.add_sequence(*genome[0])
        st.set_variants(path=str(tmp_path / fn))
        assert run_all(st, 10 ** 9) == want
        st.close()
    # unsorted positions / unsorted contigs / missing header are errors, not silent output
    rows = vcf.strip().split("\n")
    head, data = rows[:2], rows[2:]
    bad = "\n".join(head + [data[5]] + data[:5]) + "\n"
    st = ClusterStage(k)
    st.add_sequence(*genome[0])
    st.set_variants(vcf_text=bad)
    with pytest.raises(ValueError, match="sorted by position"):
        st.next_unit(10 ** 9)
    st.close()
    st = ClusterStage(k)
    with pytest.raises(ValueError, match="#CHROM"):
        st.set_variants(vcf_text="chr1\t5\tx\tA\tC\t.\t.\t.\n")
    with pytest.raises(ValueError, match="neither"):
        st.set_variants(path=str(tmp_path / "c.txt"))
    st.close()
    st = ClusterStage(k)
    st.add_sequence("a", "ACGT")
    with pytest.raises(ValueError, match="multiple times"):
        st.add_sequence("a", "ACGT")
    st.close()


def test_allele_helpers_and_copy_number_extension():
    """a tandem duplication pushes the group's end over the repeated copies: a variant that would otherwise start its own group
    joins... no — stays in its own cluster but in the SAME group as long as it lies inside the extension"""
    k = 15
    rng = np.random.default_rng(21)
    unit = "".join(rng.choice(list(NT), 3 * k))
    left = "".join(rng.choice(list(NT), 200))
    right = "".join(rng.choice(list(NT), 400))
    seq = left + unit + unit + right            # two copies of the unit in the reference
    p = len(left) - 1                           # insertion of a third copy right before them
    far = len(left) + 2 * len(unit) - 5         # an SNV near the end of the second copy: > k away from the insertion's flank
    alt_snv = NT[(NT.find(seq[far]) + 1) % 4]
    vcf = ("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + f"c\t{p + 1}\tdup\t{seq[p]}\t{seq[p] + unit}\t.\t.\t.\n" + f"c\t{far + 1}\tsnv\t{seq[far]}\t{alt_snv}\t.\t.\t.\n")
    units, *_ = parse_dump(host_text(vcf, [["c", seq, False]], k, 10 ** 9))
    assert len(units[0]) == 1 and len(units[0][0]["vertices"]) == 2      # one group, two clusters
    units, *_ = parse_dump(host_text(vcf, [["c", seq, False]], k, 10 ** 9, thr=1.1))   # extension disabled (threshold above any fraction)
    assert len(units[0]) == 2


def test_parsed_clusters_become_graphs():
    """VCF -> clusters -> VariantClusterGraph: the graph the stage builds from a parsed cluster equals the one bth_graph_build makes
    from the same cluster's records as they appear in the dump (variants, redundant nucleotides, dependency flags, contained
    clusters), for every cluster of an SV-rich unit; nested clusters appear as nested-cluster vertices in their parent."""
    import test_host_graph_cpu as G
    from bayestyper_amd.host.cluster_stage import ClusterStage, fetch_graph

    k = G.K
    rng = np.random.default_rng(31)
    genome = random_genome(rng, [60000], ())
    vcf = make_vcf(rng, genome, k, 40, False, extra_contig=False, sv_blocks=30)
    st = ClusterStage(k)
    st.add_sequence(*genome[0])
    st.set_variants(vcf_text=vcf)
    assert st.next_unit(10 ** 9)
    units, *_ = parse_dump("UNIT 1\n" + st.unit_text())
    code = {c: i for i, c in enumerate("ACGT")}
    n_nested_vertices = 0
    sizes = st.unit_sizes()
    assert len(sizes) == len(units[0])
    for gi, g in enumerate(units[0]):
        assert sizes[gi] == len(g["vertices"])
        for vi, v in enumerate(g["vertices"]):
            variants = [{"pos": pos, "alts": [(rl, [code[c] for c in seq]) for rl, seq in alts], "has_dependency": bool(dep), "num_redundant": red}
                        for (pos, vid, dep, alts), red in zip(v["vars"], v["red"])]
            want = G._build(genome[0][1].encode(), variants, [(lf, rf, ci) for ci, lf, rf in v["contained"]])
            got = fetch_graph(st.graph(gi, vi), len(variants))
            for key in want:
                assert np.array_equal(want[key], got[key]), (gi, vi, key)
            nested = got["nested"][got["nested"] != 0xFFFFFFFF]
            assert sorted(set(nested.tolist())) == sorted(ci for ci, _, _ in v["contained"])
            n_nested_vertices += len(nested)
            assert np.array_equal(got["num_alleles"], [1 + len(x["alts"]) + int(x["has_dependency"]) for x in variants])
    assert n_nested_vertices > 5
    st.close()
