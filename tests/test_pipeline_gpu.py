"""VCF in -> VCF out: the whole path on the GPU through the C ABI and the host layer, stage by stage against the oracle.

  candidate VCF + genome --VariantFileParser--> clusters / groups / intercluster regions --VariantClusterGraph--> graphs
  sample KMC databases --makeBloom--> sample filters --findSamplePaths--> best paths --countPathKmers--> path filter
  --parseSampleKmers + countInterclusterKmers--> count table --classifyPathKmers / getHaplotypeCandidates--> cluster bundles
  --estimateGenotypes (nested groups included)--> samples --getGenotypes + GenotypeWriter--> VCF

The product side uses bayestyper_amd (libbtgpu.so kernels, libbthost.so host classes); the oracle side its own restatement of every
stage (oracle_cluster.cpp, oracle_graph.cpp, oracle_kmer.cpp, oracle_gibbs.cpp, oracle_writer.py; the flattening and the batch assembly of the
oracle side are the test's own, tests/test_cli_gpu.py — no module of the product package runs on the oracle side).  Every hand-over is
compared, and the two VCFs must be the same text."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

import _oracle  # noqa: E402
import test_cluster_stage_cpu as T  # noqa: E402
from _oracle import OrcBloom, OrcGraphs, OrcKmc, OrcTable  # noqa: E402

pytestmark = pytest.mark.gpu
K, S, SEED = 55, 2, 42
NT = "ACGT"
CODE = {c: i for i, c in enumerate(NT)}


def sample_haplotype(rng, seq, records):
    """one haplotype of a sample: a random subset of non-overlapping candidate alleles applied to the reference"""
    out, at = [], 0
    for pos, ref, alts in records:
        if pos < at or rng.random() < 0.5:
            continue
        out.append(seq[at:pos])
        out.append(alts[int(rng.integers(len(alts)))])
        at = pos + len(ref)
    out.append(seq[at:])
    return "".join(out)


def graph_from_fetch(out):
    from bayestyper_amd import synth_graphs

    g = synth_graphs.Graph()
    nv = len(out["var"])
    for v in range(nv):
        g.new_vertex()
        g.seq[v] = out["seq"][int(out["seq_off"][v]):int(out["seq_off"][v + 1])].copy()
        g.var[v], g.allele[v] = int(out["var"][v]), int(out["allele"][v])
        g.refvars[v] = [int(x) for x in out["refvar"][out["refvar_off"][v]:out["refvar_off"][v + 1]]]
        g.nested[v] = int(out["nested"][v])
        g.disconnected[v] = bool(out["flags"][v] & 1)
        g.redundant[v] = bool(out["flags"][v] & 2)
    for a, b in out["edges"]:
        g.edge(int(a), int(b))
    g.num_alleles = [int(x) for x in out["num_alleles"]]
    g.has_dep = [int(x) for x in out["dep"]]
    return g


def test_vcf_to_vcf(gpu_ctx, oracle, tmp_path):
    import oracle_writer
    from bayestyper_amd import lib, synth_graphs
    from bayestyper_amd.host import count_model, genotypes
    from bayestyper_amd.host.cluster_stage import ClusterStage, GenotypeWriter, fetch_graph

    oracle.l.orc_cluster_stage.restype = C.c_ulonglong
    oracle.l.orc_cluster_stage.argtypes = [C.c_char_p, C.c_ulonglong, C.c_uint, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_ulonglong), C.c_char_p, C.c_uint, C.c_uint,
                                           C.c_float, C.c_uint, C.c_char_p, C.c_ulonglong]
    rng = np.random.default_rng(2024)
    genome = [[f"chr{i + 1}", "".join(rng.choice(list(NT), n)), False] for i, n in enumerate([40000, 25000])]
    vcf = T.make_vcf(rng, genome, K, 45, False, extra_contig=False, sv_blocks=3)
    records = {}
    for line in vcf.split("\n"):
        if line and line[0] != "#":
            c, p, _, ref, alt = line.split("\t")[:5]
            records.setdefault(c, []).append((int(p) - 1, ref, [a for a in alt.split(",") if a != "*"]))

    # ---- cluster stage front end: product and oracle, each from the VCF ----
    st = ClusterStage(K)
    for g in genome:
        st.add_sequence(*g)
    st.set_variants(vcf_text=vcf)
    assert st.next_unit(10 ** 9)
    units_h, *_ = T.parse_dump("UNIT 1\n" + st.unit_text())
    st.sort_regions()
    regions_h = [tuple(r.split("\t")) for r in st.regions_text().strip().split("\n")]
    text_o = T.oracle_text(oracle, vcf, genome, K, 10 ** 9)
    units_o, _, regions_sorted_o, _ = T.parse_dump(text_o)
    assert units_h == units_o and [(c, str(d), str(s), str(e)) for c, d, s, e in regions_sorted_o] == regions_h
    groups_h = units_h[0]
    seqs = {name: seq for name, seq, _ in genome}
    # clusters in unit order; group structure
    where, groups, sources, out_edges, cluster_ids = [], [], [], [], []
    for gi, g in enumerate(groups_h):
        groups.append(list(range(len(where), len(where) + len(g["vertices"]))))
        sources.append(g["sources"])
        for vi, v in enumerate(g["vertices"]):
            where.append((gi, vi))
            out_edges.append(v["edges"])
            cluster_ids.append(v["cluster_idx"])
    NC = len(where)
    assert any(len(g) > 1 for g in groups) and any(out_edges)

    # ---- graphs: host C++ from the parsed clusters / the oracle's own restatement of the constructor (oracle/oracle_graph.cpp) from the oracle's dump.
    #      The oracle side is assembled by the test's own helpers (tests/test_cli_gpu.py): nothing of bayestyper_amd.synth_graphs runs on it. ----
    from test_cli_gpu import _flatten as oracle_flatten, _gibbs_batch as oracle_gibbs_batch, _graph_arrays as oracle_graph_arrays

    gs_h, gs_o = [], []
    for gi, vi in where:
        v = units_o[0][gi]["vertices"][vi]
        gs_h.append(graph_from_fetch(fetch_graph(st.graph(gi, vi), len(v["vars"]))))
        gs_o.append(oracle_graph_arrays(oracle, seqs[v["chrom"]].encode(), v["vars"], v["red"], v["contained"]))
    f_h, f_o = synth_graphs.flatten(gs_h), oracle_flatten(gs_o)
    for name in f_o:
        if name not in ("num_clusters", "refvar"):
            assert np.array_equal(np.asarray(f_h[name]).astype(np.asarray(f_o[name]).dtype), f_o[name]), name
    # a vertex's reference_variant_indices: the reference keeps them in the iteration order of an unordered_set (as the oracle does) and only
    # ever uses them as a set (VariantClusterGraph.cpp:1001-1011, 1124-1130); the product stores them ascending
    for v in range(len(f_o["refvar_off"]) - 1):
        a, b = int(f_o["refvar_off"][v]), int(f_o["refvar_off"][v + 1])
        assert sorted(f_o["refvar"][a:b].tolist()) == f_h["refvar"][a:b].tolist(), v

    # ---- samples: reads of two haplotypes each -> KMC database -> sample filter (makeBloom) -> best paths ----
    og, gf = OrcGraphs(oracle, f_o, K), lib.FindPaths(gpu_ctx, f_h, K, 32, S)
    dbs = []
    for s in range(S):
        text = "N".join(sample_haplotype(rng, seqs[c], records[c]) for c in seqs for _ in range(2))
        km, va = oracle.kmers_from_sequence(text.encode(), K)
        present = np.unique(km[va == 1], axis=0)
        cnt = (rng.poisson(14, len(present)) + 1).astype(np.uint32)
        pref = str(tmp_path / f"sample{s}")
        asc = oracle.unpack(present, K).reshape(-1, K)
        order = np.lexsort(asc.T[::-1])                     # KMC order = ascending ASCII order
        oracle.kmc_write(pref, np.ascontiguousarray(asc[order]).reshape(-1), cnt[order], K, 7, 1)
        db = OrcKmc(oracle, pref)
        dbs.append(db)
        ob = OrcBloom(oracle, db.total, 1e-3, K)
        ob.insert(db.list()[0].reshape(-1))
        gb = lib.Bloom.create(gpu_ctx, db.total, 1e-3, K, threaded=False)
        sc = lib.KmcScan(gpu_ctx, db.k, db.p, db.counter_size, db.total, db.lut())
        buf = gpu_ctx.to_device(db.payload())
        sc.make_bloom(gb, buf.ptr, 0, db.total)
        gpu_ctx.sync()
        assert np.array_equal(gb.bits(0), ob.bits(0))
        # seed + (group index + 1) * (sample + 1) + cluster index (KmerCounter.cpp:65, VariantClusterGroup.cpp:142)
        seeds = np.array([SEED + (gi + 1) * (s + 1) + cluster_ids[c] for c, (gi, _) in enumerate(where)], np.uint32)
        bo = og.find_sample_paths(ob, seeds, 32)
        gf.sample(gb, seeds)
        for x in (ob, gb, sc):
            x.close()
        buf.free()
    bg = gf.best_paths()
    for c in range(NC):
        assert bo[c].shape == bg[c].shape and np.array_equal(bo[c], bg[c]), c
        gs_h[c].paths = bg[c]
    assert sum(b.shape[0] for b in bg) > NC + NC // 3
    og.close(), gf.close()
    f_h, f_o = synth_graphs.flatten(gs_h), oracle_flatten(gs_o, bo)

    # ---- path k-mers -> path filter; sample counts and intercluster multiplicities -> count table ----
    og, gp = OrcGraphs(oracle, f_o, K), lib.Paths(gpu_ctx, f_h, K)
    n_path = 400_000
    ob, gb = OrcBloom(oracle, n_path, 1e-4, K, threaded=True), lib.Bloom.create(gpu_ctx, n_path, 1e-4, K, threaded=True)
    og.count_kmers(ob)
    gp.count_kmers(gb)
    ot, gt = OrcTable(oracle, S, K), lib.Table(gpu_ctx, 200_000, S, K)
    for chrom, decoy, a, b in regions_h:   # countInterclusterKmers over the sorted regions (autosomes: ploidy 2 for both genders)
        piece = seqs[chrom][int(a):int(b) + 1].encode()
        ot.count_intercluster(ob, piece, int(decoy), 2, 2)
        gt.count_intercluster(gb, piece, int(decoy), 2, 2)
    for s, db in enumerate(dbs):
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
    assert co["kmer_has_counts"].sum() > len(co["kmer_has_counts"]) // 2 and co["kmer_counts"].sum() > 10000   # the samples' reads reached the bundles

    # ---- Gibbs over the parsed group structure (nested clusters follow their parents) ----
    kw = dict(seed=SEED, chains=2, burn=10, iters=40)
    flat_o = oracle_gibbs_batch(co, f_o, groups, S, np.full((len(groups), S), 2, np.uint8), [0] * S, cluster_ids, sources, out_edges)
    flat_h = synth_graphs.gibbs_batch_from_candidates(cg, f_h, groups, S, cluster_ids=cluster_ids, sources=sources, out_edges=out_edges)
    lut_o = _oracle.build_luts(oracle, S, mean=15.0, var=30.0)
    lut_h = count_model.build_luts(S, mean=15.0, var=30.0)
    assert np.array_equal(lut_o[0], lut_h[0]) and np.array_equal(lut_o[1], lut_h[1])
    ogb = _oracle.OrcGibbs(oracle, flat_o, *lut_o, **kw)
    ogb.run(8)
    ro = ogb.results()
    ogb.close()
    ggb = lib.Gibbs(gpu_ctx, flat_h, *lut_h, **kw)
    ggb.run()
    rg = ggb.results()
    ggb.close()
    for name in ("dip_off", "h1", "h2", "freq", "cell_off"):
        assert np.array_equal(ro[name], rg[name]), name
    assert np.array_equal(ro["stats"][:, :, 0], rg["stats"][:, :, 0]) and np.allclose(ro["stats"], rg["stats"], rtol=1e-9, atol=1e-12)

    # ---- output VCF ----
    names = ["sampleA", "sampleB"]
    mf = genotypes.min_fraction_observed_kmers([15.0] * S)
    ploidy = np.full(S, 2, np.uint8)
    w = GenotypeWriter(st, names)
    fn = oracle.l.orc_cluster_output_columns
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_uint, C.c_uint, C.c_uint] + [C.c_void_p] * 3 + [C.c_ulonglong] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_char_p, C.c_ulonglong]
    lines = {}
    hv = np.concatenate([[0], np.cumsum(flat_o["num_haplotypes"].astype(np.int64) * flat_o["num_variants"].astype(np.int64))])
    for c, (gi, vi) in enumerate(where):
        g, v = units_o[0][gi], units_o[0][gi]["vertices"][vi]
        H = int(flat_h["num_haplotypes"][c])
        e0, e1 = int(rg["dip_off"][c]), int(rg["dip_off"][c + 1])
        cells = slice(int(rg["cell_off"][c]), int(rg["cell_off"][c + 1]))
        w.add_cluster(gi, vi, S, H, flat_h["hap_allele"][hv[c]:hv[c + 1]], rg["h1"][e0:e1], rg["h2"][e0:e1], rg["freq"][e0:e1], rg["stats"][cells], ploidy, mf)
        cols = genotypes.cluster_output_columns(flat_o, ro, c, ploidy, mf, fn=fn)
        infos = v["vars"]
        vcr = "%s:%d-%d" % (v["chrom"], infos[0][0] + 1, max(pos + max(rl for rl, _ in alts) for pos, _, _, alts in infos))
        for (pos, vid, dep, alts), col, aco in zip(infos, cols, v["aco"]):
            full = [(rl, seq, a) for (rl, seq), a in zip(alts, aco)]
            lines.setdefault(v["chrom"], []).append((pos, oracle_writer.vcf_line(v["chrom"], seqs[v["chrom"]], pos, vid, bool(dep), full, col, len(infos), vcr, len(g["vertices"]), g["region"],
                                                                                int(flat_o["num_haplotypes"][c]))))
    want = oracle_writer.vcf_header("genome.fa", genome, "", "", names)
    for name, _, _ in genome:
        for _, line in sorted(lines.get(name, [])):
            want += line
    got = w.text("genome.fa", "", "")
    assert got == want
    n_var = sum(len(x) for x in lines.values())
    assert w.finalise(str(tmp_path / "calls"), True) == n_var and n_var > 100
    body = [x.split("\t") for x in got.split("\n") if x and x[0] != "#"]
    assert sum(x[6] == "PASS" for x in body) > n_var // 2              # most variants get called genotypes
    assert any(x[9].split(":")[0] in ("0/1", "1/1") for x in body)     # and non-reference calls exist
    for x in (og, gp, ob, gb, ot, gt, omg, gmg, w, st):
        x.close()
