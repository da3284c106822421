"""The two command lines end to end on the GPU: `bayesTyper cluster` then `bayesTyper genotype` run as subprocesses on the plumbing
dataset of BASELINE.json configs[0] (C1: synthetic reference, SNV candidates, one sample with KMC database + sample Bloom filter,
tests/c1_dataset.py), and every file they write compared with what the ORACLE pipeline produces from the same inputs and seed:

  oracle_cluster.cpp (VCF -> clusters / groups / inter-cluster regions)  ->  oracle_graph.cpp (graphs)  ->  oracle_kmer.cpp (best paths,
  path k-mers, multigroup k-mers, parameter k-mers; their selection order from the REFERENCE's own HybridHash when oracle/_ref is built)
  ->  count table (inter-cluster multiplicities, KMC scan, classification), k-mer statistics -> NB fit  ->  oracle_gibbs.cpp
  (estimateNoise, estimateGenotypes, getGenotypes)  ->  oracle_writer.py (VCF lines)

No product code runs on the oracle side (the group / batch assembly below is the test's own)."""
import ctypes as C
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _oracle  # noqa: E402
import c1_dataset  # noqa: E402
import test_cluster_stage_cpu as T  # noqa: E402
from _oracle import OrcBloom, OrcGraphs, OrcKmc, OrcTable  # noqa: E402

pytestmark = pytest.mark.gpu
K = 55
EXE = os.path.join(ROOT, "bayestyper_amd", "bayesTyper")
CODE = {c: i for i, c in enumerate("ACGT")}
KC_PARAMETER, KC_DECOY = 0x20, 0x08


def _graph_arrays(orc, chrom_ascii, vars_, red, contained):
    """one cluster's graph from the oracle's restatement of the reference constructor (oracle/oracle_graph.cpp)"""
    L = orc.l
    L.orc_graph_build.restype = C.c_void_p
    L.orc_graph_build.argtypes = [C.c_uint, C.c_char_p, C.c_ulonglong, C.c_uint] + [C.c_void_p] * 6 + [C.c_char_p, C.c_uint] + [C.c_void_p] * 3
    L.orc_graph_free.argtypes = [C.c_void_p]
    L.orc_graph_sizes.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_graph_fetch.argtypes = [C.c_void_p] * 12
    pos = np.array([v[0] for v in vars_], np.uint32)
    nalt = np.array([len(v[3]) for v in vars_], np.uint32)
    redn = np.array(red, np.uint32)
    dep = np.array([v[2] for v in vars_], np.uint8)
    ref_len = np.array([a[0] for v in vars_ for a in v[3]], np.uint32)
    seqs = [a[1].encode() for v in vars_ for a in v[3]]
    off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint32)
    ci = np.array([c[0] for c in contained] + [0], np.uint32)
    cl = np.array([c[1] for c in contained] + [0], np.uint32)
    cr = np.array([c[2] for c in contained] + [0], np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    h = L.orc_graph_build(K, chrom_ascii, len(chrom_ascii), len(vars_), p(pos), p(nalt), p(redn), p(dep), p(ref_len), p(off), b"".join(seqs), len(contained), p(cl), p(cr), p(ci))
    sizes = np.zeros(4, np.uint64)
    L.orc_graph_sizes(h, p(sizes))
    nv, ne, nnt, nref = [int(x) for x in sizes]
    out = {"seq_off": np.zeros(nv + 1, np.uint64), "seq": np.zeros(max(nnt, 1), np.uint8), "var": np.zeros(nv, np.uint16), "allele": np.zeros(nv, np.uint16),
           "flags": np.zeros(nv, np.uint8), "nested": np.zeros(nv, np.uint32), "refvar_off": np.zeros(nv + 1, np.uint32), "refvar": np.zeros(max(nref, 1), np.uint16),
           "edges": np.zeros(max(2 * ne, 1), np.uint32), "num_alleles": np.zeros(len(vars_), np.uint16), "dep": np.zeros(len(vars_), np.uint8)}
    L.orc_graph_fetch(h, *[p(out[k]) for k in ("seq_off", "seq", "var", "allele", "flags", "nested", "refvar_off", "refvar", "edges", "num_alleles", "dep")])
    L.orc_graph_free(h)
    out["seq"], out["refvar"], out["edges"] = out["seq"][:nnt], out["refvar"][:nref], out["edges"][: 2 * ne].reshape(-1, 2)
    return out


def _flatten(graphs, paths=None):
    """oracle graphs (+ best paths) -> the flat arrays OrcGraphs takes (bt_paths_batch field names)"""
    f = {"num_clusters": len(graphs)}
    vertex_off, seq_off, roff, poff, voff, in_off = [0], [0], [0], [0], [0], [0]
    seqs, rv, pv, in_src, num_paths = [], [], [], [], []
    for c, g in enumerate(graphs):
        nv = len(g["var"])
        ins = [[] for _ in range(nv)]
        for a, b in g["edges"]:
            ins[int(b)].append(int(a))
        for v in range(nv):
            in_src.extend(ins[v])
            in_off.append(len(in_src))
        vertex_off.append(vertex_off[-1] + nv)
        seqs.append(g["seq"])
        seq_off.extend((g["seq_off"][1:] + np.uint64(seq_off[-1])).tolist())
        rv.append(g["refvar"])
        roff.extend((g["refvar_off"][1:] + np.uint32(roff[-1])).tolist())
        pm = np.zeros((0, nv), np.uint8) if paths is None else paths[c]
        num_paths.append(pm.shape[0])
        pv.append(pm.reshape(-1))
        poff.append(poff[-1] + pm.size)
        voff.append(voff[-1] + len(g["num_alleles"]))
    cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs).astype(dt)) if len(xs) else np.zeros(0, dt)   # noqa: E731
    f.update(vertex_off=np.asarray(vertex_off, np.uint32), num_paths=np.asarray(num_paths, np.uint32), seq_off=np.asarray(seq_off, np.uint64), seq=cat(seqs, np.uint8),
             vertex_variant=cat([g["var"] for g in graphs], np.uint16), vertex_allele=cat([g["allele"] for g in graphs], np.uint16),
             vertex_flags=cat([g["flags"] for g in graphs], np.uint8), vertex_nested=cat([g["nested"] for g in graphs], np.uint32), refvar_off=np.asarray(roff, np.uint32),
             refvar=cat(rv, np.uint16), path_off=np.asarray(poff, np.uint64), path_vertices=cat(pv, np.uint8), var_off=np.asarray(voff, np.uint32),
             var_num_alleles=cat([g["num_alleles"] for g in graphs], np.uint16), var_has_dependency=cat([g["dep"] for g in graphs], np.uint8),
             in_off=np.asarray(in_off, np.uint32), in_src=np.asarray(in_src, np.uint32))
    return f


def _gibbs_batch(cand, f, groups, S, ploidy, gender, cluster_ids, sources, out_edges):
    """the unit's bundles + group structure as the oracle sampler's input (the test's own assembly: VariantClusterGroup.hpp:60-89)"""
    R = int(cand["kmer_off"][-1])
    shared = np.full(R, -1, np.int32)
    num_shared = []
    for g in groups:
        keys = {}
        for c in g:
            r0 = int(cand["kmer_off"][c])
            for r in cand["multi_idx"][cand["multi_off"][c]:cand["multi_off"][c + 1]]:
                key = (int(cand["kmer_key"][2 * (r0 + r)]), int(cand["kmer_key"][2 * (r0 + r) + 1]))
                shared[r0 + r] = keys.setdefault(key, len(keys))
        num_shared.append(len(keys))
    G = len(groups)
    goff = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.uint32)
    out = {"S": S, "gender": np.asarray(gender, np.uint8), "num_groups": G, "num_clusters": f["num_clusters"], "group_index": np.arange(G, dtype=np.uint32),
           "group_cluster_off": goff, "group_ploidy": np.ascontiguousarray(np.asarray(ploidy, np.uint8).reshape(-1)),
           "group_source_off": np.concatenate([[0], np.cumsum([len(x) for x in sources])]).astype(np.uint32),
           "group_sources": np.concatenate([np.asarray(x, np.uint32) for x in sources]), "group_num_shared": np.asarray(num_shared, np.uint32),
           "cluster_idx": np.asarray(cluster_ids, np.uint32), "edge_off": np.concatenate([[0], np.cumsum([len(x) for x in out_edges])]).astype(np.uint32),
           "edges": np.concatenate([np.asarray(x, np.uint32) for x in out_edges] + [np.zeros(0, np.uint32)]), "num_haplotypes": f["num_paths"].astype(np.uint32),
           "num_variants": (f["var_off"][1:] - f["var_off"][:-1]).astype(np.uint32), "kmer_shared": shared, "var_num_alleles": f["var_num_alleles"],
           "var_has_dependency": f["var_has_dependency"]}
    for name in ("kmer_off", "hap_kmer_mult", "kmer_has_counts", "kmer_counts", "kmer_ic_mult", "kv_off", "kv_var", "kv_bits", "unique_off", "unique_idx", "multi_off", "multi_idx",
                 "hap_allele", "hapnest_off", "hapnest_idx", "nestdep_off", "nestdep_cluster", "nestdep_var_off", "nestdep_var"):
        out[name] = cand[name]
    return out


def _fmt(x):
    """default ostream formatting of a double (6 significant digits)"""
    return "%g" % x


def oracle_pipeline(oracle, ref, ds, seed, gibbs, noise_genotyping=False, min_unit_variants=10 ** 9, unit=0):
    """the cluster stage over all units (they share the path filter, the multigroup table and the parameter k-mers), then the genotype stage of unit `unit`"""
    import oracle_writer
    from bayestyper_amd.host import genotypes as G   # (ctypes signatures of the oracle's genotype functions only; `fn=` selects the oracle)

    out = {}
    genome = ds.get("contigs") or [["chr1", ds["genome"], False]]   # [name, sequence, is_decoy]
    seqs = {name: seq for name, seq, _ in genome}
    chrom_rank = {name: i for i, (name, _, _) in enumerate(genome)}

    def chrom_ploidy(name):   # ChromosomePloidy.cpp:40-180: the ploidy file's (female, male), else the defaults by contig name
        if ds.get("ploidy") and name in ds["ploidy"]:
            return ds["ploidy"][name]
        return {"x": (2, 1), "chrx": (2, 1), "y": (0, 1), "chry": (0, 1)}.get(name.lower(), (2, 2))

    vcf = open(os.path.join(ds["dir"], "candidates.vcf")).read()
    sample_rows = [line.rstrip("\n").split("\t") for line in open(os.path.join(ds["dir"], "samples.tsv"))]
    S = len(sample_rows)
    gender = [0 if r[1] in ("F", "Female") else 1 for r in sample_rows]
    # ---- cluster: clusters, groups, regions ----
    oracle.l.orc_cluster_stage.restype = C.c_ulonglong
    oracle.l.orc_cluster_stage.argtypes = [C.c_char_p, C.c_ulonglong, C.c_uint, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_ulonglong), C.c_char_p, C.c_uint, C.c_uint,
                                           C.c_float, C.c_uint, C.c_char_p, C.c_ulonglong]
    units, _, regions_sorted, _ = T.parse_dump(T.oracle_text(oracle, vcf, genome, K, min_unit_variants))
    out["num_units"] = len(units)
    out["regions_text"] = "".join(f"{c}\t{d}\t{s}\t{e}\n" for c, d, s, e in regions_sorted)
    genome_len = sum(len(seq) for _, seq, decoy in genome if not decoy)
    expected_path = int(np.ceil(genome_len * (1 + 0.05 * 2 * S)))
    pb = OrcBloom(oracle, expected_path, 1e-4, K, threaded=True)
    mg_table = OrcTable(oracle, 1, K)
    per_unit = []
    for groups_u in units:
        where, groups, sources, out_edges, cluster_ids = [], [], [], [], []
        for gi, g in enumerate(groups_u):
            groups.append(list(range(len(where), len(where) + len(g["vertices"]))))
            sources.append(g["sources"])
            for vi, v in enumerate(g["vertices"]):
                where.append((gi, vi))
                out_edges.append(v["edges"])
                cluster_ids.append(v["cluster_idx"])
        graphs = [_graph_arrays(oracle, seqs[groups_u[gi]["vertices"][vi]["chrom"]].encode(), groups_u[gi]["vertices"][vi]["vars"], groups_u[gi]["vertices"][vi]["red"], groups_u[gi]["vertices"][vi]["contained"]) for gi, vi in where]
        # ---- best paths per sample ----
        og = OrcGraphs(oracle, _flatten(graphs), K)
        for s in range(S):
            sb = OrcBloom.load(oracle, sample_rows[s][2], K)
            seeds = np.array([seed + (gi + 1) * (s + 1) + cluster_ids[c] for c, (gi, _) in enumerate(where)], np.uint32)
            paths = og.find_sample_paths(sb, seeds, 32)
            sb.close()
        og.close()
        f = _flatten(graphs, paths)
        og = OrcGraphs(oracle, f, K)
        # ---- multigroup k-mers, path k-mer count (filter and table are shared by the units) ----
        num_path_kmers = og.count_multigroup(np.array([gi for gi, _ in where], np.uint32), pb, mg_table)
        og.close()
        per_unit.append((groups_u, where, groups, sources, out_edges, cluster_ids, f, num_path_kmers))
    groups_o, where, groups, sources, out_edges, cluster_ids, f, num_path_kmers = per_unit[unit]
    NC = len(where)
    og = OrcGraphs(oracle, f, K)
    mg_keys = mg_table.export()[0]
    # ---- parameter k-mers ----
    num_region_kmers = sum(e - s + 1 for _, _, s, e in regions_sorted) - len(regions_sorted) * (K - 1)
    fraction = min(1.0, np.float32(3_000_000) / np.float32(num_region_kmers))
    ptab = OrcTable(oracle, 1, K)
    for i, (c, d, s, e) in enumerate(regions_sorted):
        ptab.count_parameter_kmers(pb, seqs[c][s:e + 1].encode(), d, seed + i, fraction)
    pk, _, pmeta = ptab.export()
    asc = oracle.unpack(pk, K)
    order = np.zeros(len(pk), np.uint32)
    if ref is not None:   # the reference's own container decides the order
        ref.l.ref_hybrid_hash_order.restype = C.c_uint64
        ref.l.ref_hybrid_hash_order.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint, C.c_int, C.c_void_p]
        assert ref.l.ref_hybrid_hash_order(asc.ctypes.data, len(pk), 4 ** 12, seed, 1, order.ctypes.data) == len(pk)
    else:
        from bayestyper_amd.host import dll

        dll.bth_hybrid_hash_order.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint64, C.c_void_p]
        dll.bth_hybrid_hash_order(np.ascontiguousarray(pk).ctypes.data, len(pk), K, seed, 4 ** 12, order.ctypes.data)
    flags = pmeta[:, 0]
    chosen = [i for i in order if (flags[i] & KC_PARAMETER) and not (flags[i] & KC_DECOY)][:1_000_000]
    param_ascii = asc.reshape(-1, K)[chosen]
    out["parameter_kmers"] = [bytes(r).decode() for r in param_ascii]
    pb.close(), ptab.close(), mg_table.close()
    # ---- genotype: tables ----
    pb = OrcBloom(oracle, num_path_kmers + 1_000_000, 1e-4, K, threaded=True)
    table = OrcTable(oracle, S, K)
    flat_params = np.ascontiguousarray(param_ascii).reshape(-1)
    pb.insert(flat_params)
    table.insert(flat_params, mark_parameter=True)
    og.count_kmers(pb)
    for c, d, s, e in regions_sorted:
        table.count_intercluster(pb, seqs[c][s:e + 1].encode(), d, *chrom_ploidy(c))
    for s in range(S):
        db = OrcKmc(oracle, sample_rows[s][2])
        table.parse_sample_kmers(pb, db, s)
        db.close()
    mgb = OrcBloom(oracle, max(len(mg_keys), 1), 1e-4, K)
    if len(mg_keys):
        mgb.insert(oracle.unpack(mg_keys, K))
    og.classify(table, mgb)
    cand = og.candidates(table)
    # ---- NB fit from the parameter k-mers (CountDistribution.cpp:66-141) ----
    _, stats = table.kmer_stats(gender)
    means, vars_ = [], []
    _oracle._gibbs_sigs(oracle.l)
    ps, sz = np.zeros(S), np.zeros(S)
    for s in range(S):
        m = 1 + int(np.argmax(stats[s, 1:33, 0]))
        cnt, mean, m2 = stats[s, m, 0], stats[s, m, 2], stats[s, m, 3]
        p_, size_ = C.c_double(), C.c_double()
        oracle.l.orc_nb_moments(mean, m2 / (cnt - 1), C.byref(p_), C.byref(size_))
        p2 = min(p_.value, 0.99)   # (the fit caps p; not reached here)
        ps[s], sz[s] = p2, size_.value / m
        means.append(sz[s] * (1 - p2) / p2)
        vars_.append(sz[s] * (1 - p2) / (p2 * p2))
    out["genomic"] = list(zip(means, vars_))
    lut_g, lut_n = np.zeros(S * 65536), np.zeros(S * 256)
    oracle.l.orc_build_luts(S, _oracle._ptr(ps), _oracle._ptr(sz), _oracle._ptr(np.full(S, 0.05)), _oracle._ptr(lut_g), _oracle._ptr(lut_n))   # (the noise table is replaced by the driver)
    # ---- Gibbs: estimateNoise, then estimateGenotypes ----
    ploidy = np.array([[chrom_ploidy(g["vertices"][0]["chrom"])[gender[s_]] for s_ in range(S)] for g in groups_o], np.uint8).reshape(len(groups), S)
    flat = _gibbs_batch(cand, f, groups, S, ploidy, gender, cluster_ids, sources, out_edges)
    kw = dict(seed=seed, chains=gibbs["chains"], burn=gibbs["burn"], iters=gibbs["samples"])
    if noise_genotyping:   # --noise-genotyping: estimateNoiseAndGenotypes (InferenceEngine.cpp:384-472), noise rates and genotypes in one loop
        ogb = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, noise_seeding=1, **kw)
        out["noise_rows"] = ogb.estimate_noise_and_genotypes()
        ro = ogb.results()
        ogb.close()
    else:
        ogb = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, noise_seeding=1, **kw)
        trace, _, final = ogb.estimate_noise()
        ogb.close()
        out["noise_rows"] = trace
        lut_n2 = np.zeros(S * 256)
        oracle.l.orc_build_luts(S, _oracle._ptr(ps), _oracle._ptr(sz), _oracle._ptr(np.ascontiguousarray(final, np.float64)), _oracle._ptr(np.zeros(S * 65536)), _oracle._ptr(lut_n2))
        ogb = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n2, **kw)
        ogb.run(8)
        ro = ogb.results()
        ogb.close()
    # ---- genotypes -> VCF lines ----
    mf = np.array([1 - np.exp(-0.275 * m) for m in means], np.float32)
    fn = oracle.l.orc_cluster_output_columns
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_uint, C.c_uint, C.c_uint] + [C.c_void_p] * 3 + [C.c_ulonglong] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p, C.c_char_p, C.c_ulonglong]
    lines = []
    for c, (gi, vi) in enumerate(where):
        g, v = groups_o[gi], groups_o[gi]["vertices"][vi]
        cols = G.cluster_output_columns(flat, ro, c, ploidy[gi], mf, fn=fn)
        infos = v["vars"]
        vcr = "%s:%d-%d" % (v["chrom"], infos[0][0] + 1, max(pos + max(rl for rl, _ in alts) for pos, _, _, alts in infos))
        for (pos, vid, dep, alts), col, aco in zip(infos, cols, v["aco"]):
            full = [(rl, seq, a) for (rl, seq), a in zip(alts, aco)]
            lines.append(((chrom_rank[v["chrom"]], pos), oracle_writer.vcf_line(v["chrom"], seqs[v["chrom"]], pos, vid, bool(dep), full, col, len(infos), vcr, len(g["vertices"]), g["region"],
                                                       int(flat["num_haplotypes"][c]))))
    out["vcf_body"] = "".join(line for _, line in sorted(lines))
    out["num_groups"], out["num_clusters"] = len(groups), NC
    out["unit_variants"] = sum(g["nvar"] for g in groups_o)
    for x in (og, pb, table, mgb):
        x.close()
    return out


@pytest.mark.parametrize("genome_len,num_snvs,num_samples,gibbs,noise_genotyping,min_unit,unit",
                         [(60_000, 300, 1, dict(chains=20, burn=100, samples=250), False, None, 0), (1_000_000, 5000, 1, dict(chains=3, burn=15, samples=40), False, None, 0),
                          (80_000, 400, 3, dict(chains=4, burn=20, samples=60), False, None, 0), (50_000, 250, 2, dict(chains=3, burn=12, samples=30), True, None, 0),
                          (90_000, 450, 1, dict(chains=3, burn=12, samples=30), False, 120, 2)],
                         ids=["small-default-schedule", "C1-1Mb-5000-SNVs", "trio-female-male-female", "noise-genotyping-two-samples", "several-units-genotype-the-third"])
def test_cluster_then_genotype_equal_the_oracle_pipeline(oracle, tmp_path, genome_len, num_snvs, num_samples, gibbs, noise_genotyping, min_unit, unit):
    ref = _oracle.load_ref()
    ds = c1_dataset.make(str(tmp_path / "data"), oracle, genome_len, num_snvs, num_samples, num_error_kmers=200_000, genders=["F", "M", "F"][:num_samples])
    seed = 42
    prefix = str(tmp_path / "bt")
    r = subprocess.run([EXE, "cluster", "-v", os.path.join(ds["dir"], "candidates.vcf"), "-s", os.path.join(ds["dir"], "samples.tsv"), "-g", os.path.join(ds["dir"], "genome.fa"), "-o", prefix,
                        "-r", str(seed)] + (["--min-number-of-unit-variants", str(min_unit)] if min_unit else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    assert "BayesTyper cluster completed succesfully!" in r.stdout
    r = subprocess.run([EXE, "genotype", "-v", prefix + f"_unit_{unit + 1}/variant_clusters.bin", "-c", prefix + "_cluster_data", "-s", os.path.join(ds["dir"], "samples.tsv"), "-g",
                        os.path.join(ds["dir"], "genome.fa"), "-o", prefix, "-r", str(seed), "--number-of-gibbs-chains", str(gibbs["chains"]), "--gibbs-burn-in", str(gibbs["burn"]),
                        "--gibbs-samples", str(gibbs["samples"])] + (["--noise-genotyping", "-z"] if noise_genotyping else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    assert "BayesTyper genotype completed succesfully!" in r.stdout

    # main.cpp:218,242: the option gives the number of units, floor(variants / option), and every unit then takes ceil(variants / units) variants
    per_unit = int(np.ceil(num_snvs / max(1, num_snvs // min_unit))) if min_unit else 10 ** 9
    want = oracle_pipeline(oracle, ref, ds, seed, gibbs, noise_genotyping, per_unit, unit)
    assert f"- {want['unit_variants']} were genotyped" in r.stdout, r.stdout[-1500:]
    if min_unit:
        assert want["num_units"] >= 3 and unit < want["num_units"] and want["unit_variants"] < num_snvs
        assert os.path.isdir(prefix + f"_unit_{want['num_units']}") and not os.path.isdir(prefix + f"_unit_{want['num_units'] + 1}")
    else:
        assert want["unit_variants"] == num_snvs
    # cluster stage files
    assert gzip.open(prefix + "_cluster_data/intercluster_regions.txt.gz", "rt").read() == want["regions_text"]
    got_params = gzip.open(prefix + "_cluster_data/parameter_kmers.fa.gz", "rt").read().split("\n")
    assert got_params[0] == ">k55" and got_params[-1] == "" and got_params[1:-1] == want["parameter_kmers"] and len(want["parameter_kmers"]) > 10_000
    assert open(prefix + "_cluster_data/multigroup_kmers.bloomMeta").read().split("\t")[2].strip() == "55"
    # genotype stage files
    rows = open(prefix + "_genomic_parameters.txt").read().split("\n")
    assert rows[0] == "Sample\tMean\tVariance"
    names = [f"sample{i + 1}" for i in range(num_samples)]
    for i in range(num_samples):
        name, m, v = rows[1 + i].split("\t")
        assert name == names[i] and abs(float(m) - want["genomic"][i][0]) < 1e-3 and abs(float(v) - want["genomic"][i][1]) < 1e-2 and abs(float(m) - 15) < 0.5
    noise = open(prefix + "_noise_parameters.txt").read().split("\n")
    assert noise[0] == "Chain\tIteration\t" + "\t".join(names)
    want_rows = ["%d\t%d\t%s" % (int(r_[0]), int(r_[1]), "\t".join(_fmt(x) for x in r_[2:])) for r_ in want["noise_rows"]]
    assert noise[1:-1] == want_rows
    vcf = gzip.open(prefix + ".vcf.gz", "rt").read() if noise_genotyping else open(prefix + ".vcf").read()   # (-z / --gzip-output in the noise-genotyping case)
    header = [x for x in vcf.split("\n") if x.startswith("#")]
    assert header[0] == "##fileformat=VCFv4.2" and sum(x.startswith("##BayesTyperOptions=command:") for x in header) == 2 and header[-1].endswith("FORMAT\t" + "\t".join(names))
    assert any('command:"cluster"' in x and 'random-seed:"42"' in x and 'max-number-of-sample-haplotypes:"32"' in x for x in header)
    body = "".join(x + "\n" for x in vcf.split("\n") if x and not x.startswith("#"))
    assert body == want["vcf_body"]
    # and the calls are right: the sample's true genotypes are recovered
    for i in range(num_samples):
        calls = {int(x.split("\t")[1]) - 1: x.split("\t")[9 + i].split(":")[0] for x in body.strip().split("\n")}
        truth = {int(p): {0: "0/0", 1: "0/1", 2: "1/1"}[int(g)] for p, g in zip(ds["pos"], ds["truth"][i]) if int(p) in calls}
        assert len(truth) == want["unit_variants"]
        agree = sum(calls[p] == t for p, t in truth.items())
        assert agree >= 0.97 * len(truth), (i, agree)


@pytest.mark.parametrize("num_samples,gibbs,noise_genotyping", [(2, dict(chains=3, burn=10, samples=25), False), (10, dict(chains=3, burn=20, samples=50), True)],
                         ids=["two-samples-default-mode", "ten-samples-noise-genotyping"])
def test_sv_rich_candidates_through_the_executable(oracle, tmp_path, num_samples, gibbs, noise_genotyping):
    """The executable on candidates of every flavour the parser distinguishes — SNVs, indels, multi-allelic records, MNVs and blocks of structural
    variants with variants nested inside their alleles (nested variant-cluster groups, '*' alleles, ACO attributes) — and samples (female,
    male alternating) whose haplotypes carry random subsets of the candidate alleles: every output file against the oracle pipeline.  The second case
    is BASELINE configs[3]'s sample count through the shipped C++ estimateNoiseAndGenotypes (--noise-genotyping, InferenceEngine.cpp:384-472): ten
    samples, nested SV groups, 3 x (20 + 50) iterations, every noise rate of every iteration and the VCF body."""
    from test_pipeline_gpu import sample_haplotype

    ref = _oracle.load_ref()
    rng = np.random.default_rng(77)
    seq = "".join(rng.choice(list("ACGT"), 120_000))
    genome = [["chr1", seq, False]]
    vcf = T.make_vcf(rng, genome, K, 70, False, extra_contig=False, sv_blocks=3)
    records = []
    for line in vcf.split("\n"):
        if line and line[0] != "#":
            _, p, _, r_, alt = line.split("\t")[:5]
            records.append((int(p) - 1, r_, [a for a in alt.split(",") if a != "*"]))
    d = tmp_path / "data"
    os.makedirs(d)
    with open(d / "genome.fa", "w") as fh:
        fh.write(">chr1\n" + "\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + "\n")
    open(d / "candidates.vcf", "w").write(vcf)
    with open(d / "samples.tsv", "w") as sf:
        for s, gender in enumerate((["F", "M"] * 5)[:num_samples]):
            text = "N".join(sample_haplotype(rng, seq, records) for _ in range(2))
            km, va = oracle.kmers_from_sequence(text.encode(), K)
            present = np.unique(km[va == 1], axis=0)
            cnt = (rng.poisson(14, len(present)) + 1).astype(np.uint32)
            asc = oracle.unpack(present, K).reshape(-1, K)
            order = np.lexsort(asc.T[::-1])                     # KMC order = ascending ASCII order
            prefix = str(d / f"sample{s + 1}")
            if s % 2 == 0:
                oracle.kmc_write(prefix, np.ascontiguousarray(asc[order]).reshape(-1), cnt[order], K, 7, 1)
            else:   # every second sample's database in the KMC2 ("0x200") layout: five signature bins
                oracle.kmc2_write(prefix, np.ascontiguousarray(asc[order]).reshape(-1), cnt[order], K, 7, 1, 5)
            bloom = OrcBloom(oracle, len(present), 1e-3, K)
            bloom.insert(np.ascontiguousarray(asc).reshape(-1))
            bloom.save(prefix)
            bloom.close()
            sf.write(f"sample{s + 1}\t{gender}\t{prefix}\n")
    ds = {"genome": seq, "dir": str(d)}
    seed = 11
    prefix = str(tmp_path / "bt")
    r = subprocess.run([EXE, "cluster", "-v", str(d / "candidates.vcf"), "-s", str(d / "samples.tsv"), "-g", str(d / "genome.fa"), "-o", prefix, "-r", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    r = subprocess.run([EXE, "genotype", "-v", prefix + "_unit_1/variant_clusters.bin", "-c", prefix + "_cluster_data", "-s", str(d / "samples.tsv"), "-g", str(d / "genome.fa"), "-o", prefix,
                        "-r", str(seed), "--number-of-gibbs-chains", str(gibbs["chains"]), "--gibbs-burn-in", str(gibbs["burn"]), "--gibbs-samples", str(gibbs["samples"])]
                       + (["--noise-genotyping"] if noise_genotyping else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    want = oracle_pipeline(oracle, ref, ds, seed, gibbs, noise_genotyping)
    assert want["num_clusters"] > want["num_groups"] > 20          # nested variant-cluster groups are present
    assert gzip.open(prefix + "_cluster_data/intercluster_regions.txt.gz", "rt").read() == want["regions_text"]
    got_params = gzip.open(prefix + "_cluster_data/parameter_kmers.fa.gz", "rt").read().split("\n")
    assert got_params[1:-1] == want["parameter_kmers"]
    noise = open(prefix + "_noise_parameters.txt").read().split("\n")
    assert noise[1:-1] == ["%d\t%d\t%s" % (int(r_[0]), int(r_[1]), "\t".join(_fmt(x) for x in r_[2:])) for r_ in want["noise_rows"]]
    body = "".join(x + "\n" for x in open(prefix + ".vcf").read().split("\n") if x and not x.startswith("#"))
    assert body == want["vcf_body"] and body.count("\n") > 100


@pytest.mark.parametrize("ploidy_file", [None, {"chr1": (2, 2), "chrX": (1, 2), "chrY": (1, 1)}], ids=["default-ploidies", "chromosome-ploidy-file"])
def test_sex_chromosomes_and_decoys_through_the_executable(oracle, tmp_path, ploidy_file):
    """Several contigs with different ploidies and a decoy file: chr1 (diploid), chrX (female 2 / male 1), chrY (female 0 / male 1; the
    reference's default ploidies by contig name, ChromosomePloidy.cpp:40-180) and a decoy contig (-d) that repeats a stretch of chr1 with
    candidates on it (their path k-mers become decoy k-mers and are excluded).  A female and a male sample; every output file against the oracle
    pipeline."""
    from test_pipeline_gpu import sample_haplotype

    ref = _oracle.load_ref()
    rng = np.random.default_rng(99)
    contigs = [["chr1", "".join(rng.choice(list("ACGT"), 50_000)), False], ["chrX", "".join(rng.choice(list("ACGT"), 30_000)), False],
               ["chrY", "".join(rng.choice(list("ACGT"), 20_000)), False]]
    decoy = "".join(rng.choice(list("ACGT"), 3_000)) + contigs[0][1][20_000:24_000] + "".join(rng.choice(list("ACGT"), 3_000))
    vcf = T.make_vcf(rng, contigs, K, 60, False, extra_contig=False, sv_blocks=1)
    records = {}
    for line in vcf.split("\n"):
        if line and line[0] != "#":
            c, p, _, r_, alt = line.split("\t")[:5]
            records.setdefault(c, []).append((int(p) - 1, r_, [a for a in alt.split(",") if a != "*"]))
    assert any(20_000 <= p < 24_000 for p, _, _ in records["chr1"])
    d = tmp_path / "data"
    os.makedirs(d)
    with open(d / "genome.fa", "w") as fh:
        for name, seq, _ in contigs:
            fh.write(f">{name}\n" + "\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + "\n")
    with open(d / "decoy.fa", "w") as fh:
        fh.write(">decoy1\n" + "\n".join(decoy[i:i + 60] for i in range(0, len(decoy), 60)) + "\n")
    open(d / "candidates.vcf", "w").write(vcf)
    copies = {"F": {"chr1": 2, "chrX": 2, "chrY": 0}, "M": {"chr1": 2, "chrX": 1, "chrY": 1}}
    with open(d / "samples.tsv", "w") as sf:
        for s, gender in enumerate(["F", "M"]):
            haps = [sample_haplotype(rng, seq, records.get(name, [])) for name, seq, _ in contigs for _ in range(copies[gender][name])] + [decoy, decoy]
            km, va = oracle.kmers_from_sequence("N".join(haps).encode(), K)
            present = np.unique(km[va == 1], axis=0)
            cnt = (rng.poisson(14, len(present)) + 1).astype(np.uint32)
            asc = oracle.unpack(present, K).reshape(-1, K)
            order = np.lexsort(asc.T[::-1])
            prefix = str(d / f"sample{s + 1}")
            oracle.kmc_write(prefix, np.ascontiguousarray(asc[order]).reshape(-1), cnt[order], K, 7, 1)
            bloom = OrcBloom(oracle, len(present), 1e-3, K)
            bloom.insert(np.ascontiguousarray(asc).reshape(-1))
            bloom.save(prefix)
            bloom.close()
            sf.write(f"sample{s + 1}\t{gender}\t{prefix}\n")
    ds = {"contigs": contigs + [["decoy1", decoy, True]], "dir": str(d), "ploidy": ploidy_file}
    seed, gibbs = 5, dict(chains=3, burn=10, samples=25)
    extra = []
    if ploidy_file:   # -y: <chromosome> <female ploidy> <male ploidy> (deliberately not the defaults)
        open(d / "ploidy.txt", "w").write("".join(f"{c}\t{f}\t{m}\n" for c, (f, m) in ploidy_file.items()))
        extra = ["-y", str(d / "ploidy.txt")]
    prefix = str(tmp_path / "bt")
    common = ["-s", str(d / "samples.tsv"), "-g", str(d / "genome.fa"), "-d", str(d / "decoy.fa"), "-o", prefix, "-r", str(seed)]
    r = subprocess.run([EXE, "cluster", "-v", str(d / "candidates.vcf")] + common, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    r = subprocess.run([EXE, "genotype", "-v", prefix + "_unit_1/variant_clusters.bin", "-c", prefix + "_cluster_data"] + common + extra +
                       ["--number-of-gibbs-chains", str(gibbs["chains"]), "--gibbs-burn-in", str(gibbs["burn"]), "--gibbs-samples", str(gibbs["samples"])], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    want = oracle_pipeline(oracle, ref, ds, seed, gibbs)
    assert gzip.open(prefix + "_cluster_data/intercluster_regions.txt.gz", "rt").read() == want["regions_text"] and "decoy1\t1\t" in want["regions_text"]
    got_params = gzip.open(prefix + "_cluster_data/parameter_kmers.fa.gz", "rt").read().split("\n")
    assert got_params[1:-1] == want["parameter_kmers"]
    noise = open(prefix + "_noise_parameters.txt").read().split("\n")
    assert noise[1:-1] == ["%d\t%d\t%s" % (int(r_[0]), int(r_[1]), "\t".join(_fmt(x) for x in r_[2:])) for r_ in want["noise_rows"]]
    body = "".join(x + "\n" for x in open(prefix + ".vcf").read().split("\n") if x and not x.startswith("#"))
    assert body == want["vcf_body"]
    rows = [x.split("\t") for x in body.strip().split("\n")]
    assert {x[0] for x in rows} == {"chr1", "chrX", "chrY"}
    if ploidy_file is None:
        # ploidy shows in the calls: the female's chrY columns are the empty-sample string (GenotypeWriter.cpp:261-330), the male's chrX / chrY genotypes are haploid
        assert all(x[9].split(":")[0] == "" for x in rows if x[0] == "chrY") and all("/" not in x[10].split(":")[0] for x in rows if x[0] in ("chrX", "chrY"))
        assert all("/" in x[9].split(":")[0] for x in rows if x[0] in ("chr1", "chrX"))
    else:
        assert all("/" not in x[9].split(":")[0] and "/" in x[10].split(":")[0] for x in rows if x[0] == "chrX")


def _genotype(prefix, unit_prefix, ds_dir, seed, gibbs, extra_args=(), env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([EXE, "genotype", "-v", unit_prefix + "_unit_1/variant_clusters.bin", "-c", unit_prefix + "_cluster_data", "-s", os.path.join(ds_dir, "samples.tsv"), "-g",
                        os.path.join(ds_dir, "genome.fa"), "-o", prefix, "-r", str(seed), "--number-of-gibbs-chains", str(gibbs["chains"]), "--gibbs-burn-in", str(gibbs["burn"]),
                        "--gibbs-samples", str(gibbs["samples"])] + list(extra_args), capture_output=True, text=True, env=e, timeout=600)
    logs = "".join(f"\n--- {f}\n" + open(os.path.join(os.path.dirname(prefix), f)).read()[-1500:] for f in sorted(os.listdir(os.path.dirname(prefix))) if f.endswith(".log"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:] + logs
    assert "BayesTyper genotype completed succesfully!" in r.stdout
    return r.stdout


def _outputs(prefix):
    vcf = [x for x in open(prefix + ".vcf").read().split("\n") if not x.startswith("##BayesTyperOptions=")]   # (the options line names the output prefix)
    return vcf, open(prefix + "_noise_parameters.txt").read(), open(prefix + "_genomic_parameters.txt").read()


def _sharded_equals_single(oracle, tmp_path, ranks_env, noise_genotyping):
    """`bayesTyper genotype` on several ranks (BT_GPUS: the process forks its ranks; host/Comm.hpp) against the same command on one rank: the KMC
    scan split by record range + count merge, the groups dealt to the ranks, the per-iteration histogram reduction of the noise drivers and the
    gather of the collected samples must leave every output file as the one-rank run writes it (which the tests above compare with the oracle)."""
    ds = c1_dataset.make(str(tmp_path / "data"), oracle, 70_000, 350, 3, num_error_kmers=150_000, genders=["F", "M", "F"])
    seed, gibbs = 7, dict(chains=3, burn=12, samples=30)
    unit_prefix = str(tmp_path / "bt")
    r = subprocess.run([EXE, "cluster", "-v", os.path.join(ds["dir"], "candidates.vcf"), "-s", os.path.join(ds["dir"], "samples.tsv"), "-g", os.path.join(ds["dir"], "genome.fa"), "-o", unit_prefix,
                        "-r", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    extra = ["--noise-genotyping"] if noise_genotyping else []
    one = str(tmp_path / "one")
    _genotype(one, unit_prefix, ds["dir"], seed, gibbs, extra)
    many = str(tmp_path / "many")
    out = _genotype(many, unit_prefix, ds["dir"], seed, gibbs, extra, env=ranks_env)
    assert "Rank 0 of " in out and "Merged the sample counts of" in out
    # the collected samples go into the gather as result strings packed ON THE DEVICE (bt_gibbs_result_words), one per launch, on every rank
    import re

    lines = [out] + [open(many + f".rank{r_}.log").read() for r_ in range(1, int(ranks_env["BT_GPUS"]))]
    for r_, text in enumerate(lines):
        m = re.search(r"\[%d\] gather: from the device, (\d+) launch\(es\), (\d+) bytes" % r_, text)
        assert m and int(m.group(2)) > 0, text[-800:]
        if "BT_MAX_GROUPS_PER_LAUNCH" in ranks_env:
            assert int(m.group(1)) > 1
    a, b = _outputs(one), _outputs(many)
    assert a[0] == b[0] and len(a[0]) > 300
    assert a[1] == b[1] and a[2] == b[2]
    n = int(ranks_env["BT_GPUS"])
    for r_ in range(1, n):   # the other ranks leave a log, no parameter files
        assert os.path.exists(many + f".rank{r_}.log") and not os.path.exists(many + f"_noise_parameters.rank{r_}.txt")
    assert not [f for f in os.listdir(tmp_path) if ".comm_id" in f]


@pytest.mark.parametrize("noise_genotyping", [False, True], ids=["default-mode", "noise-genotyping"])
def test_three_ranks_sharing_one_gpu(oracle, tmp_path, noise_genotyping):
    """three ranks on GPU 0 exchanging through files (BT_COMM_TRANSPORT=files: RCCL forms no communicator over ranks that share a GPU) — the
    sharded run's logic on a one-GPU box"""
    env = {"BT_GPUS": "3", "BT_COMM_TRANSPORT": "files", "BT_DEVICE": "0"}
    if not noise_genotyping:
        env["BT_MAX_GROUPS_PER_LAUNCH"] = "23"   # (several launches per rank: their strings are concatenated on the device)
    _sharded_equals_single(oracle, tmp_path, env, noise_genotyping)


def test_two_ranks_over_rccl_when_two_gpus_are_visible(oracle, tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _sharded_equals_single(oracle, tmp_path, {"BT_GPUS": "2", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, True)


def test_make_bloom_command_line_feeds_cluster(oracle, tmp_path):
    """`bayesTyperTools makeBloom -k <kmc prefix>` (src/bayesTyperTools/main.cpp:101-146, MakeBloom.cpp:39-51,200-295): the .bloomMeta / .bloomData it
    writes are byte-identical to the reference KmerBloom's (the dataset's filters come from the oracle), for a KMC1 and a KMC2 database, and
    `bayesTyper cluster` on samples that point to them writes what it writes on the dataset's own filters"""
    import shutil

    tools = os.path.join(os.path.dirname(EXE), "bayesTyperTools")
    ds = c1_dataset.make(str(tmp_path / "data"), oracle, 40_000, 200, 2, num_error_kmers=80_000, genders=["F", "M"])
    r = subprocess.run([tools, "makeBloom"], capture_output=True, text=True)
    assert r.returncode == 1 and "--false-positive-rate arg (=0.001)" in r.stdout          # help screen, returns 1 (main.cpp:137-141)
    with open(os.path.join(ds["dir"], "samples_mb.tsv"), "w") as sf:
        for s, gender in ((1, "F"), (2, "M")):
            src, dst = os.path.join(ds["dir"], f"sample{s}"), os.path.join(ds["dir"], f"mb{s}")
            for ext in (".kmc_pre", ".kmc_suf"):
                shutil.copy(src + ext, dst + ext)
            r = subprocess.run([tools, "makeBloom", "-k", dst] + (["--false-positive-rate", "0.001", "-p", "4"] if s == 2 else []), capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr
            assert "Making bloom filter of" in r.stdout and "Completed saving bloom filter" in r.stdout
            for ext in (".bloomMeta", ".bloomData"):
                assert open(dst + ext, "rb").read() == open(src + ext, "rb").read(), ext
            sf.write(f"sample{s}\t{gender}\t{dst}\n")
    outs = []
    for tag, samples in (("a", "samples.tsv"), ("b", "samples_mb.tsv")):
        prefix = str(tmp_path / tag)
        r = subprocess.run([EXE, "cluster", "-v", os.path.join(ds["dir"], "candidates.vcf"), "-s", os.path.join(ds["dir"], samples), "-g", os.path.join(ds["dir"], "genome.fa"), "-o", prefix, "-r", "5"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr
        outs.append((gzip.open(prefix + "_cluster_data/parameter_kmers.fa.gz", "rt").read(), gzip.open(prefix + "_cluster_data/intercluster_regions.txt.gz", "rt").read(),
                     open(prefix + "_cluster_data/multigroup_kmers.bloomData", "rb").read()))
    assert outs[0] == outs[1] and len(outs[0][0]) > 1000
    r = subprocess.run([tools, "makeBloom", "-k", os.path.join(ds["dir"], "nothing_here")], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR" in r.stderr
