"""getGenotypes (GPP / APP / GQ / allele filters / calls / AC, ACP): the C++ host layer against the oracle's restatement on sampler
results produced by the oracle (CPU only), plus the invariants a VCF reader relies on."""
import numpy as np

import _oracle
from bayestyper_amd import synth
from bayestyper_amd.host import genotypes


def _batch(S):
    rng = np.random.default_rng(4)
    groups = [synth.group_shape_A(rng, i) for i in range(10)] + [synth.group_shape_B(rng, 100 + i) for i in range(4)] + [synth.group_shape_C(rng, 200 + 3 * i, root_H=8, root_kpa=60) for i in range(2)]
    ploidy = np.full((len(groups), S), 2, np.uint8)
    ploidy[::4, 1] = 1
    ploidy[3, 2] = 0
    return synth.flatten(groups, S, rng, ploidy=ploidy, gender=[0, 1, 1][:S]), ploidy


def test_host_genotypes_match_oracle_and_are_consistent(oracle):
    S = 3
    flat, ploidy = _batch(S)
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    og = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, seed=11, chains=3, burn=10, iters=40)
    og.run(4)
    res = og.results()
    og.close()
    mf = genotypes.min_fraction_observed_kmers([15.0] * S)
    goff = flat["group_cluster_off"]
    called = 0
    for g in range(flat["num_groups"]):
        for c in range(goff[g], goff[g + 1]):
            # nested clusters take their ploidy from the parent's diplotype per sweep; the summaries use the group's chromosome ploidy
            a = genotypes.cluster_genotypes(flat, res, c, ploidy[g], mf)
            b = genotypes.cluster_genotypes(flat, res, c, ploidy[g], mf, fn=oracle.l.orc_cluster_genotypes)
            for k in a:
                assert np.array_equal(a[k], b[k]), (c, k)
            V = a["gpp"].shape[0]
            for v in range(V):
                A = int(a["num_alleles"][v])
                for s in range(S):
                    if ploidy[g, s] == 0:
                        assert a["gpp"][v, s].sum() == 0 and (a["estimate"][v, s] == 0xFFFF).all()
                        continue
                    G = A * (A + 1) // 2 if ploidy[g, s] == 2 else A
                    assert abs(a["gpp"][v, s, :G].sum() - 1) < 1e-5 and a["gpp"][v, s, G:].sum() == 0
                    assert 0 <= a["gq"][v, s] <= 99 and (a["filters"][v, s] <= 3).all()
                    est = a["estimate"][v, s]
                    if est[0] != 0xFFFF:
                        called += 1
                        gi = int(est[1]) * (int(est[1]) + 1) // 2 + int(est[0]) if ploidy[g, s] == 2 else int(est[0])
                        assert a["gpp"][v, s, gi] >= 0.99 - 1e-6 and a["gpp"][v, s, gi] == a["gpp"][v, s].max()
                assert a["total_count"][v] == sum(int(x != 0xFFFF) for s in range(S) for x in a["estimate"][v, s][: int(ploidy[g, s])])
    assert called > 10


def test_output_columns_match_oracle(oracle):
    """QUAL / FILTER / AC,AF,AN,ACP,ANC and the per-sample GT:GQ:GPP:APP:NAK:FAK:MAC:SAF columns, character for character"""
    S = 3
    flat, ploidy = _batch(S)
    lut_g, lut_n = _oracle.build_luts(oracle, S)
    og = _oracle.OrcGibbs(oracle, flat, lut_g, lut_n, seed=5, chains=2, burn=10, iters=25)
    og.run(4)
    res = og.results()
    og.close()
    mf = genotypes.min_fraction_observed_kmers([15.0] * S)
    goff = flat["group_cluster_off"]
    seen_pass = seen_an0 = seen_null = False
    for g in range(flat["num_groups"]):
        for c in range(goff[g], goff[g + 1]):
            a = genotypes.cluster_output_columns(flat, res, c, ploidy[g], mf)
            b = genotypes.cluster_output_columns(flat, res, c, ploidy[g], mf, fn=oracle.l.orc_cluster_output_columns)
            assert a == b and len(a) == int(flat["num_variants"][c])
            for line in a:
                cols = line.split("\t")
                assert len(cols) == 3 + S and cols[1] in ("PASS", "AN0") and cols[2].startswith("AC=")
                seen_pass |= cols[1] == "PASS"
                seen_an0 |= cols[1] == "AN0"
                seen_null |= any(x == ":.:.:.:.:.:." for x in cols[3:])
                for x in cols[3:]:
                    assert x == ":.:.:.:.:.:." or x.count(":") == 7
    assert seen_pass and seen_null
