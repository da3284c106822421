"""Tiny end-to-end pass used by __graft_entry__.smoke(): KMC scan -> count table, then the Gibbs schedule for a few groups,
GPU (C ABI) against the oracle on the same inputs and seed."""
import numpy as np

import _oracle


def run(ctx, orc):
    from bayestyper_amd import lib, shard, synth

    # ---- Gibbs: one nested group, a few single-cluster groups; diplotype sampling frequencies must agree
    S = 2
    flat = synth.concat([synth.make_batch("A", 12, S, seed=11, templates=3), synth.make_batch("B", 3, S, seed=12), synth.make_batch("C", 1, S, seed=13)])
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    lut_g, lut_n = _oracle.build_luts(orc, S)
    kw = dict(seed=7, chains=2, burn=5, iters=20)
    og = _oracle.OrcGibbs(orc, flat, lut_g, lut_n, **kw)
    og.run(2)
    ro = og.results()
    og.close()
    gg = lib.Gibbs(ctx, flat, lut_g, lut_n, **kw)
    gg.run()
    rg, summ = gg.results(), gg.posterior_summary()
    gg.close()
    assert np.array_equal(ro["dip_off"], rg["dip_off"]) and np.array_equal(ro["h1"], rg["h1"]) and np.array_equal(ro["h2"], rg["h2"]), "sampled diplotype sets differ"
    assert np.array_equal(ro["freq"], rg["freq"]), "diplotype sampling frequencies differ"
    assert np.array_equal(ro["stats"][:, :, 0], rg["stats"][:, :, 0]) and np.allclose(ro["stats"], rg["stats"], rtol=1e-9, atol=1e-12), "allele k-mer statistics differ"
    assert np.array_equal(summ, shard.summary_from_results(ro, flat["num_clusters"], S)), "posterior summary differs"
