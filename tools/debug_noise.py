"""Debug aid: the --noise-genotyping loop driven from Python over lib.Gibbs (BTGPU_LIB selects the build) against the oracle, trace by trace.
usage: python tools/debug_noise.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle
from bayestyper_amd import lib, synth
from bayestyper_amd.host import count_model

orc = _oracle.load_oracle()
S = 10
flat = synth.concat([synth.make_hetero_batch("D", 1, S, seed=61), synth.make_hetero_batch("C", 5, S, seed=62), synth.make_hetero_batch("B", 10, S, seed=63), synth.make_hetero_batch("A", 40, S, seed=64)])
flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
kw = dict(seed=2468, chains=3, burn=20, iters=50)
def cd():
    d = count_model.CountDistribution(S, prior=(1.0, 0.01), seed=kw["seed"])
    for s in range(S): d.set_genomic(s, 15.0, 30.0)
    return d
n_sw = kw["chains"] * (kw["burn"] + kw["iters"])
cd_o, cd_g = cd(), cd()
og = _oracle.OrcGibbs(orc, flat, *cd_o.tables(), noise_seeding=1, **kw)
og.trace_enable(n_sw)
ctx = lib.Ctx(0)
gg = lib.Gibbs(ctx, flat, *cd_g.tables(), noise_seeding=1, **kw)
gg.trace_enable(n_sw)
first_hist = None
for chain in range(kw["chains"]):
    gg.set_noise_lut(cd_g.noise_table()); gg.init_chain(chain)
    og.set_noise_lut(cd_o.noise_table()); og.init_chain(chain)
    for it in range(1, kw["burn"] + kw["iters"] + 1):
        gg.sweep(1, it > kw["burn"]); og.sweep(1, it > kw["burn"])
        hg, ho = gg.noise_counts(), og.noise_counts()
        if first_hist is None and not np.array_equal(hg, ho): first_hist = (chain, it)
        cd_g.sample_noise_parameters(hg); cd_o.sample_noise_parameters(ho)
        gg.set_noise_lut(cd_g.noise_table()); og.set_noise_lut(cd_o.noise_table())
    cd_g.reset_noise_rates(); cd_o.reset_noise_rates()
print("first iteration with different noise histograms:", first_hist)
ro = og.results()
rg = gg.results()
tg = gg.trace()
goff = flat["group_cluster_off"]
bad = 0
for g in range(flat["num_groups"]):
    nv = int(goff[g + 1] - goff[g])
    to = og.trace(g, nv, n_sw)
    if len(to) and not np.array_equal(to, tg[g]):
        w = np.argwhere(to != tg[g])[0]
        print("group", g, "clusters", goff[g], goff[g+1], "first trace difference at sweep/vertex/sample", w, "oracle", hex(to[tuple(w)]), "gpu", hex(tg[g][tuple(w)])); bad += 1
print("trace-different groups:", bad)
for c in range(flat["num_clusters"]):
    e0, e1 = int(ro["dip_off"][c]), int(ro["dip_off"][c + 1]); f0, f1 = int(rg["dip_off"][c]), int(rg["dip_off"][c + 1])
    ko = {(int(ro["h1"][e]), int(ro["h2"][e])): ro["freq"][e] for e in range(e0, e1)}
    kg = {(int(rg["h1"][e]), int(rg["h2"][e])): rg["freq"][e] for e in range(f0, f1)}
    if set(ko) != set(kg) or any((ko[k] != kg[k]).any() for k in ko):
        diff = [(k, ko.get(k), kg.get(k)) for k in sorted(set(ko) | set(kg)) if k not in ko or k not in kg or (ko[k] != kg[k]).any()]
        print("cluster", c, "H", flat["num_haplotypes"][c], "V", flat["num_variants"][c], "entries oracle/gpu", len(ko), len(kg), "freq tables differ:")
        for k, a, b in diff: print("   ", k, None if a is None else a.tolist(), None if b is None else b.tolist())
so, sg = ro["stats"], rg["stats"]
print("stats equal counts:", np.array_equal(so[:, :, 0], sg[:, :, 0]), "close:", np.allclose(so, sg, rtol=1e-9, atol=1e-12))
