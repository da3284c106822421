"""Placement independence on the GPU: the first groups of a mixture batch run as a batch of their own give the posterior summaries they give inside the batch.
usage: python tools/check_placement.py [S] [groups] [subset]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayestyper_amd import lib, shard, synth
from bayestyper_amd.host import count_model

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
G = int(sys.argv[2]) if len(sys.argv) > 2 else 600_000
n_sub = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = lib.Ctx(0)
flat = synth.make_mixture(G, S, seed=1000)
lg, ln = count_model.build_luts(S, mean=15.0, var=30.0, noise_rate=0.05)


def summary(f):
    g = lib.Gibbs(ctx, f, lg, ln, seed=42)
    g.run()
    out = g.posterior_summary().reshape(-1)
    g.close()
    return out.reshape(f["num_clusters"], S * 2)


whole = summary(flat)
bounds, at = {}, 0
for shape in ("D", "C", "B", "A"):
    n = flat["mixture"].get(shape, 0)
    bounds[shape] = (at, at + n)
    at += n
for name, ids in (("first", np.arange(min(n_sub, G))), ("D", np.arange(*bounds["D"])[:1024]), ("C", np.arange(*bounds["C"])[:1024]), ("B", np.arange(*bounds["B"])[:2048]),
                  ("A", np.arange(*bounds["A"])[:4096])):
    sub = shard.take_groups(flat, ids)
    cl = shard.cluster_ids_of(flat, ids)
    got = summary(sub)
    bad = np.nonzero((got != whole[cl]).any(axis=1))[0]
    print(name, "groups", len(ids), "clusters", len(cl), "differing clusters", len(bad), bad[:10].tolist(), flush=True)
