"""The graph stages alone (best-path search, path k-mer enumeration, classification, candidates, the multigroup pass) on synthetic SNV / indel clusters —
the driver for the rocprofv3 counter passes of find_paths_kernel and mg_order_kernel (tools/profile_round.sh).  usage: graph_stages.py [clusters]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bayestyper_amd import lib, synth_graphs

K, S = 55, 3
n_cl = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
ctx = lib.Ctx(0)
prng = np.random.default_rng(11)
gs = [synth_graphs.random_cluster(prng, K, int(prng.integers(1, 4)), int(prng.integers(2, 5)), kinds=("snv", "snv", "snv", "ins", "del")) for _ in range(n_cl)]
fg = synth_graphs.flatten(gs)


def timed(fn):
    ctx.sync()
    t = time.perf_counter()
    r = fn()
    ctx.sync()
    return r, time.perf_counter() - t


gp, t_create = timed(lambda: lib.Paths(ctx, fg, K))
W = gp.num_windows
pb = lib.Bloom.create(ctx, W + 1_000_000, 1e-4, K, threaded=True)
_, t_bloom = timed(lambda: gp.count_kmers(pb))
for g_ in gs:
    g_.paths = None
fg2 = synth_graphs.flatten(gs)
gf, _ = timed(lambda: lib.FindPaths(ctx, fg2, K, 32, 1))
_, t_find = timed(lambda: gf.sample(pb, np.arange(n_cl, dtype=np.uint32) + 7))
pb2 = lib.Bloom.create(ctx, W + 1_000_000, 1e-4, K, threaded=True)
mgt = lib.Table(ctx, max(W // 8, 1024), 1, K)
_, t_mg = timed(lambda: gp.count_multigroup(np.arange(n_cl, dtype=np.uint32), pb2, mgt))
print(json.dumps({"clusters": n_cl, "kmer_windows": int(W), "enumerate_s": t_create, "find_sample_paths_s": t_find, "multigroup_s": t_mg,
                  "find_sample_paths_clusters_per_sec": n_cl / t_find, "multigroup_windows_per_sec": W / t_mg}))
