"""The shape of the bench's synthetic mixture in the buckets `BT_STAGE_TIMES=1 bayesTyper genotype` prints for a real unit ("unit shape: ..." lines on stderr):
haplotype candidates, path k-mers and variants per cluster, clusters per group (upper bucket bound: count).  usage: python tools/mixture_shape.py [groups 600320] [S 3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayestyper_amd import synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 600320
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = synth.make_mixture(G, S, seed=1000)


def buckets(x):
    b = np.ones_like(x)
    while (b < x).any():
        b = np.where(b < x, b * 2, b)
    u, c = np.unique(b, return_counts=True)
    return " ".join(f"{int(a)}:{int(n)}" for a, n in zip(u, c))


H = f["num_haplotypes"].astype(np.int64)
K = (f["kmer_off"][1:].astype(np.int64) - f["kmer_off"][:-1].astype(np.int64))
print(f"unit shape: {f['num_groups']} groups, {f['num_clusters']} clusters, sum over clusters of k-mers x haplotype candidates {int((K * H).sum())}  (synth.make_mixture {f['mixture']})")
print("unit shape: haplotype candidates per cluster (upper bucket bound: count)", buckets(H))
print("unit shape: path k-mers per cluster (upper bucket bound: count)", buckets(K))
print("unit shape: variants per cluster (upper bucket bound: count)", buckets(f["num_variants"].astype(np.int64)))
print("unit shape: clusters per group (upper bucket bound: count)", buckets((f["group_cluster_off"][1:].astype(np.int64) - f["group_cluster_off"][:-1].astype(np.int64))))
