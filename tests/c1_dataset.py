"""The plumbing dataset of BASELINE.json configs[0] / SURVEY §8(d) ("C1"): a synthetic reference, SNV candidates, one (or more) samples
with known genotypes, their KMC databases and sample Bloom filters — everything `bayesTyper cluster` / `genotype` read from disk.

  reference   `genome_len` nt uniform ACGT (seed 1), one chromosome "chr1"
  candidates  `num_snvs` SNVs at positions drawn without replacement from [55, genome_len - 110] (seed 2), alt = (ref + 1 + U{0,1,2}) mod 4
  samples     genotype per variant ~ {0/0: 0.25, 0/1: 0.5, 1/1: 0.25} (seed 3 + sample); k-mer counts = NB(mean 15, var 30) per copy on every
              55-mer of the two haplotypes + `num_error_kmers` random error 55-mers with count 1 (seed 4 + sample); KMC1 database
              (p = 7, 1 counter byte) + sample Bloom filter (fpr 1e-3: <prefix>.bloomMeta / .bloomData)

Test tooling (uses the oracle's packers and writers).  usage: python tests/c1_dataset.py <out dir> [genome_len] [num_snvs] [num_samples]"""
import os
import sys

import numpy as np

K = 55
NT = np.frombuffer(b"ACGT", np.uint8)


def _sorted_kmc_order(orc, packed):
    """KMC order = ascending ASCII order of the k-mers"""
    codes = np.searchsorted(NT, orc.unpack(packed, K).reshape(-1, K)).astype(np.uint64)
    key1 = np.zeros(len(codes), np.uint64)
    key2 = np.zeros(len(codes), np.uint64)
    for i in range(32):
        key1 |= codes[:, i] << np.uint64(2 * (31 - i))
    for i in range(32, K):
        key2 |= codes[:, i] << np.uint64(2 * (K - 1 - i))
    return np.lexsort((key2, key1))


def make(out_dir, orc, genome_len=1_000_000, num_snvs=5000, num_samples=1, num_error_kmers=1_000_000, genders=None, mean=15.0, var=30.0):
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 4, genome_len).astype(np.uint8)
    genome = NT[codes].tobytes().decode()
    with open(os.path.join(out_dir, "genome.fa"), "w") as f:
        f.write(">chr1 synthetic\n")
        for i in range(0, genome_len, 60):
            f.write(genome[i:i + 60] + "\n")
    rng = np.random.default_rng(2)
    pos = np.sort(rng.choice(np.arange(55, genome_len - 110), num_snvs, replace=False))
    alt = (codes[pos] + 1 + rng.integers(0, 3, num_snvs)) % 4
    with open(os.path.join(out_dir, "candidates.vcf"), "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=chr1,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" % genome_len)
        for i in range(num_snvs):
            f.write("chr1\t%d\tsnv%d\t%s\t%s\t.\t.\t.\n" % (pos[i] + 1, i, "ACGT"[codes[pos[i]]], "ACGT"[alt[i]]))
    genders = genders or ["F"] * num_samples
    p, size = mean / var, mean * mean / (var - mean)
    truth = []
    with open(os.path.join(out_dir, "samples.tsv"), "w") as sf:
        for s in range(num_samples):
            rng = np.random.default_rng(3 + s)
            gt = rng.choice(3, num_snvs, p=[0.25, 0.5, 0.25])            # number of alt copies
            truth.append(gt)
            haps = []
            which = rng.integers(0, 2, num_snvs)                           # the haplotype that carries a heterozygous variant's alt allele
            for h in range(2):
                c = codes.copy()
                carries = (gt == 2) | ((gt == 1) & (which == h))
                c[pos[carries]] = alt[carries]
                haps.append(NT[c].tobytes())
            km, va = orc.kmers_from_sequence(haps[0] + b"N" + haps[1], K)
            uniq, mult = np.unique(km[va == 1], axis=0, return_counts=True)
            rng = np.random.default_rng(4 + s)
            cnt = rng.negative_binomial(size * mult, p)
            err = orc.pack(NT[rng.integers(0, 4, num_error_kmers * K)].copy(), K) if num_error_kmers else np.zeros((0, 2), np.uint64)
            if len(err):   # canonical form of the error k-mers (the database holds canonical k-mers)
                asc = orc.unpack(err, K).reshape(-1, K)
                comp = np.zeros(256, np.uint8)
                comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
                rc = comp[asc[:, ::-1]]
                diff = asc != rc
                first = diff.argmax(axis=1)                                 # first position where the k-mer and its reverse complement differ
                rows = np.arange(len(asc))
                lower = ~diff.any(axis=1) | (asc[rows, first] < rc[rows, first])
                err = orc.pack(np.ascontiguousarray(np.where(lower[:, None], asc, rc)).reshape(-1), K)
            keep = cnt > 0
            allk = np.concatenate([uniq[keep], err])
            allc = np.concatenate([np.minimum(cnt[keep], 255), np.ones(len(err), np.int64)])
            allk, first = np.unique(allk, axis=0, return_index=True)
            allc = allc[first]
            order = _sorted_kmc_order(orc, allk)
            prefix = os.path.join(out_dir, f"sample{s + 1}")
            orc.kmc_write(prefix, orc.unpack(allk[order], K), allc[order].astype(np.uint32), K, 7, 1)
            from _oracle import OrcBloom

            bloom = OrcBloom(orc, len(allk), 1e-3, K)
            bloom.insert(orc.unpack(allk, K))
            bloom.save(prefix)
            bloom.close()
            sf.write(f"sample{s + 1}\t{genders[s]}\t{prefix}\n")
    return {"genome": genome, "pos": pos, "alt": alt, "truth": truth, "dir": out_dir}


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _oracle

    a = sys.argv
    d = make(a[1], _oracle.load_oracle(), int(a[2]) if len(a) > 2 else 1_000_000, int(a[3]) if len(a) > 3 else 5000, int(a[4]) if len(a) > 4 else 1)
    print("wrote", d["dir"], len(d["pos"]), "variants")
