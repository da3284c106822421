"""Frozen outputs of the ORACLE (oracle/oracle_gibbs.cpp, the scalar restatement of the reference's sampler) for small groups of the four shape
classes at S = 1, 3, 10 — SURVEY §8(c) fixtures (7), (9), (11) as far as the oracle exposes them:

  per case (shape x S): the flattened input batch, the first 50 sweeps' diplotype trace of every group, the collected samples of a short schedule
  (2 chains x (20 + 30) sweeps: diplotype sampling frequencies, allele k-mer statistics), a digest of the count-model tables the run used;
  one toy unit (S = 3, 40 groups of all classes): the rows of the noise parameter file of estimateNoiseAndGenotypes (3 chains x (10 + 20) iterations).

THESE FIXTURES PIN NOTHING AGAINST THE REFERENCE: the sampler classes of the reference include Boost headers and cannot be built in this image
("parity unpinned").  They freeze the oracle: tests/test_golden_oracle_cpu.py fails when oracle_gibbs.cpp drifts from them, and
tests/test_golden_gpu.py checks the GPU path against them without needing the oracle at all.

usage: python tests/golden/make_oracle_fixtures.py     (rewrites tests/golden/oracle_gibbs_fixtures.npz)"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SCHEDULE = dict(seed=77, chains=2, burn=20, iters=30)
TRACE_SWEEPS = 50
CASES = [(shape, S) for S in (1, 3, 10) for shape in "ABCD"]
NOISE = dict(seed=9, chains=3, burn=10, iters=20, noise_seeding=1)
FLAT_KEYS = None   # every array of the batch dict


def case_batch(shape, S):
    from bayestyper_amd import synth

    n = {"A": 6, "B": 3, "C": 2, "D": 1}[shape]
    return synth.make_batch(shape, n, S, seed=500 + 10 * S + "ABCD".index(shape), templates=min(n, 3))


def toy_unit():
    from bayestyper_amd import synth

    return synth.concat([synth.make_batch("A", 30, 3, seed=61, templates=5), synth.make_batch("B", 6, 3, seed=62, templates=2), synth.make_batch("C", 3, 3, seed=63),
                         synth.make_batch("D", 1, 3, seed=64)])


def noise_count_distribution():
    """the run's count model as the executable sets it up (host layer, CPU code): NB(15, 30) genomic counts, noise-rate prior (1, 0.01)"""
    from bayestyper_amd.host import count_model

    d = count_model.CountDistribution(3, prior=(1.0, 0.01), seed=NOISE["seed"])
    for s in range(3):
        d.set_genomic(s, 15.0, 30.0)
    return d


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def pack_flat(prefix, flat, out):
    for k, v in flat.items():
        if isinstance(v, np.ndarray):
            out[f"{prefix}/flat/{k}"] = v
        elif isinstance(v, (int, np.integer)):
            out[f"{prefix}/flat/{k}"] = np.asarray(v, np.int64)


def unpack_flat(prefix, z):
    flat = {}
    for k in z.files:
        if k.startswith(prefix + "/flat/"):
            v = z[k]
            flat[k[len(prefix) + 6:]] = int(v) if v.shape == () else v
    return flat


def oracle_outputs(orc, _oracle):
    """everything the fixture file holds, computed by the oracle"""
    out = {}
    for shape, S in CASES:
        key = f"{shape}{S}"
        flat = case_batch(shape, S)
        lut_g, lut_n = _oracle.build_luts(orc, S)
        og = _oracle.OrcGibbs(orc, flat, lut_g, lut_n, **SCHEDULE)
        og.trace_enable(TRACE_SWEEPS)
        og.run(1)
        goff = flat["group_cluster_off"]
        for g in range(flat["num_groups"]):
            out[f"{key}/trace/{g}"] = np.asarray(og.trace(g, int(goff[g + 1] - goff[g]), TRACE_SWEEPS))
        r = og.results()
        for k in ("dip_off", "h1", "h2", "freq", "cell_off", "stats"):
            out[f"{key}/res/{k}"] = np.asarray(r[k])
        out[f"{key}/lut_digest"] = np.frombuffer(bytes.fromhex(digest(lut_g, lut_n)), np.uint8)
        og.close()
        pack_flat(key, flat, out)
    flat = toy_unit()
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    og = _oracle.OrcGibbs(orc, flat, *noise_count_distribution().tables(), **NOISE)
    out["noise/rows"] = np.asarray(og.estimate_noise_and_genotypes(prior=(1.0, 0.01), threads=1), np.float64)
    r = og.results()
    for k in ("dip_off", "h1", "h2", "freq"):
        out[f"noise/res/{k}"] = np.asarray(r[k])
    og.close()
    pack_flat("noise", flat, out)
    return out


if __name__ == "__main__":
    import _oracle

    out = oracle_outputs(_oracle.load_oracle(), _oracle)
    dst = os.path.join(HERE, "oracle_gibbs_fixtures.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "arrays")
