"""The GPU path (through the C ABI and the C++ InferenceEngine) against the oracle's FROZEN outputs (tests/golden/oracle_gibbs_fixtures.npz): a check
that does not depend on rebuilding the oracle on the GPU box.  Per shape class (two-haplotype, multi-variant, nested SV, many-candidate) and S = 1, 3, 10:
the first 50 sweeps' diplotypes of every group, the diplotype sampling frequencies (exact), the allele k-mer statistics (1e-9); one toy unit through
estimateNoiseAndGenotypes: every row of the noise parameter file (exact) and the collected samples."""
import hashlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


@pytest.fixture(scope="module")
def frozen():
    return np.load(os.path.join(GOLDEN, "oracle_gibbs_fixtures.npz"))


@pytest.mark.parametrize("S", [1, 3, 10])
@pytest.mark.parametrize("shape", list("ABCD"))
def test_sampler_against_frozen_oracle_outputs(gpu_ctx, frozen, shape, S):
    import make_oracle_fixtures as mk
    from bayestyper_amd import lib
    from bayestyper_amd.host import count_model

    key = f"{shape}{S}"
    flat = mk.unpack_flat(key, frozen)
    lut_g, lut_n = count_model.build_luts(S, mean=15.0, var=30.0, noise_rate=0.05)
    assert mk.digest(lut_g, lut_n) == bytes(frozen[f"{key}/lut_digest"]).hex(), "the count-model tables differ from the ones the fixtures were made with"
    g = lib.Gibbs(gpu_ctx, flat, lut_g, lut_n, **mk.SCHEDULE)
    g.trace_enable(mk.TRACE_SWEEPS)
    g.run()
    gpu_ctx.sync()
    tr = g.trace()
    for gi in range(flat["num_groups"]):
        assert np.array_equal(np.asarray(tr[gi])[: mk.TRACE_SWEEPS], frozen[f"{key}/trace/{gi}"]), f"group {gi}: diplotype trace"
    r = g.results()
    g.close()
    for k in ("dip_off", "h1", "h2", "freq", "cell_off"):
        assert np.array_equal(np.asarray(r[k]), frozen[f"{key}/res/{k}"]), k
    assert np.allclose(np.asarray(r["stats"]), frozen[f"{key}/res/stats"], rtol=1e-9, atol=1e-12)


def test_noise_genotyping_against_frozen_oracle_rows(gpu_ctx, frozen):
    import make_oracle_fixtures as mk
    from bayestyper_amd.host.inference_engine import InferenceEngine

    flat = mk.unpack_flat("noise", frozen)
    kw = mk.NOISE
    eng = InferenceEngine(gpu_ctx, kw["seed"], burn=kw["burn"], samples=kw["iters"], chains=kw["chains"])
    gg, rows = eng.estimate_noise_and_genotypes(flat, mk.noise_count_distribution())
    r = gg.results()
    gg.close()
    want = frozen["noise/rows"]
    assert rows.shape == want.shape and np.array_equal(rows, want)
    for k in ("dip_off", "h1", "h2", "freq"):
        assert np.array_equal(np.asarray(r[k]), frozen[f"noise/res/{k}"]), k
