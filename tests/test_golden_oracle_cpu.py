"""The oracle against its own frozen outputs (tests/golden/oracle_gibbs_fixtures.npz, written by tests/golden/make_oracle_fixtures.py): per-sweep diplotype
traces, collected samples and noise-driver rows of small groups of every shape class at S = 1, 3, 10.  The fixtures pin nothing against the reference
(its sampler classes cannot be built here: "parity unpinned"); they stop oracle_gibbs.cpp from drifting silently — every GPU parity test is a comparison
with this oracle."""
import os
import sys

import numpy as np

import _oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)


def test_oracle_reproduces_its_frozen_outputs(oracle):
    import make_oracle_fixtures as mk

    frozen = np.load(os.path.join(GOLDEN, "oracle_gibbs_fixtures.npz"))
    now = mk.oracle_outputs(oracle, _oracle)
    assert sorted(now) == sorted(frozen.files)
    for k in frozen.files:
        a, b = np.asarray(now[k]), frozen[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype.kind == "f":   # (statistics and rates pass through libm: identical on this image, compared within 1e-12 to survive a libm update)
            assert np.allclose(a, b, rtol=1e-12, atol=0, equal_nan=True), k
        else:
            assert np.array_equal(a, b), k
