"""CPU tests of the libstdc++-compatible primitives the Gibbs kernels are built on (bt_diag_* run the same
__host__ __device__ code on the host): draw streams and unordered_set iteration order must equal libstdc++'s."""
import ctypes as C

import numpy as np
import pytest

import _oracle
from _oracle import _ptr


@pytest.fixture(scope="module")
def orc(oracle):
    _oracle._gibbs_sigs(oracle.l)
    return oracle


def _both(orc, seed, kind, a, b, n, out_n=None):
    from bayestyper_amd import lib

    a = None if a is None else np.ascontiguousarray(a, np.float64)
    b = None if b is None else np.ascontiguousarray(b, np.float64)
    o1 = np.zeros(out_n or n)
    o2 = np.zeros(out_n or n)
    orc.l.orc_rng(seed, kind, None if a is None else _ptr(a), None if b is None else _ptr(b), n, _ptr(o1))
    lib.check(lib.bt_diag_rng(seed, kind, None if a is None else _ptr(a), None if b is None else _ptr(b), n, _ptr(o2)))
    return o1, o2


@pytest.mark.parametrize("seed", [0, 5, 42, 12345, 4294967295])
def test_rng_streams(orc, seed):
    rng = np.random.default_rng(seed)
    for kind in (0, 1):
        o1, o2 = _both(orc, seed, kind, None, None, 3000)
        assert np.array_equal(o1, o2)
    # uniform_int over assorted ranges (Lemire rejection path included)
    a = rng.choice([0, 1, 2, 6, 9, 99, 1000, 65535, 2 ** 31 - 2, 3 * 2 ** 29], 2000).astype(np.float64)
    o1, o2 = _both(orc, seed, 3, a, None, len(a))
    assert np.array_equal(o1, o2)
    o1, o2 = _both(orc, seed, 4, [0.1], None, 5000)
    assert np.array_equal(o1, o2)
    for n in (0, 1, 2, 3, 10, 11, 110, 441, 4000):
        o1, o2 = _both(orc, seed, 5, [float(n)], None, 0, out_n=max(n, 1))
        assert np.array_equal(o1, o2)


def test_gamma_stream_matches_libstdcxx(orc):
    """gamma draws interleaved over changing shapes: same accept/reject decisions; values equal up to libm ulps (here both
    sides run on the host, so they are bit-identical)"""
    rng = np.random.default_rng(3)
    a = rng.choice([1.0, 2.0, 3.0, 11.0, 21.0, 0.5, 1.5, 250.0], 4000)
    b = rng.choice([1.0, 0.01, 2.5], 4000)
    o1, o2 = _both(orc, 7, 2, a, b, 4000)
    assert np.array_equal(o1, o2)
    assert o1[0] != o1[1]


def test_shuffle_large_path(orc):
    """n >= 65536 takes std::shuffle's one-draw-per-element path"""
    o1, o2 = _both(orc, 9, 5, [70000.0], None, 0, out_n=70000)
    assert np.array_equal(o1, o2)
    assert sorted(o1.astype(int).tolist()) == list(range(70000))


@pytest.mark.parametrize("universe", [1, 2, 10, 13, 14, 29, 30, 32, 60, 128, 256, 1000])
def test_unordered_set_order(orc, universe):
    from bayestyper_amd import lib

    rng = np.random.default_rng(universe)
    for trial in range(8):
        ops, vals = [], []
        present = set()
        # the access pattern of SparseFrequencyDistribution: fill, move some elements out and back, clear, refill
        for v in range(universe):
            ops.append(0), vals.append(v), present.add(v)
        for _ in range(int(rng.integers(0, 4 * universe + 5))):
            r = rng.random()
            if r < 0.45 and present:
                v = int(rng.choice(sorted(present)))
                ops.append(1), vals.append(v), present.discard(v)
            elif r < 0.95:
                v = int(rng.integers(0, universe))
                if v not in present:
                    ops.append(0), vals.append(v), present.add(v)
            else:
                ops.append(2), vals.append(0), present.clear()
        ops = np.asarray(ops, np.uint8)
        vals = np.asarray(vals, np.uint32)
        o1, o2 = np.zeros(universe, np.uint32), np.zeros(universe, np.uint32)
        n1, n2 = C.c_uint32(), C.c_uint32()
        orc.l.orc_uset_replay(universe, _ptr(ops), _ptr(vals), len(ops), _ptr(o1), C.byref(n1))
        lib.check(lib.bt_diag_uset_replay(universe, _ptr(ops), _ptr(vals), len(ops), _ptr(o2), C.byref(n2)))
        assert n1.value == n2.value == len(present)
        assert np.array_equal(o1[: n1.value], o2[: n2.value])
