"""CountDistribution LUTs (host C++: bayestyper_amd/host/CountDistribution.cpp) for bench.py and the tests."""
import ctypes as C

import numpy as np

from . import dll

vp = C.c_void_p
dll.bth_build_luts.argtypes = [C.c_uint, vp, vp, vp, vp, vp, vp]
dll.bth_count_distribution_new.restype = vp
dll.bth_count_distribution_new.argtypes = [C.c_uint, C.c_float, C.c_float, C.c_uint]
dll.bth_count_distribution_free.argtypes = [vp]
dll.bth_count_distribution_set_genomic.argtypes = [vp, C.c_uint, C.c_double, C.c_double, C.c_uint]
dll.bth_count_distribution_noise_rates.argtypes = [vp, vp]
dll.bth_count_distribution_set_noise_rates.argtypes = [vp, vp, C.c_uint]
dll.bth_count_distribution_reset_noise_rates.argtypes = [vp]
dll.bth_count_distribution_sample_noise.argtypes = [vp, vp, C.c_uint]
dll.bth_count_distribution_tables.argtypes = [vp, vp, vp, C.c_uint]
dll.bth_count_distribution_export_generator.argtypes = [vp, vp, vp]
dll.bth_count_distribution_import_generator.argtypes = [vp, vp, C.c_double]


def build_luts(S, mean=15.0, var=30.0, noise_rate=0.05, multiplicity=1):
    m = np.full(S, mean, np.float64)
    v = np.full(S, var, np.float64)
    mu = np.full(S, multiplicity, np.uint32)
    nr = np.full(S, noise_rate, np.float64) if np.isscalar(noise_rate) else np.ascontiguousarray(noise_rate, np.float64)
    g = np.zeros(S * 65536, np.float64)
    n = np.zeros(S * 256, np.float64)
    rc = dll.bth_build_luts(S, m.ctypes.data, v.ctypes.data, mu.ctypes.data, nr.ctypes.data, g.ctypes.data, n.ctypes.data)
    if rc != 0:
        raise ValueError("bth_build_luts failed")
    return g, n


class CountDistribution:
    """CountDistribution(samples, options) of the reference: noise rates drawn from mt19937(seed) + gamma, LUT rebuilds"""

    def __init__(self, S, prior=(1.0, 0.01), seed=42):
        self.S = S
        self.h = dll.bth_count_distribution_new(S, prior[0], prior[1], seed)

    def set_genomic(self, s, mean, var, multiplicity=1):
        dll.bth_count_distribution_set_genomic(self.h, s, mean, var, multiplicity)

    def noise_rates(self):
        out = np.zeros(self.S)
        dll.bth_count_distribution_noise_rates(self.h, out.ctypes.data)
        return out

    def set_noise_rates(self, rates):
        r = np.ascontiguousarray(rates, np.float64)
        dll.bth_count_distribution_set_noise_rates(self.h, r.ctypes.data, self.S)

    def reset_noise_rates(self):
        dll.bth_count_distribution_reset_noise_rates(self.h)

    def sample_noise_parameters(self, hist):
        h = np.ascontiguousarray(hist, np.uint64)
        dll.bth_count_distribution_sample_noise(self.h, h.ctypes.data, self.S)

    def export_generator(self):
        """-> (uint32[626]: the 624 mt19937 words, the next index, the normal distribution's saved flag; the saved variate)"""
        w, s = np.zeros(626, np.uint32), C.c_double()
        dll.bth_count_distribution_export_generator(self.h, w.ctypes.data, C.byref(s))
        return w, s.value

    def import_generator(self, words, saved):
        w = np.ascontiguousarray(words, np.uint32)
        dll.bth_count_distribution_import_generator(self.h, w.ctypes.data, saved)

    def tables(self):
        g = np.zeros(self.S * 65536)
        n = np.zeros(self.S * 256)
        dll.bth_count_distribution_tables(self.h, g.ctypes.data, n.ctypes.data, self.S)
        return g, n

    def noise_table(self):
        n = np.zeros(self.S * 256)
        dll.bth_count_distribution_tables(self.h, None, n.ctypes.data, self.S)
        return n

    def close(self):
        if self.h:
            dll.bth_count_distribution_free(self.h)
            self.h = None
