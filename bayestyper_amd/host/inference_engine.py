"""ctypes binding of bthost::InferenceEngine (bayestyper_amd/host/InferenceEngine.cpp), the three drivers of the genotyping stage:

  estimate_genotypes            <- estimateGenotypes            (InferenceEngine.cpp:335-382, per group :278-333)   default mode
  estimate_noise                <- estimateNoise                (:135-276)                                          noise rates from single-cluster groups
  estimate_noise_and_genotypes  <- estimateNoiseAndGenotypes    (:384-472)                                          --noise-genotyping

There is ONE implementation of the drivers — the C++ class the `bayesTyper` executable runs; this module only marshals a batch of
groups (`flat`, the bayestyper_amd.synth layout = bt_gibbs_batch) and the run's CountDistribution into it and fetches what it
collected.  By default the engine samples on the GPU (bt_gibbs_* on `ctx`).  `sampler` / `reduce_hist` hand the C++ engine another
sampler / the cross-rank histogram reduction through its callback interface (InferenceEngine.hpp: GibbsSampler, HistReducer): the CPU
tests pass the oracle's sampler and a gloo all-reduce to exercise the C++ driver and shard logic without a GPU (tests only — nothing
in the product provides a CPU sampler).
"""
import ctypes as C
import os
import tempfile

import numpy as np

from . import dll

vp = C.c_void_p
dll.bth_noise_selector_new.restype = vp
dll.bth_noise_selector_new.argtypes = [vp, vp, C.c_uint32, C.c_uint, C.c_uint32]
dll.bth_noise_selector_free.argtypes = [vp]
dll.bth_noise_selector_next_chain.restype = C.c_uint32
dll.bth_noise_selector_next_chain.argtypes = [vp, vp, vp]
dll.bth_noise_parameter_row.restype = C.c_uint
dll.bth_noise_parameter_row.argtypes = [C.c_uint, C.c_uint, vp, C.c_uint, C.c_char_p, C.c_uint]
dll.bth_engine_new.restype = vp
dll.bth_engine_new.argtypes = [vp, C.c_uint, vp, C.c_char_p, C.c_uint, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32]
dll.bth_engine_free.argtypes = [vp]
dll.bth_engine_set_sampler.argtypes = [vp, vp, vp]
dll.bth_engine_set_hist_reducer.argtypes = [vp, vp, vp]
dll.bth_engine_estimate_noise.argtypes = [vp, vp, vp, C.c_char_p, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_char_p, C.c_uint]
dll.bth_engine_estimate_genotypes.argtypes = [vp, vp, vp, C.c_char_p, C.c_uint]
dll.bth_engine_estimate_noise_and_genotypes.argtypes = [vp, vp, vp, C.c_char_p, C.c_char_p, C.c_uint]
dll.bth_engine_result_sizes.argtypes = [vp, vp]
dll.bth_engine_result_fetch.argtypes = [vp] * 8
dll.bth_engine_noise_rows.restype = C.c_uint64
dll.bth_engine_noise_rows.argtypes = [vp, vp, C.c_uint64]

NOISE_VARIANTS_BATCH_SIZE = 100000   # InferenceEngine.cpp:50


class NoiseGroupSelector:
    """bthost::NoiseGroupSelector: the groups each chain of estimateNoise runs on (InferenceEngine.cpp:141-151,172-189)"""

    def __init__(self, clusters_per_group, variants_per_group, seed, batch_size=NOISE_VARIANTS_BATCH_SIZE):
        c = np.ascontiguousarray(clusters_per_group, np.uint32)
        v = np.ascontiguousarray(variants_per_group, np.uint32)
        assert len(c) == len(v)
        self.n = len(c)
        self.h = dll.bth_noise_selector_new(c.ctypes.data, v.ctypes.data, self.n, seed, batch_size)
        self.num_variants = 0

    def next_chain(self):
        out = np.zeros(max(self.n, 1), np.uint32)
        nv = C.c_uint32()
        n = dll.bth_noise_selector_next_chain(self.h, out.ctypes.data, C.addressof(nv))
        self.num_variants = nv.value
        return out[:n].copy()

    def close(self):
        if self.h:
            dll.bth_noise_selector_free(self.h)
            self.h = None


def noise_parameter_row(chain, iteration, rates):
    r = np.ascontiguousarray(rates, np.float64)
    buf = C.create_string_buffer(64 + 32 * len(r))
    dll.bth_noise_parameter_row(chain, iteration, r.ctypes.data, len(r), buf, len(buf))
    return buf.value.decode()


def unit_group_shape(flat):
    """(clusters per group, variants per group) of a batch"""
    off = flat["group_cluster_off"].astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(flat["num_variants"].astype(np.int64))])
    return (off[1:] - off[:-1]).astype(np.uint32), (csum[off[1:]] - csum[off[:-1]]).astype(np.uint32)


class CollectedSamples:
    """what a driver collected (bt_gibbs_result_fetch layout, clusters in batch order); `parts` = groups per launch"""

    def __init__(self, res, S, num_clusters, parts):
        self._res, self.S, self.C, self.parts = res, S, num_clusters, parts

    def results(self):
        return self._res

    def posterior_summary(self):
        from .. import shard

        return shard.summary_from_results(self._res, self.C, self.S)

    def close(self):
        self._res = None


# ---- callback interface of the C++ engine (bthost_c.cpp: bth_sampler_vtable) ----
_CREATE = C.CFUNCTYPE(vp, vp, vp, vp)
_DESTROY = C.CFUNCTYPE(None, vp, vp)
_SET_LUT = C.CFUNCTYPE(None, vp, vp, vp, vp)
_SET_NOISE = C.CFUNCTYPE(None, vp, vp, vp)
_INIT = C.CFUNCTYPE(None, vp, vp, C.c_uint32)
_SWEEP = C.CFUNCTYPE(None, vp, vp, C.c_uint32, C.c_int)
_RUN = C.CFUNCTYPE(None, vp, vp)
_NOISE = C.CFUNCTYPE(None, vp, vp, vp)
_SIZES = C.CFUNCTYPE(None, vp, vp, vp, vp)
_FETCH = C.CFUNCTYPE(None, vp, vp, vp, vp, vp, vp, vp, vp)
_REDUCE = C.CFUNCTYPE(None, vp, C.c_uint64, vp)


class _VTable(C.Structure):
    _fields_ = [("create", _CREATE), ("destroy", _DESTROY), ("set_lut", _SET_LUT), ("set_noise_lut", _SET_NOISE), ("init_chain", _INIT), ("sweep", _SWEEP), ("run", _RUN),
                ("noise_counts", _NOISE), ("result_sizes", _SIZES), ("result_fetch", _FETCH)]


def _arr(ptr, dtype, n):
    dt = np.dtype(dtype)
    return np.frombuffer((C.c_char * (int(n) * dt.itemsize)).from_address(ptr), dt, int(n))


class _SamplerBridge:
    """turns a Python sampler factory `sampler(flat, lut_g, lut_n, **kw)` (objects with set_noise_lut / init_chain / sweep / run /
    noise_counts / results / close) into the C++ engine's GibbsSampler callbacks"""

    def __init__(self, factory, S):
        from .. import synth

        self.factory, self.S, self.live, self.next_id = factory, S, {}, 1
        self.error = None

        def guard(fn):
            def wrapped(*a):
                try:
                    return fn(*a)
                except BaseException as e:   # an exception must not unwind through the C++ frames
                    self.error = e
                    return None
            return wrapped

        def create(_user, params_p, batch_p):
            p = synth.GibbsParams.from_address(params_p)
            b = synth.GibbsBatch.from_address(batch_p)
            gender = _arr(p.gender, np.uint8, S).copy()
            flat = synth.from_ctypes(b, S, gender)
            kw = dict(seed=p.seed, chains=p.num_chains, burn=p.burn_in, iters=p.num_iterations, rate=p.kmer_subsampling_rate, max_hvk=p.max_haplotype_variant_kmers,
                      noise_seeding=p.noise_seeding)
            i = self.next_id
            self.next_id += 1
            self.live[i] = {"flat": flat, "kw": kw, "obj": None}
            return i

        def destroy(_user, h):
            e = self.live.pop(h, None)
            if e and e["obj"] is not None:
                e["obj"].close()

        def set_lut(_user, h, g_p, n_p):   # the sampler object is built here: its constructor takes the tables
            e = self.live[h]
            e["obj"] = self.factory(e["flat"], _arr(g_p, np.float64, S * 65536).copy(), _arr(n_p, np.float64, S * 256).copy(), **e["kw"])

        def set_noise(_user, h, n_p):
            self.live[h]["obj"].set_noise_lut(_arr(n_p, np.float64, S * 256).copy())

        def init_chain(_user, h, chain):
            self.live[h]["obj"].init_chain(chain)

        def sweep(_user, h, n, collect):
            self.live[h]["obj"].sweep(n, bool(collect))

        def run(_user, h):
            self.live[h]["obj"].run()

        def noise(_user, h, out_p):
            _arr(out_p, np.uint64, S * 256)[:] = self.live[h]["obj"].noise_counts()

        def sizes(_user, h, nd_p, nc_p):
            r = self.live[h]["res"] = self.live[h]["obj"].results()
            _arr(nd_p, np.uint64, 1)[0] = len(r["h1"])
            _arr(nc_p, np.uint64, 1)[0] = len(r["stats"])

        def fetch(_user, h, dip_off, h1, h2, freq, cell_off, stats):
            r = self.live[h].pop("res")
            nd, nc, Cn = len(r["h1"]), len(r["stats"]), len(r["dip_off"]) - 1
            _arr(dip_off, np.uint64, Cn + 1)[:] = r["dip_off"]
            _arr(cell_off, np.uint64, Cn + 1)[:] = r["cell_off"]
            if nd:
                _arr(h1, np.uint16, nd)[:] = r["h1"]
                _arr(h2, np.uint16, nd)[:] = r["h2"]
                _arr(freq, np.uint32, nd * S)[:] = np.asarray(r["freq"]).reshape(-1)
            if nc:
                _arr(stats, np.float64, nc * 12)[:] = np.asarray(r["stats"]).reshape(-1)

        self.vt = _VTable(_CREATE(guard(create)), _DESTROY(guard(destroy)), _SET_LUT(guard(set_lut)), _SET_NOISE(guard(set_noise)), _INIT(guard(init_chain)), _SWEEP(guard(sweep)),
                          _RUN(guard(run)), _NOISE(guard(noise)), _SIZES(guard(sizes)), _FETCH(guard(fetch)))


class InferenceEngine:
    def __init__(self, ctx, seed, burn=100, samples=250, chains=20, rate=0.1, max_hvk=500, sampler=None, reduce_hist=None):
        """ctx: lib.Ctx (the GPU the default sampler runs on; None only together with `sampler`).
        reduce_hist: callable(np.uint64[S*256]) -> the histogram summed over all ranks (None: single rank)"""
        if sampler is None and ctx is None:
            raise ValueError("InferenceEngine needs a GPU context (lib.Ctx): there is no CPU path")
        self.ctx, self.seed, self.burn, self.samples, self.chains, self.rate, self.max_hvk = ctx, seed, burn, samples, chains, rate, max_hvk
        self.sampler, self.reduce_hist = sampler, reduce_hist
        self.low_variant_warning = False

    # ---- a C++ engine for one call ----
    class _Handle:
        def __init__(self, eng, flat, sample_names, max_groups_per_launch=0):
            from .. import synth

            S = flat["S"]
            self.S = S
            gender = np.ascontiguousarray(flat["gender"], np.uint8)
            names = "\t".join(sample_names if sample_names is not None else [f"sample_{s}" for s in range(S)]).encode()
            self.h = dll.bth_engine_new(eng.ctx.h if eng.ctx is not None else None, S, gender.ctypes.data, names, eng.seed, eng.burn, eng.samples, eng.chains, eng.rate, eng.max_hvk,
                                        max_groups_per_launch or 0)
            self.bridge = None
            if eng.sampler is not None:
                self.bridge = _SamplerBridge(eng.sampler, S)
                dll.bth_engine_set_sampler(self.h, C.addressof(self.bridge.vt), None)
            self.reduce_cb = None
            if eng.reduce_hist is not None:
                def reduce(hist_p, n, _user):
                    a = _arr(hist_p, np.uint64, n)
                    a[:] = eng.reduce_hist(a.copy())
                self.reduce_cb = _REDUCE(reduce)
                dll.bth_engine_set_hist_reducer(self.h, C.cast(self.reduce_cb, vp), None)
            _, self.batch, self.keep = synth.to_ctypes(flat)
            self.err = C.create_string_buffer(1024)

        def check(self, rc):
            if self.bridge is not None and self.bridge.error is not None:
                raise self.bridge.error
            if rc != 0:
                raise RuntimeError(self.err.value.decode())

        def rows(self):
            n = dll.bth_engine_noise_rows(self.h, None, 0)
            out = np.zeros(n)
            dll.bth_engine_noise_rows(self.h, out.ctypes.data, n)
            return out.reshape(-1, 2 + self.S)

        def collected(self):
            sz = np.zeros(4, np.uint64)
            dll.bth_engine_result_sizes(self.h, sz.ctypes.data)
            Cn, nd, nc, nl = (int(x) for x in sz)
            dip_off, cell_off = np.zeros(Cn + 1, np.uint64), np.zeros(Cn + 1, np.uint64)
            h1, h2 = np.zeros(max(nd, 1), np.uint16), np.zeros(max(nd, 1), np.uint16)
            freq, stats, parts = np.zeros(max(nd, 1) * self.S, np.uint32), np.zeros(max(nc, 1) * 12), np.zeros(max(nl, 1), np.uint32)
            dll.bth_engine_result_fetch(self.h, dip_off.ctypes.data, h1.ctypes.data, h2.ctypes.data, freq.ctypes.data, cell_off.ctypes.data, stats.ctypes.data, parts.ctypes.data)
            res = {"dip_off": dip_off, "h1": h1[:nd], "h2": h2[:nd], "freq": freq[: nd * self.S].reshape(nd, self.S), "cell_off": cell_off, "stats": stats[: nc * 12].reshape(nc, 3, 4)}
            return CollectedSamples(res, self.S, Cn, list(parts[:nl]))

        def close(self):
            if self.h:
                dll.bth_engine_free(self.h)
                self.h = None

    # ---- default mode --------------------------------------------------------------------------------------
    def estimate_genotypes(self, flat, count_distribution, max_groups_per_launch=None):
        """-> CollectedSamples of this rank's groups.  A unit whose sampler state does not fit the GPU at once is run as consecutive
        launches (max_groups_per_launch groups each, or halves when a launch does not fit): groups are independent and keep their
        unit-wide index, so the split changes nothing but the peak memory (InferenceEngine.cpp:335-382 hands groups to its threads
        in batches the same way)."""
        h = self._Handle(self, flat, None, max_groups_per_launch)
        try:
            h.check(dll.bth_engine_estimate_genotypes(h.h, C.addressof(h.batch), count_distribution.h, h.err, len(h.err)))
            return h.collected()
        finally:
            h.close()

    # ---- estimateNoise -------------------------------------------------------------------------------------
    def estimate_noise(self, count_distribution, flat, unit_shape=None, output_prefix=None, sample_names=None, variants_batch_size=NOISE_VARIANTS_BATCH_SIZE):
        """Sets count_distribution's noise rates to the mean of the post-burn-in draws of all chains and returns the rows of the
        noise parameter file in full precision [(chain, iteration, rate_0, ...)].  unit_shape = (clusters per group, variants per
        group) of the WHOLE unit (every rank passes the same arrays); default: `flat` is the whole unit."""
        h = self._Handle(self, flat, sample_names)
        tmp = None
        try:
            if output_prefix is None:
                tmp = tempfile.TemporaryDirectory()
                output_prefix = os.path.join(tmp.name, "noise")
            cl = va = None
            n_unit = 0
            if unit_shape is not None:
                cl, va = np.ascontiguousarray(unit_shape[0], np.uint32), np.ascontiguousarray(unit_shape[1], np.uint32)
                n_unit = len(cl)
            warn = C.c_int(0)
            h.check(dll.bth_engine_estimate_noise(h.h, count_distribution.h, C.addressof(h.batch), output_prefix.encode(), variants_batch_size, cl.ctypes.data if cl is not None else None,
                                                  va.ctypes.data if va is not None else None, n_unit, C.addressof(warn), h.err, len(h.err)))
            self.low_variant_warning = bool(warn.value)
            return h.rows()
        finally:
            h.close()
            if tmp is not None:
                tmp.cleanup()

    # ---- estimateNoiseAndGenotypes -------------------------------------------------------------------------
    def estimate_noise_and_genotypes(self, flat, count_distribution, output_prefix=None, sample_names=None):
        """-> (CollectedSamples, rows of the noise parameter file in full precision)"""
        h = self._Handle(self, flat, sample_names)
        tmp = None
        try:
            if output_prefix is None:
                tmp = tempfile.TemporaryDirectory()
                output_prefix = os.path.join(tmp.name, "noise")
            h.check(dll.bth_engine_estimate_noise_and_genotypes(h.h, C.addressof(h.batch), count_distribution.h, output_prefix.encode(), h.err, len(h.err)))
            return h.collected(), h.rows()
        finally:
            h.close()
            if tmp is not None:
                tmp.cleanup()
