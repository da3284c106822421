"""InferenceEngine: the three drivers of the genotyping stage above the C ABI of libbtgpu.so.

Mirrors the reference's interface (include/bayesTyper/InferenceEngine.hpp:60-62, src/bayesTyper/InferenceEngine.cpp):

  estimate_genotypes            <- estimateGenotypes            (:335-382, per group :278-333)   default mode, one launch
  estimate_noise                <- estimateNoise                (:135-276)                       noise rates from single-cluster groups
  estimate_noise_and_genotypes  <- estimateNoiseAndGenotypes    (:384-472)                       --noise-genotyping

In the two noise drivers every iteration is: one sweep of all (selected) groups, the noise-count histogram of all of them added
up (CountAllocation, merged under a mutex in the reference; across GPUs: one all-reduce of S*256 counters), one gamma draw per
sample from the run's CountDistribution generator, a rebuild of the noise log-pmf table.  The sweep and the histogram run on the
GPU (bt_gibbs_sweep, bt_gibbs_noise_counts), the draw on the host — every rank draws from an identically seeded generator, so
all ranks hold the same rates without a broadcast.

`flat` is this rank's batch of groups (bayestyper_amd.synth layout; `group_index` = index of each group in the whole unit).
`sampler` builds the object that runs a batch: the default is lib.Gibbs on the engine's GPU context; the CPU tests of the
multi-rank logic pass the oracle's sampler instead (tests only — there is no CPU path in the product).
"""
import ctypes as C

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
    """results of consecutive launches over disjoint, consecutive group ranges, presented like one sampler's"""

    def __init__(self, parts):
        self.parts = parts

    def results(self):
        out = {}
        for key, off in (("h1", None), ("h2", None), ("freq", None), ("stats", None)):
            out[key] = np.concatenate([r[key] for r, _ in self.parts])
        for off_key, payload in (("dip_off", "h1"), ("cell_off", "stats")):
            pieces, base = [], 0
            for r, _ in self.parts:
                pieces.append(r[off_key][:-1].astype(np.uint64) + np.uint64(base))
                base += int(r[off_key][-1])
            out[off_key] = np.concatenate(pieces + [np.array([base], np.uint64)])
        return out

    def posterior_summary(self):
        return np.concatenate([s for _, s in self.parts])

    def close(self):
        self.parts = []


class InferenceEngine:
    def __init__(self, ctx, seed, burn=100, samples=250, chains=20, rate=0.1, max_hvk=500, sampler=None, reduce_hist=None):
        """reduce_hist: callable(np.uint64[S*256]) -> the histogram summed over all ranks (None: single rank)"""
        self.ctx, self.seed, self.burn, self.samples, self.chains = ctx, seed, burn, samples, chains
        self.rate, self.max_hvk = rate, max_hvk
        self.reduce_hist = reduce_hist
        if sampler is None:
            from .. import lib

            if ctx is None:
                raise ValueError("InferenceEngine needs a GPU context (lib.Ctx): there is no CPU path")
            sampler = lambda flat, lut_g, lut_n, **kw: lib.Gibbs(ctx, flat, lut_g, lut_n, **kw)   # noqa: E731
        self.sampler = sampler

    def _kw(self, noise_seeding):
        return dict(seed=self.seed, chains=self.chains, burn=self.burn, iters=self.samples, rate=self.rate, max_hvk=self.max_hvk, noise_seeding=noise_seeding)

    # ---- default mode --------------------------------------------------------------------------------------
    def estimate_genotypes(self, flat, count_distribution, max_groups_per_launch=None):
        """-> an object holding the collected samples of this rank's groups (results(), posterior_summary(), close()).
        A unit whose state does not fit the GPU at once (one lane of state per group: ~0.14 MB for an SNV group, tens of MB for a
        nested SV group) is run as consecutive launches of at most max_groups_per_launch groups — groups are independent and keep
        their unit-wide index, so the split changes nothing but the peak memory (InferenceEngine.cpp:335-382 hands groups to its
        threads in batches the same way)."""
        lut_g, lut_n = count_distribution.tables()
        if max_groups_per_launch is None or flat["num_groups"] <= max_groups_per_launch:
            g = self.sampler(flat, lut_g, lut_n, **self._kw(0))
            g.run()
            return g
        from .. import shard

        parts = []
        for a in range(0, flat["num_groups"], max_groups_per_launch):
            sub = shard.take_groups(flat, np.arange(a, min(a + max_groups_per_launch, flat["num_groups"])))
            g = self.sampler(sub, lut_g, lut_n, **self._kw(0))
            g.run()
            parts.append((g.results(), g.posterior_summary() if hasattr(g, "posterior_summary") else None))
            g.close()   # frees the launch's HBM before the next one is built
        return CollectedSamples(parts)

    # ---- shared iteration of the two noise drivers (sampleGenotypesCallback + sampleNoiseParameters) --------
    def _iteration(self, g, S, count_distribution, collect):
        if g is not None:
            g.sweep(1, collect)
            hist = g.noise_counts()
        else:   # a rank without groups in this chain still takes part in the reduction
            hist = np.zeros(S * 256, np.uint64)
        if self.reduce_hist is not None:
            hist = self.reduce_hist(hist)
        count_distribution.sample_noise_parameters(hist)
        if g is not None:
            g.set_noise_lut(count_distribution.noise_table())

    @staticmethod
    def _log(f, trace, chain, iteration, rates):
        trace.append(np.concatenate([[chain, iteration], rates]))
        if f is not None:
            f.write(noise_parameter_row(chain, iteration, rates))

    @staticmethod
    def _open(output_prefix, sample_names, S):
        if output_prefix is None:
            return None
        f = open(output_prefix + ".txt", "w")
        names = sample_names if sample_names is not None else [f"sample_{s}" for s in range(S)]
        f.write("Chain\tIteration" + "".join("\t" + n for n in names) + "\n")
        return f

    # ---- estimateNoise -------------------------------------------------------------------------------------
    def estimate_noise(self, count_distribution, flat, unit_shape=None, output_prefix=None, sample_names=None, variants_batch_size=NOISE_VARIANTS_BATCH_SIZE):
        """Sets count_distribution's noise rates to the mean of the post-burn-in draws of all chains and returns the rows of the
        noise parameter file as an array [(chain, iteration, rate_0, ...)].  unit_shape = (clusters per group, variants per group)
        of the WHOLE unit (every rank passes the same arrays); default: `flat` is the whole unit."""
        from .. import shard

        S = flat["S"]
        if unit_shape is None:
            if not np.array_equal(flat["group_index"], np.arange(flat["num_groups"])):
                raise ValueError("estimate_noise: pass unit_shape when `flat` is a shard of the unit")
            unit_shape = unit_group_shape(flat)
        sel = NoiseGroupSelector(unit_shape[0], unit_shape[1], self.seed, variants_batch_size)
        local_pos = {int(g): i for i, g in enumerate(flat["group_index"])}
        lut_g = count_distribution.tables()[0]
        mean = np.zeros(S)
        trace = []
        f = self._open(output_prefix, sample_names, S)
        for chain in range(self.chains):
            chosen = sel.next_chain()
            mine = [local_pos[int(g)] for g in chosen if int(g) in local_pos]
            g = None
            if mine:
                g = self.sampler(shard.take_groups(flat, mine), lut_g, count_distribution.noise_table(), **self._kw(1))
                g.init_chain(chain)   # a fresh sampler: genotypers are constructed with seed + (i+1)(chain+1) (:70), as after resetGroup
            self._log(f, trace, chain + 1, 0, count_distribution.noise_rates())
            for iteration in range(1, self.burn + self.samples + 1):
                self._iteration(g, S, count_distribution, False)
                rates = count_distribution.noise_rates()
                self._log(f, trace, chain + 1, iteration, rates)
                if self.burn < iteration:
                    mean += rates
            if g is not None:
                g.close()   # resetGroupsCallback: the genotypers of this chain are deleted (:240-251)
            count_distribution.reset_noise_rates()
        mean /= self.samples * self.chains
        count_distribution.set_noise_rates(mean)
        self._log(f, trace, 0, 0, count_distribution.noise_rates())
        self.low_variant_warning = sel.num_variants < variants_batch_size   # the reference's warning (:270-274)
        sel.close()
        if f is not None:
            f.close()
        return np.array(trace)

    # ---- estimateNoiseAndGenotypes -------------------------------------------------------------------------
    def estimate_noise_and_genotypes(self, flat, count_distribution, output_prefix=None, sample_names=None):
        """-> (sampler holding the collected samples, rows of the noise parameter file)"""
        S = flat["S"]
        trace = []
        f = self._open(output_prefix, sample_names, S)
        g = None
        if flat["num_groups"]:
            lut_g, lut_n = count_distribution.tables()
            g = self.sampler(flat, lut_g, lut_n, **self._kw(1))
        for chain in range(self.chains):
            if g is not None:
                g.set_noise_lut(count_distribution.noise_table())
                g.init_chain(chain)
            self._log(f, trace, chain + 1, 0, count_distribution.noise_rates())
            for iteration in range(1, self.burn + self.samples + 1):
                self._iteration(g, S, count_distribution, iteration > self.burn)
                self._log(f, trace, chain + 1, iteration, count_distribution.noise_rates())
            count_distribution.reset_noise_rates()
        if f is not None:
            f.close()
        return g, np.array(trace)
