// Device-side Gibbs genotyper for one variant-cluster group per lane (gfx950).
//
// Reference behaviour restated (not copied); every function cites what it mirrors:
//   VariantClusterGenotyper   src/bayesTyper/VariantClusterGenotyper.cpp:59-206,569-785
//   VariantClusterHaplotypes  src/bayesTyper/VariantClusterHaplotypes.cpp:45-372
//   (Sparse)FrequencyDistribution / HaplotypeFrequencyDistribution
//                             src/bayesTyper/FrequencyDistribution.cpp:43-303, HaplotypeFrequencyDistribution.cpp:79-138
//   SparsityEstimator         src/bayesTyper/SparsityEstimator.cpp:41-87
//   LogDiscreteSampler        src/bayesTyper/DiscreteSampler.cpp:100-125, Utils::logAddition Utils.hpp:105-124
//   KmerStats                 src/bayesTyper/KmerStats.cpp:51-121
//   VariantClusterGroup       src/bayesTyper/VariantClusterGroup.cpp:171-250
//
// Layout: all inputs and all sampler state of a cluster live in HBM behind one ClusterDev record (pointers
// into typed pools allocated by bt_gibbs_create).  A lane owns a whole group for the duration of a launch, so
// nothing here needs atomics except the cross-group noise histogram.
#pragma once
#include "bt_rng_device.hpp"

namespace bt {

constexpr uint16_t NOHAP = 0xFFFF;
constexpr double BT_LN2 = 0.693147180559945309417232121458176568;
constexpr double BT_DBL_EPS = 2.220446049250313080847263336181640625e-16;

// scalar slots of ClusterDev::sc
enum { SC_USE_MULTI = 0, SC_NSUB_U, SC_NSUB_M, SC_HAP_COUNT, SC_MISSING_COUNT, SC_CONSTRUCTED, SC_DIP_ENTRIES, SC_DIP_OVERFLOW, SC_COUNT };

struct GParams {
    uint32_t S, seed, num_chains, burn_in, num_iterations, max_hvk, noise_seeding;
    double rate;                // (double)(float)kmer_subsampling_rate, as bernoulli_distribution stores it
    uint8_t gender[32];
    const double *lut_g;        // [S][256][256]
    const double *lut_n;        // [S][256]
};

struct ClusterDev {
    // dimensions
    uint32_t H, V, K, HW, nu, nm, cid, Dc, D2, cache_mode, cache_mask, A, dip_cap, nd_n, ne, kv_e0;
    // inputs
    const uint8_t *M, *has_counts, *counts, *ic;
    const int32_t *shared_idx;
    const uint32_t *kv_off;       // -> kv_off[first row of the cluster]; values are absolute entry ids
    const uint16_t *kv_var;       // global array
    const uint32_t *kv_bits;      // first word of the cluster's first entry
    const uint16_t *hap_allele;   // [H][V]
    const uint32_t *hapnest_off;  // -> hapnest_off[first haplotype of the cluster]
    const uint32_t *hapnest_idx;  // global array
    const uint16_t *var_na;       // [V]
    const uint8_t *var_dep;       // [V]
    const uint32_t *allele_base;  // [V+1] prefix sums of var_na
    const uint32_t *nd_cluster;   // [nd_n]
    const uint32_t *nd_var_off;   // [nd_n + 1] absolute into nd_var
    const uint16_t *nd_var;       // global array
    const uint32_t *uniq0, *multi0;   // unique_kmer_indices / multicluster_kmer_indices as getHaplotypeCandidates built them
    uint8_t *shared_mult;         // group's shared multicluster records [num_shared][S]
    // state
    uint32_t *prng, *fprng;
    NormalState *fnd;
    uint32_t *uniq, *multi, *usub, *msub;
    uint8_t *smm;                 // sample_multicluster_kmer_multiplicities [nm][S]
    uint16_t *dip;                // [S][2]
    double *freq;
    uint32_t *obs;
    uint8_t *nz;
    uint32_t *zhdr, *zbkt, *phdr, *pbkt, *unext;
    uint32_t *hvcount;            // [H][V] scratch
    double *ucache;               // dense [S][Dc] or direct-mapped [cache_mask+1]
    uint32_t *ucache_tag;         // direct-mapped tags
    double *cum;                  // [max(D2, H, 1)] scratch
    uint16_t *nzlist;             // [H] scratch
    double *simplex;              // [H+1] scratch
    double *ksc;                  // kmer_stats_cache [S][2][V][4]
    uint8_t *ksc_upd;             // [S]
    uint32_t *dip_keys, *dip_freq;   // diplotype_sampling_frequencies: open addressing, key = (h1 | h2<<16) + 1... see dip_key()
    double *astats;               // allele_kmer_stats [S][A][3][4]
    uint8_t *nest_ploidy, *nest_n;   // NestedVariantClusterInfo of this vertex for the current sweep [S]
    double *nest_stats;           // [S][2][4]
    uint32_t *sc;                 // scalars (SC_*)
    uint32_t *edges;              // mutable out-edge list of this vertex (local vertex ids)
    uint8_t *cover_rows;          // [K] scratch (sparsity estimator)
    double sparsity;
    uint32_t is_sparse, pad;
};

struct GroupDev {
    uint32_t index, c0, nvert, nsrc, nshared, pad;
    uint32_t *sources;            // mutable
    const uint8_t *ploidy;        // [S]
    uint32_t *stack;              // [2*nvert] traversal stack
    uint32_t *brng;               // 625 words: branch-order generator
    uint32_t *trace;              // optional [max_sweeps][nvert][S]
};

// ---- Utils::logAddition (Utils.hpp:105-124) ----
__device__ inline double log_addition(double a, double b) {
    if (a < b) return b + log1p(exp(a - b));
    return a + log1p(exp(b - a));
}

// ---- KmerStats (KmerStats.cpp:51-63): ks = {count, fraction, mean, M2} ----
__device__ inline void ks_reset(double *ks) { ks[0] = 0; ks[1] = 0; ks[2] = 0; ks[3] = 0; }
__device__ inline void ks_add(double *ks, double value) {
    const double count = ks[0] + 1.0;
    ks[0] = count;
    // !doubleCompare(value, 0): value == 0 <=> equal (Utils.hpp:81-87 with b = 0)
    ks[1] += ((value == 0.0 ? 0.0 : 1.0) - ks[1]) / count;
    const double delta = value - ks[2];
    ks[2] += delta / count;
    ks[3] += delta * (value - ks[2]);
}
// AlleleKmerStats::addKmerStats (KmerStats.cpp:114-121): cell = [3][4]
__device__ inline void aks_add(double *cell, const double *ks) {
    ks_add(cell, ks[0]);                      // count_stats   <- (getCount(), true)
    if (ks[0] != 0.0) {
        ks_add(cell + 4, ks[1]);              // fraction_stats <- getFraction()  (skipped when count == 0)
        ks_add(cell + 8, ks[2]);              // mean_stats     <- getMean()
    }
}

__device__ inline double count_log_prob(const GParams &P, uint32_t s, uint8_t mult, uint8_t count) {   // CountDistribution.cpp:255-265
    if (mult == 0) return P.lut_n[s * 256u + count];
    return P.lut_g[((size_t)s * 256u + mult) * 256u + count];
}

// ---- VariantClusterHaplotypes multiplicity getters (VariantClusterHaplotypes.cpp:45-108), uchar arithmetic ----
__device__ inline uint8_t dip_mult(const ClusterDev *c, uint32_t k, uint16_t h1, uint16_t h2) {
    uint8_t m = 0;
    const uint8_t *row = c->M + (size_t)k * c->H;
    if (h1 != NOHAP) m = (uint8_t)(m + row[h1]);
    if (h2 != NOHAP) m = (uint8_t)(m + row[h2]);
    return m;
}
__device__ inline uint8_t unique_mult(const ClusterDev *c, uint32_t k, uint16_t h1, uint16_t h2, uint8_t gender) {
    uint8_t m = dip_mult(c, k, h1, h2);
    if (c->has_counts[k]) m = (uint8_t)(m + c->ic[2 * k + gender]);
    return m;
}
__device__ inline uint8_t multi_mult(const ClusterDev *c, const GParams &P, uint32_t k, uint16_t h1, uint16_t h2, uint16_t p1, uint16_t p2, uint32_t s) {
    const uint8_t icm = c->ic[2 * k + P.gender[s]];
    if (c->counts[(size_t)k * P.S + s] == 0) return (uint8_t)(dip_mult(c, k, h1, h2) + icm);
    const uint8_t shared = c->shared_mult[(size_t)c->shared_idx[k] * P.S + s];
    return (uint8_t)(shared - dip_mult(c, k, p1, p2) + dip_mult(c, k, h1, h2) + icm);
}

__device__ inline bool kv_bit(const ClusterDev *c, uint32_t e, uint32_t h) {
    return (c->kv_bits[(size_t)(e - c->kv_e0) * c->HW + (h >> 5)] >> (h & 31u)) & 1u;
}

// ---- unordered_set views ----
__device__ inline USet zero_set(const ClusterDev *c) { return USet{c->zhdr, c->zbkt, c->unext}; }
__device__ inline USet plus_set(const ClusterDev *c) { return USet{c->phdr, c->pbkt, c->unext}; }

// ---- FrequencyDistribution::reset / SparseFrequencyDistribution::reset (FrequencyDistribution.cpp:49-54,104-115) ----
__device__ inline void freq_reset(const ClusterDev *c) {
    const double f = 1 / (double)c->H;
    for (uint32_t h = 0; h < c->H; ++h) {
        c->obs[h] = 0;
        c->freq[h] = f;
        c->nz[h] = 1;
    }
    if (c->is_sparse) {
        uset_clear(plus_set(c));
        USet z = zero_set(c);
        uset_clear(z);
        for (uint32_t h = 0; h < c->H; ++h) uset_insert(z, h);
    }
}

// ---- SparsityEstimator::estimateMinimumColumnCover (SparsityEstimator.cpp:41-87), unweighted ----
// returns the cover size; uses `rng` (freshly seeded by the caller), c->cover_rows, c->obs (column sums), c->nzlist
__device__ inline uint32_t sparsity_cover(const ClusterDev *c, uint32_t *rng) {
    uint32_t remaining = 0;
    for (uint32_t k = 0; k < c->K; ++k) {
        c->cover_rows[k] = c->has_counts[k] ? 1 : 0;
        remaining += c->cover_rows[k];
    }
    uint32_t cover = 0;
    while (remaining > 0) {
        for (uint32_t h = 0; h < c->H; ++h) c->obs[h] = 0;
        for (uint32_t k = 0; k < c->K; ++k)
            if (c->cover_rows[k]) {
                const uint8_t *row = c->M + (size_t)k * c->H;
                for (uint32_t h = 0; h < c->H; ++h) c->obs[h] += row[h];
            }
        uint32_t best = 0;
        for (uint32_t h = 0; h < c->H; ++h) best = c->obs[h] > best ? c->obs[h] : best;
        if (best == 0) break;   // the reference asserts max_row_cover > 0
        uint32_t m = 0;
        for (uint32_t h = 0; h < c->H; ++h)
            if (c->obs[h] == best) c->nzlist[m++] = (uint16_t)h;
        // DiscreteSampler with outcomes 1,1,...: cum = 1..m; sample = upper_bound(cum, canonical * m) (DiscreteSampler.cpp:61-87)
        const double x = rng_canonical(rng) * (double)m;
        uint32_t pick = 0;
        if (m > 1) {
            while (pick < m && !((double)(pick + 1) > x)) ++pick;
            if (pick >= m) pick = m - 1;
        }
        const uint32_t col = c->nzlist[pick];
        ++cover;
        for (uint32_t k = 0; k < c->K; ++k)
            if (c->cover_rows[k] && c->M[(size_t)k * c->H + col] != 0) {
                c->cover_rows[k] = 0;
                --remaining;
            }
    }
    return cover;
}

// ---- VariantClusterGenotyper ctor (VariantClusterGenotyper.cpp:59-106) ----
__device__ inline void genotyper_construct(ClusterDev *c, const GParams &P, uint32_t prng_seed) {
    mt_seed(c->prng, prng_seed);
    for (uint32_t i = 0; i < SC_COUNT; ++i) c->sc[i] = 0;
    // a (re)built genotyper starts from the k-mer index lists in first-seen order (they are shuffled in place per chain)
    for (uint32_t i = 0; i < c->nu; ++i) c->uniq[i] = c->uniq0[i];
    for (uint32_t i = 0; i < c->nm; ++i) c->multi[i] = c->multi0[i];
    for (uint32_t s = 0; s < P.S; ++s) {
        c->dip[2 * s] = NOHAP;
        c->dip[2 * s + 1] = NOHAP;
        c->ksc_upd[s] = 1;
    }
    for (size_t i = 0; i < (size_t)P.S * c->A * 12; ++i) c->astats[i] = 0;
    for (size_t i = 0; i < (size_t)P.S * 2 * c->V * 4; ++i) c->ksc[i] = 0;
    for (uint32_t i = 0; i < c->dip_cap; ++i) c->dip_keys[i] = 0;
    for (size_t i = 0; i < (size_t)c->dip_cap * P.S; ++i) c->dip_freq[i] = 0;
    // SparsityEstimator(prng_seed), then (Sparse)FrequencyDistribution(.., prng_seed) with a fresh generator
    mt_seed(c->fprng, prng_seed);
    const uint32_t cover = sparsity_cover(c, c->fprng);
    mt_seed(c->fprng, prng_seed);
    c->fnd->saved = 0;
    c->fnd->available = 0;
    c->is_sparse = cover > 0 ? 1u : 0u;   // HaplotypeFrequencyDistribution.cpp:79-89
    if (c->is_sparse) {
        double sp = (double)cover / (double)c->H;
        const double cap = 1 - BT_DBL_EPS * 100;
        c->sparsity = sp < cap ? sp : cap;    // FrequencyDistribution.cpp:98
        uset_init(zero_set(c));
        uset_init(plus_set(c));
    }
    freq_reset(c);
    c->sc[SC_CONSTRUCTED] = 1;
}

__device__ inline void cache_clear(const ClusterDev *c, const GParams &P) {   // VariantClusterGenotyper::clearCache (:131-138)
    if (c->cache_mode == 0) {
        const size_t n = (size_t)P.S * c->Dc;
        for (size_t i = 0; i < n; ++i) c->ucache[i] = __longlong_as_double(0x7ff8000000000000LL);
    } else if (c->cache_mode == 1) {
        for (uint32_t i = 0; i <= c->cache_mask; ++i) c->ucache_tag[i] = 0;
    }
}

// ---- VariantClusterHaplotypes::sampleKmerSubset (+ isMaxHaplotypeVariantKmer) (VariantClusterHaplotypes.cpp:110-177) ----
__device__ inline bool is_max_hv_kmer(const ClusterDev *c, uint32_t k, uint32_t maxk) {
    bool is_max = true;
    for (uint32_t e = c->kv_off[k]; e < c->kv_off[k + 1]; ++e) {
        const uint32_t var = c->kv_var[e];
        for (uint32_t h = 0; h < c->H; ++h) {
            if (kv_bit(c, e, h)) {
                uint32_t *cnt = &c->hvcount[(size_t)h * c->V + var];
                if (*cnt < maxk) {
                    *cnt += 1;
                    is_max = false;
                }
            }
        }
    }
    return is_max;
}
__device__ inline void sample_kmer_subset(const ClusterDev *c, const GParams &P) {
    for (size_t i = 0; i < (size_t)c->H * c->V; ++i) c->hvcount[i] = 0;
    uint32_t nsu = 0, nsm = 0;
    rng_shuffle_u32(c->prng, c->uniq, c->nu);
    for (uint32_t i = 0; i < c->nu; ++i) {
        const uint32_t k = c->uniq[i];
        if (rng_bernoulli(c->prng, P.rate))
            if (!is_max_hv_kmer(c, k, P.max_hvk)) c->usub[nsu++] = k;
    }
    rng_shuffle_u32(c->prng, c->multi, c->nm);
    for (uint32_t i = 0; i < c->nm; ++i) {
        const uint32_t k = c->multi[i];
        if (rng_bernoulli(c->prng, P.rate))
            if (!is_max_hv_kmer(c, k, P.max_hvk)) c->msub[nsm++] = k;
    }
    c->sc[SC_NSUB_U] = nsu;
    c->sc[SC_NSUB_M] = nsm;
    for (size_t i = 0; i < (size_t)nsm * P.S; ++i) c->smm[i] = 0;
    for (uint32_t s = 0; s < P.S; ++s) c->ksc_upd[s] = 1;
}

// ---- VariantClusterGenotyper::reset (VariantClusterGenotyper.cpp:113-129) ----
__device__ inline void genotyper_reset(const ClusterDev *c, const GParams &P) {
    c->sc[SC_USE_MULTI] = 0;
    sample_kmer_subset(c, P);
    cache_clear(c, P);
    freq_reset(c);   // HaplotypeFrequencyDistribution::reset (counts are 0 here, as the reference asserts)
}

__device__ inline uint32_t dip_index(const ClusterDev *c, uint16_t h1, uint16_t h2) {
    if (h2 == NOHAP) return c->D2 + h1;
    return (uint32_t)h1 * c->H - ((uint32_t)h1 * ((uint32_t)h1 - 1u)) / 2u + ((uint32_t)h2 - (uint32_t)h1);
}

// unique part of calcDiplotypeLogProb with its per-(sample, diplotype) cache (VariantClusterGenotyper.cpp:619-643).
// The cached value is a pure function of (sample, diplotype, k-mer subset), so a dense table, a direct-mapped table
// or no table at all give bit-identical sums (same summation order).
__device__ inline double unique_log_prob(const ClusterDev *c, const GParams &P, uint32_t s, uint16_t h1, uint16_t h2) {
    const uint32_t idx = dip_index(c, h1, h2);
    uint32_t slot = 0;
    if (c->cache_mode == 0) {
        const double v = c->ucache[(size_t)s * c->Dc + idx];
        if (v == v) return v;
    } else if (c->cache_mode == 1) {
        const uint32_t key = s * c->Dc + idx + 1u;
        slot = (key * 2654435761u) & c->cache_mask;
        if (c->ucache_tag[slot] == key) return c->ucache[slot];
    }
    double acc = 0;
    const uint32_t n = c->sc[SC_NSUB_U];
    const uint8_t gender = P.gender[s];
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t k = c->usub[i];
        const uint8_t m = unique_mult(c, k, h1, h2, gender);
        const uint8_t cnt = c->has_counts[k] ? c->counts[(size_t)k * P.S + s] : 0;
        acc += count_log_prob(P, s, m, cnt);
    }
    if (c->cache_mode == 0) c->ucache[(size_t)s * c->Dc + idx] = acc;
    else if (c->cache_mode == 1) {
        c->ucache_tag[slot] = s * c->Dc + idx + 1u;
        c->ucache[slot] = acc;
    }
    return acc;
}

// multicluster part (VariantClusterGenotyper.cpp:647-661).  The reference keeps a second cache that it patches
// incrementally (:569-595); the patched value equals this direct sum up to floating-point re-association.
__device__ inline double multi_log_prob(const ClusterDev *c, const GParams &P, uint32_t s, uint16_t h1, uint16_t h2) {
    double acc = 0;
    const uint32_t n = c->sc[SC_NSUB_M];
    const uint16_t p1 = c->dip[2 * s], p2 = c->dip[2 * s + 1];
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t k = c->msub[i];
        const uint8_t m = multi_mult(c, P, k, h1, h2, p1, p2, s);
        acc += count_log_prob(P, s, m, c->counts[(size_t)k * P.S + s]);
    }
    return acc;
}

__device__ inline double diplotype_log_prob(const ClusterDev *c, const GParams &P, uint32_t s, uint16_t h1, uint16_t h2) {   // :597-666
    double lp = 0;
    if (h2 == NOHAP) lp += log(c->freq[h1]);
    else if (h1 == h2) lp += 2 * log(c->freq[h1]);
    else lp += BT_LN2 + log(c->freq[h1]) + log(c->freq[h2]);
    lp += unique_log_prob(c, P, s, h1, h2);
    if (c->sc[SC_USE_MULTI]) lp += multi_log_prob(c, P, s, h1, h2);
    return lp;
}

// ---- HaplotypeFrequencyDistribution::incrementCount (HaplotypeFrequencyDistribution.cpp:113-125) ----
__device__ inline void hfd_increment(const ClusterDev *c, uint16_t h) {
    if (h == NOHAP) {
        c->sc[SC_MISSING_COUNT] += 1;
        return;
    }
    c->sc[SC_HAP_COUNT] += 1;
    if (c->is_sparse && c->obs[h] == 0) {   // SparseFrequencyDistribution::incrementObservationCount (:198-207)
        // the reference inserts into plus, then erases from zero; the sets share their `next` words here, so leave
        // the zero list first (the resulting containers are identical)
        uset_erase(zero_set(c), h);
        uset_insert(plus_set(c), h);
    }
    c->obs[h] += 1;
}

// ---- diplotype_sampling_frequencies (VariantClusterGenotyper.cpp:692-696) as an open-addressing table ----
__device__ inline void dip_table_add(const ClusterDev *c, const GParams &P, uint16_t h1, uint16_t h2, uint32_t s) {
    const uint32_t key = ((uint32_t)h1 | ((uint32_t)h2 << 16));
    // stored tag: key + 1 (0 = empty slot); the null diplotype (NOHAP, NOHAP) would wrap to 0 and is stored as 0xFFFFFFFF,
    // which no other key + 1 can equal because haplotype indices are < 0xFFFE
    const uint32_t want = key == 0xFFFFFFFFu ? 0xFFFFFFFFu : key + 1u;
    const uint32_t mask = c->dip_cap - 1u;
    uint32_t slot = (key * 2654435761u) & mask;
    for (uint32_t probes = 0; probes < c->dip_cap; ++probes) {
        const uint32_t tag = c->dip_keys[slot];
        if (tag == 0) {
            c->dip_keys[slot] = want;
            c->sc[SC_DIP_ENTRIES] += 1;
            c->dip_freq[(size_t)slot * P.S + s] += 1;
            return;
        }
        if (tag == want) {
            c->dip_freq[(size_t)slot * P.S + s] += 1;
            return;
        }
        slot = (slot + 1u) & mask;
    }
    c->sc[SC_DIP_OVERFLOW] = 1;
}

// ---- VariantClusterHaplotypes::updateMulticlusterKmerMultiplicities (VariantClusterHaplotypes.cpp:197-233) ----
__device__ inline void update_multicluster_multiplicities(const ClusterDev *c, const GParams &P, uint16_t h1, uint16_t h2, uint16_t p1, uint16_t p2, uint32_t s) {
    if (h1 != p1 || h2 != p2) {
        c->ksc_upd[s] = 1;
        for (uint32_t i = 0; i < c->nm; ++i) {
            const uint32_t k = c->multi[i];
            const uint8_t cur = dip_mult(c, k, h1, h2), pre = dip_mult(c, k, p1, p2);
            if (cur != pre) {
                uint8_t *m = &c->shared_mult[(size_t)c->shared_idx[k] * P.S + s];
                *m = (uint8_t)(*m - pre);
                *m = (uint8_t)(*m + cur);
            }
        }
    }
    const uint32_t nsm = c->sc[SC_NSUB_M];
    for (uint32_t sub = 0; sub < nsm; ++sub) {
        const uint32_t k = c->msub[sub];
        const uint8_t shared = c->shared_mult[(size_t)c->shared_idx[k] * P.S + s];
        if (dip_mult(c, k, h1, h2) > 0 && c->counts[(size_t)k * P.S + s] > 0 && shared != c->smm[(size_t)sub * P.S + s]) c->ksc_upd[s] = 1;
        c->smm[(size_t)sub * P.S + s] = shared;
    }
}

// ---- updateKmerStatsCache / updateAlleleKmerStats (VariantClusterHaplotypes.cpp:235-372) ----
__device__ inline double *ksc_slot(const ClusterDev *c, uint32_t s, uint32_t which, uint32_t v) { return c->ksc + (((size_t)s * 2 + which) * c->V + v) * 4; }

__device__ inline void update_kmer_stats_cache(const ClusterDev *c, const GParams &P, uint32_t k, uint16_t h1, uint16_t h2, uint32_t s, uint8_t mult) {
    double kmer_count = 0;
    if (c->has_counts[k]) kmer_count = c->counts[(size_t)k * P.S + s] / (double)mult;
    for (uint32_t e = c->kv_off[k]; e < c->kv_off[k + 1]; ++e) {
        const uint32_t var = c->kv_var[e];
        if (kv_bit(c, e, h1)) ks_add(ksc_slot(c, s, 0, var), kmer_count);
        if (h2 != NOHAP && kv_bit(c, e, h2)) ks_add(ksc_slot(c, s, 1, var), kmer_count);
    }
}
__device__ inline double *astats_cell(const ClusterDev *c, uint32_t s, uint32_t v, uint32_t a) { return c->astats + ((size_t)s * c->A + c->allele_base[v] + a) * 12; }
__device__ inline bool is_missing(const ClusterDev *c, uint32_t v, uint32_t a) { return c->var_dep[v] && a == (uint32_t)c->var_na[v] - 1u; }

__device__ inline void add_haplotype_kmer_stats(const ClusterDev *c, uint32_t s, uint32_t which, uint16_t h) {   // :332-358
    uint32_t last_non_missing = 0xFFFFFFFFu;
    for (uint32_t v = 0; v < c->V; ++v) {
        const uint32_t a = c->hap_allele[(size_t)h * c->V + v];
        if (is_missing(c, v, a)) {
            if (last_non_missing != 0xFFFFFFFFu) aks_add(astats_cell(c, s, v, a), ksc_slot(c, s, which, last_non_missing));
        } else {
            aks_add(astats_cell(c, s, v, a), ksc_slot(c, s, which, v));
            last_non_missing = v;
        }
    }
}

__device__ inline void update_allele_kmer_stats(const ClusterDev *c, const GParams &P) {   // :235-298
    for (uint32_t s = 0; s < P.S; ++s) {
        const uint16_t h1 = c->dip[2 * s], h2 = c->dip[2 * s + 1];
        if (c->ksc_upd[s]) {
            c->ksc_upd[s] = 0;
            for (uint32_t v = 0; v < c->V; ++v) {
                ks_reset(ksc_slot(c, s, 0, v));
                ks_reset(ksc_slot(c, s, 1, v));
            }
            if (h1 != NOHAP) {
                const uint32_t nsu = c->sc[SC_NSUB_U], nsm = c->sc[SC_NSUB_M];
                for (uint32_t i = 0; i < nsu; ++i) {
                    const uint32_t k = c->usub[i];
                    if (dip_mult(c, k, h1, h2) > 0) update_kmer_stats_cache(c, P, k, h1, h2, s, unique_mult(c, k, h1, h2, P.gender[s]));
                }
                for (uint32_t i = 0; i < nsm; ++i) {
                    const uint32_t k = c->msub[i];
                    if (dip_mult(c, k, h1, h2) > 0) update_kmer_stats_cache(c, P, k, h1, h2, s, multi_mult(c, P, k, h1, h2, h1, h2, s));
                }
            }
        }
        if (h1 != NOHAP) add_haplotype_kmer_stats(c, s, 0, h1);
        if (h2 != NOHAP) add_haplotype_kmer_stats(c, s, 1, h2);
        const uint32_t nn = c->nest_n[s];
        for (uint32_t j = 0; j < nn; ++j)   // addNestedHaplotypeKmerStats (:360-372)
            for (uint32_t v = 0; v < c->V; ++v) aks_add(astats_cell(c, s, v, (uint32_t)c->var_na[v] - 1u), c->nest_stats + ((size_t)s * 2 + j) * 4);
    }
}

// ---- sampleDiplotypes / sampleDiplotype (VariantClusterGenotyper.cpp:668-755) ----
__device__ inline void sample_diplotypes(const ClusterDev *c, const GParams &P, bool collect, uint32_t *trace_row) {
    uint32_t nnz = 0;
    for (uint32_t h = 0; h < c->H; ++h)
        if (c->nz[h]) c->nzlist[nnz++] = (uint16_t)h;
    for (uint32_t s = 0; s < P.S; ++s) {
        const uint16_t p1 = c->dip[2 * s], p2 = c->dip[2 * s + 1];
        const uint8_t ploidy = c->nest_ploidy[s];
        // candidates in the reference's order; cumulative log-sum-exp exactly as LogDiscreteSampler::addOutcome
        uint32_t ncand = 0;
        double run = 0;
        if (ploidy == 2) {
            for (uint32_t a = 0; a < nnz; ++a)
                for (uint32_t b = a; b < nnz; ++b) {
                    const double lp = diplotype_log_prob(c, P, s, c->nzlist[a], c->nzlist[b]);
                    run = ncand == 0 ? lp : log_addition(lp, run);
                    c->cum[ncand++] = run;
                }
        } else if (ploidy == 1) {
            for (uint32_t a = 0; a < nnz; ++a) {
                const double lp = diplotype_log_prob(c, P, s, c->nzlist[a], NOHAP);
                run = ncand == 0 ? lp : log_addition(lp, run);
                c->cum[ncand++] = run;
            }
        } else {
            c->cum[0] = 0;
            ncand = 1;
            run = 0;
        }
        // LogDiscreteSampler::sample (DiscreteSampler.cpp:120-125): the draw happens even for a single outcome
        const double u = log(rng_canonical(c->prng)) + run;
        uint32_t pick = 0;
        if (ncand > 1) {
            uint32_t lo = 0, hi = ncand;   // upper_bound: first index with cum > u
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (u < c->cum[mid]) hi = mid;
                else lo = mid + 1;
            }
            pick = lo < ncand ? lo : ncand - 1;
        }
        uint16_t h1 = NOHAP, h2 = NOHAP;
        if (ploidy == 2) {
            // invert the (a, b >= a) enumeration
            uint32_t a = 0, rem = pick;
            while (rem >= nnz - a) {
                rem -= nnz - a;
                ++a;
            }
            h1 = c->nzlist[a];
            h2 = c->nzlist[a + rem];
        } else if (ploidy == 1) {
            h1 = c->nzlist[pick];
        }
        c->dip[2 * s] = h1;
        c->dip[2 * s + 1] = h2;
        hfd_increment(c, h1);
        hfd_increment(c, h2);
        update_multicluster_multiplicities(c, P, h1, h2, p1, p2, s);
        if (trace_row) trace_row[s] = (uint32_t)h1 | ((uint32_t)h2 << 16);
        if (collect) dip_table_add(c, P, h1, h2, s);
    }
    if (collect) update_allele_kmer_stats(c, P);
    c->sc[SC_USE_MULTI] = c->sc[SC_NSUB_M] != 0 ? 1u : 0u;
}

// ---- SparseFrequencyDistribution::updateCachedSimplexProbVector (FrequencyDistribution.cpp:143-196) -> c->simplex, returns length ----
__device__ inline uint32_t simplex_prob_vector(const ClusterDev *c, uint32_t total_obs, uint32_t plus_size) {
    const uint32_t Hn = c->H;
    const double sparsity = c->sparsity;
    double prob_z_log = plus_size * log(sparsity) + (Hn - plus_size) * log(1 - sparsity);
    double prob_t_log = lgamma(plus_size * 1.0) - lgamma(total_obs + plus_size * 1.0);
    double prob_eq_z_log = 0.0 + prob_z_log + prob_t_log;
    double row_sum = prob_eq_z_log;
    uint32_t n = 0;
    c->simplex[n++] = row_sum;
    for (uint32_t j = plus_size + 1; j < Hn + 1; ++j) {
        const double cardinal = lgamma((double)(Hn - plus_size + 1)) - (lgamma((double)(j - plus_size + 1)) + lgamma((double)(Hn - j + 1)));
        prob_z_log = j * log(sparsity) + (Hn - j) * log(1 - sparsity);
        prob_t_log = lgamma(j * 1.0) - lgamma(total_obs + j * 1.0);
        prob_eq_z_log = cardinal + prob_z_log + prob_t_log;
        row_sum += log(1 + exp(prob_eq_z_log - row_sum));
        c->simplex[n++] = row_sum;
        const double a = c->simplex[n - 1], b = c->simplex[n - 2];
        const double mn = a < b ? a : b;
        if (a == b || fabs(a - b) < fabs(mn) * BT_DBL_EPS * 100) break;   // Utils::doubleCompare
    }
    for (uint32_t i = 0; i < n; ++i) c->simplex[i] = exp(c->simplex[i] - row_sum);
    return n;
}

// ---- sampleHaplotypeFrequencies (VariantClusterGenotyper.cpp:781-785 -> HaplotypeFrequencyDistribution.cpp:127-138
//      -> FrequencyDistribution.cpp:75-93 / 209-303) ----
__device__ inline void sample_haplotype_frequencies(const ClusterDev *c) {
    const uint32_t n_obs = c->sc[SC_HAP_COUNT];
    if (n_obs > 0) {
        if (!c->is_sparse) {
            double norm = 0;
            for (uint32_t h = 0; h < c->H; ++h) {
                const double f = rng_gamma(c->fprng, c->fnd, (double)(c->obs[h] + 1u), 1.0);
                c->freq[h] = f;
                norm += f;
                c->obs[h] = 0;
            }
            for (uint32_t h = 0; h < c->H; ++h) c->freq[h] /= norm;
        } else {
            USet plus = plus_set(c), zero = zero_set(c);
            const uint32_t plus_size = uset_size(plus);
            const uint32_t len = simplex_prob_vector(c, n_obs, plus_size);
            const double u = rng_canonical(c->fprng);
            uint32_t ub = 0;
            while (ub < len && !(u < c->simplex[ub])) ++ub;   // upper_bound over a non-decreasing vector
            const uint32_t simplex_size = ub + plus_size;
            double norm = 0;
            for (uint32_t e = uset_begin(plus); e != US_NONE; e = c->unext[e]) {
                const double f = rng_gamma(c->fprng, c->fnd, (double)c->obs[e] + 1.0, 1.0);
                c->freq[e] = f;
                norm += f;
                c->nz[e] = 1;
            }
            while (uset_size(plus) < simplex_size) {
                const uint32_t pos = rng_uniform_int(c->fprng, uset_size(zero));   // uniform_int(0, |zero| - 1)
                uint32_t e = uset_begin(zero);
                for (uint32_t i = 0; i < pos; ++i) e = c->unext[e];
                const double f = rng_gamma(c->fprng, c->fnd, 1.0, 1.0);
                c->freq[e] = f;
                norm += f;
                c->nz[e] = 1;
                // the two sets share the `next` words: leave the zero list before entering the plus list
                uset_erase(zero, e);
                uset_insert(plus, e);
            }
            for (uint32_t e = uset_begin(zero); e != US_NONE; e = c->unext[e]) {
                c->freq[e] = 0;
                c->nz[e] = 0;
                c->obs[e] = 0;
            }
            // "for p in plus: freq /= norm; zero.insert(p); obs = 0" then plus.clear(): record the plus iteration order
            // first (shared `next` words), clear plus, then insert into zero in that order — same final containers
            uint32_t np = 0;
            for (uint32_t e = uset_begin(plus); e != US_NONE; e = c->unext[e]) c->nzlist[np++] = (uint16_t)e;
            uset_clear(plus);
            for (uint32_t i = 0; i < np; ++i) {
                const uint32_t e = c->nzlist[i];
                c->freq[e] /= norm;
                uset_insert(zero, e);
                c->obs[e] = 0;
            }
        }
    }
    c->sc[SC_HAP_COUNT] = 0;
    c->sc[SC_MISSING_COUNT] = 0;
}

}  // namespace bt
