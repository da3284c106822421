// libbtgpu: Gibbs genotyping of variant-cluster groups (host side of bt_gibbs_* + the kernels).
//
//   bt_gibbs_init_chain  <- VariantClusterGroup::initGenotyper + shuffleBranchOrdering (VariantClusterGroup.cpp:171-218)
//   bt_gibbs_sweep       <- VariantClusterGroup::estimateGenotypes / runGibbsSample (VariantClusterGroup.cpp:220-250)
//   bt_gibbs_run         <- InferenceEngine::estimateGenotypesCallback inner loops (InferenceEngine.cpp:290-306)
//   bt_gibbs_noise_counts<- VariantClusterGroup::getNoiseCounts + clearGenotyperCache (InferenceEngine.cpp:90-92)
//
// Parallelisation: variant-cluster groups are the reference's unit of independence (one std::thread works a group at a
// time).  Here one LANE owns one group for a whole launch: the sweep is a strictly sequential chain of dependent random
// draws (samples within a sweep, sweeps within a chain, chains within a genotyper), so the parallel axis is the group
// index, and a launch carries 10^5-10^6 groups.  All state is in HBM (ClusterDev); workgroups are one wavefront wide so
// that divergence between groups costs only within a wave and small batches still spread over all 256 CUs.
#include "bt_gibbs_device.hpp"
#include "bt_internal.hpp"

#include <algorithm>
#include <cstring>

using namespace bt;

namespace {

constexpr unsigned GIBBS_BLOCK = 64;

enum GibbsOp { OP_RUN = 0, OP_INIT_CHAIN = 1, OP_SWEEP = 2, OP_NOISE = 3, OP_RESET = 4 };

struct TraceCfg {
    uint32_t max_sweeps;   // 0 = off
    uint32_t *counter;     // [G] sweeps recorded per group
};

__device__ inline void group_init_chain(const GroupDev *g, ClusterDev *cl, const GParams &P, uint32_t chain) {
    const uint32_t gseed = P.noise_seeding ? P.seed + (g->index + 1u) * (chain + 1u) : P.seed + (g->index + 1u);
    for (uint32_t v = 0; v < g->nvert; ++v) {
        ClusterDev *c = &cl[g->c0 + v];
        if (!c->sc[SC_CONSTRUCTED]) genotyper_construct(c, P, gseed + c->cid);   // VariantClusterGroup.cpp:179-182
        genotyper_reset(c, P);
    }
    // shuffleBranchOrdering (VariantClusterGroup.cpp:208-218)
    mt_seed(g->brng, P.seed + (g->index + 1u) * (chain + 1u));
    rng_shuffle_u32(g->brng, g->sources, g->nsrc);
    for (uint32_t v = 0; v < g->nvert; ++v) rng_shuffle_u32(g->brng, cl[g->c0 + v].edges, cl[g->c0 + v].ne);
}

// VariantClusterGenotyper::updateNestedVariantClusterInfo (VariantClusterGenotyper.cpp:140-206): child's info := parent's info, updated
__device__ inline void prepare_nested(const ClusterDev *c, const ClusterDev *cc, const GParams &P) {
    for (uint32_t s = 0; s < P.S; ++s) {
        uint8_t ploidy = c->nest_ploidy[s];
        uint32_t n = c->nest_n[s];
        for (uint32_t j = 0; j < n; ++j)
            for (int q = 0; q < 4; ++q) cc->nest_stats[((size_t)s * 2 + j) * 4 + q] = c->nest_stats[((size_t)s * 2 + j) * 4 + q];
        for (uint32_t which = 0; which < 2; ++which) {
            const uint16_t h = c->dip[2 * s + which];
            if (h == NOHAP) continue;
            // binary_search(nested_variant_cluster_indices of h, child cluster idx)
            bool found = false;
            uint32_t lo = c->hapnest_off[h], hi = c->hapnest_off[h + 1];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t val = c->hapnest_idx[mid];
                if (val == cc->cid) {
                    found = true;
                    break;
                }
                if (val < cc->cid) lo = mid + 1;
                else hi = mid;
            }
            if (found) continue;
            ploidy = ploidy == 2 ? 1 : 0;   // updateNestedPloidy
            uint32_t variant_idx = 0xFFFFFFFFu;
            for (uint32_t d = 0; d < c->nd_n; ++d) {
                if (c->nd_cluster[d] != cc->cid) continue;
                for (uint32_t i = c->nd_var_off[d]; i < c->nd_var_off[d + 1]; ++i) {
                    const uint32_t nv = c->nd_var[i];
                    const uint32_t a = c->hap_allele[(size_t)h * c->V + nv];
                    if (!is_missing(c, nv, a)) {
                        variant_idx = nv;
                        break;
                    }
                }
                break;
            }
            if (variant_idx != 0xFFFFFFFFu && n < 2) {
                const double *src = ksc_slot(c, s, which, variant_idx);
                for (int q = 0; q < 4; ++q) cc->nest_stats[((size_t)s * 2 + n) * 4 + q] = src[q];
                ++n;
            }
        }
        cc->nest_ploidy[s] = ploidy;
        cc->nest_n[s] = (uint8_t)n;
    }
}

__device__ inline void visit_vertex(const GroupDev *g, ClusterDev *cl, const GParams &P, uint32_t v, bool collect, uint32_t *trace_row) {
    const ClusterDev *c = &cl[g->c0 + v];
    sample_diplotypes(c, P, collect, trace_row ? trace_row + (size_t)v * P.S : nullptr);
    sample_haplotype_frequencies(c);
}

// VariantClusterGroup::estimateGenotypes + runGibbsSample (VariantClusterGroup.cpp:220-250), recursion unrolled on an explicit stack
__device__ inline void group_sweep(const GroupDev *g, ClusterDev *cl, const GParams &P, bool collect, uint32_t *trace_row) {
    if (trace_row)
        for (uint32_t i = 0; i < g->nvert * P.S; ++i) trace_row[i] = 0xFFFFFFFFu;
    for (uint32_t si = 0; si < g->nsrc; ++si) {
        const uint32_t sv = g->sources[si];
        const ClusterDev *root = &cl[g->c0 + sv];
        for (uint32_t s = 0; s < P.S; ++s) {
            root->nest_ploidy[s] = g->ploidy[s];
            root->nest_n[s] = 0;
        }
        visit_vertex(g, cl, P, sv, collect, trace_row);
        uint32_t sp = 0;
        g->stack[0] = sv;
        g->stack[1] = 0;
        sp = 1;
        while (sp > 0) {
            const uint32_t v = g->stack[2 * (sp - 1)];
            const uint32_t i = g->stack[2 * (sp - 1) + 1];
            const ClusterDev *c = &cl[g->c0 + v];
            if (i < c->ne) {
                g->stack[2 * (sp - 1) + 1] = i + 1;
                const uint32_t t = c->edges[i];
                prepare_nested(c, &cl[g->c0 + t], P);
                visit_vertex(g, cl, P, t, collect, trace_row);
                g->stack[2 * sp] = t;
                g->stack[2 * sp + 1] = 0;
                ++sp;
            } else
                --sp;
        }
    }
}

__device__ inline uint32_t *trace_row_for(const GroupDev *g, const GParams &P, const TraceCfg &tr, uint32_t gi) {
    if (!tr.max_sweeps || !g->trace) return nullptr;
    const uint32_t n = tr.counter[gi];
    if (n >= tr.max_sweeps) return nullptr;
    tr.counter[gi] = n + 1;
    return g->trace + (size_t)n * g->nvert * P.S;
}

__global__ __launch_bounds__(GIBBS_BLOCK) void gibbs_kernel(const GroupDev *__restrict__ groups, ClusterDev *__restrict__ clusters, uint32_t num_groups,
                                                            GParams P, int op, uint32_t arg0, uint32_t arg1, unsigned long long *__restrict__ hist, TraceCfg tr) {
    const uint32_t gi = blockIdx.x * GIBBS_BLOCK + threadIdx.x;
    if (gi >= num_groups) return;
    const GroupDev *g = &groups[gi];
    if (op == OP_RUN) {
        for (uint32_t chain = 0; chain < P.num_chains; ++chain) {
            group_init_chain(g, clusters, P, chain);
            for (uint32_t i = 0; i < P.burn_in; ++i) group_sweep(g, clusters, P, false, trace_row_for(g, P, tr, gi));
            for (uint32_t i = 0; i < P.num_iterations; ++i) group_sweep(g, clusters, P, true, trace_row_for(g, P, tr, gi));
        }
    } else if (op == OP_INIT_CHAIN) {
        group_init_chain(g, clusters, P, arg0);
    } else if (op == OP_SWEEP) {
        for (uint32_t i = 0; i < arg0; ++i) group_sweep(g, clusters, P, arg1 != 0, trace_row_for(g, P, tr, gi));
    } else if (op == OP_NOISE) {
        // VariantClusterGenotyper::getNoiseCounts (:757-779) for every vertex, then clearCache
        for (uint32_t v = 0; v < g->nvert; ++v) {
            const ClusterDev *c = &clusters[g->c0 + v];
            const uint32_t nsu = c->sc[SC_NSUB_U];
            for (uint32_t s = 0; s < P.S; ++s) {
                const uint16_t h1 = c->dip[2 * s], h2 = c->dip[2 * s + 1];
                for (uint32_t i = 0; i < nsu; ++i) {
                    const uint32_t k = c->usub[i];
                    if (unique_mult(c, k, h1, h2, P.gender[s]) == 0) {
                        const uint32_t cnt = c->has_counts[k] ? c->counts[(size_t)k * P.S + s] : 0;
                        atomicAdd(&hist[s * 256u + cnt], 1ULL);
                    }
                }
            }
            cache_clear(c, P);
        }
    } else if (op == OP_RESET) {
        // VariantClusterGroup::resetGroup: genotypers are deleted; the shared KmerCounts multiplicities are NOT reset
        for (uint32_t v = 0; v < g->nvert; ++v) clusters[g->c0 + v].sc[SC_CONSTRUCTED] = 0;
    }
}

// per (cluster, sample): most frequently sampled diplotype and its frequency -> the compact posterior summary that is
// gathered to rank 0; ties resolve to the smallest (h1, h2)
__global__ __launch_bounds__(256) void summary_kernel(const ClusterDev *__restrict__ clusters, uint32_t num_clusters, uint32_t S, uint32_t *__restrict__ out) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= num_clusters) return;
    const ClusterDev *d = &clusters[c];
    for (uint32_t s = 0; s < S; ++s) {
        uint32_t best_key = 0xFFFFFFFFu, best = 0;
        for (uint32_t slot = 0; slot < d->dip_cap; ++slot) {
            const uint32_t tag = d->dip_keys[slot];
            if (!tag) continue;
            const uint32_t key = tag == 0xFFFFFFFFu ? 0xFFFFFFFFu : tag - 1u;
            const uint32_t f = d->dip_freq[(size_t)slot * S + s];
            const uint32_t kk = (key << 16) | (key >> 16);   // order by (h1, h2)
            const uint32_t bk = (best_key << 16) | (best_key >> 16);
            if (f > best || (f == best && f > 0 && kk < bk)) {
                best = f;
                best_key = key;
            }
        }
        out[((size_t)c * S + s) * 2] = best_key;
        out[((size_t)c * S + s) * 2 + 1] = best;
    }
}

struct PoolPlan {
    uint64_t size = 0;
    uint64_t take(uint64_t bytes, uint64_t align = 8) {
        size = (size + align - 1) / align * align;
        uint64_t off = size;
        size += bytes;
        return off;
    }
};

template <typename T>
int upload(bt_ctx *ctx, const T *h, uint64_t n, T **d, uint64_t *total) {
    const uint64_t bytes = std::max<uint64_t>(n, 1) * sizeof(T);
    BT_HIP(hipMalloc(reinterpret_cast<void **>(d), bytes));
    if (n) BT_HIP(hipMemcpyAsync(*d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *total += bytes;
    return BT_OK;
}

}  // namespace

struct bt_gibbs {
    bt_ctx *ctx = nullptr;
    GParams P{};
    uint32_t G = 0, C = 0, S = 0;
    std::vector<void *> allocs;          // every device allocation, freed in destroy
    uint64_t device_bytes = 0;
    GroupDev *d_groups = nullptr;
    ClusterDev *d_clusters = nullptr;
    uint8_t *d_pool = nullptr;
    uint64_t pool_bytes = 0;
    double *d_lut_g = nullptr, *d_lut_n = nullptr;
    bool lut_set = false;
    // host copies needed to fetch results
    std::vector<ClusterDev> h_clusters;
    std::vector<uint32_t> h_A;           // alleles per cluster
    // trace
    uint32_t trace_sweeps = 0;
    uint32_t *d_trace = nullptr, *d_trace_counter = nullptr;
    std::vector<uint64_t> trace_off;     // per group word offset
    uint64_t trace_words = 0;
    std::vector<uint32_t> h_nvert;
};

namespace {

int launch(bt_gibbs *g, int op, uint32_t a0, uint32_t a1, unsigned long long *hist) {
    if (!g->lut_set && (op == OP_RUN || op == OP_SWEEP)) return fail("bt_gibbs: count-model LUTs not set (bt_gibbs_set_lut)");
    BT_HIP(hipSetDevice(g->ctx->device));
    TraceCfg tr{g->trace_sweeps, g->d_trace_counter};
    hipLaunchKernelGGL(gibbs_kernel, dim3((g->G + GIBBS_BLOCK - 1) / GIBBS_BLOCK), dim3(GIBBS_BLOCK), 0, g->ctx->stream, g->d_groups, g->d_clusters, g->G,
                       g->P, op, a0, a1, hist, tr);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

}  // namespace

extern "C" {

int bt_gibbs_create(bt_ctx *ctx, const bt_gibbs_params *params, const bt_gibbs_batch *B, bt_gibbs **out) {
    if (!ctx || !params || !B || !out) return fail("bt_gibbs_create: null argument");
    const uint32_t S = params->num_samples, G = B->num_groups, C = B->num_clusters;
    if (S < 1 || S > 30) return fail("bt_gibbs_create: number of samples must be in 1..30");   // main.cpp:72
    if (!params->gender) return fail("bt_gibbs_create: gender array missing");
    if (G == 0 || C == 0) return fail("bt_gibbs_create: empty batch");
    if (B->group_cluster_off[G] != C) return fail("bt_gibbs_create: group_cluster_off[G] != num_clusters");
    BT_HIP(hipSetDevice(ctx->device));
    bt_gibbs *g = new bt_gibbs();
    g->ctx = ctx;
    g->G = G;
    g->C = C;
    g->S = S;
    GParams &P = g->P;
    P.S = S;
    P.seed = params->seed;
    P.num_chains = params->num_chains;
    P.burn_in = params->burn_in;
    P.num_iterations = params->num_iterations;
    P.max_hvk = params->max_haplotype_variant_kmers;
    P.noise_seeding = params->noise_seeding;
    P.rate = (double)params->kmer_subsampling_rate;
    for (uint32_t s = 0; s < S; ++s) P.gender[s] = params->gender[s] ? 1 : 0;

#define BT_TRY(x)                \
    do {                         \
        int _rc = (x);           \
        if (_rc != BT_OK) {      \
            bt_gibbs_destroy(g); \
            return _rc;          \
        }                        \
    } while (0)
#define UP(field, T, n)                                                 \
    T *d_##field = nullptr;                                             \
    BT_TRY(upload<T>(ctx, B->field, (n), &d_##field, &g->device_bytes)); \
    g->allocs.push_back(d_##field)

    // ---- sizes of the flat input arrays ----
    const uint64_t R = B->kmer_off[C];
    const uint64_t NNZ = B->kv_off[R];
    uint64_t sumH = 0, sumV = 0, multBytes = 0, kvWords = 0, hapvar = 0, nShared = 0;
    std::vector<uint64_t> mult_off(C + 1, 0), kvb_off(C + 1, 0), hapvar_off(C + 1, 0);
    std::vector<uint32_t> hap_base(C + 1, 0), var_base(C + 1, 0);
    for (uint32_t c = 0; c < C; ++c) {
        const uint32_t H = B->num_haplotypes[c], V = B->num_variants[c], K = B->kmer_off[c + 1] - B->kmer_off[c];
        if (H < 1 || H >= 65535 || V < 1) {
            bt_gibbs_destroy(g);
            return fail("bt_gibbs_create: cluster with no haplotype / variant or too many haplotypes");
        }
        const uint32_t nnz = B->kv_off[B->kmer_off[c + 1]] - B->kv_off[B->kmer_off[c]];
        mult_off[c + 1] = mult_off[c] + (uint64_t)K * H;
        kvb_off[c + 1] = kvb_off[c] + (uint64_t)nnz * ((H + 31) / 32);
        hapvar_off[c + 1] = hapvar_off[c] + (uint64_t)H * V;
        hap_base[c + 1] = hap_base[c] + H;
        var_base[c + 1] = var_base[c] + V;
    }
    sumH = hap_base[C];
    sumV = var_base[C];
    multBytes = mult_off[C];
    kvWords = kvb_off[C];
    hapvar = hapvar_off[C];
    for (uint32_t gi = 0; gi < G; ++gi) nShared += B->group_num_shared[gi];
    const uint64_t nEdges = B->edge_off[C], nSrc = B->group_source_off[G];
    const uint64_t nND = B->nestdep_off[C];

    UP(hap_kmer_mult, uint8_t, multBytes);
    UP(kmer_has_counts, uint8_t, R);
    UP(kmer_counts, uint8_t, R * S);
    UP(kmer_ic_mult, uint8_t, R * 2);
    UP(kmer_shared, int32_t, R);
    UP(kv_off, uint32_t, R + 1);
    UP(kv_var, uint16_t, NNZ);
    UP(kv_bits, uint32_t, kvWords);
    UP(hap_allele, uint16_t, hapvar);
    UP(hapnest_off, uint32_t, sumH + 1);
    UP(hapnest_idx, uint32_t, B->hapnest_off[sumH]);
    UP(var_num_alleles, uint16_t, sumV);
    UP(var_has_dependency, uint8_t, sumV);
    UP(nestdep_cluster, uint32_t, nND);
    UP(nestdep_var_off, uint32_t, nND + 1);
    UP(nestdep_var, uint16_t, B->nestdep_var_off[nND]);
    UP(group_ploidy, uint8_t, (uint64_t)G * S);
    UP(unique_idx, uint32_t, B->unique_off[C]);
    UP(multi_idx, uint32_t, B->multi_off[C]);

    // ---- plan the state pool ----
    PoolPlan plan;
    std::vector<ClusterDev> &hc = g->h_clusters;
    hc.assign(C, ClusterDev{});
    g->h_A.assign(C, 0);
    struct Offs {
        uint64_t allele_base, prng, fprng, fnd, uniq, multi, usub, msub, smm, dip, freq, obs, nz, zhdr, zbkt, phdr, pbkt, unext, hvcount, ucache, ucache_tag,
            cum, nzlist, simplex, ksc, ksc_upd, dip_keys, dip_freq, astats, nest_ploidy, nest_n, nest_stats, sc, edges, cover_rows;
    };
    std::vector<Offs> offs(C);
    const uint64_t collect_total = (uint64_t)std::max<uint32_t>(params->num_chains, 1) * std::max<uint32_t>(params->num_iterations, 1) * S;
    for (uint32_t c = 0; c < C; ++c) {
        ClusterDev &d = hc[c];
        Offs &o = offs[c];
        d.H = B->num_haplotypes[c];
        d.V = B->num_variants[c];
        d.K = B->kmer_off[c + 1] - B->kmer_off[c];
        d.HW = (d.H + 31) / 32;
        d.nu = B->unique_off[c + 1] - B->unique_off[c];
        d.nm = B->multi_off[c + 1] - B->multi_off[c];
        d.cid = B->cluster_idx[c];
        d.D2 = d.H * (d.H + 1) / 2;
        d.Dc = d.D2 + d.H;
        d.nd_n = B->nestdep_off[c + 1] - B->nestdep_off[c];
        d.ne = B->edge_off[c + 1] - B->edge_off[c];
        d.kv_e0 = B->kv_off[B->kmer_off[c]];
        uint32_t A = 0;
        for (uint32_t v = 0; v < d.V; ++v) A += B->var_num_alleles[var_base[c] + v];
        d.A = A;
        g->h_A[c] = A;
        // unique log-prob cache: dense when small, else direct-mapped
        const uint64_t dense = (uint64_t)S * d.Dc;
        uint64_t cache_entries;
        if (dense <= 8192) {
            d.cache_mode = 0;
            cache_entries = dense;
            d.cache_mask = 0;
        } else {
            d.cache_mode = 1;
            cache_entries = 16384;
            d.cache_mask = (uint32_t)cache_entries - 1;
        }
        const uint64_t Dtot = (uint64_t)d.Dc + 1;
        uint64_t cap = 4;
        while (cap < 2 * std::min<uint64_t>(Dtot, collect_total)) cap <<= 1;
        d.dip_cap = (uint32_t)cap;
        const uint32_t bcap = uset_bucket_capacity(d.H);
        o.allele_base = plan.take((uint64_t)(d.V + 1) * 4, 4);
        o.prng = plan.take(MT_WORDS * 4, 4);
        o.fprng = plan.take(MT_WORDS * 4, 4);
        o.fnd = plan.take(sizeof(NormalState), 8);
        o.uniq = plan.take((uint64_t)d.nu * 4, 4);
        o.multi = plan.take((uint64_t)d.nm * 4, 4);
        o.usub = plan.take((uint64_t)d.nu * 4, 4);
        o.msub = plan.take((uint64_t)d.nm * 4, 4);
        o.smm = plan.take((uint64_t)d.nm * S, 1);
        o.dip = plan.take((uint64_t)S * 4, 2);
        o.freq = plan.take((uint64_t)d.H * 8, 8);
        o.obs = plan.take((uint64_t)d.H * 4, 4);
        o.nz = plan.take(d.H, 1);
        o.zhdr = plan.take(16, 4);
        o.zbkt = plan.take((uint64_t)bcap * 4, 4);
        o.phdr = plan.take(16, 4);
        o.pbkt = plan.take((uint64_t)bcap * 4, 4);
        o.unext = plan.take((uint64_t)d.H * 4, 4);
        o.hvcount = plan.take((uint64_t)d.H * d.V * 4, 4);
        o.ucache = plan.take(cache_entries * 8, 8);
        o.ucache_tag = plan.take(d.cache_mode == 1 ? cache_entries * 4 : 4, 4);
        o.cum = plan.take((uint64_t)std::max<uint32_t>(d.D2, 1) * 8, 8);
        o.nzlist = plan.take((uint64_t)d.H * 2, 2);
        o.simplex = plan.take((uint64_t)(d.H + 1) * 8, 8);
        o.ksc = plan.take((uint64_t)S * 2 * d.V * 4 * 8, 8);
        o.ksc_upd = plan.take(S, 1);
        o.dip_keys = plan.take(cap * 4, 4);
        o.dip_freq = plan.take(cap * S * 4, 4);
        o.astats = plan.take((uint64_t)S * A * 12 * 8, 8);
        o.nest_ploidy = plan.take(S, 1);
        o.nest_n = plan.take(S, 1);
        o.nest_stats = plan.take((uint64_t)S * 2 * 4 * 8, 8);
        o.sc = plan.take(SC_COUNT * 4, 4);
        o.edges = plan.take((uint64_t)d.ne * 4, 4);
        o.cover_rows = plan.take(d.K, 1);
    }
    struct GOffs {
        uint64_t sources, stack, brng, shared;
    };
    std::vector<GOffs> goffs(G);
    for (uint32_t gi = 0; gi < G; ++gi) {
        const uint32_t nv = B->group_cluster_off[gi + 1] - B->group_cluster_off[gi];
        const uint32_t ns = B->group_source_off[gi + 1] - B->group_source_off[gi];
        goffs[gi].sources = plan.take((uint64_t)ns * 4, 4);
        goffs[gi].stack = plan.take((uint64_t)(nv + 1) * 2 * 4, 4);
        goffs[gi].brng = plan.take(MT_WORDS * 4, 4);
        goffs[gi].shared = plan.take((uint64_t)B->group_num_shared[gi] * S, 1);
    }
    g->pool_bytes = plan.size + 64;
    {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&g->d_pool), g->pool_bytes);
        if (e != hipSuccess) {
            bt_gibbs_destroy(g);
            return fail(std::string("bt_gibbs_create: state pool of ") + std::to_string(g->pool_bytes) + " bytes: " + hipGetErrorString(e));
        }
    }
    g->allocs.push_back(g->d_pool);
    g->device_bytes += g->pool_bytes;
    BT_TRY(hipMemsetAsync(g->d_pool, 0, g->pool_bytes, ctx->stream) == hipSuccess ? BT_OK : fail("bt_gibbs_create: memset failed"));

    // ---- host image of the mutable, non-zero-initialised parts of the pool ----
    // (index lists, edges, sources, allele_base); built in one staging buffer and copied over the zeroed pool
    std::vector<uint8_t> stage(g->pool_bytes, 0);
    for (uint32_t c = 0; c < C; ++c) {
        const ClusterDev &d = hc[c];
        const Offs &o = offs[c];
        uint32_t *ab = reinterpret_cast<uint32_t *>(stage.data() + o.allele_base);
        uint32_t acc = 0;
        for (uint32_t v = 0; v < d.V; ++v) {
            ab[v] = acc;
            acc += B->var_num_alleles[var_base[c] + v];
        }
        ab[d.V] = acc;
        std::memcpy(stage.data() + o.uniq, B->unique_idx + B->unique_off[c], (size_t)d.nu * 4);
        std::memcpy(stage.data() + o.multi, B->multi_idx + B->multi_off[c], (size_t)d.nm * 4);
        std::memcpy(stage.data() + o.edges, B->edges + B->edge_off[c], (size_t)d.ne * 4);
    }
    for (uint32_t gi = 0; gi < G; ++gi) {
        const uint32_t ns = B->group_source_off[gi + 1] - B->group_source_off[gi];
        std::memcpy(stage.data() + goffs[gi].sources, B->group_sources + B->group_source_off[gi], (size_t)ns * 4);
    }
    BT_TRY(hipMemcpyAsync(g->d_pool, stage.data(), g->pool_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess ? BT_OK
                                                                                                                        : fail("bt_gibbs_create: pool upload failed"));
    BT_TRY(hipStreamSynchronize(ctx->stream) == hipSuccess ? BT_OK : fail("bt_gibbs_create: sync failed"));

    // ---- device records ----
    std::vector<GroupDev> hg(G);
    for (uint32_t gi = 0; gi < G; ++gi) {
        GroupDev &gd = hg[gi];
        gd.index = B->group_index[gi];
        gd.c0 = B->group_cluster_off[gi];
        gd.nvert = B->group_cluster_off[gi + 1] - gd.c0;
        gd.nsrc = B->group_source_off[gi + 1] - B->group_source_off[gi];
        gd.nshared = B->group_num_shared[gi];
        gd.sources = reinterpret_cast<uint32_t *>(g->d_pool + goffs[gi].sources);
        gd.ploidy = d_group_ploidy + (uint64_t)gi * S;
        gd.stack = reinterpret_cast<uint32_t *>(g->d_pool + goffs[gi].stack);
        gd.brng = reinterpret_cast<uint32_t *>(g->d_pool + goffs[gi].brng);
        gd.trace = nullptr;
        uint8_t *shared = g->d_pool + goffs[gi].shared;
        for (uint32_t c = gd.c0; c < gd.c0 + gd.nvert; ++c) {
            ClusterDev &d = hc[c];
            const Offs &o = offs[c];
            const uint64_t r0 = B->kmer_off[c];
            uint8_t *P0 = g->d_pool;
            d.M = d_hap_kmer_mult + mult_off[c];
            d.has_counts = d_kmer_has_counts + r0;
            d.counts = d_kmer_counts + r0 * S;
            d.ic = d_kmer_ic_mult + r0 * 2;
            d.shared_idx = d_kmer_shared + r0;
            d.kv_off = d_kv_off + r0;
            d.kv_var = d_kv_var;
            d.kv_bits = d_kv_bits + kvb_off[c];
            d.hap_allele = d_hap_allele + hapvar_off[c];
            d.hapnest_off = d_hapnest_off + hap_base[c];
            d.hapnest_idx = d_hapnest_idx;
            d.var_na = d_var_num_alleles + var_base[c];
            d.var_dep = d_var_has_dependency + var_base[c];
            d.allele_base = reinterpret_cast<uint32_t *>(P0 + o.allele_base);
            d.nd_cluster = d_nestdep_cluster + B->nestdep_off[c];
            d.nd_var_off = d_nestdep_var_off + B->nestdep_off[c];
            d.nd_var = d_nestdep_var;
            d.uniq0 = d_unique_idx + B->unique_off[c];
            d.multi0 = d_multi_idx + B->multi_off[c];
            d.shared_mult = shared;
            d.prng = reinterpret_cast<uint32_t *>(P0 + o.prng);
            d.fprng = reinterpret_cast<uint32_t *>(P0 + o.fprng);
            d.fnd = reinterpret_cast<NormalState *>(P0 + o.fnd);
            d.uniq = reinterpret_cast<uint32_t *>(P0 + o.uniq);
            d.multi = reinterpret_cast<uint32_t *>(P0 + o.multi);
            d.usub = reinterpret_cast<uint32_t *>(P0 + o.usub);
            d.msub = reinterpret_cast<uint32_t *>(P0 + o.msub);
            d.smm = P0 + o.smm;
            d.dip = reinterpret_cast<uint16_t *>(P0 + o.dip);
            d.freq = reinterpret_cast<double *>(P0 + o.freq);
            d.obs = reinterpret_cast<uint32_t *>(P0 + o.obs);
            d.nz = P0 + o.nz;
            d.zhdr = reinterpret_cast<uint32_t *>(P0 + o.zhdr);
            d.zbkt = reinterpret_cast<uint32_t *>(P0 + o.zbkt);
            d.phdr = reinterpret_cast<uint32_t *>(P0 + o.phdr);
            d.pbkt = reinterpret_cast<uint32_t *>(P0 + o.pbkt);
            d.unext = reinterpret_cast<uint32_t *>(P0 + o.unext);
            d.hvcount = reinterpret_cast<uint32_t *>(P0 + o.hvcount);
            d.ucache = reinterpret_cast<double *>(P0 + o.ucache);
            d.ucache_tag = reinterpret_cast<uint32_t *>(P0 + o.ucache_tag);
            d.cum = reinterpret_cast<double *>(P0 + o.cum);
            d.nzlist = reinterpret_cast<uint16_t *>(P0 + o.nzlist);
            d.simplex = reinterpret_cast<double *>(P0 + o.simplex);
            d.ksc = reinterpret_cast<double *>(P0 + o.ksc);
            d.ksc_upd = P0 + o.ksc_upd;
            d.dip_keys = reinterpret_cast<uint32_t *>(P0 + o.dip_keys);
            d.dip_freq = reinterpret_cast<uint32_t *>(P0 + o.dip_freq);
            d.astats = reinterpret_cast<double *>(P0 + o.astats);
            d.nest_ploidy = P0 + o.nest_ploidy;
            d.nest_n = P0 + o.nest_n;
            d.nest_stats = reinterpret_cast<double *>(P0 + o.nest_stats);
            d.sc = reinterpret_cast<uint32_t *>(P0 + o.sc);
            d.edges = reinterpret_cast<uint32_t *>(P0 + o.edges);
            d.cover_rows = P0 + o.cover_rows;
        }
    }
    g->h_nvert.resize(G);
    for (uint32_t gi = 0; gi < G; ++gi) g->h_nvert[gi] = hg[gi].nvert;
    BT_TRY(upload<GroupDev>(ctx, hg.data(), G, &g->d_groups, &g->device_bytes));
    g->allocs.push_back(g->d_groups);
    BT_TRY(upload<ClusterDev>(ctx, hc.data(), C, &g->d_clusters, &g->device_bytes));
    g->allocs.push_back(g->d_clusters);
    // LUT buffers
    BT_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_lut_g), (size_t)S * 65536 * 8) == hipSuccess ? BT_OK : fail("bt_gibbs_create: LUT alloc failed"));
    g->allocs.push_back(g->d_lut_g);
    BT_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_lut_n), (size_t)S * 256 * 8) == hipSuccess ? BT_OK : fail("bt_gibbs_create: LUT alloc failed"));
    g->allocs.push_back(g->d_lut_n);
    g->device_bytes += (uint64_t)S * (65536 + 256) * 8;
    g->P.lut_g = g->d_lut_g;
    g->P.lut_n = g->d_lut_n;
    BT_TRY(hipStreamSynchronize(ctx->stream) == hipSuccess ? BT_OK : fail("bt_gibbs_create: sync failed"));
#undef UP
#undef BT_TRY
    (void)nEdges;
    (void)nSrc;
    (void)nShared;
    *out = g;
    return BT_OK;
}

int bt_gibbs_destroy(bt_gibbs *g) {
    if (!g) return BT_OK;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    for (void *p : g->allocs)
        if (p) (void)hipFree(p);
    if (g->d_trace) (void)hipFree(g->d_trace);
    if (g->d_trace_counter) (void)hipFree(g->d_trace_counter);
    delete g;
    return BT_OK;
}

int bt_gibbs_set_lut(bt_gibbs *g, const double *h_genomic, const double *h_noise) {
    if (!g || !h_genomic || !h_noise) return fail("bt_gibbs_set_lut: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipMemcpyAsync(g->d_lut_g, h_genomic, (size_t)g->S * 65536 * 8, hipMemcpyHostToDevice, g->ctx->stream));
    BT_HIP(hipMemcpyAsync(g->d_lut_n, h_noise, (size_t)g->S * 256 * 8, hipMemcpyHostToDevice, g->ctx->stream));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    g->lut_set = true;
    return BT_OK;
}

int bt_gibbs_set_noise_lut(bt_gibbs *g, const double *h_noise) {
    if (!g || !h_noise) return fail("bt_gibbs_set_noise_lut: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipMemcpyAsync(g->d_lut_n, h_noise, (size_t)g->S * 256 * 8, hipMemcpyHostToDevice, g->ctx->stream));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    return BT_OK;
}

int bt_gibbs_init_chain(bt_gibbs *g, uint32_t chain_idx) {
    if (!g) return fail("bt_gibbs_init_chain: null handle");
    return launch(g, OP_INIT_CHAIN, chain_idx, 0, nullptr);
}

int bt_gibbs_sweep(bt_gibbs *g, uint32_t num_sweeps, int collect_samples) {
    if (!g) return fail("bt_gibbs_sweep: null handle");
    if (num_sweeps == 0) return BT_OK;
    return launch(g, OP_SWEEP, num_sweeps, collect_samples ? 1u : 0u, nullptr);
}

int bt_gibbs_run(bt_gibbs *g) {
    if (!g) return fail("bt_gibbs_run: null handle");
    return launch(g, OP_RUN, 0, 0, nullptr);
}

int bt_gibbs_noise_counts(bt_gibbs *g, uint64_t *d_hist, int zero_first) {
    if (!g || !d_hist) return fail("bt_gibbs_noise_counts: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    if (zero_first) BT_HIP(hipMemsetAsync(d_hist, 0, (size_t)g->S * 256 * 8, g->ctx->stream));
    return launch(g, OP_NOISE, 0, 0, reinterpret_cast<unsigned long long *>(d_hist));
}

int bt_gibbs_reset_groups(bt_gibbs *g) {
    if (!g) return fail("bt_gibbs_reset_groups: null handle");
    return launch(g, OP_RESET, 0, 0, nullptr);
}

int bt_gibbs_posterior_summary(bt_gibbs *g, uint32_t *d_out) {
    if (!g || !d_out) return fail("bt_gibbs_posterior_summary: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    hipLaunchKernelGGL(summary_kernel, dim3((g->C + 255) / 256), dim3(256), 0, g->ctx->stream, g->d_clusters, g->C, g->S, d_out);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_gibbs_device_bytes(bt_gibbs *g, uint64_t *bytes) {
    if (!g || !bytes) return fail("bt_gibbs_device_bytes: null argument");
    *bytes = g->device_bytes;
    return BT_OK;
}

static int fetch_scalars(bt_gibbs *g, std::vector<uint32_t> &sc) {
    // the scalar blocks are scattered in the pool: copy them one by one (diagnostic path, C small) or the whole pool when large
    sc.assign((size_t)g->C * SC_COUNT, 0);
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    if (g->C > 4096) {
        std::vector<uint8_t> pool(g->pool_bytes);
        BT_HIP(hipMemcpy(pool.data(), g->d_pool, g->pool_bytes, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < g->C; ++c)
            std::memcpy(&sc[(size_t)c * SC_COUNT], pool.data() + (reinterpret_cast<uint8_t *>(g->h_clusters[c].sc) - g->d_pool), SC_COUNT * 4);
    } else {
        for (uint32_t c = 0; c < g->C; ++c) BT_HIP(hipMemcpy(&sc[(size_t)c * SC_COUNT], g->h_clusters[c].sc, SC_COUNT * 4, hipMemcpyDeviceToHost));
    }
    return BT_OK;
}

int bt_gibbs_result_sizes(bt_gibbs *g, uint64_t *num_diplotype_entries, uint64_t *num_allele_cells) {
    if (!g) return fail("bt_gibbs_result_sizes: null handle");
    std::vector<uint32_t> sc;
    int rc = fetch_scalars(g, sc);
    if (rc != BT_OK) return rc;
    uint64_t nd = 0, nc = 0;
    for (uint32_t c = 0; c < g->C; ++c) {
        if (sc[(size_t)c * SC_COUNT + SC_DIP_OVERFLOW]) return fail("bt_gibbs: diplotype frequency table overflowed");
        nd += sc[(size_t)c * SC_COUNT + SC_DIP_ENTRIES];
        nc += (uint64_t)g->h_A[c] * g->S;
    }
    if (num_diplotype_entries) *num_diplotype_entries = nd;
    if (num_allele_cells) *num_allele_cells = nc;
    return BT_OK;
}

int bt_gibbs_result_fetch(bt_gibbs *g, uint64_t *h_dip_off, uint16_t *h_dip_h1, uint16_t *h_dip_h2, uint32_t *h_dip_freq, uint64_t *h_cell_off,
                          double *h_stats) {
    if (!g || !h_dip_off || !h_cell_off) return fail("bt_gibbs_result_fetch: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    std::vector<uint8_t> pool(g->pool_bytes);
    BT_HIP(hipMemcpy(pool.data(), g->d_pool, g->pool_bytes, hipMemcpyDeviceToHost));
    const uint32_t S = g->S;
    uint64_t e = 0, cell = 0;
    std::vector<std::pair<uint32_t, uint32_t>> order;   // (key, slot)
    for (uint32_t c = 0; c < g->C; ++c) {
        const ClusterDev &d = g->h_clusters[c];
        const uint32_t *keys = reinterpret_cast<const uint32_t *>(pool.data() + (reinterpret_cast<uint8_t *>(d.dip_keys) - g->d_pool));
        const uint32_t *freq = reinterpret_cast<const uint32_t *>(pool.data() + (reinterpret_cast<uint8_t *>(d.dip_freq) - g->d_pool));
        const double *astats = reinterpret_cast<const double *>(pool.data() + (reinterpret_cast<uint8_t *>(d.astats) - g->d_pool));
        h_dip_off[c] = e;
        h_cell_off[c] = cell;
        order.clear();
        for (uint32_t slot = 0; slot < d.dip_cap; ++slot) {
            const uint32_t tag = keys[slot];
            if (!tag) continue;
            const uint32_t key = tag == 0xFFFFFFFFu ? 0xFFFFFFFFu : tag - 1u;
            order.emplace_back(key, slot);
        }
        // entries sorted by (h1, h2)
        std::sort(order.begin(), order.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) {
            const uint32_t a1 = a.first & 0xFFFF, a2 = a.first >> 16, b1 = b.first & 0xFFFF, b2 = b.first >> 16;
            return a1 != b1 ? a1 < b1 : a2 < b2;
        });
        for (auto &kv : order) {
            if (h_dip_h1) h_dip_h1[e] = (uint16_t)(kv.first & 0xFFFF);
            if (h_dip_h2) h_dip_h2[e] = (uint16_t)(kv.first >> 16);
            if (h_dip_freq)
                for (uint32_t s = 0; s < S; ++s) h_dip_freq[e * S + s] = freq[(size_t)kv.second * S + s];
            ++e;
        }
        if (h_stats) std::memcpy(h_stats + cell * 12, astats, (size_t)S * d.A * 12 * 8);
        cell += (uint64_t)S * d.A;
    }
    h_dip_off[g->C] = e;
    h_cell_off[g->C] = cell;
    return BT_OK;
}

int bt_gibbs_trace_enable(bt_gibbs *g, uint32_t max_sweeps) {
    if (!g) return fail("bt_gibbs_trace_enable: null handle");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    if (g->d_trace) {
        BT_HIP(hipFree(g->d_trace));
        g->d_trace = nullptr;
    }
    if (g->d_trace_counter) {
        BT_HIP(hipFree(g->d_trace_counter));
        g->d_trace_counter = nullptr;
    }
    g->trace_sweeps = max_sweeps;
    std::vector<GroupDev> hg(g->G);
    BT_HIP(hipMemcpy(hg.data(), g->d_groups, (size_t)g->G * sizeof(GroupDev), hipMemcpyDeviceToHost));
    g->trace_off.assign(g->G + 1, 0);
    for (uint32_t gi = 0; gi < g->G; ++gi) g->trace_off[gi + 1] = g->trace_off[gi] + (uint64_t)max_sweeps * g->h_nvert[gi] * g->S;
    g->trace_words = g->trace_off[g->G];
    if (max_sweeps) {
        BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_trace), std::max<uint64_t>(g->trace_words, 1) * 4));
        BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_trace_counter), (size_t)g->G * 4));
        BT_HIP(hipMemset(g->d_trace, 0xFF, std::max<uint64_t>(g->trace_words, 1) * 4));
        BT_HIP(hipMemset(g->d_trace_counter, 0, (size_t)g->G * 4));
    }
    for (uint32_t gi = 0; gi < g->G; ++gi) hg[gi].trace = max_sweeps ? g->d_trace + g->trace_off[gi] : nullptr;
    BT_HIP(hipMemcpy(g->d_groups, hg.data(), (size_t)g->G * sizeof(GroupDev), hipMemcpyHostToDevice));
    return BT_OK;
}

int bt_gibbs_trace_fetch(bt_gibbs *g, uint32_t *h_trace, uint64_t max_words, uint64_t *num_sweeps_recorded) {
    if (!g || !h_trace) return fail("bt_gibbs_trace_fetch: null argument");
    if (!g->d_trace) return fail("bt_gibbs_trace_fetch: tracing is off");
    if (max_words < g->trace_words) return fail("bt_gibbs_trace_fetch: buffer too small");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    BT_HIP(hipMemcpy(h_trace, g->d_trace, g->trace_words * 4, hipMemcpyDeviceToHost));
    if (num_sweeps_recorded) {
        uint32_t n0 = 0;
        BT_HIP(hipMemcpy(&n0, g->d_trace_counter, 4, hipMemcpyDeviceToHost));
        *num_sweeps_recorded = n0;
    }
    return BT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// diagnostics (host-side execution of the __host__ __device__ primitives)
// ---------------------------------------------------------------------------------------------------------
int bt_diag_uset_replay(uint32_t universe, const uint8_t *ops, const uint32_t *values, uint64_t num_ops, uint32_t *h_order, uint32_t *n) {
    if (!ops || !values || !h_order || !n) return fail("bt_diag_uset_replay: null argument");
    std::vector<uint32_t> hdr(4), bkt(uset_bucket_capacity(universe)), next(std::max<uint32_t>(universe, 1));
    std::vector<uint8_t> present(universe, 0);
    USet s{hdr.data(), bkt.data(), next.data()};
    uset_init(s);
    for (uint64_t i = 0; i < num_ops; ++i) {
        if (ops[i] == 2) {
            uset_clear(s);
            std::fill(present.begin(), present.end(), 0);
            continue;
        }
        const uint32_t v = values[i];
        if (v >= universe) return fail("bt_diag_uset_replay: value outside the universe");
        if (ops[i] == 0 && !present[v]) {
            uset_insert(s, v);
            present[v] = 1;
        } else if (ops[i] == 1 && present[v]) {
            uset_erase(s, v);
            present[v] = 0;
        }
    }
    uint32_t j = 0;
    for (uint32_t e = uset_begin(s); e != US_NONE; e = next[e]) h_order[j++] = e;
    *n = j;
    return BT_OK;
}

int bt_diag_rng(uint32_t seed, int kind, const double *a, const double *b, uint64_t n, double *h_out) {
    if (!h_out) return fail("bt_diag_rng: null argument");
    std::vector<uint32_t> st(MT_WORDS);
    mt_seed(st.data(), seed);
    NormalState nd{0, 0};
    switch (kind) {
        case 0:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = (double)mt_next(st.data());
            break;
        case 1:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = rng_canonical(st.data());
            break;
        case 2:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = rng_gamma(st.data(), &nd, a[i], b[i]);
            break;
        case 3:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = (double)rng_uniform_int(st.data(), (uint32_t)a[i] + 1u);
            break;
        case 4:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = rng_bernoulli(st.data(), (double)(float)a[0]) ? 1.0 : 0.0;
            break;
        case 5: {
            std::vector<uint32_t> v((size_t)a[0]);
            for (size_t i = 0; i < v.size(); ++i) v[i] = (uint32_t)i;
            rng_shuffle_u32(st.data(), v.data(), (uint32_t)v.size());
            for (size_t i = 0; i < v.size(); ++i) h_out[i] = v[i];
            break;
        }
        default:
            return fail("bt_diag_rng: unknown kind");
    }
    return BT_OK;
}

}  // extern "C"
