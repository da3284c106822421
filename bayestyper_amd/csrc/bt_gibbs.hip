// libbtgpu: Gibbs genotyping of variant-cluster groups (host side of bt_gibbs_* + the kernels).
//
//   bt_gibbs_init_chain  <- VariantClusterGroup::initGenotyper + shuffleBranchOrdering (VariantClusterGroup.cpp:171-218)
//   bt_gibbs_sweep       <- VariantClusterGroup::estimateGenotypes / runGibbsSample (VariantClusterGroup.cpp:220-250)
//   bt_gibbs_run         <- InferenceEngine::estimateGenotypesCallback inner loops (InferenceEngine.cpp:290-306)
//   bt_gibbs_noise_counts<- VariantClusterGroup::getNoiseCounts + clearGenotyperCache (InferenceEngine.cpp:90-92)
//
// Parallelisation: variant-cluster groups are the reference's unit of independence (one std::thread works a group at a
// time).  Here one LANE owns one group for a whole launch: a sweep is a strictly sequential chain of dependent random
// draws (samples within a sweep, sweeps within a chain, chains within a genotyper), so the parallel axis is the group.
// Groups are sorted by shape and cut into TILES of 64; a tile is one wavefront's worth of lane-interleaved HBM
// (bt_gibbs_tile.hpp) so that the wave's memory traffic coalesces.  One workgroup = one wavefront = one tile.
#ifndef BT_SWEEP_OUTLINE
#define BT_SWEEP_INLINE   // the per-visit sweep functions are part of the kernel body (bt_rng_device.hpp: BT_SWEEPFN)
#endif
#include "bt_gibbs_kernel.hpp"
#include "bt_internal.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>

using namespace bt;

namespace bt {
// defined in bt_gibbs_simple_kernel.hip
hipError_t launch_gibbs_simple_kernel(unsigned grid, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, int op, uint32_t a0, uint32_t a1,
                                      unsigned long long *hist, TraceCfg tr, const uint32_t *tile_list);
hipError_t prepare_gibbs_simple_kernel(int max_lds);
// defined in bt_gibbs_hot_kernel.hip
hipError_t launch_gibbs_hot_kernel(unsigned grid, unsigned block, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, int op, uint32_t a0, uint32_t a1,
                                   unsigned long long *hist, TraceCfg tr, const uint32_t *tile_list);
hipError_t prepare_gibbs_hot_kernel(int max_lds);
hipError_t occupancy_gibbs_hot_kernel(int *blocks_per_cu, int block, uint32_t lds);
// defined in bt_gibbs_single_kernel.hip
hipError_t launch_gibbs_single_kernel(unsigned grid, unsigned block, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, int op, uint32_t a0, uint32_t a1,
                                      unsigned long long *hist, TraceCfg tr, const uint32_t *tile_list);
hipError_t prepare_gibbs_single_kernel(int max_lds);
hipError_t occupancy_gibbs_single_kernel(int *blocks_per_cu, int block, uint32_t lds);
// defined in bt_gibbs_chain_kernel.hip
hipError_t launch_gibbs_chain_kernel(unsigned grid, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, const NoiseChainCtl *ctl, TraceCfg tr);
hipError_t occupancy_gibbs_chain_kernel(int *blocks_per_cu, uint32_t lds);
hipError_t prepare_gibbs_chain_kernel(int max_lds);
#ifdef BT_PROF
hipError_t simple_prof_read(unsigned long long *h_out32, int reset);
hipError_t hot_prof_read(unsigned long long *h_out32, int reset);
hipError_t single_prof_read(unsigned long long *h_out32, int reset);
#endif
}  // namespace bt

namespace {

__global__ __launch_bounds__(LANES * 8, GIBBS_WAVES) void gibbs_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg, int op,
                                                       uint32_t arg0, uint32_t arg1, unsigned long long *__restrict__ hist, TraceCfg tr, const uint32_t *__restrict__ tile_list) {
    gibbs_body<false>(tiles, pool, Pg, op, arg0, arg1, hist, tr, tile_list);
}

// per (cluster, sample): most frequently sampled diplotype and its frequency -> the compact posterior summary that is
// gathered to rank 0; ties resolve to the smallest (h1, h2).  One lane per cluster.
struct ClusterLoc {
    uint32_t tile, lane, v, pad;
};
__global__ __launch_bounds__(256) void summary_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const ClusterLoc *__restrict__ loc,
                                                      uint32_t num_clusters, uint32_t S, uint32_t *__restrict__ out) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= num_clusters) return;
    Tile t;
    t.d = (const TileDesc BT_CAS *)&tiles[loc[c].tile];
    t.base = (uint8_t BT_GAS *)(pool + t.d->base);
    t.lane = t.plane = loc[c].lane;   // (no LDS here)
    t.wsh = t.d->wsh;
    t.hot = nullptr;
    t.lds0 = 0;
    t.resident = 0xFFFFFFFFu;
    const Vx x = make_vx(t, loc[c].v);
    TPtr<uint32_t> keys = x.dip_keys(), freq = x.dip_freq();
    const uint32_t cap = t.d->dip_cap;
    for (uint32_t s = 0; s < S; ++s) {
        uint32_t best_key = 0xFFFFFFFFu, best = 0;
        for (uint32_t slot = 0; slot < cap; ++slot) {
            const uint32_t tag = keys[slot];
            if (!tag) continue;
            const uint32_t key = tag == 0xFFFFFFFFu ? 0xFFFFFFFFu : tag - 1u;
            const uint32_t f = freq[(size_t)slot * S + s];
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

// ---- the collected samples of a launch in the flat layout bt_gibbs_result_fetch hands out, built ON THE DEVICE ----------------------------------
// (round 3 copied three arrays of every tile to the host, one synchronous copy each, and rebuilt the layout cluster by cluster on one host thread)
__device__ inline Vx result_vx(const TileDesc *tiles, uint8_t *pool, const ClusterLoc &L) {
    Tile t;
    t.d = (const TileDesc BT_CAS *)&tiles[L.tile];
    t.base = (uint8_t BT_GAS *)(pool + t.d->base);
    t.lane = t.plane = L.lane;   // (no LDS here; pool_lane0 is 0 since the tiles have rows of their own width)
    t.plane += t.d->pool_lane0;
    t.wsh = t.d->wsh;
    t.part = 0;
    t.copies = 1;
    t.hot = nullptr;
    t.lds0 = 0;
    t.resident = 0xFFFFFFFFu;
    return make_vx(t, L.v);
}
// entries of every cluster's diplotype table (VariantClusterGenotyper: diplotype_sampling_frequencies), overflow flag
__global__ __launch_bounds__(256) void result_count_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const ClusterLoc *__restrict__ loc, uint32_t num_clusters,
                                                           uint32_t *__restrict__ n_ent, uint32_t *__restrict__ overflow) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= num_clusters) return;
    const Vx x = result_vx(tiles, pool, loc[c]);
    SPtrF<uint32_t, LANES> sc = x.sc();
    n_ent[c] = sc[SC_DIP_ENTRIES];
    if (sc[SC_DIP_OVERFLOW]) atomicOr(overflow, 1u);
}
// one wavefront per cluster: the cluster's diplotype entries ordered by (h1, h2) with their per-sample counts, its allele k-mer statistics
constexpr uint32_t kPackLds = 2048;   // entries ranked from LDS; larger tables are ranked from the table itself
__global__ __launch_bounds__(64) void result_pack_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const ClusterLoc *__restrict__ loc, uint32_t S,
                                                         const uint64_t *__restrict__ dip_off, const uint64_t *__restrict__ cell_off, const uint32_t *__restrict__ alleles,
                                                         uint16_t *__restrict__ out_h1, uint16_t *__restrict__ out_h2, uint32_t *__restrict__ out_freq, double *__restrict__ out_stats,
                                                         uint32_t *__restrict__ out_keys) {   // out_keys: h1 | h2 << 16 per entry (the wire string of bt_gibbs_result_words)
    __shared__ uint32_t l_key[kPackLds], l_slot[kPackLds];
    __shared__ uint32_t l_n;
    const uint32_t c = blockIdx.x;
    const Vx x = result_vx(tiles, pool, loc[c]);
    TPtr<uint32_t> keys = x.dip_keys(), freq = x.dip_freq();
    const uint32_t cap = x.d().dip_cap;
    const uint64_t e0 = dip_off[c];
    const uint32_t n = (uint32_t)(dip_off[c + 1] - e0);
    auto order_key = [](uint32_t tag) {   // (h1, h2) lexicographic: h1 in the upper half
        const uint32_t key = tag == 0xFFFFFFFFu ? 0xFFFFFFFFu : tag - 1u;
        return (key << 16) | (key >> 16);
    };
    if (out_h1 || out_h2 || out_freq || out_keys) {
        if (n <= kPackLds) {
            if (threadIdx.x == 0) l_n = 0;
            __syncthreads();
            for (uint32_t slot = threadIdx.x; slot < cap; slot += 64u) {
                const uint32_t tag = keys[slot];
                if (!tag) continue;
                const uint32_t i = atomicAdd(&l_n, 1u);
                if (i < kPackLds) {
                    l_key[i] = order_key(tag);
                    l_slot[i] = slot;
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += 64u) {
                const uint32_t kk = l_key[i], slot = l_slot[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n; ++j) rank += l_key[j] < kk ? 1u : 0u;   // (keys are distinct)
                const uint64_t e = e0 + rank;
                if (out_h1) out_h1[e] = (uint16_t)(kk >> 16);
                if (out_h2) out_h2[e] = (uint16_t)(kk & 0xFFFFu);
                if (out_keys) out_keys[e] = (kk >> 16) | (kk << 16);
                if (out_freq)
                    for (uint32_t s = 0; s < S; ++s) out_freq[e * S + s] = freq[slot * S + s];
            }
        } else {
            for (uint32_t slot = threadIdx.x; slot < cap; slot += 64u) {
                const uint32_t tag = keys[slot];
                if (!tag) continue;
                const uint32_t kk = order_key(tag);
                uint32_t rank = 0;
                for (uint32_t j = 0; j < cap; ++j) {
                    const uint32_t tj = keys[j];
                    rank += (tj && order_key(tj) < kk) ? 1u : 0u;
                }
                const uint64_t e = e0 + rank;
                if (out_h1) out_h1[e] = (uint16_t)(kk >> 16);
                if (out_h2) out_h2[e] = (uint16_t)(kk & 0xFFFFu);
                if (out_keys) out_keys[e] = (kk >> 16) | (kk << 16);
                if (out_freq)
                    for (uint32_t s = 0; s < S; ++s) out_freq[e * S + s] = freq[slot * S + s];
            }
        }
    }
    if (out_stats) {
        const uint32_t A = alleles[c], Am = x.d().Am, row = A * 12u;
        TPtr<double> as = x.a<double>(A_ASTATS, S * Am * 12u);
        double *dst = out_stats + cell_off[c] * 12u;
        for (uint32_t i = threadIdx.x; i < S * row; i += 64u) {
            const uint32_t s = i / row, r = i - s * row;
            dst[i] = as[s * Am * 12u + r];
        }
    }
}

// entries and cells per cluster, interleaved (the size table of the wire string)
__global__ __launch_bounds__(64) void wire_sizes_kernel(const uint64_t *__restrict__ dip_off, const uint64_t *__restrict__ cell_off, uint32_t num_clusters, uint32_t *__restrict__ sizes) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= num_clusters) return;
    sizes[2 * c] = (uint32_t)(dip_off[c + 1] - dip_off[c]);
    sizes[2 * c + 1] = (uint32_t)(cell_off[c + 1] - cell_off[c]);
}

// ---- the noise model's update of one iteration (bt_gibbs_noise_chain) ----------------------------------------------------------------
// std::mt19937 as libstdc++ runs it: the whole block of 624 words is regenerated when it is used up (bits/random.tcc _M_gen_rand)
struct MtBlock {
    uint32_t *x;
    uint32_t p;
    __device__ void gen_rand() {
        for (uint32_t k = 0; k < MT_N - MT_M; ++k) x[k] = mt_twist(x[k], x[k + 1], x[k + MT_M]);
        for (uint32_t k = MT_N - MT_M; k < MT_N - 1; ++k) x[k] = mt_twist(x[k], x[k + 1], x[k + MT_M - MT_N]);
        x[MT_N - 1] = mt_twist(x[MT_N - 1], x[0], x[MT_M - 1]);
        p = 0;
    }
    __device__ uint32_t next() {
        if (p >= MT_N) gen_rand();
        return mt_temper(x[p++]);
    }
    template <unsigned N>
    __device__ void next_n(uint32_t (&out)[N]) {
        for (uint32_t k = 0; k < N; ++k) out[k] = next();
    }
};
struct NoiseDev {   // device copy of bt_noise_rng + the per-sample priors / rates
    uint32_t mt[MT_N];
    uint32_t mt_pos, saved_available;
    double saved;
};
// one workgroup of 256 threads: sufficient statistics of the histogram (calcCountSuffStats), S gamma draws by thread 0 (the stream is
// sequential), then the S x 256 Poisson log-pmf table with the tail of counts >= 255 folded into entry 255 (CountDistribution.cpp:314-347)
__global__ __launch_bounds__(256) void noise_update_kernel(NoiseDev *__restrict__ st, const float *__restrict__ prior, uint32_t S, const unsigned long long *__restrict__ hist,
                                                           const double *__restrict__ lgamma_c /* lgamma(c + 1), c = 0..255 */, double *__restrict__ rates_row, double *__restrict__ rates_cur,
                                                           double *__restrict__ lut_n) {
    __shared__ unsigned long long s_obs[32], s_sum[32];
    __shared__ double s_rate[32];
    const uint32_t t = threadIdx.x;
    for (uint32_t s = 0; s < S; ++s) {   // num_observations = sum_i n_i, count_sum = sum_i i * n_i
        unsigned long long o = hist[s * 256u + t], w = o * t;
        for (int off = 32; off > 0; off >>= 1) {
            o += __shfl_down(o, off);
            w += __shfl_down(w, off);
        }
        __shared__ unsigned long long part_o[4], part_w[4];
        if ((t & 63u) == 0) {
            part_o[t >> 6] = o;
            part_w[t >> 6] = w;
        }
        __syncthreads();
        if (t == 0) {
            s_obs[s] = part_o[0] + part_o[1] + part_o[2] + part_o[3];
            s_sum[s] = part_w[0] + part_w[1] + part_w[2] + part_w[3];
        }
        __syncthreads();
    }
    if (t == 0) {
        MtBlock g{st->mt, st->mt_pos};
        NormalState nd{&st->saved, &st->saved_available};
        for (uint32_t s = 0; s < S; ++s) {
            // CountDistribution.cpp:173-186: the priors are floats and the counts unsigned longs there, so both arguments are computed in FLOAT
            // arithmetic (usual arithmetic conversions) before sampleGamma widens them
            const float shape = prior[2 * s], scale = prior[2 * s + 1];
            const float shape_f = shape + (float)s_sum[s];
            const float scale_f = scale / ((float)s_obs[s] * scale + 1.0f);
            const double r = rng_gamma(g, nd, (double)shape_f, (double)scale_f);
            s_rate[s] = r;
            rates_row[s] = r;
            rates_cur[s] = r;
        }
        st->mt_pos = g.p;
    }
    __syncthreads();
    for (uint32_t s = 0; s < S; ++s) {
        const double rate = s_rate[s], lr = log(rate);
        double v = (double)t * lr - rate - lgamma_c[t];   // poissonLogProb (:349-352)
        if (t == 255) {
            unsigned limit = 255;
            double prev = 0;
            bool more = true;
            while (more) {
                limit++;
                prev = v;
                const double q = (double)limit * lr - rate - lgamma((double)limit + 1.0);
                v = v < q ? q + log1p(exp(v - q)) : v + log1p(exp(q - v));   // Utils::logAddition
                if (v > 0) {
                    v = 0;
                    break;
                }
                const double mn = prev < v ? prev : v;
                more = !((prev == v) || (fabs(prev - v) < fabs(mn) * BT_DBL_EPS * 100));   // Utils::doubleCompare
            }
        }
        lut_n[s * 256u + t] = v;
    }
}

// ---- tile builder on the device (bt_gibbs_create) --------------------------------------------------------------------------------
// The batch arrives as flat arrays (include/btgpu.h: bt_gibbs_batch).  One workgroup per cluster scatters the cluster's slices into its
// tile's arrays (rows interleaved over the tile's width, TileDesc::wsh); the host computes only the descriptors below.
struct BuildCluster {
    uint32_t tile, lane, v, H, V, K, r0, u0, nu, m0, nm, nd0, nd, e0, ne, cid, A, hap_base, var_base, pad;
    uint64_t mult_off, kvb_off, hapvar_off;
};
struct BuildGroup {
    uint32_t tile, lane, nvert, src0, nsrc, group_index, g, pad;
};
struct BuildBatch {
    const BuildCluster *clusters;
    const BuildGroup *groups;
    const uint8_t *group_ploidy;
    const uint32_t *group_sources, *edges;
    const uint8_t *hap_kmer_mult, *kmer_has_counts, *kmer_counts, *kmer_ic_mult;
    const int32_t *kmer_shared;
    const uint32_t *kv_off;
    const uint16_t *kv_var;
    const uint32_t *kv_bits, *unique_idx, *multi_idx;
    const uint16_t *hap_allele;
    const uint32_t *hapnest_off, *hapnest_idx;
    const uint16_t *var_num_alleles;
    const uint8_t *var_has_dependency;
    const uint32_t *nestdep_cluster, *nestdep_var_off;
    const uint16_t *nestdep_var;
};
template <typename T>
__device__ inline void tput(uint8_t *tile_base, const TileDesc &d, int arr, uint64_t idx, uint32_t lane, T value) {
    reinterpret_cast<T *>(tile_base + d.off[arr])[(idx << d.wsh) + lane] = value;
}
// Workgroups of ONE wavefront for everything a sampler's construction launches: a helper thread builds the next noise chain's sampler while the current
// chain's launch is resident (gibbs_chain_kernel: up to two 256-register wavefronts per SIMD on most SIMDs), and the dispatcher only places a workgroup of
// several wavefronts on a CU that has room on all its SIMDs — measured: a 256-thread kernel on another stream waited for the chain to end, a 64-thread one
// ran next to it.  Hence no hipMemsetAsync for the pool either (the runtime's fill kernel has large workgroups).
__global__ __launch_bounds__(64) void zero_fill_kernel(uint4 *__restrict__ p, uint64_t n16, uint64_t per_wg) {
    // a contiguous stretch per workgroup (a grid-stride loop over a 20 GB pool touched another page every iteration: 115 GB/s)
    const uint64_t a = (uint64_t)blockIdx.x * per_wg, b = a + per_wg < n16 ? a + per_wg : n16;
    for (uint64_t i = a + threadIdx.x; i < b; i += 64u) p[i] = uint4{0u, 0u, 0u, 0u};
}
__global__ __launch_bounds__(64) void copy_words_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 64u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 64u) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void build_tiles_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, BuildBatch b, uint32_t S) {
    const BuildCluster x = b.clusters[blockIdx.x];
    const TileDesc &d = tiles[x.tile];
    uint8_t *tb = pool + d.base;
    const uint32_t l = x.lane, t = threadIdx.x, NT = blockDim.x;
    const uint64_t v = x.v;
    const uint32_t H = x.H, V = x.V, K = x.K, HW = (H + 31) / 32;
    if (t < 8) {
        const uint32_t dims[8] = {H, V, K, x.nu, x.nm, x.nd, x.ne, x.cid};
        tput<uint32_t>(tb, d, A_VDIMS, v * 8 + t, l, dims[t]);
    }
    if (t == 8) tput<uint32_t>(tb, d, A_VDIMS2, v * 2, l, x.A);
    // K x H multiplicities: per-lane contiguous in narrow tiles, interleaved rows otherwise
    const uint8_t *M = b.hap_kmer_mult + x.mult_off;
    for (uint64_t i = t; i < (uint64_t)K * H; i += NT) {
        const uint32_t k = (uint32_t)(i / H), h = (uint32_t)(i - (uint64_t)k * H);
        if (d.mat_width) (tb + d.off[A_M])[((v * d.mat_width + l) * d.Km + k) * d.Hm + h] = M[i];
        else tput<uint8_t>(tb, d, A_M, (v * d.Km + k) * d.Hm + h, l, M[i]);
    }
    const uint32_t e0 = b.kv_off[x.r0];
    for (uint32_t k = t; k < K; k += NT) {
        const uint64_t r = (uint64_t)x.r0 + k;
        tput<uint8_t>(tb, d, A_HASC, v * d.Km + k, l, b.kmer_has_counts[r]);
        for (uint32_t s = 0; s < S; ++s) tput<uint8_t>(tb, d, A_COUNTS, (v * d.Km + k) * S + s, l, b.kmer_counts[r * S + s]);
        tput<uint8_t>(tb, d, A_IC, (v * d.Km + k) * 2, l, b.kmer_ic_mult[2 * r]);
        tput<uint8_t>(tb, d, A_IC, (v * d.Km + k) * 2 + 1, l, b.kmer_ic_mult[2 * r + 1]);
        tput<int32_t>(tb, d, A_SHARED, v * d.Km + k, l, b.kmer_shared[r]);
        tput<uint32_t>(tb, d, A_KVOFF, v * (d.Km + 1) + k, l, b.kv_off[r] - e0);
    }
    const uint32_t nnz = b.kv_off[(uint64_t)x.r0 + K] - e0;
    if (t == 0) tput<uint32_t>(tb, d, A_KVOFF, v * (d.Km + 1) + K, l, nnz);
    for (uint32_t e = t; e < nnz; e += NT) {
        tput<uint16_t>(tb, d, A_KVVAR, v * d.NNZm + e, l, b.kv_var[e0 + e]);
        for (uint32_t w = 0; w < HW; ++w) tput<uint32_t>(tb, d, A_KVBITS, (v * d.NNZm + e) * d.HWm + w, l, b.kv_bits[x.kvb_off + (uint64_t)e * HW + w]);
    }
    const uint32_t hn0 = b.hapnest_off[x.hap_base];
    for (uint32_t h = t; h < H; h += NT) {
        // A_HAPCELL: addHaplotypeKmerStats' source variant (VariantClusterHaplotypes.cpp:332-358) and the allele cell, per (haplotype, variant)
        uint32_t last_non_missing = 0xFFFFu, albase = 0;
        for (uint32_t vv = 0; vv < V; ++vv) {
            const uint32_t al = b.hap_allele[x.hapvar_off + (uint64_t)h * V + vv], na = b.var_num_alleles[x.var_base + vv];
            tput<uint16_t>(tb, d, A_HAPAL, (v * d.Hm + h) * d.Vm + vv, l, (uint16_t)al);
            const bool missing = b.var_has_dependency[x.var_base + vv] && al == na - 1u;
            if (!missing) last_non_missing = vv;
            tput<uint32_t>(tb, d, A_HAPCELL, (v * d.Hm + h) * d.Vm + vv, l, last_non_missing | ((albase + al) << 16));
            albase += na;
        }
        tput<uint32_t>(tb, d, A_HNOFF, v * (d.Hm + 1) + h, l, b.hapnest_off[x.hap_base + h] - hn0);
    }
    const uint32_t hn_n = b.hapnest_off[x.hap_base + H] - hn0;
    if (t == 0) tput<uint32_t>(tb, d, A_HNOFF, v * (d.Hm + 1) + H, l, hn_n);
    for (uint32_t i = t; i < hn_n; i += NT) tput<uint32_t>(tb, d, A_HNIDX, v * d.HNm + i, l, b.hapnest_idx[hn0 + i]);
    if (t == 0) {   // per-variant arrays with a running allele base (a few entries)
        uint32_t acc = 0;
        for (uint32_t vv = 0; vv < V; ++vv) {
            tput<uint16_t>(tb, d, A_VARNA, v * d.Vm + vv, l, b.var_num_alleles[x.var_base + vv]);
            tput<uint8_t>(tb, d, A_VARDEP, v * d.Vm + vv, l, b.var_has_dependency[x.var_base + vv]);
            tput<uint32_t>(tb, d, A_ALBASE, v * (d.Vm + 1) + vv, l, acc);
            acc += b.var_num_alleles[x.var_base + vv];
        }
        tput<uint32_t>(tb, d, A_ALBASE, v * (d.Vm + 1) + V, l, acc);
        const uint32_t ndv0 = b.nestdep_var_off[x.nd0];
        const uint32_t NDm1 = d.NDm > 1 ? d.NDm : 1, NDVm1 = d.NDVm > 1 ? d.NDVm : 1;
        for (uint32_t i = 0; i < x.nd; ++i) {
            tput<uint32_t>(tb, d, A_NDCL, v * NDm1 + i, l, b.nestdep_cluster[x.nd0 + i]);
            tput<uint32_t>(tb, d, A_NDVOFF, v * (d.NDm + 1) + i, l, b.nestdep_var_off[x.nd0 + i] - ndv0);
        }
        tput<uint32_t>(tb, d, A_NDVOFF, v * (d.NDm + 1) + x.nd, l, b.nestdep_var_off[x.nd0 + x.nd] - ndv0);
        for (uint32_t i = 0, n = b.nestdep_var_off[x.nd0 + x.nd] - ndv0; i < n; ++i) tput<uint16_t>(tb, d, A_NDVAR, v * NDVm1 + i, l, b.nestdep_var[ndv0 + i]);
        const uint32_t NEm1 = d.NEm > 1 ? d.NEm : 1;
        for (uint32_t i = 0; i < x.ne; ++i) tput<uint32_t>(tb, d, A_EDGES0, v * NEm1 + i, l, b.edges[x.e0 + i]);
    }
    for (uint32_t i = t; i < x.nu; i += NT) tput<uint32_t>(tb, d, A_UNIQ0, v * d.NUm + i, l, b.unique_idx[x.u0 + i]);
    for (uint32_t i = t; i < x.nm; i += NT) tput<uint32_t>(tb, d, A_MULTI0, v * d.NMm + i, l, b.multi_idx[x.m0 + i]);
}
__global__ __launch_bounds__(64) void build_groups_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, BuildBatch b, uint32_t G, uint32_t S) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= G) return;
    const BuildGroup x = b.groups[i];
    const TileDesc &d = tiles[x.tile];
    uint8_t *tb = pool + d.base;
    tput<uint32_t>(tb, d, A_GDIMS, 0, x.lane, x.nvert);
    tput<uint32_t>(tb, d, A_GDIMS, 1, x.lane, x.nsrc);
    tput<uint32_t>(tb, d, A_GDIMS, 2, x.lane, x.group_index);
    tput<uint32_t>(tb, d, A_GDIMS, 3, x.lane, 1u);
    for (uint32_t q = 0; q < x.nsrc; ++q) tput<uint32_t>(tb, d, A_SOURCES0, q, x.lane, b.group_sources[x.src0 + q]);
    for (uint32_t s = 0; s < S; ++s) tput<uint8_t>(tb, d, A_PLOIDY, s, x.lane, b.group_ploidy[(uint64_t)x.g * S + s]);
}

// ---- host-side tile builder -------------------------------------------------------------------------------------
struct TilePlan {
    TileDesc d;
    uint64_t in_bytes;      // the input arrays occupy [0, in_bytes) of the tile
    uint64_t total_bytes;
};

template <typename T>
inline void put(std::vector<uint8_t> &img, const TileDesc &d, int arr, size_t idx, uint32_t lane, T value) {
    T *p = reinterpret_cast<T *>(img.data() + d.off[arr]) + (idx << d.wsh) + lane;   // rows interleaved over the tile's own width
    *p = value;
}

}  // namespace


// Host -> device copies of a sampler's construction: through the context's pinned arena and copy_words_kernel (one-wavefront workgroups reading host
// memory), not hipMemcpy — the runtime stages small pageable copies through a shader copy with large workgroups, which cannot be placed while a resident noise
// chain fills the SIMDs (the "tables" phase of a construction on the helper thread took 55 - 60 ms, i.e. until the running chain ended, instead of 0.5 ms).
// arena_reserve: at the start of a construction, while the context's stream is idle.  Sizes are multiples of four bytes.
static hipError_t arena_reserve(bt_ctx *ctx, size_t bytes) {
    ctx->pin_used = 0;
    if (ctx->pin_bytes >= bytes) return hipSuccess;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return e;
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    ctx->pin = nullptr;
    ctx->pin_bytes = 0;
    const size_t want = bytes + bytes / 2 + 65536;
    e = hipHostMalloc(reinterpret_cast<void **>(&ctx->pin), want, hipHostMallocDefault);
    if (e == hipSuccess) ctx->pin_bytes = want;
    return e;
}
static hipError_t staged_upload(bt_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    const size_t need = (bytes + 15) & ~(size_t)15;
    if ((bytes & 3u) || ctx->pin_used + need > ctx->pin_bytes) return hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream);   // (not reserved for: the runtime's copy)
    uint8_t *p = ctx->pin + ctx->pin_used;
    ctx->pin_used += need;
    std::memcpy(p, h_src, bytes);
    const uint64_t n = bytes / 4;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)std::min<uint64_t>((n + 63) / 64, 16384)), dim3(64), 0, ctx->stream, reinterpret_cast<uint32_t *>(d_dst), reinterpret_cast<const uint32_t *>(p), n);
    return hipGetLastError();
}

// ---- a batch's flat arrays on the device + what the tile planner needs of it on the host (bt_gibbs_source) ----
// per cluster: its dimensions and where its slices start in the flat arrays
struct ClusterDims {
    uint32_t H, V, K, nnz, nu, nm, nd, ndv, ne, hn, A, cid;
    uint32_t r0, u0, m0, nd0, e0, hap_base, var_base, pad;
    uint64_t mult_off, kvb_off, hapvar_off;
};
struct GroupDims {
    uint32_t c0, nv;          // first cluster (index into the ClusterDims array handed to the planner), number of clusters
    uint32_t src0, nsrc;      // slice of group_sources
    uint32_t num_shared, group_index;
    uint32_t g_abs, pad;      // position of the group in the source batch (group_ploidy row)
};

#ifndef BT_HOT_BUDGET
#define BT_HOT_BUDGET 155648
#endif
constexpr uint32_t kHotBudget = BT_HOT_BUDGET;   // LDS bytes a tile may claim for its hot arrays (larger tiles stay in HBM)
struct FillChunk {
    uint64_t off;      // bytes from the pool
    uint32_t words, pad;
};
struct PrefillItem {
    uint32_t tile, lane, v, pad;
};

struct bt_gibbs {
    bt_ctx *ctx = nullptr;
    GParams P{};
    uint32_t G = 0, C = 0, S = 0, ntiles = 0;
    std::vector<void *> allocs;
    uint64_t device_bytes = 0;
    TileDesc *d_tiles = nullptr;
    uint8_t *d_pool = nullptr;
    uint64_t pool_alloc_bytes = 0;   // size of the allocation behind d_pool (a pool taken over from the context's cache may be larger than pool_bytes)
    bool keep_pool = false;          // a noise driver's sampler: its pool goes to the context's cache when it is destroyed
    bool recycles = false;           // ... and so do its pinned words and class streams (gibbs_create_impl)
    uint64_t pool_bytes = 0;
    ClusterLoc *d_loc = nullptr;
    double *d_lut_g = nullptr, *d_lut_n = nullptr, *d_lgamma = nullptr;
    GParams *d_params = nullptr;
    bool lut_set = false;
    std::vector<TileDesc> tiles;
    std::vector<ClusterLoc> loc;           // per cluster (batch order)
    std::vector<uint32_t> h_A;             // alleles per cluster
    std::vector<uint32_t> group_tile, group_lane, group_nvert;
    // trace
    // Tiles are launched in classes of similar LDS need (a launch has ONE dynamic LDS size: that of its hungriest tile), concurrently on
    // separate streams, so that a few LDS-hungry tiles do not cap the occupancy of all the others
    struct LaunchClass {
        uint32_t lds = 0, split = 1;      // dynamic LDS per workgroup, wavefronts per tile (tile_lane())
        bool simple = false;              // every tile runs simple_sweeps(): launched as gibbs_simple_kernel
        bool hot = false;                 // every tile keeps all vertices' hot arrays in LDS for the launch: sampling operations launched as gibbs_hot_kernel
        bool single = false;              // ... and every group is one cluster without multicluster k-mers: sampling operations launched as gibbs_single_kernel (three wavefronts per SIMD)
        std::vector<uint32_t> tiles;
        uint32_t *d_tiles = nullptr;
        // hot / single classes: the PACKED launch of the sampling operations (bt_gibbs_tile.hpp: BT_PACKED) — pack_wgs workgroups of pack_waves wavefronts, each
        // wavefront a slot (tile, LDS offset) of d_pack, pack_lds bytes of LDS per workgroup
        uint32_t *d_pack = nullptr;
        uint32_t pack_wgs = 0, pack_waves = 1, pack_lds = 0;
        hipStream_t stream = nullptr;     // nullptr: the context's stream
        hipEvent_t done = nullptr;
        hipEvent_t ready = nullptr;       // recorded on the class's stream right before its launch: the next class waits for it (launch(): start order)
        // the tiles' large dense tables of unique-k-mer sums in pieces of <= 256 KB: clearGenotyperCache between two iterations of the noise
        // drivers invalidates them with the whole GPU (nan_fill_kernel) instead of the tile's own lanes
        FillChunk *d_fill = nullptr;
        uint32_t num_fill = 0;
        // (tile, lane, vertex) of every vertex with such a table: ucache_prefill_kernel
        struct PrefillItem *d_prefill = nullptr;
        uint32_t num_prefill = 0;
    };
    bool stepwise_run = false;   // bt_gibbs_run chain by chain (BT_GIBBS_STEPWISE)
    bool noise_in_gibbs_kernel = false;   // BT_GIBBS_NOISE_GLOBAL_ATOMICS: the OP_NOISE branch of gibbs_kernel instead of gibbs_noise_kernel
    bool prefill_armed = false, wide_fill = true;   // wide_fill: BT_GIBBS_NO_WIDE_FILL unset
    std::vector<LaunchClass> classes;      // hungriest first
    hipEvent_t ev_fork = nullptr;
    uint32_t trace_sweeps = 0;
    uint32_t *d_trace = nullptr, *d_trace_counter = nullptr;
    uint32_t *d_wire = nullptr;   // bt_gibbs_result_words' string
    uint64_t wire_cap = 0;
    uint64_t trace_words = 0;
    // bt_gibbs_noise_iteration: pinned staging of the histogram (device -> host) and of the noise table (host -> device), device histogram
    uint64_t *h_pin_hist = nullptr, *d_iter_hist = nullptr;
    double *h_pin_noise = nullptr;
    // bt_gibbs_noise_chain_{begin,step,end}: a chain of a noise driver as one resident launch (bt_noise_chain.hpp)
    std::vector<double> h_lut_n;          // host mirror of d_lut_n
    struct NoiseChainState {
        uint8_t *h_mail = nullptr;        // pinned, fine-grained: [S*256] u64 histogram | [S*256] f64 table | hist_seq, table_seq (a cache line each)
        uint8_t *d_sync = nullptr;        // device: [S*256] u64 histogram | arrived | abort | table_seq copies
        NoiseChainCtl *d_ctl = nullptr;
        NoiseChainCtl ctl{};
        HelpItem *d_help_items = nullptr;       // the help phase (bt_noise_help.hpp): every (tile, lane, vertex) with a large table
        uint32_t *d_help_words = nullptr;       // [0] the unit counter, [64 ..) units finished per tile
        uint32_t *d_tile_units = nullptr;
        uint32_t help_units = 0;
        bool help_built = false;
        uint32_t *h_phase = nullptr;            // BT_NOISE_CHAIN_DEBUG_FLAGS & 2: where every workgroup is (pinned)
        unsigned long long *d_busy = nullptr;   // BT_NOISE_CHAIN_PROF: per tile, the ticks its workgroup worked (the rest of a chain it waited for the others / the host)
        uint32_t num_wgs = 0;   // tiles + helpers
        bool active = false, launched = false;
        bool fallback = false;   // the launch's roll call failed (its workgroups were not resident together): the chain's remaining steps are launches per iteration
        uint32_t n = 0, next = 0;
        uint32_t it_begin = 0, first_collect = 0, lds = 0;   // it_begin = 1: iteration 0 of the chain runs as ordinary launches (whole-GPU table refill), the resident launch starts with iteration 1
    } nc;
};

namespace {

// VariantClusterGenotyper::getNoiseCounts (:757-779) + clearCache for every vertex of every group of a launch class.  Every group adds
// |subset| x S counts to S x 256 bins of which a few dozen are ever hit: tallied with global atomics (the OP_NOISE branch of gibbs_kernel) the
// 90 240 two-haplotype groups of the ten-sample batch spent 10.5 ms per iteration queueing on those words.  Here a workgroup tallies the tiles it
// walks in LDS and adds its non-empty bins once.
__global__ __launch_bounds__(LANES * 8) void gibbs_noise_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg, uint32_t wide,
                                                              unsigned long long *__restrict__ hist, const uint32_t *__restrict__ tile_list, uint32_t ntiles) {
    const GParams BT_CAS &P = *(const GParams BT_CAS *)Pg;
    uint32_t *lh = reinterpret_cast<uint32_t *>(lds_block());
    const uint32_t nb = P.S * 256u;
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    // the LDS bins are 32 bits wide: they are added to the 64-bit histogram before the tiles walked since the last flush could have put 2^31 counts
    // into one of them (a tile adds at most lanes x vertices x subset k-mers to a bin; the bound is the same in every thread of the workgroup)
    uint64_t bound = 0;
    for (uint32_t b = blockIdx.x; b < ntiles; b += gridDim.x) {
        const uint32_t tile = tile_list ? tile_list[b] : b;
        Tile t;
        t.d = (const TileDesc BT_CAS *)&tiles[tile];
        {
            const uint64_t add = (uint64_t)t.d->num_lanes * t.d->nvm * (t.d->NUm > 1 ? t.d->NUm : 1);
            if (bound + add > (1ull << 31)) {
                __syncthreads();
                for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) {
                    const uint32_t n = lh[i];
                    if (n) atomicAdd(&hist[i], (unsigned long long)n);
                    lh[i] = 0;
                }
                __syncthreads();
                bound = 0;
            }
            bound += add;
        }
        if (!tile_thread_active(t.d->split, t.d->copies)) continue;
        t.base = (uint8_t BT_GAS *)(pool + t.d->base);
        t.lane = tile_lane(t.d->split, t.d->copies);
        t.plane = t.lane + t.d->pool_lane0;
        t.wsh = t.d->wsh;
        t.part = tile_part(t.d->copies);
        t.copies = t.d->copies;
        t.hot = nullptr;
        t.lds0 = 0;
        t.resident = 0xFFFFFFFFu;
        TPtr<uint32_t> gd = t.arr<uint32_t>(A_GDIMS);
        if (!gd[3]) continue;   // padding lane of the last tile
        const uint32_t nvert = gd[0];
        for (uint32_t v = 0; v < nvert; ++v) {
            const Vx c = make_vx(t, v);
            const uint32_t nsu = c.sc()[SC_NSUB_U];
            TPtr<uint32_t> usub = c.usub();
            for (uint32_t s = 0; s < P.S; ++s) {
                const uint16_t h1 = c.dip()[2 * s], h2 = c.dip()[2 * s + 1];
                for (uint32_t i = t.part; i < nsu; i += t.copies) {   // the copies of a narrow tile's group share the k-mers of the subset (a tally: any order)
                    const uint32_t k = usub[i];
                    if (unique_mult(c, k, h1, h2, P.gender[s]) == 0) {
                        const uint32_t cnt = c.has_counts(k) ? c.count(k, s) : 0;
                        atomicAdd(&lh[s * 256u + cnt], 1u);
                    }
                }
            }
            if (t.part == 0) cache_clear(c, P, false, wide != 0);   // (the other copies read nothing this touches)
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) {
        const uint32_t n = lh[i];
        if (n) atomicAdd(&hist[i], (unsigned long long)n);
    }
}

__global__ __launch_bounds__(256) void nan_fill_kernel(uint8_t *__restrict__ pool, const FillChunk *__restrict__ chunks) {
    const FillChunk c = chunks[blockIdx.x];
    unsigned long long *p = reinterpret_cast<unsigned long long *>(pool + c.off);
    for (uint32_t i = threadIdx.x; i < c.words; i += 256) p[i] = 0x7FF8000000000000ULL;   // "not computed" (unique_log_prob tests v == v)
}

// The sums a sweep of the noise drivers will ask for, computed by the whole GPU before the sweep starts.  clearGenotyperCache empties the
// tables after every iteration, so each sweep evaluates its candidates' sums over the k-mer subset again; inside the sweep that work is bound to the
// lanes of the group's own tile (an iteration then lasts as long as the largest group).  The candidates of a vertex visit are the pairs (and, for a
// nested vertex with one parental copy, the singles) of the haplotypes with a non-zero frequency, and the frequencies of a vertex only change at the
// end of its own visit: the set is known when the sweep starts.  One workgroup per (vertex of a group, sample); every entry is the sum
// unique_log_prob_block computes, stored where the sweep looks it up; whatever is missing is still computed on demand.
constexpr uint32_t kPrefillMaxH = 2048;
__global__ __launch_bounds__(256) void ucache_prefill_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg,
                                                              const PrefillItem *__restrict__ items) {
    __shared__ uint16_t nzl[kPrefillMaxH];
    __shared__ uint32_t nnz_sh;
    const PrefillItem it = items[blockIdx.x];
    const uint32_t s = blockIdx.y;
    const GParams BT_CAS &P = *(const GParams BT_CAS *)Pg;
    Tile t;
    t.d = (const TileDesc BT_CAS *)&tiles[it.tile];
    t.base = (uint8_t BT_GAS *)(pool + t.d->base);
    t.lane = it.lane;
    t.plane = it.lane + t.d->pool_lane0;
    t.wsh = t.d->wsh;
    t.part = 0;
    t.copies = t.d->copies;
    t.hot = nullptr;
    t.lds0 = 0;
    t.resident = 0xFFFFFFFFu;
    const Vx c = make_vx(t, it.v);
    SPtrF<uint32_t, LANES> sc = c.sc();
    if (!sc[SC_CONSTRUCTED] || sc[SC_UC_DIRTY] || c.H > kPrefillMaxH) return;   // (a table to be rebuilt whole is rebuilt by the sweep)
    if (threadIdx.x == 0) nnz_sh = 0;
    __syncthreads();
    {
        SPtrF<uint8_t, LANES> nz = c.nz();
        for (uint32_t h = threadIdx.x; h < c.H; h += blockDim.x)
            if (nz[h]) nzl[atomicAdd(&nnz_sh, 1u)] = (uint16_t)h;   // (any order: the entries are independent)
    }
    __syncthreads();
    const uint32_t nnz = nnz_sh, pairs = nnz * (nnz + 1) / 2, total = pairs + (t.d->nvm > 1 ? nnz : 0u);
    const uint32_t nsub_u = sc[SC_NSUB_U];
    const TileDesc BT_CAS &d = c.d();
    const Vx::UCPtr uc = c.ucache();
    for (uint32_t base = EVB * threadIdx.x; base < total; base += EVB * blockDim.x) {
        uint16_t ha[EVB], hb[EVB];
        bool need[EVB], any = false;
        uint32_t a = 0, b = 0;
        if (base < pairs) {   // row a of the triangle holds nnz - a pairs
            uint32_t q = base, left = nnz;
            while (q >= left) {
                q -= left;
                ++a;
                --left;
            }
            b = a + q;
        }
#pragma unroll
        for (uint32_t q = 0; q < EVB; ++q) {
            const uint32_t i = base + q;
            need[q] = false;
            ha[q] = 0;
            hb[q] = NOHAP;
            if (i >= total) continue;
            if (i < pairs) {
                uint16_t x = nzl[a], y = nzl[b];
                if (x > y) {   // (the list is unordered here; dip_index wants h1 <= h2)
                    const uint16_t z = x;
                    x = y;
                    y = z;
                }
                ha[q] = x;
                hb[q] = y;
                if (++b == nnz) {
                    ++a;
                    b = a;
                }
            } else {
                ha[q] = nzl[i - pairs];
            }
            const double v = uc[(uint32_t)s * d.Dcm + dip_index(c, ha[q], hb[q])];
            need[q] = !(v == v);
            any = any || need[q];
        }
        if (any) {
            double out[EVB];
            unique_log_prob_block(c, P, s, ha, hb, need, nsub_u, out);
        }
    }
}

// every call that enqueues on the sampler's stream would queue behind a resident chain's launch — which is itself waiting for the host
inline bool chain_in_flight(const bt_gibbs *g) { return g->nc.active && g->nc.launched; }

int launch(bt_gibbs *g, int op, uint32_t a0, uint32_t a1, unsigned long long *hist) {
    if (!g->lut_set && (op == OP_RUN || op == OP_SWEEP)) return fail("bt_gibbs: count-model LUTs not set (bt_gibbs_set_lut)");
    if (g->nc.active && g->nc.launched) return fail("bt_gibbs: a resident noise chain is in progress (bt_gibbs_noise_chain_end)");
    BT_HIP(hipSetDevice(g->ctx->device));
    if (op == OP_NOISE && !g->wide_fill) a0 = 0;
    if (op == OP_INIT_CHAIN) a1 = g->wide_fill ? 1u : 0u;
    const bool wide = (op == OP_NOISE && a0 != 0) || (op == OP_INIT_CHAIN && a1 != 0);
    TraceCfg tr{g->trace_sweeps, g->d_trace_counter, g->d_trace};
    const bool fork = g->classes.size() > 1;
    if (fork) BT_HIP(hipEventRecord(g->ev_fork, g->ctx->stream));
    // BT_GIBBS_ORDER (tuning): "hot_first" = the class of the two-haplotype tiles (gibbs_simple_kernel, on the context's stream) starts when the other classes'
    // launches are through; unset = all classes start together
    const char *order_env = getenv("BT_GIBBS_ORDER");
    const bool hot_first = order_env && std::strcmp(order_env, "hot_first") == 0 && (op == OP_RUN || op == OP_SWEEP);
    // START ORDER.  The classes are listed hungriest first — their tiles are the long ones — and a schedule is shortest when the long tiles are resident
    // first.  When the launches are enqueued while the GPU is still busy (a KMC scan, the previous schedule) all classes become runnable at the same instant
    // and the dispatcher deals the first wave slots round robin: the bench batch then took 4.34 s instead of 3.83 s.  So a class's stream waits for the
    // previous class's stream to have reached its launch (an event recorded right before it): the classes become runnable in list order, microseconds apart,
    // whatever the host's timing.
    const bool ordered = (op == OP_RUN || op == OP_SWEEP) && !getenv("BT_GIBBS_UNORDERED_START");
    const bt_gibbs::LaunchClass *prev = nullptr;
    for (auto &c : g->classes) {
        hipStream_t st = c.stream ? c.stream : g->ctx->stream;
        if (c.stream) BT_HIP(hipStreamWaitEvent(st, g->ev_fork, 0));
        if (ordered && prev) BT_HIP(hipStreamWaitEvent(st, prev->ready, 0));
        if (hot_first && !c.stream)
            for (auto &o : g->classes)
                if (o.stream) BT_HIP(hipStreamWaitEvent(st, o.done, 0));
        if (op == OP_SWEEP && g->prefill_armed && c.num_prefill) {
            // one wavefront per (vertex, sample) when there are many of them: between two iterations a vertex has a handful of candidates, and a
            // 256-thread workgroup with one busy wavefront holds four wavefront slots for the length of a subset walk
            const unsigned threads = (uint64_t)c.num_prefill * g->S >= 4096 ? 64u : 256u;
            hipLaunchKernelGGL(ucache_prefill_kernel, dim3(c.num_prefill, g->S), dim3(threads), 0, st, (const TileDesc *)g->d_tiles, g->d_pool, (const GParams *)g->d_params,
                               (const PrefillItem *)c.d_prefill);
            BT_CHECK_LAUNCH();
        }
        if (ordered) {
            BT_HIP(hipEventRecord(c.ready, st));
            prev = &c;
        }
        if (op == OP_NOISE && (size_t)g->S * 1024 <= kHotBudget && !g->noise_in_gibbs_kernel) {
            const unsigned grid = (unsigned)std::min<size_t>(c.tiles.size(), (size_t)g->ctx->num_cu * 4);
            hipLaunchKernelGGL(gibbs_noise_kernel, dim3(grid), dim3(LANES * c.split), (size_t)g->S * 1024, st, (const TileDesc *)g->d_tiles, g->d_pool, (const GParams *)g->d_params, a0, hist,
                               (const uint32_t *)c.d_tiles, (uint32_t)c.tiles.size());
        } else if (c.simple && (op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN))
            BT_HIP(launch_gibbs_simple_kernel((unsigned)c.tiles.size(), c.lds, st, g->d_tiles, g->d_pool, g->d_params, op, a0, a1, hist, tr, (const uint32_t *)c.d_tiles));
        else if (c.single && (op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN))
            BT_HIP(launch_gibbs_single_kernel(c.pack_wgs, LANES * c.pack_waves, c.pack_lds, st, g->d_tiles, g->d_pool, g->d_params, op, a0, a1, hist, tr, (const uint32_t *)c.d_pack));
        else if (c.hot && (op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN))
            BT_HIP(launch_gibbs_hot_kernel(c.pack_wgs, LANES * c.pack_waves, c.pack_lds, st, g->d_tiles, g->d_pool, g->d_params, op, a0, a1, hist, tr, (const uint32_t *)c.d_pack));
        else
            hipLaunchKernelGGL(gibbs_kernel, dim3((unsigned)c.tiles.size()), dim3(LANES * c.split), c.lds, st, g->d_tiles, g->d_pool, g->d_params, op, a0, a1, hist, tr,
                               (const uint32_t *)c.d_tiles);
        BT_CHECK_LAUNCH();
        if (wide && c.num_fill) {
            hipLaunchKernelGGL(nan_fill_kernel, dim3(c.num_fill), dim3(256), 0, st, g->d_pool, (const FillChunk *)c.d_fill);
            BT_CHECK_LAUNCH();
        }
        if (c.stream) BT_HIP(hipEventRecord(c.done, st));
    }
    for (auto &c : g->classes)
        if (c.stream) BT_HIP(hipStreamWaitEvent(g->ctx->stream, c.done, 0));
    g->prefill_armed = wide;   // the sweep that follows a chain start or a cache-clearing noise count starts with the prefill
    return BT_OK;
}

inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
// dynamic LDS a workgroup of this tile needs: one block per vertex when all vertices are resident
inline uint32_t tile_lds_bytes(const TileDesc &d) { return d.lds_all ? d.hot_bytes * d.nvm : d.hot_bytes; }
// a cluster's [S][D] tables are dense up to 256 MB per tile (64 MB when the batch would not fit the GPU otherwise): a hashed table far
// smaller than the set of live (sample, diplotype) pairs thrashes (256 candidates x 10 samples: 12x slower than dense)
constexpr uint32_t kMinTileWidth = 4;           // groups per wavefront of the narrowest tiles (the other lanes run copies)
constexpr uint32_t kLightLds = 24576;            // tiles above this are "heavy" (they get fewer wavefronts per tile)
// upper LDS bounds of the launch classes: at most four, the streams of more classes than hardware queues would run one after another
const uint32_t kClassLds[] = {16384, 32768, 0xFFFFFFFFu};   // + one class of the simple tiles (two-haplotype clusters), launched as gibbs_simple_kernel

// element sizes per array, in TileArr order
const uint32_t kElemSize[A_COUNT] = {
    /*A_M*/ 1, /*HASC*/ 1, /*COUNTS*/ 1, /*IC*/ 1, /*SHARED*/ 4, /*KVOFF*/ 4, /*KVVAR*/ 2, /*KVBITS*/ 4, /*HAPAL*/ 2, /*HAPCELL*/ 4, /*HNOFF*/ 4, /*HNIDX*/ 4, /*VARNA*/ 2,
    /*VARDEP*/ 1, /*ALBASE*/ 4, /*NDCL*/ 4, /*NDVOFF*/ 4, /*NDVAR*/ 2, /*UNIQ0*/ 4, /*MULTI0*/ 4, /*EDGES0*/ 4, /*VDIMS*/ 4, /*VDIMS2*/ 4, /*GDIMS*/ 4,
    /*SOURCES0*/ 4, /*PLOIDY*/ 1,
    /*MT*/ 4, /*FNDSAVED*/ 8, /*SPARSITY*/ 8, /*UNIQ*/ 4, /*MULTI*/ 4, /*USUB*/ 4, /*MSUB*/ 4, /*SMM*/ 1, /*DIP*/ 2, /*FREQ*/ 8, /*OBS*/ 4, /*NZ*/ 1,
    /*ZHDR*/ 4, /*ZBKT*/ 4, /*PHDR*/ 4, /*PBKT*/ 4, /*UNEXT*/ 4, /*HVCOUNT*/ 4, /*UCACHE*/ 8, /*UCTAG*/ 4, /*CUM*/ 8, /*NZLIST*/ 2, /*SIMPLEX*/ 8,
    /*SCACHE*/ 8, /*SCLEN*/ 4, /*KSC*/ 8, /*KSCUPD*/ 1, /*DIPKEYS*/ 4, /*DIPFREQ*/ 4, /*ASTATS*/ 8, /*NESTPL*/ 1, /*NESTN*/ 1, /*NESTST*/ 8, /*KSCKEY*/ 4, /*KSCDATA*/ 8, /*EVLOG*/ 4, /*EVN*/ 1, /*NVER*/ 4, /*PENDNEST*/ 8, /*SC*/ 4,
    /*EDGES*/ 4, /*COVER*/ 1, /*MCACHE*/ 8, /*MCTAG*/ 4, /*MCGEN*/ 4, /*MGEN*/ 4, /*OTH*/ 1, /*SUBM*/ 1, /*SUBCNT*/ 1, /*SUBIC*/ 1, /*SKVOFF*/ 4, /*SKVVAR*/ 2, /*SKVBITS*/ 4, /*KSCTMP*/ 8, /*LOGF*/ 8, /*PEND*/ 4, /*PENDDIP*/ 2, /*PENDVALID*/ 1, /*SOURCES*/ 4, /*STACK*/ 4, /*BRNG*/ 4, /*SHMULT*/ 1, /*MSUBM*/ 1, /*MSUBC*/ 1, /*MSUBIC*/ 1, /*MSUBSH*/ 4, /*RING*/ 4};

}  // namespace

extern "C" {

struct bt_gibbs_source {
    bt_ctx *ctx = nullptr;
    uint32_t S = 0, G = 0, C = 0;
    std::vector<ClusterDims> cd;   // every cluster of the batch
    std::vector<GroupDims> gd;     // every group (c0 = index into cd)
    BuildBatch dev{};              // the flat arrays on the device (clusters / groups are per sampler)
    std::vector<void *> allocs;
    uint64_t device_bytes = 0, flat_bytes = 0;
    bool uploaded = false;
};

// a malformed batch must not make the tile builder read out of bounds: offsets monotone, indices in range
static int validate_batch(const bt_gibbs_batch *B) {
    const uint32_t G = B->num_groups, C = B->num_clusters;
    if (G == 0 || C == 0) return fail("bt_gibbs_create: empty batch");
    if (B->group_cluster_off[G] != C) return fail("bt_gibbs_create: group_cluster_off[G] != num_clusters");
    {
        auto monotone = [](const uint32_t *off, uint64_t n) {
            for (uint64_t i = 0; i < n; ++i)
                if (off[i + 1] < off[i]) return false;
            return true;
        };
        if (B->group_cluster_off[0] != 0 || !monotone(B->group_cluster_off, G) || !monotone(B->group_source_off, G) || !monotone(B->edge_off, C) || !monotone(B->kmer_off, C) ||
            !monotone(B->unique_off, C) || !monotone(B->multi_off, C) || !monotone(B->nestdep_off, C) || !monotone(B->kv_off, B->kmer_off[C]) ||
            !monotone(B->nestdep_var_off, B->nestdep_off[C]))
            return fail("bt_gibbs_create: an offset array of the batch is not monotone");
        uint64_t hap_at = 0, var_at = 0, hv_at = 0;
        for (uint32_t gi = 0; gi < G; ++gi) {
            const uint32_t c0 = B->group_cluster_off[gi], c1 = B->group_cluster_off[gi + 1], nv = c1 - c0;
            if (nv == 0) return fail("bt_gibbs_create: group without clusters");
            for (uint32_t i = B->group_source_off[gi]; i < B->group_source_off[gi + 1]; ++i)
                if (B->group_sources[i] >= nv) return fail("bt_gibbs_create: source vertex outside its group");
            for (uint32_t c = c0; c < c1; ++c) {
                const uint32_t H = B->num_haplotypes[c], V = B->num_variants[c], K = B->kmer_off[c + 1] - B->kmer_off[c];
                for (uint32_t i = B->edge_off[c]; i < B->edge_off[c + 1]; ++i)
                    if (B->edges[i] >= nv) return fail("bt_gibbs_create: edge target outside its group");
                for (uint32_t i = B->unique_off[c]; i < B->unique_off[c + 1]; ++i)
                    if (B->unique_idx[i] >= K) return fail("bt_gibbs_create: unique k-mer index outside its cluster");
                for (uint32_t i = B->multi_off[c]; i < B->multi_off[c + 1]; ++i)
                    if (B->multi_idx[i] >= K) return fail("bt_gibbs_create: multicluster k-mer index outside its cluster");
                for (uint32_t r = B->kmer_off[c]; r < B->kmer_off[c + 1]; ++r) {
                    if (B->kmer_shared[r] >= (int32_t)B->group_num_shared[gi]) return fail("bt_gibbs_create: shared k-mer record outside its group");
                    for (uint32_t e = B->kv_off[r]; e < B->kv_off[r + 1]; ++e)
                        if (B->kv_var[e] >= V) return fail("bt_gibbs_create: k-mer overlaps a variant outside its cluster");
                }
                for (uint64_t i = 0; i < (uint64_t)H * V; ++i)
                    if (B->hap_allele[hv_at + i] >= B->var_num_alleles[var_at + i % V]) return fail("bt_gibbs_create: haplotype allele index outside its variant");
                if (B->hapnest_off[hap_at + H] < B->hapnest_off[hap_at]) return fail("bt_gibbs_create: hapnest_off is not monotone");
                hap_at += H;
                var_at += V;
                hv_at += (uint64_t)H * V;
            }
        }
    }
    return BT_OK;
}

// dimensions + slice starts of every cluster and group of a batch; upload: the flat arrays go to the device (they stay there until the source is destroyed)
static int source_build(bt_ctx *ctx, uint32_t S, const bt_gibbs_batch *B, bool upload, bt_gibbs_source **out) {
    const uint32_t G = B->num_groups, C = B->num_clusters;
    std::unique_ptr<bt_gibbs_source> src(new bt_gibbs_source());
    src->ctx = ctx;
    src->S = S;
    src->G = G;
    src->C = C;
    src->cd.resize(C);
    src->gd.resize(G);
    uint64_t mult = 0, kvb = 0, hapvar = 0;
    uint32_t hap = 0, var = 0;
    for (uint32_t c = 0; c < C; ++c) {
        ClusterDims &x = src->cd[c];
        x.H = B->num_haplotypes[c];
        x.V = B->num_variants[c];
        x.r0 = B->kmer_off[c];
        x.K = B->kmer_off[c + 1] - x.r0;
        if (x.H < 1 || x.H >= 65535 || x.V < 1) return fail("bt_gibbs_create: cluster with no haplotype / variant or too many haplotypes");
        x.nnz = B->kv_off[B->kmer_off[c + 1]] - B->kv_off[B->kmer_off[c]];
        x.u0 = B->unique_off[c];
        x.nu = B->unique_off[c + 1] - x.u0;
        x.m0 = B->multi_off[c];
        x.nm = B->multi_off[c + 1] - x.m0;
        x.nd0 = B->nestdep_off[c];
        x.nd = B->nestdep_off[c + 1] - x.nd0;
        x.ndv = B->nestdep_var_off[B->nestdep_off[c + 1]] - B->nestdep_var_off[B->nestdep_off[c]];
        x.e0 = B->edge_off[c];
        x.ne = B->edge_off[c + 1] - x.e0;
        x.cid = B->cluster_idx[c];
        x.hap_base = hap;
        x.var_base = var;
        x.hn = B->hapnest_off[hap + x.H] - B->hapnest_off[hap];
        x.mult_off = mult;
        x.kvb_off = kvb;
        x.hapvar_off = hapvar;
        x.pad = 0;
        uint32_t A = 0;
        for (uint32_t v = 0; v < x.V; ++v) A += B->var_num_alleles[var + v];
        x.A = A;
        if (x.V >= 0xFFFFu || A >= 65536u) return fail("bt_gibbs_create: cluster with 65535 or more variants / 65536 or more alleles");   // A_HAPCELL packs a variant index and an allele cell into 16 bits each
        mult += (uint64_t)x.K * x.H;
        kvb += (uint64_t)x.nnz * ((x.H + 31) / 32);
        hapvar += (uint64_t)x.H * x.V;
        hap += x.H;
        var += x.V;
    }
    for (uint32_t gi = 0; gi < G; ++gi)
        src->gd[gi] = GroupDims{B->group_cluster_off[gi], B->group_cluster_off[gi + 1] - B->group_cluster_off[gi], B->group_source_off[gi], B->group_source_off[gi + 1] - B->group_source_off[gi],
                                B->group_num_shared[gi], B->group_index[gi], gi, 0};
    const uint64_t R = B->kmer_off[C], NNZ = B->kv_off[R], ND = B->nestdep_off[C];
    src->flat_bytes = (uint64_t)G * S + (uint64_t)B->group_source_off[G] * 4 + (uint64_t)B->edge_off[C] * 4 + mult + R * (S + 11ull) + 4 + NNZ * 2 + kvb * 4 +
                      ((uint64_t)B->unique_off[C] + B->multi_off[C]) * 4 + hapvar * 2 + ((uint64_t)hap + 1) * 4 + (uint64_t)B->hapnest_off[hap] * 4 + (uint64_t)var * 3 + ND * 8 + 4 +
                      (uint64_t)B->nestdep_var_off[ND] * 2;
    if (upload) {
        BT_HIP(hipSetDevice(ctx->device));
        hipError_t e = hipSuccess;
        auto up = [&](const void *h, size_t bytes, const void **d_out) -> hipError_t {
            void *d = nullptr;
            hipError_t er = hipMalloc(&d, std::max<size_t>(bytes, 16));
            if (er != hipSuccess) return er;
            src->allocs.push_back(d);
            src->device_bytes += std::max<size_t>(bytes, 16);
            *d_out = d;
            return bytes ? hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
        };
        BuildBatch &bb = src->dev;
#define BT_UP(field, ptr, bytes) \
    if (e == hipSuccess) e = up(ptr, bytes, reinterpret_cast<const void **>(&bb.field))
        BT_UP(group_ploidy, B->group_ploidy, (size_t)G * S);
        BT_UP(group_sources, B->group_sources, (size_t)B->group_source_off[G] * 4);
        BT_UP(edges, B->edges, (size_t)B->edge_off[C] * 4);
        BT_UP(hap_kmer_mult, B->hap_kmer_mult, (size_t)mult);
        BT_UP(kmer_has_counts, B->kmer_has_counts, (size_t)R);
        BT_UP(kmer_counts, B->kmer_counts, (size_t)R * S);
        BT_UP(kmer_ic_mult, B->kmer_ic_mult, (size_t)R * 2);
        BT_UP(kmer_shared, B->kmer_shared, (size_t)R * 4);
        BT_UP(kv_off, B->kv_off, (size_t)(R + 1) * 4);
        BT_UP(kv_var, B->kv_var, (size_t)NNZ * 2);
        BT_UP(kv_bits, B->kv_bits, (size_t)kvb * 4);
        BT_UP(unique_idx, B->unique_idx, (size_t)B->unique_off[C] * 4);
        BT_UP(multi_idx, B->multi_idx, (size_t)B->multi_off[C] * 4);
        BT_UP(hap_allele, B->hap_allele, (size_t)hapvar * 2);
        BT_UP(hapnest_off, B->hapnest_off, ((size_t)hap + 1) * 4);
        BT_UP(hapnest_idx, B->hapnest_idx, (size_t)B->hapnest_off[hap] * 4);
        BT_UP(var_num_alleles, B->var_num_alleles, (size_t)var * 2);
        BT_UP(var_has_dependency, B->var_has_dependency, (size_t)var);
        BT_UP(nestdep_cluster, B->nestdep_cluster, (size_t)ND * 4);
        BT_UP(nestdep_var_off, B->nestdep_var_off, (size_t)(ND + 1) * 4);
        BT_UP(nestdep_var, B->nestdep_var, (size_t)B->nestdep_var_off[ND] * 2);
#undef BT_UP
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);   // (the caller's arrays may go away)
        if (e != hipSuccess || e2 != hipSuccess) {
            for (void *d : src->allocs) (void)hipFree(d);
            return fail(std::string("bt_gibbs: uploading the batch: ") + hipGetErrorString(e != hipSuccess ? e : e2));
        }
        src->uploaded = true;
    }
    *out = src.release();
    return BT_OK;
}

static int gibbs_create_impl(const bt_gibbs_source *src, bt_ctx *ctx, const bt_gibbs_params *params, const uint32_t *group_ids, uint32_t num_ids, bt_gibbs **out, uint64_t *plan_bytes);

int bt_gibbs_source_create(bt_ctx *ctx, uint32_t num_samples, const bt_gibbs_batch *B, bt_gibbs_source **out) {
    if (!ctx || !B || !out) return fail("bt_gibbs_source_create: null argument");
    if (num_samples < 1 || num_samples > 30) return fail("bt_gibbs_source_create: number of samples must be in 1..30");
    const int rc = validate_batch(B);
    if (rc != BT_OK) return rc;
    return source_build(ctx, num_samples, B, true, out);
}

int bt_gibbs_source_destroy(bt_gibbs_source *src) {
    if (!src) return BT_OK;
    (void)hipSetDevice(src->ctx->device);
    for (void *d : src->allocs) (void)hipFree(d);
    delete src;
    return BT_OK;
}

int bt_gibbs_source_device_bytes(bt_gibbs_source *src, uint64_t *bytes) {
    if (!src || !bytes) return fail("bt_gibbs_source_device_bytes: null argument");
    *bytes = src->device_bytes;
    return BT_OK;
}

int bt_gibbs_create_from_source(bt_gibbs_source *src, bt_ctx *ctx, const bt_gibbs_params *params, const uint32_t *group_ids, uint32_t num_groups, bt_gibbs **out) {
    if (!src || !params || !out) return fail("bt_gibbs_create_from_source: null argument");
    if (!src->uploaded) return fail("bt_gibbs_create_from_source: the source holds no device arrays");
    return gibbs_create_impl(src, ctx, params, group_ids, num_groups, out, nullptr);
}

int bt_gibbs_state_bytes_from_source(bt_gibbs_source *src, const bt_gibbs_params *params, const uint32_t *group_ids, uint32_t num_groups, uint64_t *bytes) {
    if (!src || !params || !bytes) return fail("bt_gibbs_state_bytes_from_source: null argument");
    return gibbs_create_impl(src, nullptr, params, group_ids, num_groups, nullptr, bytes);
}

int bt_gibbs_create(bt_ctx *ctx, const bt_gibbs_params *params, const bt_gibbs_batch *B, bt_gibbs **out) {
    if (!out) return fail("bt_gibbs_create: null argument");
    if (!ctx || !params || !B) return fail("bt_gibbs_create: null argument");
    if (params->num_samples < 1 || params->num_samples > 30) return fail("bt_gibbs_create: number of samples must be in 1..30");   // main.cpp:72
    int rc = validate_batch(B);
    if (rc != BT_OK) return rc;
    bt_gibbs_source *src = nullptr;
    rc = source_build(ctx, params->num_samples, B, true, &src);
    if (rc != BT_OK) return rc;
    rc = gibbs_create_impl(src, nullptr, params, nullptr, 0, out, nullptr);
    bt_gibbs_source_destroy(src);   // (the sampler holds its tiles; the flat arrays were only the builder's input)
    return rc;
}

int bt_gibbs_state_bytes(bt_ctx *ctx, const bt_gibbs_params *params, const bt_gibbs_batch *B, uint64_t *bytes) {
    if (!bytes) return fail("bt_gibbs_state_bytes: null argument");
    if (!ctx || !params || !B) return fail("bt_gibbs_create: null argument");
    if (params->num_samples < 1 || params->num_samples > 30) return fail("bt_gibbs_create: number of samples must be in 1..30");
    int rc = validate_batch(B);
    if (rc != BT_OK) return rc;
    bt_gibbs_source *src = nullptr;
    rc = source_build(ctx, params->num_samples, B, false, &src);
    if (rc != BT_OK) return rc;
    rc = gibbs_create_impl(src, nullptr, params, nullptr, 0, nullptr, bytes);
    bt_gibbs_source_destroy(src);
    return rc;
}

// plan_bytes != nullptr: lay the selected groups out only; *plan_bytes = device bytes a sampler over them would allocate.
// group_ids: the groups of the source the sampler runs (in this order; nullptr: all of them)
static int gibbs_create_impl(const bt_gibbs_source *src, bt_ctx *ctx, const bt_gibbs_params *params, const uint32_t *group_ids, uint32_t num_ids, bt_gibbs **out, uint64_t *plan_bytes) {
    // A noise driver's chain samplers (bt_gibbs_create_from_source with noise seeding: one sampler per chain over a subset of the same unit) hand their pool,
    // pinned words and class streams on to the next one through the context.  Every other sampler allocates and releases its own: a default-mode sampler
    // on class streams taken over from its predecessor ran a ten-sample schedule in 17.1 s instead of 13.1 s (round 5, bench sub-record; not understood).
    const bool recycles = ctx != nullptr && params->noise_seeding != 0 && !ctx_caches_off() && !getenv("BT_GIBBS_NO_POOL_CACHE");
    if (!ctx) ctx = src->ctx;
    if (ctx->device != src->ctx->device) return fail("bt_gibbs_create_from_source: the context is on another device than the source");
    const uint32_t S = params->num_samples;
    if (S != src->S) return fail("bt_gibbs_create: the sampler's number of samples differs from its source's");
    if (!params->gender) return fail("bt_gibbs_create: gender array missing");
    // the planner's view: the selected groups and their clusters, compact
    std::vector<ClusterDims> cd_sel;
    std::vector<GroupDims> gd_sel;
    const ClusterDims *cd = src->cd.data();
    const GroupDims *gd = src->gd.data();
    uint32_t G = src->G, C = src->C;
    if (group_ids) {
        G = num_ids;
        gd_sel.resize(G);
        uint32_t nc = 0;
        for (uint32_t i = 0; i < G; ++i) {
            if (group_ids[i] >= src->G) return fail("bt_gibbs_create_from_source: group index outside the source");
            nc += src->gd[group_ids[i]].nv;
        }
        cd_sel.reserve(nc);
        for (uint32_t i = 0; i < G; ++i) {
            GroupDims x = src->gd[group_ids[i]];
            const uint32_t c0 = x.c0;
            x.c0 = (uint32_t)cd_sel.size();
            for (uint32_t c = 0; c < x.nv; ++c) cd_sel.push_back(src->cd[c0 + c]);
            gd_sel[i] = x;
        }
        C = (uint32_t)cd_sel.size();
        cd = cd_sel.data();
        gd = gd_sel.data();
    }
    if (G == 0 || C == 0) return fail("bt_gibbs_create: empty batch");
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
#define BT_TRYHIP(x)                                                     \
    do {                                                                 \
        hipError_t _e = (x);                                             \
        if (_e != hipSuccess) {                                          \
            bt_gibbs_destroy(g);                                         \
            return fail(std::string(#x) + ": " + hipGetErrorString(_e)); \
        }                                                                \
    } while (0)

    g->h_A.assign(C, 0);
    uint64_t flat_sel = 0;   // bytes of the selected groups' slices of the flat arrays (bt_gibbs_state_bytes: what a create from a host batch uploads next to the pool)
    for (uint32_t c = 0; c < C; ++c) {
        g->h_A[c] = cd[c].A;
        flat_sel += (uint64_t)cd[c].K * cd[c].H + (uint64_t)cd[c].K * (S + 11ull) + (uint64_t)cd[c].nnz * (2 + 4 * ((cd[c].H + 31) / 32)) + ((uint64_t)cd[c].nu + cd[c].nm) * 4 +
                    (uint64_t)cd[c].H * cd[c].V * 2 + (uint64_t)cd[c].H * 4 + (uint64_t)cd[c].hn * 4;
    }

    // BT_GIBBS_DEBUG: where a construction spends its time
    const bool timing = getenv("BT_GIBBS_DEBUG") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    std::string t_text;
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof buf, " %s %.1f ms;", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_text += buf;
        t_prev = now;
    };
    // ---- order groups by shape (vertices, haplotypes, k-mers: descending) and cut into tiles of 64 ----
    struct GShape {
        uint32_t nv, Hmax, Kmax, g;
    };
    std::vector<GShape> shapes(G);
    for (uint32_t gi = 0; gi < G; ++gi) {
        GShape s{gd[gi].nv, 0, 0, gi};
        for (uint32_t c = gd[gi].c0; c < gd[gi].c0 + gd[gi].nv; ++c) {
            s.Hmax = std::max(s.Hmax, cd[c].H);
            s.Kmax = std::max(s.Kmax, cd[c].K);
        }
        shapes[gi] = s;
    }
    std::stable_sort(shapes.begin(), shapes.end(), [](const GShape &a, const GShape &b) {
        if (a.nv != b.nv) return a.nv > b.nv;
        if (a.Hmax != b.Hmax) return a.Hmax > b.Hmax;
        return a.Kmax > b.Kmax;
    });
    // Tiles hold up to 64 groups (one per lane).  The few most expensive groups of a batch (nested groups, clusters with many
    // haplotypes) run thousands of sequential sweeps each and would otherwise occupy a handful of wavefronts while the rest of
    // the chip idles after the cheap groups finish: they are cut into narrower tiles (fewer groups per wavefront => less
    // divergence per step and more compute units working on the tail).  Cheap groups always fill 64-lane tiles.
    // The pool's arrays are interleaved over 64 lanes whatever a tile's width, so narrow tiles trade HBM for speed (64 / width times the
    // state of their groups).  plan_tiles(relax) lays the batch out with the one-group-per-wavefront class 2^relax times wider (up
    // to the width of the other expensive groups), then with the smaller dense-table limit; the first layout that fits the free HBM is used.  A batch that does not fit even
    // then is refused with the size it needs: the host splits the unit (host/inference_engine.py: max_groups_per_launch).
    std::vector<uint32_t> tile_start, blk_first, blk_last;   // (blk_*: first / last tile of the pool block a tile shares, see plan_tiles)
    std::vector<TilePlan> plans;
    uint64_t pool = 0;
    uint32_t ntiles = 0;
    uint32_t noise_level = 0;
    auto plan_tiles = [&](uint32_t relax) {
    tile_start.clear();
    pool = 0;
    const uint64_t dense_limit = relax >= 4 ? (64ull << 20) : (256ull << 20);
    {
        // two classes of expensive groups (the batch is sorted, so they are prefixes): X = nested groups and clusters with >= 16
        // haplotype candidates, Y = single clusters with 6..15 candidates.  Narrower is faster per group (measured: 2 304 nested
        // groups of shape C 271 / 259 / 245 / 237 ms at 16 / 8 / 4 / 2 groups per wavefront; 16 384 ten-haplotype clusters x 10
        // samples 730 -> 549 ms at 64 -> 16: fewer diverging lanes, and the idle lanes run copies that share the data-parallel phases)
        // and needs less LDS per tile, so these tiles co-reside with everything else instead of queueing for a CU — as long as they
        // are all resident at once: X gets <= 3 tiles per CU and >= 4 groups per tile, X + Y together <= 6 per CU, Y >= 16 per tile
        // (narrower loses again: too many wavefronts for the slots).
        const uint32_t tail_h = getenv("BT_GIBBS_TAIL_H") ? (uint32_t)atoi(getenv("BT_GIBBS_TAIL_H")) : 16u;   // env: tuning overrides
        uint32_t n_x = 0;
        while (n_x < G && (shapes[n_x].nv > 1 || shapes[n_x].Hmax >= tail_h)) ++n_x;
        uint32_t n_y = n_x;
        while (n_y < G && shapes[n_y].Hmax >= 6) ++n_y;
        // Width policy (measured on MI355X, r02: 600 000-group mixture batches at S = 3): these tiles are bound by the latency of their
        // slowest lane's sequential program, so few groups per wavefront (+ copies that share the data-parallel phases) finish a group
        // sooner AND need little LDS per tile, which keeps many tiles resident; launches with more narrow tiles than wavefront slots
        // simply run them in rounds next to the two-haplotype tiles (12.0 s at widths 8 / 16, 12.2 s at 4 / 16, 14.6 s at 16 / 32,
        // 21.3 s at 64 / 64 before the copies worked in teams).  Four per wavefront; single clusters with 6..15 candidates sixteen.
        // `relax` (the batch does not fit the free HBM) doubles both.
        uint32_t width_x = kMinTileWidth, width_y = 16;   // (r02, after the copies' teams: 10.2 s at 4 / 16 against 10.6 s at 8 / 16 and 11.4 s at 8 / 32)
        // The sampler of a noise driver (noise_seeding): a chain's iteration waits for its slowest tile (150 us at 16 groups per tile, 100 at 8: eight copies
        // share the table fill) — as long as the narrower tiles do not push the tile count over what can be resident together (seven wavefronts of the resident
        // launch per CU is what its LDS charge leaves at ten samples): the widest level whose tile count fits is taken (below).
        const bool noise_widths = params->noise_seeding && !getenv("BT_GIBBS_NO_NOISE_WIDTHS");
        uint32_t width_z3 = LANES;
        if (noise_widths) {   // noise_level: 0 = the narrowest widths; raised by the caller of plan_tiles when the tile count does not fit a resident chain
            if (noise_level == 0) width_y = 8, width_z3 = 32;
            else if (noise_level == 1) width_y = 16, width_z3 = 32;
            else width_y = 16, width_z3 = LANES;
        }
        width_x = std::min<uint32_t>(LANES, width_x << relax);
        width_y = std::min<uint32_t>(LANES, width_y << relax);
        if (const char *e = getenv("BT_GIBBS_TAIL_WIDTH")) {
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) width_x = (uint32_t)v;
        }
        if (const char *e = getenv("BT_GIBBS_MID_WIDTH")) {
            const int v = atoi(e);
            if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) width_y = (uint32_t)v;
        }
        // inside X, clusters with more than 65 536 (sample, diplotype) pairs (hundreds of haplotype candidates times several samples) evaluate tens of thousands of candidates per sample in the first sweep of every chain: they get a wavefront of
        // their own (63 copies share that work; 8 such groups x 30 samples: 10.7 / 5.7 / 3.0 s at 4 / 2 / 1 per wavefront)
        auto hashed = [&](uint32_t i) { return (uint64_t)S * ((uint64_t)shapes[i].Hmax * (shapes[i].Hmax + 1) / 2 + shapes[i].Hmax) > 65536; };
        uint32_t n_w = 0;
        for (uint32_t i = 0; i < n_x; ++i) n_w += hashed(i) ? 1u : 0u;
        uint32_t width_w = 1;
        while (width_w < width_x && (uint64_t)n_w > (uint64_t)width_w * (4 * 256)) width_w *= 2;
        if (const char *e = getenv("BT_GIBBS_HUGE_WIDTH")) {
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) width_w = (uint32_t)v;
        }
        width_w = std::min<uint32_t>(width_x, width_w << relax);
        uint32_t at = 0;
        while (at < n_x) {   // runs of equal kind
            const bool hw = hashed(at);
            uint32_t run_end = at;
            while (run_end < n_x && hashed(run_end) == hw) ++run_end;
            const uint32_t w = hw ? width_w : width_x;
            while (at < run_end) {
                tile_start.push_back(at);
                at += std::min<uint32_t>(w, run_end - at);
            }
        }
        while (at < n_y) {
            tile_start.push_back(at);
            at += std::min<uint32_t>(width_y, n_y - at);
        }
        // The sampler of a noise driver (noise_seeding): its chains run as ONE resident launch in which every workgroup is charged the LDS of the hungriest
        // tile and an iteration lasts as long as its slowest tile (bt_gibbs_noise_chain_begin).  Single clusters with 4 or 5 candidates in 64-group tiles
        // are both (36 - 44 KB of LDS; 190 - 260 us per iteration with their hot arrays pushed out to HBM against 75 us of the others): they get 16 groups
        // per wavefront (four copies share the table fills), like the clusters with 6..15 candidates above.
        uint32_t n_z = n_y, n_z3 = n_y;
        if (noise_widths) {
            while (n_z < G && shapes[n_z].Hmax >= 4) ++n_z;
            n_z3 = n_z;
            while (n_z3 < G && shapes[n_z3].Hmax >= 3) ++n_z3;   // three candidates: 29 KB at 64 groups per tile, half of that at 32
        }
        while (at < n_z) {
            tile_start.push_back(at);
            at += std::min<uint32_t>(width_y, n_z - at);
        }
        while (at < n_z3) {
            tile_start.push_back(at);
            at += std::min<uint32_t>(width_z3, n_z3 - at);
        }
        while (at < G) {
            tile_start.push_back(at);
            at += std::min<uint32_t>(LANES, G - at);
        }
        tile_start.push_back(G);
    }
    ntiles = (uint32_t)tile_start.size() - 1;
    g->ntiles = ntiles;
    g->group_tile.assign(G, 0);
    g->group_lane.assign(G, 0);
    g->group_nvert.assign(G, 0);
    g->loc.assign(C, ClusterLoc{0, 0, 0, 0});
    const uint64_t collect_total = (uint64_t)std::max<uint32_t>(params->num_chains, 1) * std::max<uint32_t>(params->num_iterations, 1) * S;

    // Pool blocks.  Every array of the pool is interleaved over 64 lanes; a narrow tile uses 4 or 16 of them.  Consecutive narrow tiles of
    // the same width therefore SHARE one pool block — 64 / width tiles side by side, tile j on pool lanes [j * width, (j + 1) * width) — laid out
    // for the maxima of all their groups (the batch is sorted by shape, so neighbours are alike).  A tile addresses HBM with its pool lane
    // (TileDesc::pool_lane0 + lane) and LDS with its own lane; nothing else changes.
    blk_first.assign(ntiles, 0);
    blk_last.assign(ntiles, 0);
    {
        auto width_of = [&](uint32_t ti) {
            uint32_t w = 1;
            while (w < tile_start[ti + 1] - tile_start[ti]) w *= 2;
            return w;
        };
        const bool share = false;   // (round 3: every tile has a pool block of its own width, see TileDesc::wsh)
        for (uint32_t ti = 0; ti < ntiles;) {
            const uint32_t w = width_of(ti);
            uint32_t n = 1;
            if (share && w < LANES)
                while (ti + n < ntiles && n < LANES / w && width_of(ti + n) == w) ++n;
            for (uint32_t j = 0; j < n; ++j) {
                blk_first[ti + j] = ti;
                blk_last[ti + j] = ti + n - 1;
            }
            ti += n;
        }
    }
    plans.assign(ntiles, TilePlan{});
    for (uint32_t ti = 0; ti < ntiles; ++ti) {
        TileDesc d{};
        d.S = S;
        d.first_group = tile_start[ti];
        d.num_lanes = tile_start[ti + 1] - tile_start[ti];
        uint32_t Am = 1, NSHm = 0;
        for (uint32_t bl = tile_start[blk_first[ti]]; bl < tile_start[blk_last[ti] + 1]; ++bl) {   // dimensions: maxima over the pool block
            const uint32_t gi = shapes[bl].g;
            const uint32_t c0 = gd[gi].c0, c1 = gd[gi].c0 + gd[gi].nv;
            d.nvm = std::max(d.nvm, c1 - c0);
            NSHm = std::max(NSHm, gd[gi].num_shared);
            for (uint32_t c = c0; c < c1; ++c) {
                const uint32_t H = cd[c].H, V = cd[c].V, K = cd[c].K;
                d.Hm = std::max(d.Hm, H);
                d.Vm = std::max(d.Vm, V);
                d.Km = std::max(d.Km, K);
                d.NUm = std::max(d.NUm, cd[c].nu);
                d.NMm = std::max(d.NMm, cd[c].nm);
                d.NNZm = std::max(d.NNZm, cd[c].nnz);
                d.HNm = std::max(d.HNm, cd[c].hn);
                d.NDm = std::max(d.NDm, cd[c].nd);
                d.NDVm = std::max(d.NDVm, cd[c].ndv);
                d.NEm = std::max(d.NEm, cd[c].ne);
                Am = std::max(Am, g->h_A[c]);
            }
        }
        d.HWm = (d.Hm + 31) / 32;
        d.NSHm = NSHm;
        d.Am = Am;
        d.Bcap = uset_bucket_capacity(d.Hm);
        d.D2m = d.Hm * (d.Hm + 1) / 2;
        d.Dcm = d.D2m + d.Hm;
        // dense [S][D] table of unique-k-mer sums when it is affordable, otherwise the tag-checked direct-mapped table.  Narrow tiles
        // keep the table per-lane contiguous (TileDesc::uc_width) and pay for their own lanes only; the multicluster tables of the
        // same size stay interleaved over 64 lanes
        const uint64_t dense = (uint64_t)S * d.Dcm;
        uint32_t tile_w = 1;
        while (tile_w < d.num_lanes) tile_w *= 2;
        d.pool_lane0 = 0;
        d.wsh = 0;
        while ((1u << d.wsh) < tile_w) ++d.wsh;
        d.mat_width = tile_w < LANES ? tile_w : 0u;   // narrow tiles: K x H matrices contiguous per lane (TileDesc::mat_width)
        const uint64_t table_bytes = dense * 8 * tile_w * d.nvm + (d.NMm ? dense * 16 * tile_w * d.nvm : 0);
        if (table_bytes <= dense_limit && dense < (1ull << 26)) {
            d.cache_mode = 0;
            d.cache_entries = (uint32_t)dense;
            d.uc_width = tile_w < LANES ? tile_w : 0u;
        } else {
            d.cache_mode = 1;
            d.cache_entries = 16384;
            d.uc_width = 0;
        }
        uint64_t cap = 4;
        while (cap < 2 * std::min<uint64_t>((uint64_t)d.Dcm + 1, collect_total)) cap <<= 1;
        d.dip_cap = (uint32_t)cap;
        d.scache_p = std::min<uint32_t>(2 * S, d.Hm);
        d.scache_len = d.Hm + 1;
        d.scache_n = 2 * S * d.scache_p;
        if ((uint64_t)d.scache_n * d.scache_len > 4096) d.scache_n = 0;
        // lengths (elements per lane) of every array
        const uint64_t nv = d.nvm;
        uint64_t len[A_COUNT];
        len[A_M] = (uint64_t)nv * d.Km * d.Hm;
        len[A_HASC] = nv * d.Km;
        len[A_COUNTS] = nv * d.Km * S;
        len[A_IC] = nv * d.Km * 2;
        len[A_SHARED] = nv * d.Km;
        len[A_KVOFF] = nv * (d.Km + 1);
        len[A_KVVAR] = nv * d.NNZm;
        len[A_KVBITS] = nv * (uint64_t)d.NNZm * d.HWm;
        len[A_HAPAL] = nv * d.Hm * d.Vm;
        len[A_HAPCELL] = nv * d.Hm * d.Vm;
        len[A_HNOFF] = nv * (d.Hm + 1);
        len[A_HNIDX] = nv * d.HNm;
        len[A_VARNA] = nv * d.Vm;
        len[A_VARDEP] = nv * d.Vm;
        len[A_ALBASE] = nv * (d.Vm + 1);
        len[A_NDCL] = nv * std::max<uint32_t>(d.NDm, 1);
        len[A_NDVOFF] = nv * (d.NDm + 1);
        len[A_NDVAR] = nv * std::max<uint32_t>(d.NDVm, 1);
        len[A_UNIQ0] = nv * d.NUm;
        len[A_MULTI0] = nv * d.NMm;
        len[A_EDGES0] = nv * std::max<uint32_t>(d.NEm, 1);
        len[A_VDIMS] = nv * 8;
        len[A_VDIMS2] = nv * 2;
        len[A_GDIMS] = 4;
        len[A_SOURCES0] = nv;
        len[A_PLOIDY] = S;
        len[A_MT] = nv * 2 * MT_RING_PAD;
        len[A_FNDSAVED] = nv;
        len[A_SPARSITY] = nv;
        len[A_UNIQ] = len[A_USUB] = nv * d.NUm;
        len[A_MULTI] = len[A_MSUB] = nv * d.NMm;
        len[A_SMM] = nv * d.NMm * S;
        len[A_DIP] = nv * 2 * S;
        len[A_FREQ] = nv * d.Hm;
        len[A_OBS] = nv * d.Hm;
        len[A_NZ] = nv * d.Hm;
        len[A_ZHDR] = len[A_PHDR] = nv * 4;
        len[A_ZBKT] = len[A_PBKT] = nv * d.Bcap;
        len[A_UNEXT] = nv * d.Hm;
        len[A_HVCOUNT] = nv * d.Hm * d.Vm;
        len[A_UCACHE] = (uint64_t)nv * d.cache_entries;
        len[A_UCTAG] = nv * (d.cache_mode == 1 ? d.cache_entries : 1);
        {   // teams of copies in sample_diplotypes (narrow tiles with dense tables): min(S, copies)
            uint32_t stride = 1;
            while (stride < d.num_lanes) stride *= 2;
            const uint32_t copies = stride <= 32 && !getenv("BT_GIBBS_NO_COPIES") ? 64u / stride : 1u;
            d.teams = copies > 1 && d.cache_mode == 0 && S > 1 && !getenv("BT_GIBBS_NO_TEAMS") ? std::min<uint32_t>(S, copies) : 1u;
        }
        len[A_CUM] = nv * std::max<uint32_t>(d.D2m, 1) * d.teams;
        len[A_NZLIST] = nv * d.Hm;
        len[A_SIMPLEX] = nv * (d.Hm + 1);
        len[A_SCACHE] = nv * (uint64_t)std::max<uint32_t>(d.scache_n, 1) * std::max<uint32_t>(d.scache_n ? d.scache_len : 1, 1);
        len[A_SCLEN] = nv * std::max<uint32_t>(d.scache_n, 1);
        len[A_KSC] = nv * S * 2 * d.Vm * 4;
        len[A_KSCUPD] = nv * S;
        len[A_DIPKEYS] = nv * d.dip_cap;
        len[A_DIPFREQ] = nv * (uint64_t)d.dip_cap * S;
        len[A_ASTATS] = nv * (uint64_t)S * d.Am * 12;
        len[A_NESTPL] = len[A_NESTN] = nv * S;
        len[A_NESTST] = nv * S * 8;
        len[A_PENDNEST] = nv * S * 8;
        len[A_NVER] = nv * S * 2;
        len[A_EVLOG] = nv * S * (2 * EV_CAP + 1);
        len[A_EVN] = nv * S;
        len[A_KSCKEY] = nv * S * (KSC_WAYS + 1);
        len[A_KSCDATA] = (uint64_t)nv * S * KSC_WAYS * 2 * d.Vm * 4;
        len[A_SC] = nv * SC_COUNT;
        len[A_EDGES] = nv * std::max<uint32_t>(d.NEm, 1);
        len[A_COVER] = nv * d.Km;
        len[A_MCACHE] = len[A_MCTAG] = len[A_MCGEN] = nv * (d.NMm ? d.cache_entries : 1);
        len[A_MGEN] = nv * S;
        len[A_OTH] = nv * std::max<uint32_t>(d.NMm, 1) * S;
        len[A_MSUBM] = nv * (uint64_t)std::max<uint32_t>(d.NMm, 1) * d.Hm;
        len[A_MSUBC] = nv * (uint64_t)std::max<uint32_t>(d.NMm, 1) * S;
        len[A_MSUBIC] = nv * (uint64_t)std::max<uint32_t>(d.NMm, 1) * 2;
        len[A_MSUBSH] = nv * (uint64_t)std::max<uint32_t>(d.NMm, 1);
        len[A_SUBM] = nv * (uint64_t)d.NUm * d.Hm;
        len[A_SUBCNT] = nv * (uint64_t)d.NUm * S;
        len[A_SUBIC] = nv * (uint64_t)d.NUm * 2;
        len[A_SKVOFF] = nv * (uint64_t)(d.NUm + 1);
        len[A_SKVVAR] = nv * (uint64_t)std::max<uint32_t>(d.NNZm, 1);
        len[A_SKVBITS] = nv * (uint64_t)std::max<uint32_t>(d.NNZm, 1) * d.HWm;
        len[A_KSCTMP] = nv * (uint64_t)2 * d.Vm * 4;
        len[A_LOGF] = nv * d.Hm;
        len[A_PEND] = nv * S;
        len[A_PENDDIP] = nv * 2 * S;
        len[A_PENDVALID] = nv * S;
        len[A_SOURCES] = nv;
        len[A_STACK] = 2 * (nv + 1);
        len[A_BRNG] = MT_PAD;
        len[A_SHMULT] = (uint64_t)std::max<uint32_t>(d.NSHm, 1) * S;
        // draw-ahead rings (MtRing): the diplotype generator consumes exactly two words per sample and visit, the frequency generator
        // a few words per non-zero haplotype; both are topped up at the start of a visit
        // (production is in chunks of four words: a ring of `cap` words can be filled up to cap - 3 ahead)
        d.ring_cap[0] = 8;
        while (d.ring_cap[0] < 2 * S + 3 && d.ring_cap[0] < 64) d.ring_cap[0] *= 2;
        // 32 words for the frequency generator: a refill in the middle of a visit happens where the lanes of a wavefront have diverged (every lane pays for
        // every other lane's refill), the top-up at the start of a visit is one pass for all of them.  Measured (round 5, the bench mixtures): 16 -> 32 words
        // for the tiles of 4 and 16 groups: S = 3 3.80 -> 3.69 s, S = 10 12.8 -> 11.7 s; 64 words cost more LDS than they save (3.75 s / 12.1 s).
        d.ring_cap[1] = 32;
        if (params->noise_seeding && !getenv("BT_GIBBS_NO_NOISE_WIDTHS")) {   // (a noise sampler's LDS block decides how many tiles its resident chain can hold)
            d.ring_cap[1] = 16;                                                  // 4 KB less per two-haplotype tile
            if (d.ring_cap[0] > 16) d.ring_cap[0] = 16;                          // seven samples and more: the diplotype draws of a visit top the ring up on the way (another 4 KB)
        }
        if (const char *e = getenv("BT_GIBBS_RING0")) d.ring_cap[0] = (uint32_t)atoi(e);
        if (const char *e = getenv("BT_GIBBS_RING1")) d.ring_cap[1] = (uint32_t)atoi(e);
        {   // (tuning) per kind of narrow tile: RING1_Y = single clusters in tiles of 8..32 groups, RING1_X = tiles of up to 4 groups; RING1_W = every other non-two-haplotype tile
            const bool two_hap = d.nvm == 1 && d.Hm == 2 && d.NMm == 0;
            const char *e = two_hap ? nullptr : (tile_w <= 4 ? getenv("BT_GIBBS_RING1_X") : (tile_w <= 32 ? getenv("BT_GIBBS_RING1_Y") : getenv("BT_GIBBS_RING1_W")));
            if (e) d.ring_cap[1] = (uint32_t)atoi(e);
            e = two_hap ? nullptr : (tile_w <= 4 ? getenv("BT_GIBBS_RING0_X") : (tile_w <= 32 ? getenv("BT_GIBBS_RING0_Y") : getenv("BT_GIBBS_RING0_W")));
            if (e) d.ring_cap[0] = (uint32_t)atoi(e);
        }
        d.ring_len = d.ring_cap[0] + d.ring_cap[1] + 2 * MT_RING_HDR;
        len[A_RING] = nv * d.ring_len;
        uint64_t off = 0, in_bytes = 0;
        for (int a = 0; a < A_COUNT; ++a) {
            off = align_up(off, 256);
            d.off[a] = off;
            off += len[a] * tile_w * kElemSize[a];
            if (a == A_PLOIDY) in_bytes = align_up(off, 256);
        }
        {
            // hot arrays (bt_gibbs_tile.hpp: Vx::harr users) -> offsets inside the wavefront's LDS block
            for (int a = 0; a < A_COUNT; ++a) d.hoff[a] = NOHOT;
            const int hot_arrs[] = {A_SC, A_DIP, A_NESTPL, A_NESTN, A_KSCUPD, A_MGEN, A_PEND, A_PENDDIP, A_PENDVALID, A_EVN, A_FREQ, A_LOGF, A_OBS, A_NZ, A_NZLIST,
                                    A_UNEXT, A_ZHDR, A_ZBKT, A_PHDR, A_PBKT, A_KSCTMP, A_CUM, A_RING, A_FNDSAVED, A_UCACHE};
            // LDS rows are interleaved over the tile's lanes only (16 / 32 / 64): a narrow tile needs a fraction of the LDS per vertex,
            // which lets every vertex of a multi-cluster group stay resident instead of being swapped around each visit
            d.lds_stride = 1;
            while (d.lds_stride < d.num_lanes) d.lds_stride *= 2;
            uint64_t ho = 0;
            // tiles of two-haplotype clusters run simple_sweeps(), which keeps the two haplotype sets and the candidate scratch in
            // registers: those arrays stay in HBM (written once per launch) and the tile's LDS block shrinks by a fifth
            bool want_simple = d.nvm == 1 && d.Hm == 2 && d.NMm == 0 && d.cache_mode == 0 && d.lds_stride == LANES && !getenv("BT_GIBBS_NO_SIMPLE") && !getenv("BT_GIBBS_NO_SIMPLE_KERNEL");
            for (uint32_t l = 0; l < d.num_lanes && want_simple; ++l) want_simple = cd[gd[shapes[tile_start[ti] + l].g].c0].H == 2;
            // tuning: BT_GIBBS_HOT_SKIP = bit mask over hot_arrs[] of arrays to leave in HBM
            const uint64_t skip_mask = getenv("BT_GIBBS_HOT_SKIP") ? strtoull(getenv("BT_GIBBS_HOT_SKIP"), nullptr, 0) : 0ull;
            int hot_i = -1;
            for (int a : hot_arrs) {
                ++hot_i;
                if ((skip_mask >> hot_i) & 1ull) continue;
                if (a == A_MGEN && d.NMm == 0) continue;          // (only clusters with multicluster k-mers read it)
                if (a == A_KSCTMP) continue;   // (the k-mer-stats rebuild accumulates in registers since round 2: the scratch rows are unused — in narrow tiles they were a third of the LDS block)
                if (a == A_CUM && d.D2m * d.teams > 16) continue;
                if (want_simple && a != A_RING) continue;   // (simple_sweeps keeps the cluster's state in registers and its own per-sample words: TileDesc::sblk)
                // the dense table of unique-k-mer sums is read for every candidate of every sample: a few entries per lane (two-haplotype
                // clusters x a few samples) stay in LDS for the launch
                if (a == A_UCACHE && !(d.cache_mode == 0 && d.uc_width == 0 && d.nvm == 1 && (uint64_t)d.cache_entries * 8 <= 160)) continue;
                const uint64_t per_vertex = len[a] / nv;   // elements per lane and vertex
                d.hoff[a] = (uint32_t)ho;
                ho = align_up(ho + per_vertex * d.lds_stride * kElemSize[a], 16);
            }
            d.sblk = 0;
            if (want_simple) {   // per sample: two weights + the packed state word (bt_gibbs_simple.hpp: SB_WORDS), lane-interleaved
                d.sblk = (uint32_t)ho;
                ho = align_up(ho + (uint64_t)3 * S * LANES * 4, 16);
            }
            d.hot_bytes = (uint32_t)std::min<uint64_t>(ho, 0xFFFFFFFFu);
            const uint64_t hot_budget = getenv("BT_GIBBS_HOT_BUDGET") ? strtoull(getenv("BT_GIBBS_HOT_BUDGET"), nullptr, 0) : kHotBudget;   // tuning
            d.lds_all = d.nvm > 1 && ho * d.nvm <= hot_budget && !getenv("BT_GIBBS_NO_LDS_ALL") ? 1u : 0u;
            if (ho > hot_budget || (d.nvm > 1 && !d.lds_all && hot_budget < kHotBudget) || (d.nvm > 1 && getenv("BT_GIBBS_NOHOT_MULTI"))) {
                for (int a = 0; a < A_COUNT; ++a) d.hoff[a] = NOHOT;
                d.hot_bytes = 0;
                d.lds_all = 0;
            }
        }
        // Wavefronts per tile (measured on MI355X, 64-group tiles): two-haplotype clusters run best on one full wavefront;
        // tiles whose lanes diverge over tens of diplotype candidates gain 15-25 % from narrower wavefronts.
        d.split = d.Hm < 6 ? 1u : (tile_lds_bytes(d) > kLightLds ? 2u : 4u);
        if (const char *e = getenv("BT_GIBBS_SPLIT")) {   // tuning override
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8) d.split = (uint32_t)v;
        }
        // narrow tiles: one wavefront, its idle lanes run identical copies of the groups and share the data-parallel phases
        d.copies = 1;
        if (d.lds_stride <= 32 && !getenv("BT_GIBBS_NO_COPIES")) {
            d.split = 1;
            d.copies = 64u / d.lds_stride;
        }
        {
            bool simple = d.nvm == 1 && d.Hm == 2 && d.NMm == 0 && d.cache_mode == 0 && d.copies == 1 && d.split == 1 && d.hot_bytes != 0 && !getenv("BT_GIBBS_NO_SIMPLE");
            for (uint32_t l = 0; l < d.num_lanes && simple; ++l) simple = cd[gd[shapes[tile_start[ti] + l].g].c0].H == 2;
            simple = simple && d.sblk != 0 && d.hoff[A_RING] != NOHOT && d.hoff[A_SC] == NOHOT;   // (the LDS block was laid out for simple_sweeps above: its per-sample words have their place)
            d.simple = simple ? 1u : 0u;
        }
        d.logged = d.nvm == 1 && d.NMm == 0 && !getenv("BT_GIBBS_NO_LOG") ? 1u : 0u;
        d.prio = (d.copies > 1 || d.num_lanes < LANES / 2) && !getenv("BT_GIBBS_NO_PRIO") ? 1u : 0u;
        d.base = ti == blk_first[ti] ? pool : plans[blk_first[ti]].d.base;
        if (ti == 0 && getenv("BT_GIBBS_DEBUG") && atoi(getenv("BT_GIBBS_DEBUG")) >= 2) {   // the arrays that make up most of tile 0
            std::vector<std::pair<uint64_t, int>> by_size;
            for (int a = 0; a < A_COUNT; ++a) by_size.emplace_back(len[a] * tile_w * kElemSize[a], a);
            std::sort(by_size.rbegin(), by_size.rend());
            fprintf(stderr, "bt_gibbs: tile 0 (%u groups, %u vertices max, Hm %u, Km %u, S %u): %.1f MB;", d.num_lanes, d.nvm, d.Hm, d.Km, S, (double)align_up(off, 256) / 1048576.0);
            for (int i = 0; i < 6; ++i) fprintf(stderr, " array %d: %.1f MB", by_size[i].second, (double)by_size[i].first / 1048576.0);
            fprintf(stderr, "\n");
        }
        plans[ti].d = d;
        plans[ti].in_bytes = in_bytes;
        plans[ti].total_bytes = align_up(off, 256);
        if (ti == blk_first[ti]) pool += plans[ti].total_bytes;
    }
    };   // plan_tiles
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        for (uint32_t relax = 0;; ++relax) {
            plan_tiles(relax);
            if (relax >= 4 || free_b == 0 || pool + 256 <= (uint64_t)(0.92 * (double)free_b)) break;
        }
        // a noise sampler: can its chain be one resident launch?  Every workgroup of that launch is charged the LDS of the two-haplotype tiles (the bulk; hungrier
        // tiles go without) + the chain's bins, at most eight 256-register wavefronts fit a CU: with more tiles than that the plan is made again with wider tiles
        // for the few-candidate clusters (fewer tiles; an iteration then waits longer for its slowest tile, but it is not a launch per iteration).
        while (params->noise_seeding && !getenv("BT_GIBBS_NO_NOISE_WIDTHS") && noise_level < 2) {
            uint32_t simple_need = 0;
            for (uint32_t ti = 0; ti < ntiles; ++ti)
                if (plans[ti].d.simple) simple_need = std::max(simple_need, tile_lds_bytes(plans[ti].d));
            const uint64_t per_cu = std::min<uint64_t>(8, 163840 / (((uint64_t)simple_need + 15) / 16 * 16 + (S * NC_BINS + 8u + NC_HELP_MAXH / 2u) * 4u));
            if ((uint64_t)ntiles <= per_cu * ctx->num_cu) break;
            ++noise_level;
            plan_tiles(0);
        }
    }
    lap("shape sort + tile plan");
    if (!plan_bytes)   // the construction's uploads: tile descriptors, cluster / group descriptors, cluster locations, tables, class lists (+ slack for fill lists)
        BT_TRYHIP(arena_reserve(ctx, (size_t)ntiles * (sizeof(TileDesc) + 64) + (size_t)C * (sizeof(BuildCluster) + sizeof(ClusterLoc) + 32) + (size_t)G * (sizeof(BuildGroup) + 16) + (8u << 20)));
    g->pool_bytes = pool + 256;
    if (plan_bytes) {   // + the count-model tables, descriptors and the transient device copy of the flat batch the tile builder reads
        const uint64_t flat = (uint64_t)C * 160 + (uint64_t)G * (32 + S) + flat_sel;
        *plan_bytes = g->pool_bytes + (uint64_t)S * (65536 + 256) * 8 + (uint64_t)ntiles * sizeof(TileDesc) + (uint64_t)C * sizeof(ClusterLoc) + flat;
        bt_gibbs_destroy(g);
        return BT_OK;
    }
    if (ctx->pool_cache && recycles && ctx->pool_cache_bytes >= g->pool_bytes && ctx->pool_cache_bytes / 2 <= g->pool_bytes + (64u << 20)) {
        g->d_pool = static_cast<uint8_t *>(ctx->pool_cache);   // the previous chain's pool (the work queued on it was waited for when its sampler was destroyed)
        g->pool_alloc_bytes = ctx->pool_cache_bytes;
        ctx->pool_cache = nullptr;
        ctx->pool_cache_bytes = 0;
    } else {
        if (ctx->pool_cache) {
            (void)hipFree(ctx->pool_cache);
            ctx->pool_cache = nullptr;
            ctx->pool_cache_bytes = 0;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&g->d_pool), g->pool_bytes);
        g->pool_alloc_bytes = g->pool_bytes;
        if (e != hipSuccess) {
            bt_gibbs_destroy(g);
            return fail(std::string("bt_gibbs_create: state pool of ") + std::to_string(g->pool_bytes) + " bytes: " + hipGetErrorString(e));
        }
    }
    g->allocs.push_back(g->d_pool);
    g->recycles = recycles;
    g->keep_pool = recycles && g->pool_alloc_bytes <= (8ull << 30);   // (a chain's sampler of estimateNoise: 1 - 6 GB)
    g->device_bytes += g->pool_bytes;
    {
        const uint64_t n16 = (g->pool_bytes + 15) / 16;   // (hipMalloc sizes are multiples of the allocation granule: the tail belongs to the allocation)
        const uint64_t per_wg = std::max<uint64_t>(4096, ((n16 + 1048575) / 1048576 + 63) / 64 * 64);   // 64 KB and more per workgroup, at most 2^20 workgroups
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n16 + per_wg - 1) / per_wg)), dim3(64), 0, ctx->stream, reinterpret_cast<uint4 *>(g->d_pool), n16, per_wg);
        BT_TRYHIP(hipGetLastError());
    }

    // ---- the input arrays of every tile: the flat batch is uploaded as it is and scattered into the tiles' layout ON THE DEVICE
    // (build_tiles_kernel, one workgroup per cluster).  The host only computes where things go: a descriptor per cluster.
    lap("pool allocation + clear (enqueued)");
    g->tiles.resize(ntiles);
    for (uint32_t ti = 0; ti < ntiles; ++ti) g->tiles[ti] = plans[ti].d;
    BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&g->d_tiles), (size_t)ntiles * sizeof(TileDesc)));
    g->allocs.push_back(g->d_tiles);
    BT_TRYHIP(staged_upload(ctx, g->d_tiles, g->tiles.data(), (size_t)ntiles * sizeof(TileDesc)));
    {
        std::vector<BuildCluster> bc(C);
        std::vector<BuildGroup> bg(G);
        for (uint32_t ti = 0; ti < ntiles; ++ti) {
            const TileDesc &d = plans[ti].d;
            for (uint32_t l = 0; l < d.num_lanes; ++l) {
                const uint32_t gi = shapes[tile_start[ti] + l].g;
                const uint32_t c0 = gd[gi].c0, c1 = gd[gi].c0 + gd[gi].nv;
                g->group_tile[gi] = ti;
                g->group_lane[gi] = l;
                g->group_nvert[gi] = c1 - c0;
                bg[gi] = BuildGroup{ti, l, c1 - c0, gd[gi].src0, gd[gi].nsrc, gd[gi].group_index, gd[gi].g_abs, 0};
                for (uint32_t c = c0; c < c1; ++c) {
                    g->loc[c] = ClusterLoc{ti, l, c - c0, 0};
                    BuildCluster &x = bc[c];
                    x.tile = ti;
                    x.lane = l;
                    x.v = c - c0;
                    const ClusterDims &q = cd[c];
                    x.H = q.H;
                    x.V = q.V;
                    x.r0 = q.r0;
                    x.K = q.K;
                    x.u0 = q.u0;
                    x.nu = q.nu;
                    x.m0 = q.m0;
                    x.nm = q.nm;
                    x.nd0 = q.nd0;
                    x.nd = q.nd;
                    x.e0 = q.e0;
                    x.ne = q.ne;
                    x.cid = q.cid;
                    x.A = q.A;
                    x.mult_off = q.mult_off;
                    x.kvb_off = q.kvb_off;
                    x.hapvar_off = q.hapvar_off;
                    x.hap_base = q.hap_base;
                    x.var_base = q.var_base;
                }
            }
        }
        // the flat batch is on the device already (bt_gibbs_source); the descriptors of this sampler's clusters and groups go up
        std::vector<void *> tmp;
        auto up = [&](const void *h, size_t bytes, const void **d_out) -> hipError_t {
            void *d = nullptr;
            hipError_t e = hipMalloc(&d, std::max<size_t>(bytes, 16));
            if (e != hipSuccess) return e;
            tmp.push_back(d);
            *d_out = d;
            return staged_upload(ctx, d, h, bytes);
        };
        BuildBatch bb = src->dev;
        hipError_t e = up(bc.data(), (size_t)C * sizeof(BuildCluster), reinterpret_cast<const void **>(&bb.clusters));
        if (e == hipSuccess) e = up(bg.data(), (size_t)G * sizeof(BuildGroup), reinterpret_cast<const void **>(&bb.groups));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(build_tiles_kernel, dim3(C), dim3(64), 0, ctx->stream, g->d_tiles, g->d_pool, bb, S);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(build_groups_kernel, dim3((G + 63) / 64), dim3(64), 0, ctx->stream, g->d_tiles, g->d_pool, bb, G, S);
            e = hipGetLastError();
        }
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);   // (bc / bg go out of scope)
        // the descriptors' device copies stay until the sampler is destroyed: hipFree waits for the whole device to idle — for the chain of a noise driver
        // that is resident on another stream while a helper thread builds this sampler (82 ms per construction instead of 20)
        for (void *d : tmp) g->allocs.push_back(d);
        if (e != hipSuccess || e2 != hipSuccess) {
            bt_gibbs_destroy(g);
            return fail(std::string("bt_gibbs_create: building the tiles: ") + hipGetErrorString(e != hipSuccess ? e : e2));
        }
    }
    lap("descriptors + build kernels (synchronised)");
    BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&g->d_loc), (size_t)C * sizeof(ClusterLoc)));
    g->allocs.push_back(g->d_loc);
    BT_TRYHIP(staged_upload(ctx, g->d_loc, g->loc.data(), (size_t)C * sizeof(ClusterLoc)));
    BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&g->d_lut_g), (size_t)S * 65536 * 8));
    g->allocs.push_back(g->d_lut_g);
    BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&g->d_lut_n), (size_t)S * 256 * 8));
    g->allocs.push_back(g->d_lut_n);
    g->device_bytes += (uint64_t)S * (65536 + 256) * 8 + (uint64_t)ntiles * sizeof(TileDesc) + (uint64_t)C * sizeof(ClusterLoc);
    g->P.lut_g = (const double BT_GAS *)g->d_lut_g;
    g->P.lut_n = (const double BT_GAS *)g->d_lut_n;
    {
        // lgamma over the integers the simplex-size distribution touches (FrequencyDistribution.cpp:143-196): <= Hmax + 2S + 1
        uint32_t Hmax = 0;
        for (uint32_t c = 0; c < C; ++c) Hmax = std::max(Hmax, cd[c].H);
        const uint32_t n = Hmax + 2 * S + 8;
        std::vector<double> lg(n, 0.0);
        for (uint32_t i = 1; i < n; ++i) lg[i] = std::lgamma((double)i);
        BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&g->d_lgamma), (size_t)n * 8));
        g->allocs.push_back(g->d_lgamma);
        BT_TRYHIP(staged_upload(ctx, g->d_lgamma, lg.data(), (size_t)n * 8));
        BT_TRYHIP(hipStreamSynchronize(ctx->stream));
        g->P.lgamma_int = (const double BT_GAS *)g->d_lgamma;
        g->P.lgamma_n = n;
        // Marsaglia-Tsang's a2 = 1 / sqrt(9 (alpha - 1/3)) for alpha = an observation count + 1 <= 2S + 1 (gamma_distribution::param_type::_M_initialize,
        // bits/random.tcc: the same two correctly rounded IEEE operations the device would execute per draw)
        {
            const uint32_t na = 2 * S + 8;
            std::vector<double> a2(na, 0.0);
            for (uint32_t i = 1; i < na; ++i) {
                const double a1 = (double)i - 1.0 / 3.0;
                a2[i] = 1.0 / std::sqrt(9.0 * a1);
            }
            double *d_a2 = nullptr;
            BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&d_a2), (size_t)na * 8));
            g->allocs.push_back(d_a2);
            BT_TRYHIP(staged_upload(ctx, d_a2, a2.data(), (size_t)na * 8));
            BT_TRYHIP(hipStreamSynchronize(ctx->stream));
            g->P.gamma_a2 = (const double BT_GAS *)d_a2;
            g->P.gamma_n = na;
        }
        BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&g->d_params), sizeof(GParams)));
        g->allocs.push_back(g->d_params);
        BT_TRYHIP(staged_upload(ctx, g->d_params, &g->P, sizeof(GParams)));
        BT_TRYHIP(hipStreamSynchronize(ctx->stream));
    }
    {
        lap("tables");
        // Launch classes.  A launch has ONE kernel and ONE dynamic LDS size — its hungriest tile's — so the tiles are classed by the kernel that runs their
        // sampling operations (KIND: gibbs_kernel / gibbs_hot_kernel / gibbs_single_kernel; the two-haplotype tiles are a class of their own) and, within a kind,
        // by LDS need.  At most three classes beside the two-haplotype one: more streams than hardware queues run one after another (round 2: ten classes
        // 5.2 -> 8.9 s).  The schedule is LDS-capacity-bound on whole-genome batch shapes (sum over the tiles of LDS x duration against 160 KB per CU), so the
        // three are dealt to the kinds, and the cuts inside a kind placed, where they minimise the LDS charged: sum over the classes of tiles x the class's
        // largest need (2 KB bins).  BT_GIBBS_LDS_CLASSES="a,b,...": upper bounds (bytes) for every kind instead (tuning; a last class takes the rest);
        // BT_GIBBS_FIXED_CLASSES=1: kClassLds for every kind; BT_GIBBS_KIND_CLASSES="g,h,s": classes per kind instead of the dealt ones.
        const bool own_kernel = !getenv("BT_GIBBS_NO_SIMPLE_KERNEL");
        const bool hot_kernel = !getenv("BT_GIBBS_NO_HOT_KERNEL");
        // gibbs_single_kernel (one-cluster tiles at 168 registers, three wavefronts per SIMD) is OPT-IN (BT_GIBBS_SINGLE_KERNEL=1): the multi-variant class alone at
        // the bench's size runs 1.12 s with two wavefronts per SIMD and 1.13 s with three — its SIMDs are VALU-issue-bound at two (49 % of a wavefront's cycles
        // issuing, profiles/r06_single_kernel.txt) — and a kind of its own splits the launch classes: the mixture 3.69 -> 3.87 s
        const bool single_kernel = hot_kernel && getenv("BT_GIBBS_SINGLE_KERNEL") && !getenv("BT_GIBBS_NO_SINGLE_KERNEL");
        auto hot_tile = [&](const TileDesc &d) {
            bool ok = hot_kernel && !d.simple && d.hot_bytes != 0 && (d.nvm == 1 || d.lds_all);
            for (int a = 0; a < A_COUNT && ok; ++a)
                if (hot_core(a)) ok = d.hoff[a] != NOHOT;
            return ok;
        };
        enum { KIND_GENERAL = 0, KIND_HOT = 1, KIND_SINGLE = 2, NKIND = 3 };
        // (the hot kinds' launches are PACKED — several one-wavefront tiles per workgroup, bt_gibbs_tile.hpp: BT_PACKED —, so a tile worked on by several wavefronts,
        // which only a batch squeezed into too little HBM has, goes through gibbs_kernel)
        auto tile_kind = [&](const TileDesc &d) { return !(hot_tile(d) && d.split == 1) ? KIND_GENERAL : (single_kernel && d.logged && d.nvm == 1 && d.NMm == 0 ? KIND_SINGLE : KIND_HOT); };
        std::vector<int> kind_of(ntiles, -1);   // -1: the two-haplotype class
        for (uint32_t ti = 0; ti < ntiles; ++ti)
            if (!(own_kernel && g->tiles[ti].simple && g->tiles[ti].split == 1)) kind_of[ti] = tile_kind(g->tiles[ti]);
        std::vector<uint32_t> kind_cuts[NKIND];   // per kind: upper LDS bounds of its classes, ascending, the last one unbounded
        if (getenv("BT_GIBBS_LDS_CLASSES") || getenv("BT_GIBBS_FIXED_CLASSES")) {
            std::vector<uint32_t> class_lds(kClassLds, kClassLds + sizeof(kClassLds) / sizeof(kClassLds[0]));
            if (const char *e = getenv("BT_GIBBS_LDS_CLASSES")) {
                class_lds.clear();
                for (const char *q = e; *q;) {
                    class_lds.push_back((uint32_t)strtoul(q, nullptr, 10));
                    while (*q && *q != ',') ++q;
                    if (*q == ',') ++q;
                }
                class_lds.push_back(0xFFFFFFFFu);
            }
            for (auto &kc : kind_cuts) kc = class_lds;
        } else {
            constexpr uint32_t NB = 81;   // 2 KB bins
            std::vector<uint64_t> cnt[NKIND];
            std::vector<uint32_t> top[NKIND];
            bool present[NKIND] = {false, false, false};
            for (int k = 0; k < NKIND; ++k) cnt[k].assign(NB, 0), top[k].assign(NB, 0);
            for (uint32_t ti = 0; ti < ntiles; ++ti) {
                const int k = kind_of[ti];
                if (k < 0) continue;
                const uint32_t hb = tile_lds_bytes(g->tiles[ti]), b = std::min<uint32_t>(NB - 1, (hb + 2047) / 2048);
                cnt[k][b] += 1;
                top[k][b] = std::max(top[k][b], hb);
                present[k] = true;
            }
            // best[k][n]: the least LDS charged to kind k's tiles in n classes (n <= MAXC), cuts[k][n]: the bins after which it cuts — dynamic programme over the bins
            constexpr int MAXC = 7;
            uint64_t best[NKIND][MAXC + 1];
            std::vector<uint32_t> cuts[NKIND][MAXC + 1];
            for (int k = 0; k < NKIND; ++k) {
                auto charged = [&](uint32_t lo, uint32_t hi) {   // bins [lo, hi]
                    uint64_t n = 0;
                    uint32_t m = 0;
                    for (uint32_t b = lo; b <= hi; ++b) n += cnt[k][b], m = std::max(m, top[k][b]);
                    return n * m;
                };
                // f[n][e]: least charge of bins [0, e] in n classes; from[n][e]: end of the class before the last one
                std::vector<std::vector<uint64_t>> f(MAXC + 1, std::vector<uint64_t>(NB, ~0ull));
                std::vector<std::vector<int>> from(MAXC + 1, std::vector<int>(NB, -1));
                for (uint32_t e = 0; e < NB; ++e) f[1][e] = charged(0, e);
                for (int n = 2; n <= MAXC; ++n)
                    for (uint32_t e = (uint32_t)n - 1; e < NB; ++e)
                        for (uint32_t m = (uint32_t)n - 2; m < e; ++m) {
                            if (f[n - 1][m] == ~0ull) continue;
                            const uint64_t v = f[n - 1][m] + charged(m + 1, e);
                            if (v < f[n][e]) f[n][e] = v, from[n][e] = (int)m;
                        }
                best[k][0] = present[k] ? ~0ull : 0;
                for (int n = 1; n <= MAXC; ++n) {
                    best[k][n] = f[n][NB - 1];
                    int e = (int)NB - 1;
                    for (int q = n; q >= 2 && e >= 0; --q) {
                        e = from[q][e];
                        if (e >= 0) cuts[k][n].insert(cuts[k][n].begin(), (uint32_t)e);
                    }
                }
            }
            int nk[NKIND] = {present[0] ? 1 : 0, present[1] ? 1 : 0, present[2] ? 1 : 0};
            // BUDGET: the classes run beside each other only on streams that sit on different hardware queues, and the runtime deals a handful to the process's
            // streams round robin.  The context counts the class streams that PROVED to overlap with its stream and with one another (ctx_class_streams): three
            // at least (as since round 2), four in a process with the runtime's default queues whose context has a stream of its own — the finer cuts charge
            // less LDS to a whole-genome batch: 3.70 -> 3.55 s per schedule —, up to seven (no faster, and GPU_MAX_HW_QUEUES=8 slows the same process's KMC
            // scans down: profiles/r06_launch_classes.txt; bench.py and the executables leave the runtime's default).
            unsigned concurrent = 3;
            // (a noise driver's sampler keeps three: its iterations are launches PER CLASS when its chain is not resident — sweep, refill, tally —, and the
            //  ten-sample batch of the bench took 28 instead of 14 ms per iteration with eight classes; a resident chain is one launch whatever the classes)
            if (params->noise_seeding && !getenv("BT_GIBBS_MAX_CLASSES")) concurrent = 3;
            else if (!getenv("BT_GIBBS_MAX_CLASSES")) {
                int prio_lo = 0, prio_hi = 0;
                BT_TRYHIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                hipStream_t probe[MAXC];
                const hipError_t pe = ctx_class_streams(ctx, MAXC, getenv("BT_GIBBS_NO_PRIO") ? prio_lo : prio_hi, probe);
                if (pe == hipSuccess) concurrent = std::max<unsigned>(3, std::min<unsigned>(MAXC, ctx->class_streams_concurrent));
                else (void)hipGetLastError();
            } else
                concurrent = (unsigned)std::max(1, std::min(MAXC, atoi(getenv("BT_GIBBS_MAX_CLASSES"))));
            const int budget = std::max<int>((int)concurrent, nk[0] + nk[1] + nk[2]);
            const bool packed_kinds = getenv("BT_GIBBS_PACK") != nullptr;   // a packed launch charges every tile its own LDS need: ONE class per hot kind
            if (const char *e = getenv("BT_GIBBS_KIND_CLASSES")) {
                int v[NKIND] = {1, 1, 1};
                sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
                for (int k = 0; k < NKIND; ++k) nk[k] = present[k] ? std::min(MAXC, std::max(1, v[k])) : 0;
            } else {
                uint64_t least = ~0ull;
                int pick[NKIND] = {nk[0], nk[1], nk[2]};
                for (int a = nk[0]; a <= (present[0] ? MAXC : 0); ++a)
                    for (int b = nk[1]; b <= (present[1] ? (packed_kinds ? 1 : MAXC) : 0); ++b)
                        for (int c = nk[2]; c <= (present[2] ? (packed_kinds ? 1 : MAXC) : 0); ++c) {
                            if (a + b + c > budget) continue;
                            if (best[0][a] == ~0ull || best[1][b] == ~0ull || best[2][c] == ~0ull) continue;
                            const uint64_t v = best[0][a] + best[1][b] + best[2][c];
                            if (v < least) least = v, pick[0] = a, pick[1] = b, pick[2] = c;
                        }
                for (int k = 0; k < NKIND; ++k) nk[k] = pick[k];
            }
            for (int k = 0; k < NKIND; ++k) {
                for (uint32_t a : cuts[k][nk[k]]) kind_cuts[k].push_back(a * 2048u);
                kind_cuts[k].push_back(0xFFFFFFFFu);
            }
        }
        std::vector<bt_gibbs::LaunchClass> byb[NKIND];
        for (int k = 0; k < NKIND; ++k) byb[k].resize(kind_cuts[k].size());
        bt_gibbs::LaunchClass simple_class;
        simple_class.simple = true;
        for (uint32_t ti = 0; ti < ntiles; ++ti) {
            const uint32_t hb = tile_lds_bytes(g->tiles[ti]);
            const int k = kind_of[ti];
            if (k < 0) {
                simple_class.tiles.push_back(ti);
                simple_class.lds = std::max(simple_class.lds, hb);
                continue;
            }
            size_t b = 0;
            while (hb > kind_cuts[k][b]) ++b;
            auto &c = byb[k][b];
            c.hot = k != KIND_GENERAL;
            c.single = k == KIND_SINGLE;
            c.tiles.push_back(ti);
            c.lds = std::max(c.lds, hb);
            c.split = std::max(c.split, g->tiles[ti].split);
        }
        {   // hungriest first: those tiles run longest
            std::vector<bt_gibbs::LaunchClass> all;
            for (int k = 0; k < NKIND; ++k)
                for (auto &c : byb[k])
                    if (!c.tiles.empty()) all.push_back(std::move(c));
            std::stable_sort(all.begin(), all.end(), [](const bt_gibbs::LaunchClass &a, const bt_gibbs::LaunchClass &b) { return a.lds > b.lds; });
            for (auto &c : all) g->classes.push_back(std::move(c));
        }
        if (!simple_class.tiles.empty()) g->classes.push_back(std::move(simple_class));
        // PACKED LAUNCHES of the hot kinds (bt_gibbs_tile.hpp: BT_PACKED).  Tiles are bucketed by a duration proxy (haplotype candidates x clusters per group,
        // in powers of two: a workgroup holds its LDS until its last wavefront ends, so its tiles should run about equally long), longest first, and within a
        // bucket packed best-fit-decreasing into slabs of T bytes and at most W tiles.  (W, T) = the pair with the most tiles resident per CU: workgroups per CU
        // (the occupancy calculator: registers, LDS granularity) x tiles per workgroup.  BT_GIBBS_NO_PACK=1: one tile per workgroup, as before round 6.
        BT_TRYHIP(prepare_gibbs_hot_kernel((int)kHotBudget));
        BT_TRYHIP(prepare_gibbs_single_kernel((int)kHotBudget));
        for (auto &c : g->classes) {
            if (!c.hot || c.simple) continue;
            // (packing is OPT-IN, BT_GIBBS_PACK="W,T" or "auto": measured slower — a workgroup holds its slab until its last wavefront ends, and a slab needs W free
            // wavefront slots and T bytes at once: the bench batch 3.69 -> 4.15 s, profiles/r06_single_kernel.txt)
            const bool no_pack = getenv("BT_GIBBS_PACK") == nullptr || getenv("BT_GIBBS_NO_PACK") != nullptr;
            struct Item { uint32_t tile, need, bucket; };
            std::vector<Item> items;
            uint32_t max_need = 64;
            for (uint32_t ti : c.tiles) {
                const TileDesc &d = g->tiles[ti];
                const uint32_t need = (uint32_t)align_up(std::max<uint32_t>(tile_lds_bytes(d), 16), 64);
                uint32_t proxy = std::max<uint32_t>(1, d.Hm * d.nvm), bucket = 0;
                while (proxy >>= 1) ++bucket;
                items.push_back(Item{ti, need, bucket});
                max_need = std::max(max_need, need);
            }
            std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.bucket != b.bucket ? a.bucket > b.bucket : a.need > b.need; });
            // slots of a packing: bins in bucket order, every bin W slots
            auto pack = [&](uint32_t W, uint32_t T, std::vector<uint32_t> *slots, uint32_t *top) {
                uint32_t nbins = 0, biggest = 0;
                std::vector<uint32_t> fill, count;   // per bin
                std::multimap<uint32_t, uint32_t> open;   // room left -> bin (bins of the current bucket that can take another tile)
                uint32_t bucket = 0xFFFFFFFFu;
                if (slots) slots->clear();
                for (const Item &it : items) {
                    if (it.bucket != bucket) {
                        open.clear();
                        bucket = it.bucket;
                    }
                    auto at = open.lower_bound(it.need);   // best fit: the open bin with the least room that still takes it
                    uint32_t b;
                    if (at == open.end()) {
                        b = nbins++;
                        fill.push_back(0);
                        count.push_back(0);
                        if (slots) slots->resize((size_t)nbins * W * 2, 0xFFFFFFFFu);
                    } else {
                        b = at->second;
                        open.erase(at);
                    }
                    if (slots) {
                        (*slots)[((size_t)b * W + count[b]) * 2] = it.tile;
                        (*slots)[((size_t)b * W + count[b]) * 2 + 1] = fill[b];
                    }
                    fill[b] += it.need;
                    count[b] += 1;
                    biggest = std::max(biggest, fill[b]);
                    if (count[b] < W && fill[b] < T) open.emplace(T - fill[b], b);
                }
                if (top) *top = biggest;
                return nbins;
            };
            auto occupancy = [&](uint32_t W, uint32_t lds, int *occ) { return c.single ? occupancy_gibbs_single_kernel(occ, (int)(LANES * W), lds) : occupancy_gibbs_hot_kernel(occ, (int)(LANES * W), lds); };
            uint32_t bestW = 1, bestT = max_need;
            double best_score = -1;
            const uint32_t widths_single[] = {1, 2, 3, 4}, widths_hot[] = {1, 2, 4, 8};
            for (uint32_t wi = 0; wi < 4 && !no_pack; ++wi) {
                const uint32_t W = c.single ? widths_single[wi] : widths_hot[wi];
                for (uint32_t per_cu = 1; per_cu <= 24; ++per_cu) {
                    uint32_t T = per_cu == 1 ? kHotBudget : (uint32_t)((163840u / per_cu) & ~255u);
                    T = std::min(T, kHotBudget);
                    if (T < max_need) break;
                    uint32_t top = 0;
                    const uint32_t nbins = pack(W, T, nullptr, &top);
                    int occ = 0;
                    BT_TRYHIP(occupancy(W, top, &occ));
                    const double score = (double)occ * (double)items.size() / (double)nbins - 1e-3 * W;   // (ties: the fewer wavefronts share a workgroup's lifetime the better)
                    if (score > best_score) best_score = score, bestW = W, bestT = T;
                }
            }
            if (const char *e = getenv("BT_GIBBS_PACK")) {   // tuning: "W,T"
                unsigned w = 0, t = 0;
                if (sscanf(e, "%u,%u", &w, &t) == 2 && w >= 1 && w <= (c.single ? 4u : 8u) && t >= max_need && t <= kHotBudget) bestW = w, bestT = t;
            }
            std::vector<uint32_t> slots;
            uint32_t top = 0;
            c.pack_wgs = pack(bestW, bestT, &slots, &top);
            c.pack_waves = bestW;
            c.pack_lds = top;
            BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&c.d_pack), slots.size() * 4));
            g->allocs.push_back(c.d_pack);
            BT_TRYHIP(staged_upload(ctx, c.d_pack, slots.data(), slots.size() * 4));
            if (getenv("BT_GIBBS_DEBUG")) {
                int occ = 0;
                (void)occupancy(bestW, top, &occ);
                uint64_t need_sum = 0;
                for (const Item &it : items) need_sum += it.need;
                fprintf(stderr, "bt_gibbs: packed %s class: %zu tiles (%.1f MB of LDS needs) in %u workgroups of %u wavefronts, %u B of LDS each (%.1f MB charged), %d workgroups per CU\n",
                        c.single ? "single" : "hot", items.size(), need_sum / 1048576.0, c.pack_wgs, bestW, top, (double)c.pack_wgs * top / 1048576.0, occ);
            }
        }
        BT_TRYHIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gibbs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHotBudget));
        BT_TRYHIP(prepare_gibbs_simple_kernel((int)kHotBudget));
        BT_TRYHIP(prepare_gibbs_hot_kernel((int)kHotBudget));
        BT_TRYHIP(prepare_gibbs_single_kernel((int)kHotBudget));
        lap("class cut");
        BT_TRYHIP(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
        lap("event");
        g->wide_fill = !getenv("BT_GIBBS_NO_WIDE_FILL");
        g->noise_in_gibbs_kernel = getenv("BT_GIBBS_NOISE_GLOBAL_ATOMICS") != nullptr;
        BT_TRYHIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gibbs_noise_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHotBudget));
        if (const char *e = getenv("BT_GIBBS_STEPWISE")) g->stepwise_run = atoi(e) != 0 && g->wide_fill;
        lap("attributes");
        hipStream_t class_streams[16] = {};
        if (g->classes.size() > 16) {
            bt_gibbs_destroy(g);
            return fail("bt_gibbs_create: more launch classes than class streams");
        }
        for (size_t i = 0; i < g->classes.size(); ++i) {
            auto &c = g->classes[i];
            BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&c.d_tiles), c.tiles.size() * 4));
            g->allocs.push_back(c.d_tiles);
            BT_TRYHIP(staged_upload(ctx, c.d_tiles, c.tiles.data(), c.tiles.size() * 4));
            std::vector<FillChunk> fill;
            for (uint32_t ti : c.tiles) {
                const TileDesc &d = g->tiles[ti];
                if (!(d.cache_mode == 0 && !d.simple && d.cache_entries > BT_UC_INVALIDATE_MIN && d.hoff[A_UCACHE] == NOHOT)) continue;   // (cache_clear's condition)
                const uint64_t words = ((uint64_t)d.nvm * d.cache_entries) << d.wsh;
                for (uint64_t w = 0; w < words; w += 32768) fill.push_back(FillChunk{d.base + d.off[A_UCACHE] + w * 8, (uint32_t)std::min<uint64_t>(32768, words - w), 0u});
            }
            std::vector<PrefillItem> pre;
            if (!fill.empty() && !getenv("BT_GIBBS_NO_PREFILL")) {
                std::vector<uint8_t> in_class(ntiles, 0);
                for (uint32_t ti : c.tiles) {
                    const TileDesc &d = g->tiles[ti];
                    in_class[ti] = d.cache_mode == 0 && !d.simple && d.cache_entries > BT_UC_INVALIDATE_MIN && d.hoff[A_UCACHE] == NOHOT;
                }
                for (uint32_t gi = 0; gi < g->G; ++gi)
                    if (in_class[g->group_tile[gi]])
                        for (uint32_t v = 0; v < g->group_nvert[gi]; ++v) pre.push_back(PrefillItem{g->group_tile[gi], g->group_lane[gi], v, 0u});
            }
            if (!pre.empty()) {
                BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&c.d_prefill), pre.size() * sizeof(PrefillItem)));
                g->allocs.push_back(c.d_prefill);
                BT_TRYHIP(staged_upload(ctx, c.d_prefill, pre.data(), pre.size() * sizeof(PrefillItem)));
                c.num_prefill = (uint32_t)pre.size();
            }
            if (!fill.empty() && !getenv("BT_GIBBS_NO_WIDE_FILL")) {
                BT_TRYHIP(hipMalloc(reinterpret_cast<void **>(&c.d_fill), fill.size() * sizeof(FillChunk)));
                g->allocs.push_back(c.d_fill);
                BT_TRYHIP(staged_upload(ctx, c.d_fill, fill.data(), fill.size() * sizeof(FillChunk)));
                c.num_fill = (uint32_t)fill.size();
            }
            lap("class lists");
            BT_TRYHIP(hipEventCreateWithFlags(&c.ready, hipEventDisableTiming));
            if (i + 1 < g->classes.size()) {   // the last class runs on the context's stream
                int prio_lo = 0, prio_hi = 0;   // the hungrier classes (created first) get the higher dispatch priority
                BT_TRYHIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                int prio = getenv("BT_GIBBS_NO_PRIO") ? prio_lo : prio_hi;
                if (const char *e = getenv("BT_GIBBS_CLASS_PRIO")) prio = atoi(e);   // tuning: 0 = the priority of the context's stream (the two-haplotype class)
                BT_TRYHIP(ctx_class_streams(ctx, (unsigned)i + 1, prio, class_streams));   // (the context's: borrowed)
                c.stream = class_streams[i];
                BT_TRYHIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
            }
            lap("class stream");
        }
        if (getenv("BT_GIBBS_DEBUG")) {   // tuning aid: the tiles' LDS need in 2 KB bins (a launch class is charged its hungriest tile's)
            std::vector<uint32_t> bins(40, 0);
            for (uint32_t ti = 0; ti < ntiles; ++ti) bins[std::min<uint32_t>(39, tile_lds_bytes(g->tiles[ti]) / 2048)] += 1;
            fprintf(stderr, "bt_gibbs: tiles by LDS need:");
            for (uint32_t b = 0; b < 40; ++b)
                if (bins[b]) fprintf(stderr, " [%u-%u KB: %u]", 2 * b, 2 * b + 2, bins[b]);
            fprintf(stderr, "\n");
        }
        if (getenv("BT_GIBBS_DEBUG")) {   // tuning aid: how the batch was tiled
            fprintf(stderr, "bt_gibbs: %u tiles in %zu launch classes:", ntiles, g->classes.size());
            for (auto &c : g->classes) fprintf(stderr, " [%zu tiles, lds %u B, split %u%s%s]", c.tiles.size(), c.lds, c.split, c.simple ? ", simple" : "", c.single ? ", single" : (c.hot ? ", hot" : ""));
            fprintf(stderr, "; tile 0: hot_bytes %u lds_stride %u copies %u\n", g->tiles[0].hot_bytes, g->tiles[0].lds_stride, g->tiles[0].copies);
        }
    }
    BT_TRYHIP(hipStreamSynchronize(ctx->stream));
    lap("tables, launch classes");
    BT_TRY(launch(g, OP_SETUP, 0, 0, nullptr));
    BT_TRYHIP(hipStreamSynchronize(ctx->stream));
#undef BT_TRY
#undef BT_TRYHIP
    lap("set-up launch (synchronised)");
    if (timing) fprintf(stderr, "bt_gibbs: construction of %u groups in %u tiles, pool %.1f MB:%s\n", G, ntiles, g->pool_bytes / 1048576.0, t_text.c_str());
    *out = g;
    return BT_OK;
}

int bt_gibbs_destroy(bt_gibbs *g) {
    if (!g) return BT_OK;
    (void)hipSetDevice(g->ctx->device);
    if (g->nc.active) (void)bt_gibbs_noise_chain_end(g);   // (releases the resident launch)
    (void)hipStreamSynchronize(g->ctx->stream);
    const bool dbg_release = getenv("BT_GIBBS_DEBUG") != nullptr;
    const auto t_rel0 = std::chrono::steady_clock::now();
    for (void *p : g->allocs) {
        if (!p) continue;
        if (p == g->d_pool && g->keep_pool && !g->ctx->pool_cache && g->pool_alloc_bytes) {   // (the stream was waited for above: nothing is in flight on the pool)
            g->ctx->pool_cache = p;
            g->ctx->pool_cache_bytes = g->pool_alloc_bytes;
            continue;
        }
        (void)hipFree(p);
    }
    const auto t_rel1 = std::chrono::steady_clock::now();
    if (g->d_trace) (void)hipFree(g->d_trace);
    if (g->d_trace_counter) (void)hipFree(g->d_trace_counter);
    if (g->d_wire) (void)hipFree(g->d_wire);
    {
        const size_t nh = (size_t)g->S * 256;
        if (g->recycles) {
            ctx_host_give(g->ctx, g->h_pin_hist, nh * 8, hipHostMallocDefault);
            ctx_host_give(g->ctx, g->h_pin_noise, nh * 8, hipHostMallocDefault);
            ctx_host_give(g->ctx, g->nc.h_mail, nh * 16 + 512, hipHostMallocCoherent | hipHostMallocMapped);
        } else {
            if (g->h_pin_hist) (void)hipHostFree(g->h_pin_hist);
            if (g->h_pin_noise) (void)hipHostFree(g->h_pin_noise);
            if (g->nc.h_mail) (void)hipHostFree(g->nc.h_mail);
        }
        if (g->nc.h_phase) (void)hipHostFree(g->nc.h_phase);
    }
    const auto t_rel1b = std::chrono::steady_clock::now();
    if (g->d_iter_hist) (void)hipFree(g->d_iter_hist);
    if (g->nc.d_sync) (void)hipFree(g->nc.d_sync);
    if (g->nc.d_ctl) (void)hipFree(g->nc.d_ctl);
    if (g->nc.d_busy) (void)hipFree(g->nc.d_busy);
    if (g->nc.d_help_items) (void)hipFree(g->nc.d_help_items);
    if (g->nc.d_help_words) (void)hipFree(g->nc.d_help_words);
    if (g->nc.d_tile_units) (void)hipFree(g->nc.d_tile_units);
    const auto t_rel1c = std::chrono::steady_clock::now();
    for (auto &c : g->classes) {
        if (c.stream) {
            (void)hipStreamSynchronize(c.stream);
            // (the stream belongs to the context)
        }
        if (c.done) (void)hipEventDestroy(c.done);
        if (c.ready) (void)hipEventDestroy(c.ready);
    }
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (dbg_release) {
        const auto t_rel2 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "bt_gibbs_destroy: device allocations %.2f ms (pool %s), pinned buffers %.2f ms, the chain's device words %.2f ms, streams + events %.2f ms\n", ms(t_rel0, t_rel1),
                     g->ctx->pool_cache == g->d_pool ? "kept for the next sampler" : "freed", ms(t_rel1, t_rel1b), ms(t_rel1b, t_rel1c), ms(t_rel1c, t_rel2));
    }
    delete g;
    return BT_OK;
}

int bt_gibbs_set_lut(bt_gibbs *g, const double *h_genomic, const double *h_noise) {
    if (!g || !h_genomic || !h_noise) return fail("bt_gibbs_set_lut: null argument");
    if (chain_in_flight(g)) return fail("bt_gibbs_set_lut: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipMemcpyAsync(g->d_lut_g, h_genomic, (size_t)g->S * 65536 * 8, hipMemcpyHostToDevice, g->ctx->stream));
    BT_HIP(hipMemcpyAsync(g->d_lut_n, h_noise, (size_t)g->S * 256 * 8, hipMemcpyHostToDevice, g->ctx->stream));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    g->h_lut_n.assign(h_noise, h_noise + (size_t)g->S * 256);
    g->lut_set = true;
    return BT_OK;
}

int bt_gibbs_set_noise_lut(bt_gibbs *g, const double *h_noise) {
    if (!g || !h_noise) return fail("bt_gibbs_set_noise_lut: null argument");
    if (chain_in_flight(g)) return fail("bt_gibbs_set_noise_lut: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipMemcpyAsync(g->d_lut_n, h_noise, (size_t)g->S * 256 * 8, hipMemcpyHostToDevice, g->ctx->stream));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    g->h_lut_n.assign(h_noise, h_noise + (size_t)g->S * 256);
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
    if (g->stepwise_run) {
        // chain by chain: the launches of a chain start are followed by the wide table refill (nan_fill_kernel, ucache_prefill_kernel) instead of
        // every tile filling its own tables with its own lanes; the same operations in the same order as OP_RUN
        for (uint32_t chain = 0; chain < g->P.num_chains; ++chain) {
            int rc = launch(g, OP_INIT_CHAIN, chain, 0, nullptr);
            if (rc == BT_OK && g->P.burn_in) rc = launch(g, OP_SWEEP, g->P.burn_in, 0, nullptr);
            if (rc == BT_OK && g->P.num_iterations) rc = launch(g, OP_SWEEP, g->P.num_iterations, 1, nullptr);
            if (rc != BT_OK) return rc;
        }
        return BT_OK;
    }
    return launch(g, OP_RUN, 0, 0, nullptr);
}

int bt_gibbs_noise_counts(bt_gibbs *g, uint64_t *d_hist, int zero_first) {
    if (!g || !d_hist) return fail("bt_gibbs_noise_counts: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    if (zero_first) BT_HIP(hipMemsetAsync(d_hist, 0, (size_t)g->S * 256 * 8, g->ctx->stream));
    return launch(g, OP_NOISE, 1, 0, reinterpret_cast<unsigned long long *>(d_hist));
}

int bt_gibbs_noise_iteration(bt_gibbs *g, const double *h_noise, int collect_samples, uint64_t *h_hist) {
    if (!g || !h_hist) return fail("bt_gibbs_noise_iteration: null argument");
    BT_HIP(hipSetDevice(g->ctx->device));
    const size_t nh = (size_t)g->S * 256;
    if (!g->h_pin_hist) {
        BT_HIP(ctx_host_take(g->ctx, reinterpret_cast<void **>(&g->h_pin_hist), nh * 8, hipHostMallocDefault));
        BT_HIP(ctx_host_take(g->ctx, reinterpret_cast<void **>(&g->h_pin_noise), nh * 8, hipHostMallocDefault));
        BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_iter_hist), nh * 8));
    }
    hipStream_t st = g->ctx->stream;
    if (h_noise) {   // (the staging buffer is free: the previous call ended with a synchronisation after its upload)
        g->h_lut_n.assign(h_noise, h_noise + nh);
        std::memcpy(g->h_pin_noise, h_noise, nh * 8);
        BT_HIP(hipMemcpyAsync(g->d_lut_n, g->h_pin_noise, nh * 8, hipMemcpyHostToDevice, st));
    }
    int rc = launch(g, OP_SWEEP, 1, collect_samples ? 1u : 0u, nullptr);
    if (rc != BT_OK) return rc;
    BT_HIP(hipMemsetAsync(g->d_iter_hist, 0, nh * 8, st));
    rc = launch(g, OP_NOISE, 1, 0, reinterpret_cast<unsigned long long *>(g->d_iter_hist));
    if (rc != BT_OK) return rc;
    BT_HIP(hipMemcpyAsync(g->h_pin_hist, g->d_iter_hist, nh * 8, hipMemcpyDeviceToHost, st));
    BT_HIP(hipStreamSynchronize(st));
    std::memcpy(h_hist, g->h_pin_hist, nh * 8);
    return BT_OK;
}


// ---- a chain of a noise driver as ONE resident launch (bt_noise_chain.hpp) ----
namespace {
constexpr uint32_t kResidentTableEntries = 1u << 21;
inline uint32_t nc_bins_bytes(uint32_t S) { return (S * NC_BINS + 8u + NC_HELP_MAXH / 2u) * 4u; }   // the bins, the flag word, (aligned) the profiling time stamp, the help phase's words + haplotype list
struct NcMail {   // layout of the pinned mailbox
    uint64_t *hist;
    double *table;
    uint32_t *hist_seq, *table_seq;
};
inline NcMail nc_mail(bt_gibbs *g) {
    const size_t nh = (size_t)g->S * 256;
    uint8_t *b = g->nc.h_mail;
    return NcMail{reinterpret_cast<uint64_t *>(b), reinterpret_cast<double *>(b + nh * 8), reinterpret_cast<uint32_t *>(b + nh * 16), reinterpret_cast<uint32_t *>(b + nh * 16 + 256)};
}
// where a chain that did not finish stood (after the launch has ended): for the error text
static std::string nc_state_text(bt_gibbs *g, uint32_t it);
static std::string nc_phase_text(bt_gibbs *g) {
    if (!g->nc.h_phase) return "";
    std::string t = " phases:";
    for (uint32_t i = 0; i < g->ntiles * 64 && i < 128; ++i) {
        char b[32];
        snprintf(b, sizeof b, " %u:%x", i, g->nc.h_phase[i]);
        t += b;
    }
    return t;
}
static std::string nc_state_text(bt_gibbs *g, uint32_t it) {
    uint32_t arrived = 0, aborted = 0, seq = 0;
    const size_t nh = (size_t)g->S * 256;
    (void)hipMemcpy(&arrived, g->nc.d_sync + nh * 8, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&aborted, g->nc.d_sync + nh * 8 + 256, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&seq, g->nc.d_sync + nh * 8 + 512, 4, hipMemcpyDeviceToHost);
    const NcMail mail = nc_mail(g);
    return " [iteration " + std::to_string(it) + " of " + std::to_string(g->nc.n) + ": " + std::to_string(arrived) + " arrivals of " +
           std::to_string(g->nc.ctl.total_wgs) + " workgroups per iteration, device table_seq " + std::to_string(seq) + ", host hist_seq " +
           std::to_string(*mail.hist_seq) + ", host table_seq " + std::to_string(*mail.table_seq) + ", device abort flag " + std::to_string(aborted) + "]";
}
inline uint32_t nc_load(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void nc_store(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
double nc_timeout_seconds() {
    if (const char *e = getenv("BT_NOISE_CHAIN_TIMEOUT_S")) return std::max(0.001, atof(e));
    return 60.0;
}
double nc_rollcall_seconds() {   // how long the workgroups of a resident launch wait for each other to have STARTED (bt_noise_chain.hpp: roll call)
    if (const char *e = getenv("BT_NOISE_CHAIN_ROLLCALL_S")) return std::max(0.001, atof(e));
    return 2.0;
}
// a roll call failed on this device: something else holds wavefront slots there (a CU mask, another process, another rank), and it will for the chains to come —
// the samplers of this process stop asking for resident launches on it instead of paying the roll call's deadline chain after chain
std::atomic<bool> g_no_resident_chain[64];
}  // namespace

int bt_gibbs_noise_chain_begin(bt_gibbs *g, uint32_t num_iterations, uint32_t first_collect, int *resident) {
    if (!g || !resident) return fail("bt_gibbs_noise_chain_begin: null argument");
    *resident = 0;
    if (g->nc.active) return fail("bt_gibbs_noise_chain_begin: a chain is already in progress");
    if (!g->lut_set) return fail("bt_gibbs: count-model LUTs not set (bt_gibbs_set_lut)");
    if (num_iterations == 0 || getenv("BT_NOISE_CHAIN_OFF")) return BT_OK;
    if (g->ctx->device >= 0 && g->ctx->device < 64 && g_no_resident_chain[g->ctx->device].load()) return BT_OK;
    BT_HIP(hipSetDevice(g->ctx->device));
    // (1) Can every tile have its workgroup resident at the same time?  One launch, one workgroup shape (gibbs_chain_kernel: a wavefront per tile, the LDS
    // need of the hungriest tile + the bins): tiles <= workgroups per CU x CUs, exactly.  Tiles with large dense tables want the whole-GPU refill between
    // iterations (nan_fill_kernel / ucache_prefill_kernel), which a resident launch cannot give them: such batches keep the launch-per-iteration path.
    const double fill_limit = getenv("BT_NOISE_CHAIN_FILL") ? atof(getenv("BT_NOISE_CHAIN_FILL")) : 1.0;
    // Tiles with large dense tables of unique-k-mer sums: the FIRST sweep of a chain asks for every entry (all haplotypes have a non-zero frequency at a chain
    // start), which the whole GPU computes far faster than the tile's own lanes (nan_fill_kernel / ucache_prefill_kernel between launches) — so iteration 0 of
    // such a chain runs as ordinary launches and the resident launch starts with iteration 1, where the sweeps ask for the pairs of the few haplotypes left and
    // the tile invalidates its table itself (cache_clear's "dirty = 2").  BT_NOISE_CHAIN_WIDE=1: resident from iteration 0 on (tests).
    // The per-iteration refill of such tables is shared out among ALL workgroups of the chain (bt_noise_help.hpp: a 256-candidate cluster at thirty samples keeps
    // some sixty haplotypes alive — 55 000 sums over the k-mer subset per iteration, 43 ms by the tile's own lanes).  Above kResidentTableEntries entries per
    // group (the owner invalidates its table itself every iteration) a batch keeps the launch-per-iteration path.
    bool has_wide = false;
    uint32_t biggest = 0;
    for (const auto &c : g->classes)
        if (c.num_fill || c.num_prefill) {
            has_wide = true;
            for (uint32_t ti : c.tiles) biggest = std::max(biggest, g->tiles[ti].cache_entries);
        }
    const bool forced = getenv("BT_NOISE_CHAIN_WIDE") != nullptr;
    if (biggest > kResidentTableEntries && !forced) return BT_OK;
    const uint32_t it_begin = has_wide && !forced ? 1u : 0u;
    if (num_iterations <= it_begin) return BT_OK;
    // LDS per workgroup: every workgroup of the launch is charged the same amount.  The cap is the largest tile need with which all tiles are still resident
    // together; the (few, many-candidate) tiles above it keep their hot arrays in HBM for the chain (RESIDENT_NEVER); the tiles of two-haplotype clusters
    // (the bulk, and they cannot do without their block) must fit.
    if (g->ntiles == 0) return BT_OK;
    std::vector<uint32_t> needs;
    uint32_t simple_need = 0;
    for (const TileDesc &d : g->tiles) {
        needs.push_back(tile_lds_bytes(d));
        if (d.simple) simple_need = std::max(simple_need, needs.back());
    }
    std::sort(needs.begin(), needs.end());
    needs.erase(std::unique(needs.begin(), needs.end()), needs.end());
    const double fits = fill_limit * g->ctx->num_cu;
    uint32_t lds_cap = 0, bins_off = 0, lds = 0;
    int occ = 0;
    bool found = false;
    for (size_t lo = 0, hi = needs.size(); lo < hi;) {   // (occupancy is monotone in the LDS need: binary search over the distinct needs)
        const size_t mid = (lo + hi) / 2;
        const uint32_t off = (needs[mid] + 15u) & ~15u, bytes = off + nc_bins_bytes(g->S);
        int o = 0;
        if (bytes <= kHotBudget) BT_HIP(occupancy_gibbs_chain_kernel(&o, bytes));
        if (o >= 1 && (double)g->ntiles <= fits * o) {
            found = true, lds_cap = needs[mid], bins_off = off, lds = bytes, occ = o;
            lo = mid + 1;
        } else
            hi = mid;
    }
    if (const char *e = getenv("BT_NOISE_CHAIN_LDS_CAP"))   // (tests: push more tiles out of LDS than residency asks for; the workgroups are still charged `lds`)
        if (found) lds_cap = std::min(lds_cap, std::max<uint32_t>((uint32_t)strtoul(e, nullptr, 0), simple_need));
    if (getenv("BT_GIBBS_DEBUG")) {
        size_t demoted = 0;
        for (const TileDesc &d : g->tiles) demoted += tile_lds_bytes(d) > lds_cap;
        fprintf(stderr, "bt_gibbs_noise_chain_begin: %u tiles (LDS needs %u .. %u B, two-haplotype tiles %u B): %s, %u B of LDS per workgroup, %d workgroups per CU x %d CUs, %zu tiles without LDS\n",
                g->ntiles, needs.front(), needs.back(), simple_need, found && lds_cap >= simple_need ? "resident" : "NOT resident", lds, occ, g->ctx->num_cu, demoted);
    }
    if (!found || lds_cap < simple_need) return BT_OK;
    const uint32_t total = g->ntiles;
    // (2) mailbox + device words
    const size_t nh = (size_t)g->S * 256;
    // (each pointer on its own: a failed allocation must not leave the next call with a mailbox but no device words)
    if (!g->nc.h_mail) BT_HIP(ctx_host_take(g->ctx, reinterpret_cast<void **>(&g->nc.h_mail), nh * 16 + 512, hipHostMallocCoherent | hipHostMallocMapped));
    if (!g->nc.d_sync) BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->nc.d_sync), nh * 8 + 512 + (size_t)NC_SEQ_COPIES * NC_SEQ_STRIDE * 4));
    if (!g->nc.d_ctl) BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->nc.d_ctl), sizeof(NoiseChainCtl)));
    int wall_khz = 0;   // rate of wall_clock64() (the deadlines of the device-side waits)
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, g->ctx->device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    if (getenv("BT_GIBBS_DEBUG")) fprintf(stderr, "bt_gibbs_noise_chain_begin: wall clock %d kHz\n", wall_khz);
    const NcMail mail = nc_mail(g);
    std::memset(g->nc.h_mail, 0, nh * 16 + 512);
    std::memcpy(mail.table, g->h_lut_n.data(), nh * 8);   // (a step without a new table hands the current one over again)
    hipStream_t st = g->ctx->stream;
    BT_HIP(hipMemsetAsync(g->nc.d_sync, 0, nh * 8 + 512 + (size_t)NC_SEQ_COPIES * NC_SEQ_STRIDE * 4, st));
    {
        NoiseChainCtl &k = g->nc.ctl;
        k.hist = reinterpret_cast<unsigned long long *>(g->nc.d_sync);
        k.arrived = reinterpret_cast<uint32_t *>(g->nc.d_sync + nh * 8);
        k.abort_flag = reinterpret_cast<uint32_t *>(g->nc.d_sync + nh * 8 + 256);
        k.rollcall = reinterpret_cast<uint32_t *>(g->nc.d_sync + nh * 8 + 128);
        k.rollcall_ticks = (unsigned long long)(std::min(nc_timeout_seconds(), nc_rollcall_seconds()) * 1e3 * wall_khz);
        k.table_seq = reinterpret_cast<uint32_t *>(g->nc.d_sync + nh * 8 + 512);
        k.lut_n = g->d_lut_n;
        k.h_hist = reinterpret_cast<unsigned long long *>(mail.hist);
        k.h_table = mail.table;
        k.h_hist_seq = mail.hist_seq;
        k.h_table_seq = mail.table_seq;
        k.total_wgs = total;
        k.bins_off = bins_off;
        k.n_iterations = num_iterations;
        k.first_collect = first_collect;
        k.S = g->S;
        k.lds_cap = lds_cap;
        k.timeout_ticks = (unsigned long long)(nc_timeout_seconds() * 1e3 * wall_khz);
        // the help phase's work list: once per sampler
        if (!g->nc.help_built) {
            std::vector<HelpItem> items;
            std::vector<uint32_t> units(g->ntiles, 0);
            for (uint32_t gi = 0; gi < g->G; ++gi) {
                const TileDesc &d = g->tiles[g->group_tile[gi]];
                if (!help_table(d.cache_mode, d.simple, d.cache_entries, d.hoff[A_UCACHE], d.Dcm)) continue;
                for (uint32_t v = 0; v < g->group_nvert[gi]; ++v) {
                    items.push_back(HelpItem{g->group_tile[gi], g->group_lane[gi], v, 0u});
                    units[g->group_tile[gi]] += g->S;
                }
            }
            g->nc.help_units = getenv("BT_NOISE_CHAIN_NO_HELP") ? 0u : (uint32_t)std::min<uint64_t>((uint64_t)items.size() * g->S, 0x7FFFFFFFu);
            if (g->nc.help_units) {
                BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->nc.d_help_items), items.size() * sizeof(HelpItem)));
                BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->nc.d_help_words), (64 + (size_t)g->ntiles) * 4));
                BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->nc.d_tile_units), (size_t)g->ntiles * 4));
                BT_HIP(hipMemcpyAsync(g->nc.d_help_items, items.data(), items.size() * sizeof(HelpItem), hipMemcpyHostToDevice, st));
                BT_HIP(hipMemcpyAsync(g->nc.d_tile_units, units.data(), (size_t)g->ntiles * 4, hipMemcpyHostToDevice, st));
                BT_HIP(hipStreamSynchronize(st));
            }
            g->nc.help_built = true;
        }
        k.help_items = g->nc.d_help_items;
        k.help_units = g->nc.help_units;
        k.help_next = g->nc.d_help_words;
        k.help_done = g->nc.d_help_words ? g->nc.d_help_words + 64 : nullptr;
        k.tile_units = g->nc.d_tile_units;
        k.num_tiles = g->ntiles;
        {   // helpers (gibbs_chain_kernel): top the launch up to what can be resident when there is help to give
            const uint64_t room = (uint64_t)(fits * occ);
            uint32_t helpers = g->nc.help_units && room > g->ntiles && !getenv("BT_NOISE_CHAIN_NO_HELPERS") ? (uint32_t)std::min<uint64_t>(room - g->ntiles, (uint64_t)g->nc.help_units) : 0u;
            if (const char *e = getenv("BT_NOISE_CHAIN_HELPERS")) helpers = std::min<uint32_t>(helpers, (uint32_t)atoi(e));
            g->nc.num_wgs = g->ntiles + helpers;
            k.total_wgs = g->nc.num_wgs;
            // TEST HOOK (tests/_rollcall_child.py): the roll call expects workgroups that are never launched — what a launch looks like from the inside when a CU
            // mask or another process keeps some of its workgroups from being resident
            if (const char *e = getenv("BT_NOISE_CHAIN_TEST_ABSENT_WGS")) k.total_wgs += (uint32_t)std::max(0, atoi(e));
        }
        if (g->nc.help_units) BT_HIP(hipMemsetAsync(g->nc.d_help_words, 0, (64 + (size_t)g->ntiles) * 4, st));
        k.h_phase = nullptr;
        if (getenv("BT_NOISE_CHAIN_DEBUG_FLAGS") && (atoi(getenv("BT_NOISE_CHAIN_DEBUG_FLAGS")) & 2)) {
            if (!g->nc.h_phase) BT_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->nc.h_phase), ((size_t)g->ntiles + 16384) * 256, hipHostMallocCoherent | hipHostMallocMapped));
            std::memset(g->nc.h_phase, 0, ((size_t)g->ntiles + 16384) * 256);
            k.h_phase = g->nc.h_phase;
        }
        k.busy = nullptr;
        k.debug_flags = getenv("BT_NOISE_CHAIN_DEBUG_FLAGS") ? (uint32_t)atoi(getenv("BT_NOISE_CHAIN_DEBUG_FLAGS")) : 0u;
        if (getenv("BT_NOISE_CHAIN_PROF")) {
            if (!g->nc.d_busy) BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->nc.d_busy), ((size_t)g->ntiles + 16384) * 8));
            BT_HIP(hipMemsetAsync(g->nc.d_busy, 0, ((size_t)g->ntiles + 16384) * 8, st));
            k.busy = g->nc.d_busy;
        }
    }
    g->nc.ctl.it_begin = it_begin;
    g->nc.it_begin = it_begin;
    g->nc.first_collect = first_collect;
    g->nc.lds = lds;
    g->nc.launched = false;
    BT_HIP(hipMemcpyAsync(g->nc.d_ctl, &g->nc.ctl, sizeof(NoiseChainCtl), hipMemcpyHostToDevice, st));
    BT_HIP(hipStreamSynchronize(st));   // (the control block is read from pageable memory)
    BT_HIP(prepare_gibbs_chain_kernel((int)kHotBudget));
    if (it_begin == 0) {
        TraceCfg tr{g->trace_sweeps, g->d_trace_counter, g->d_trace};
        BT_HIP(launch_gibbs_chain_kernel(g->nc.num_wgs, lds, st, g->d_tiles, g->d_pool, g->d_params, g->nc.d_ctl, tr));
        g->prefill_armed = false;
        g->nc.launched = true;
    }
    g->nc.active = true;
    g->nc.fallback = false;
    g->nc.n = num_iterations;
    g->nc.next = 0;
    *resident = 1;
    return BT_OK;
}

int bt_gibbs_noise_chain_step(bt_gibbs *g, const double *h_noise, uint64_t *h_hist) {
    if (!g || !h_hist) return fail("bt_gibbs_noise_chain_step: null argument");
    if (!g->nc.active) return fail("bt_gibbs_noise_chain_step: no chain in progress (bt_gibbs_noise_chain_begin)");
    const uint32_t it = g->nc.next;
    if (it >= g->nc.n) return fail("bt_gibbs_noise_chain_step: the chain has no iteration left");
    if (it == 0 && h_noise) return fail("bt_gibbs_noise_chain_step: the first iteration of a chain runs with the sampler's table (bt_gibbs_set_noise_lut before the chain)");
    const size_t nh = (size_t)g->S * 256;
    const NcMail mail = nc_mail(g);
    if (g->nc.fallback) {   // the resident launch never started its sweeps (roll call): the chain continues launch by launch, the same values
        const int rc = bt_gibbs_noise_iteration(g, it > 0 ? h_noise : nullptr, it >= g->nc.first_collect ? 1 : 0, h_hist);
        if (rc != BT_OK) {
            g->nc.active = g->nc.fallback = false;
            return rc;
        }
        g->nc.next = it + 1;
        return BT_OK;
    }
    if (!g->nc.launched && it < g->nc.it_begin) {   // iteration 0 of a chain with large tables: ordinary launches (sweep with the whole-GPU refill, tally, one synchronisation)
        const int rc = bt_gibbs_noise_iteration(g, nullptr, it >= g->nc.first_collect ? 1 : 0, h_hist);
        if (rc != BT_OK) {
            g->nc.active = false;
            return rc;
        }
        g->nc.next = it + 1;
        return BT_OK;
    }
    if (!g->nc.launched) {   // it == it_begin > 0: this iteration's table goes up the ordinary way, the large tables are refilled by the whole GPU, then the launch
        BT_HIP(hipSetDevice(g->ctx->device));
        hipStream_t st = g->ctx->stream;
        if (h_noise) {
            std::memcpy(mail.table, h_noise, nh * 8);
            g->h_lut_n.assign(h_noise, h_noise + nh);
        }
        BT_HIP(hipMemcpyAsync(g->d_lut_n, mail.table, nh * 8, hipMemcpyHostToDevice, st));
        if (g->prefill_armed)
            for (auto &c : g->classes)
                if (c.num_prefill) {
                    const unsigned threads = (uint64_t)c.num_prefill * g->S >= 4096 ? 64u : 256u;
                    hipLaunchKernelGGL(ucache_prefill_kernel, dim3(c.num_prefill, g->S), dim3(threads), 0, st, (const TileDesc *)g->d_tiles, g->d_pool, (const GParams *)g->d_params,
                                       (const PrefillItem *)c.d_prefill);
                    BT_CHECK_LAUNCH();
                }
        TraceCfg tr{g->trace_sweeps, g->d_trace_counter, g->d_trace};
        BT_HIP(launch_gibbs_chain_kernel(g->nc.num_wgs, g->nc.lds, st, g->d_tiles, g->d_pool, g->d_params, g->nc.d_ctl, tr));
        g->prefill_armed = false;
        g->nc.launched = true;
    } else if (it > 0) {   // the table of this iteration's sweep: the workgroup that published histogram `it` is waiting for it
        if (h_noise) {
            std::memcpy(mail.table, h_noise, nh * 8);
            g->h_lut_n.assign(h_noise, h_noise + nh);
        }
        nc_store(mail.table_seq, it);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = nc_timeout_seconds();
    uint32_t v = nc_load(mail.hist_seq);
    for (uint64_t spins = 0; v < it + 1u; ++spins) {
        __builtin_ia32_pause();
        v = nc_load(mail.hist_seq);
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            nc_store(mail.table_seq, NC_ABORT);
            if (g->nc.h_phase) fprintf(stderr, "bt_gibbs_noise_chain_step: deadline passed at iteration %u;%s\n", it, nc_phase_text(g).c_str());
            (void)bt_gibbs_noise_chain_end(g);
            return fail("bt_gibbs_noise_chain_step: no histogram from the device within the deadline (BT_NOISE_CHAIN_TIMEOUT_S)" + nc_state_text(g, it));
        }
    }
    if (v == NC_ABORT) {
        (void)bt_gibbs_noise_chain_end(g);   // (waits for the launch to have ended)
        uint32_t code = 0;
        if (it == g->nc.it_begin && hipMemcpy(&code, g->nc.d_sync + nh * 8 + 256, 4, hipMemcpyDeviceToHost) == hipSuccess && code == 2u) {
            // ROLL CALL FAILED: the launch's workgroups did not all start — this GPU is not this process's alone — and none of them ran a sweep.  This and the
            // chain's remaining iterations run as launches per iteration; later chains on this device do not try again.
            if (g->ctx->device >= 0 && g->ctx->device < 64) g_no_resident_chain[g->ctx->device].store(true);
            if (getenv("BT_GIBBS_DEBUG") || getenv("BT_STAGE_TIMES"))
                fprintf(stderr, "bt_gibbs_noise_chain_step: the chain's %u workgroups were not resident together within %.1f s (is the GPU shared?): launches per iteration from here on\n",
                        g->nc.num_wgs, std::min(nc_timeout_seconds(), nc_rollcall_seconds()));
            g->nc.active = true;
            g->nc.fallback = true;
            g->nc.next = it;
            return bt_gibbs_noise_chain_step(g, h_noise, h_hist);
        }
        return fail("bt_gibbs_noise_chain_step: the resident launch gave up waiting (workgroups not resident together, or the host too slow: BT_NOISE_CHAIN_TIMEOUT_S)" + nc_state_text(g, it));
    }
    std::memcpy(h_hist, mail.hist, nh * 8);
    g->nc.next = it + 1;
    return BT_OK;
}

int bt_gibbs_noise_chain_end(bt_gibbs *g) {
    if (!g) return fail("bt_gibbs_noise_chain_end: null handle");
    if (!g->nc.active) return BT_OK;
    if (g->nc.fallback) {   // (nothing resident)
        g->nc.active = g->nc.fallback = false;
        return BT_OK;
    }
    (void)hipSetDevice(g->ctx->device);
    const NcMail mail = nc_mail(g);
    const bool complete = g->nc.next >= g->nc.n;
    if (!complete) nc_store(mail.table_seq, NC_ABORT);   // the launch ends at its next exchange
    g->nc.active = false;
    g->nc.launched = false;
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    if (complete && nc_load(mail.hist_seq) == NC_ABORT) return fail("bt_gibbs_noise_chain_end: the resident launch was aborted");
    if (g->nc.d_busy && g->nc.ctl.busy) {   // BT_NOISE_CHAIN_PROF: which tiles an iteration waits for
        std::vector<unsigned long long> busy(g->ntiles);
        BT_HIP(hipMemcpy(busy.data(), g->nc.d_busy, busy.size() * 8, hipMemcpyDeviceToHost));
        std::vector<uint32_t> order(g->ntiles);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return busy[a] > busy[b]; });
        const double per_it = 1e-2 / std::max(1u, g->nc.next);   // ticks of 10 ns -> microseconds per iteration
        fprintf(stderr, "bt_gibbs_noise_chain_end: busy microseconds per iteration and tile (of %u tiles): median %.1f;", g->ntiles, busy[order[g->ntiles / 2]] * per_it);
        for (uint32_t q = 0; q < std::min<uint32_t>(8, g->ntiles); ++q) {
            const TileDesc &d = g->tiles[order[q]];
            fprintf(stderr, " [tile %u: %.1f us, %u lanes, H %u, K %u, NU %u, simple %u, lds %u%s]", order[q], busy[order[q]] * per_it, d.num_lanes, d.Hm, d.Km, d.NUm, d.simple, tile_lds_bytes(d),
                    tile_lds_bytes(d) > g->nc.ctl.lds_cap ? " (in HBM)" : "");
        }
        double simple_sum = 0, other_sum = 0;
        uint32_t ns = 0, no = 0;
        for (uint32_t t = 0; t < g->ntiles; ++t) (g->tiles[t].simple ? (simple_sum += busy[t], ++ns) : (other_sum += busy[t], ++no));
        fprintf(stderr, "; mean two-haplotype tiles %.1f us (%u), others %.1f us (%u)\n", ns ? simple_sum * per_it / ns : 0.0, ns, no ? other_sum * per_it / no : 0.0, no);
    }
    return BT_OK;
}

struct bt_noise_model {
    bt_ctx *ctx = nullptr;
    uint32_t S = 0;
    NoiseDev *d_state = nullptr;
    float *d_prior = nullptr;
    double *d_lgamma = nullptr, *d_rates_cur = nullptr, *d_lut_spare = nullptr;
    unsigned long long *d_hist = nullptr;
};

int bt_noise_model_create(bt_ctx *ctx, uint32_t S, const float *h_prior, bt_noise_model **out) {
    if (!ctx || !h_prior || !out) return fail("bt_noise_model_create: null argument");
    if (S < 1 || S > 30) return fail("bt_noise_model_create: number of samples must be in 1..30");
    BT_HIP(hipSetDevice(ctx->device));
    bt_noise_model *m = new bt_noise_model();
    m->ctx = ctx;
    m->S = S;
    std::vector<double> lg(256);
    for (uint32_t c = 0; c < 256; ++c) lg[c] = std::lgamma((double)c + 1.0);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->d_state), sizeof(NoiseDev));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_prior), 2 * S * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_lgamma), 256 * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_rates_cur), S * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_lut_spare), (size_t)S * 256 * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_hist), (size_t)S * 256 * 8);
    if (e == hipSuccess) e = hipMemcpy(m->d_prior, h_prior, 2 * S * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_lgamma, lg.data(), 256 * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(m->d_state, 0, sizeof(NoiseDev));
    if (e != hipSuccess) {
        bt_noise_model_destroy(m);
        return fail(std::string("bt_noise_model_create: ") + hipGetErrorString(e));
    }
    *out = m;
    return BT_OK;
}

int bt_noise_model_destroy(bt_noise_model *m) {
    if (!m) return BT_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (void *p : {(void *)m->d_state, (void *)m->d_prior, (void *)m->d_lgamma, (void *)m->d_rates_cur, (void *)m->d_lut_spare, (void *)m->d_hist})
        if (p) (void)hipFree(p);
    delete m;
    return BT_OK;
}

int bt_noise_model_set_rng(bt_noise_model *m, const bt_noise_rng *h) {
    if (!m || !h) return fail("bt_noise_model_set_rng: null argument");
    if (h->mt_pos > MT_N) return fail("bt_noise_model_set_rng: generator position outside 0..624");
    static_assert(sizeof(bt_noise_rng) == sizeof(NoiseDev), "bt_noise_rng and its device copy differ");
    BT_HIP(hipSetDevice(m->ctx->device));
    BT_HIP(hipMemcpyAsync(m->d_state, h, sizeof(NoiseDev), hipMemcpyHostToDevice, m->ctx->stream));
    BT_HIP(hipStreamSynchronize(m->ctx->stream));
    return BT_OK;
}

int bt_noise_model_get_rng(bt_noise_model *m, bt_noise_rng *h) {
    if (!m || !h) return fail("bt_noise_model_get_rng: null argument");
    BT_HIP(hipSetDevice(m->ctx->device));
    BT_HIP(hipMemcpyAsync(h, m->d_state, sizeof(NoiseDev), hipMemcpyDeviceToHost, m->ctx->stream));
    BT_HIP(hipStreamSynchronize(m->ctx->stream));
    return BT_OK;
}

int bt_gibbs_noise_chain(bt_gibbs *g, bt_noise_model *m, uint32_t num_iterations, uint32_t first_collect, int (*reduce)(void *user, uint64_t *d_hist, uint64_t n), void *user,
                         double *h_rates) {
    if (!m || !h_rates) return fail("bt_gibbs_noise_chain: null argument");
    if (g && (g->S != m->S || g->ctx != m->ctx)) return fail("bt_gibbs_noise_chain: sampler and noise model differ in samples or context");
    if (num_iterations == 0) return BT_OK;
    BT_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const size_t nh = (size_t)m->S * 256;
    double *d_rates = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_rates), (size_t)num_iterations * m->S * 8));
    int rc = BT_OK;
    hipError_t e = hipSuccess;
    for (uint32_t it = 0; it < num_iterations && rc == BT_OK && e == hipSuccess; ++it) {
        if (g) rc = launch(g, OP_SWEEP, 1, it >= first_collect ? 1u : 0u, nullptr);
        if (rc != BT_OK) break;
        e = hipMemsetAsync(m->d_hist, 0, nh * 8, st);
        if (e != hipSuccess) break;
        if (g) rc = launch(g, OP_NOISE, 1, 0, m->d_hist);
        if (rc != BT_OK) break;
        if (reduce && reduce(user, reinterpret_cast<uint64_t *>(m->d_hist), nh) != 0) {
            rc = fail("bt_gibbs_noise_chain: the reduction of the noise counts failed");
            break;
        }
        hipLaunchKernelGGL(noise_update_kernel, dim3(1), dim3(256), 0, st, m->d_state, m->d_prior, m->S, m->d_hist, m->d_lgamma, d_rates + (size_t)it * m->S, m->d_rates_cur,
                           g ? g->d_lut_n : m->d_lut_spare);
        e = hipGetLastError();
    }
    if (rc == BT_OK && e == hipSuccess) e = hipMemcpyAsync(h_rates, d_rates, (size_t)num_iterations * m->S * 8, hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_rates);
    if (rc != BT_OK) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(std::string("bt_gibbs_noise_chain: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return BT_OK;
}

int bt_gibbs_reset_groups(bt_gibbs *g) {
    if (!g) return fail("bt_gibbs_reset_groups: null handle");
    return launch(g, OP_RESET, 0, 0, nullptr);
}

int bt_gibbs_posterior_summary(bt_gibbs *g, uint32_t *d_out) {
    if (!g || !d_out) return fail("bt_gibbs_posterior_summary: null argument");
    if (chain_in_flight(g)) return fail("bt_gibbs_posterior_summary: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    BT_HIP(hipSetDevice(g->ctx->device));
    hipLaunchKernelGGL(summary_kernel, dim3((g->C + 255) / 256), dim3(256), 0, g->ctx->stream, g->d_tiles, g->d_pool, g->d_loc, g->C, g->S, d_out);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_gibbs_device_bytes(bt_gibbs *g, uint64_t *bytes) {
    if (!g || !bytes) return fail("bt_gibbs_device_bytes: null argument");
    *bytes = g->device_bytes;
    return BT_OK;
}

}  // extern "C"

// copy one array of one tile to the host (all vertices, all lanes)
template <typename T>
static int fetch_array(bt_gibbs *g, uint32_t ti, int arr, uint64_t elems_per_lane, std::vector<T> &out) {
    out.resize(elems_per_lane << g->tiles[ti].wsh);
    BT_HIP(hipMemcpy(out.data(), g->d_pool + g->tiles[ti].base + g->tiles[ti].off[arr], out.size() * sizeof(T), hipMemcpyDeviceToHost));
    return BT_OK;
}

extern "C" {

// entries per cluster (device count -> host prefix sums); fails when a table overflowed
static int result_offsets(bt_gibbs *g, std::vector<uint64_t> &dip_off, std::vector<uint64_t> &cell_off) {
    BT_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    const uint32_t C = g->C;
    uint32_t *d_n = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_n), ((size_t)C + 1) * 4));
    struct Free {
        void *p;
        ~Free() { (void)hipFree(p); }
    } fr{d_n};
    BT_HIP(hipMemsetAsync(d_n + C, 0, 4, st));
    hipLaunchKernelGGL(result_count_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const TileDesc *)g->d_tiles, g->d_pool, (const ClusterLoc *)g->d_loc, C, d_n, d_n + C);
    BT_CHECK_LAUNCH();
    std::vector<uint32_t> n((size_t)C + 1);
    BT_HIP(hipMemcpyAsync(n.data(), d_n, ((size_t)C + 1) * 4, hipMemcpyDeviceToHost, st));
    BT_HIP(hipStreamSynchronize(st));
    if (n[C]) return fail("bt_gibbs: diplotype frequency table overflowed");
    dip_off.resize((size_t)C + 1);
    cell_off.resize((size_t)C + 1);
    uint64_t e = 0, cells = 0;
    for (uint32_t c = 0; c < C; ++c) {
        dip_off[c] = e;
        cell_off[c] = cells;
        e += n[c];
        cells += (uint64_t)g->S * g->h_A[c];
    }
    dip_off[C] = e;
    cell_off[C] = cells;
    return BT_OK;
}

int bt_gibbs_result_sizes(bt_gibbs *g, uint64_t *num_diplotype_entries, uint64_t *num_allele_cells) {
    if (!g) return fail("bt_gibbs_result_sizes: null handle");
    if (chain_in_flight(g)) return fail("bt_gibbs_result_sizes: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    std::vector<uint64_t> dip_off, cell_off;
    const int rc = result_offsets(g, dip_off, cell_off);
    if (rc != BT_OK) return rc;
    if (num_diplotype_entries) *num_diplotype_entries = dip_off[g->C];
    if (num_allele_cells) *num_allele_cells = cell_off[g->C];
    return BT_OK;
}

// The flat layout is built by result_pack_kernel (one wavefront per cluster: entries ranked by (h1, h2), statistics gathered from the tile's
// interleaved rows) and comes back in one copy per output array.
int bt_gibbs_result_fetch(bt_gibbs *g, uint64_t *h_dip_off, uint16_t *h_dip_h1, uint16_t *h_dip_h2, uint32_t *h_dip_freq, uint64_t *h_cell_off,
                          double *h_stats) {
    if (!g || !h_dip_off || !h_cell_off) return fail("bt_gibbs_result_fetch: null argument");
    if (chain_in_flight(g)) return fail("bt_gibbs_result_fetch: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    std::vector<uint64_t> dip_off, cell_off;
    {
        const int rc = result_offsets(g, dip_off, cell_off);
        if (rc != BT_OK) return rc;
    }
    const uint32_t C = g->C, S = g->S;
    hipStream_t st = g->ctx->stream;
    const uint64_t nd = dip_off[C], nc = cell_off[C];
    std::memcpy(h_dip_off, dip_off.data(), ((size_t)C + 1) * 8);
    std::memcpy(h_cell_off, cell_off.data(), ((size_t)C + 1) * 8);
    if (C == 0) return BT_OK;
    std::vector<void *> tmp;
    struct FreeAll {
        std::vector<void *> &v;
        ~FreeAll() {
            for (void *p : v) (void)hipFree(p);
        }
    } fr{tmp};
    auto dev = [&](size_t bytes, void **out) -> hipError_t {
        hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) tmp.push_back(*out);
        return e;
    };
    uint64_t *d_dip_off = nullptr, *d_cell_off = nullptr;
    uint32_t *d_alleles = nullptr, *d_freq = nullptr;
    uint16_t *d_h1 = nullptr, *d_h2 = nullptr;
    double *d_stats = nullptr;
    BT_HIP(dev(((size_t)C + 1) * 8, reinterpret_cast<void **>(&d_dip_off)));
    BT_HIP(dev(((size_t)C + 1) * 8, reinterpret_cast<void **>(&d_cell_off)));
    BT_HIP(dev((size_t)C * 4, reinterpret_cast<void **>(&d_alleles)));
    if (h_dip_h1) BT_HIP(dev(nd * 2, reinterpret_cast<void **>(&d_h1)));
    if (h_dip_h2) BT_HIP(dev(nd * 2, reinterpret_cast<void **>(&d_h2)));
    if (h_dip_freq) BT_HIP(dev(nd * S * 4, reinterpret_cast<void **>(&d_freq)));
    if (h_stats) BT_HIP(dev(nc * 12 * 8, reinterpret_cast<void **>(&d_stats)));
    BT_HIP(hipMemcpyAsync(d_dip_off, dip_off.data(), ((size_t)C + 1) * 8, hipMemcpyHostToDevice, st));
    BT_HIP(hipMemcpyAsync(d_cell_off, cell_off.data(), ((size_t)C + 1) * 8, hipMemcpyHostToDevice, st));
    BT_HIP(hipMemcpyAsync(d_alleles, g->h_A.data(), (size_t)C * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(result_pack_kernel, dim3(C), dim3(64), 0, st, (const TileDesc *)g->d_tiles, g->d_pool, (const ClusterLoc *)g->d_loc, S, (const uint64_t *)d_dip_off,
                       (const uint64_t *)d_cell_off, (const uint32_t *)d_alleles, d_h1, d_h2, d_freq, d_stats, (uint32_t *)nullptr);
    BT_CHECK_LAUNCH();
    if (h_dip_h1 && nd) BT_HIP(hipMemcpyAsync(h_dip_h1, d_h1, nd * 2, hipMemcpyDeviceToHost, st));
    if (h_dip_h2 && nd) BT_HIP(hipMemcpyAsync(h_dip_h2, d_h2, nd * 2, hipMemcpyDeviceToHost, st));
    if (h_dip_freq && nd) BT_HIP(hipMemcpyAsync(h_dip_freq, d_freq, nd * S * 4, hipMemcpyDeviceToHost, st));
    if (h_stats && nc) BT_HIP(hipMemcpyAsync(h_stats, d_stats, nc * 12 * 8, hipMemcpyDeviceToHost, st));
    BT_HIP(hipStreamSynchronize(st));
    return BT_OK;
}

// The same results as ONE word string in device memory — what a rank hands to bt_comm_gather_summaries (InferenceEngine.cpp:335-382: the
// reference's threads push their genotypes into one queue; with one process per GPU the queue is the gather to rank 0) — without the host
// round trip of bt_gibbs_result_fetch:  [C, entries, cells, S]  [entries, cells] per cluster  h1 | h2 << 16 per entry  counts [entry][S]
// (one pad word if the count so far is odd)  statistics [cell][12] doubles.  The buffer belongs to the sampler (valid until the next call
// or bt_gibbs_destroy) and is complete when the call returns.
int bt_gibbs_result_words(bt_gibbs *g, const uint32_t **d_words, uint64_t *num_words) {
    if (!g || !d_words || !num_words) return fail("bt_gibbs_result_words: null argument");
    if (chain_in_flight(g)) return fail("bt_gibbs_result_words: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    std::vector<uint64_t> dip_off, cell_off;
    {
        const int rc = result_offsets(g, dip_off, cell_off);
        if (rc != BT_OK) return rc;
    }
    const uint32_t C = g->C, S = g->S;
    hipStream_t st = g->ctx->stream;
    const uint64_t nd = dip_off[C], nc = cell_off[C];
    if (nd >> 32 || nc >> 32) return fail("bt_gibbs_result_words: more than 2^32 entries in one launch");
    const uint64_t at_sizes = 4, at_keys = at_sizes + 2ull * C, at_freq = at_keys + nd, at_stats = (at_freq + nd * S + 1) & ~1ull, total = at_stats + nc * 24;
    if (g->wire_cap < total) {
        if (g->d_wire) BT_HIP(hipFree(g->d_wire));
        g->d_wire = nullptr;
        g->wire_cap = 0;
        BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_wire), total * 4));
        g->wire_cap = total;
    }
    uint32_t *w = g->d_wire;
    const uint32_t head[4] = {C, (uint32_t)nd, (uint32_t)nc, S};
    BT_HIP(hipMemcpyAsync(w, head, sizeof head, hipMemcpyHostToDevice, st));
    if (at_stats != at_freq + nd * S) BT_HIP(hipMemsetAsync(w + at_stats - 1, 0, 4, st));
    void *d_tmp = nullptr;
    struct Free {
        void *&p;
        ~Free() {
            if (p) (void)hipFree(p);
        }
    } fr{d_tmp};
    if (C) {
        const size_t off_bytes = ((size_t)C + 1) * 8;
        BT_HIP(hipMalloc(&d_tmp, 2 * off_bytes + (size_t)C * 4));
        uint64_t *d_dip_off = (uint64_t *)d_tmp, *d_cell_off = d_dip_off + C + 1;
        uint32_t *d_alleles = (uint32_t *)(d_cell_off + C + 1);
        BT_HIP(hipMemcpyAsync(d_dip_off, dip_off.data(), off_bytes, hipMemcpyHostToDevice, st));
        BT_HIP(hipMemcpyAsync(d_cell_off, cell_off.data(), off_bytes, hipMemcpyHostToDevice, st));
        BT_HIP(hipMemcpyAsync(d_alleles, g->h_A.data(), (size_t)C * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(wire_sizes_kernel, dim3((C + 63) / 64), dim3(64), 0, st, (const uint64_t *)d_dip_off, (const uint64_t *)d_cell_off, C, w + at_sizes);
        BT_CHECK_LAUNCH();
        hipLaunchKernelGGL(result_pack_kernel, dim3(C), dim3(64), 0, st, (const TileDesc *)g->d_tiles, g->d_pool, (const ClusterLoc *)g->d_loc, S, (const uint64_t *)d_dip_off,
                           (const uint64_t *)d_cell_off, (const uint32_t *)d_alleles, (uint16_t *)nullptr, (uint16_t *)nullptr, w + at_freq, reinterpret_cast<double *>(w + at_stats),
                           w + at_keys);
        BT_CHECK_LAUNCH();
    }
    BT_HIP(hipStreamSynchronize(st));
    *d_words = w;
    *num_words = total;
    return BT_OK;
}

int bt_gibbs_trace_enable(bt_gibbs *g, uint32_t max_sweeps) {
    if (!g) return fail("bt_gibbs_trace_enable: null handle");
    if (chain_in_flight(g)) return fail("bt_gibbs_trace_enable: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
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
    uint64_t words = 0;
    for (uint32_t ti = 0; ti < g->ntiles; ++ti) {
        g->tiles[ti].trace_base = words;
        words += (uint64_t)max_sweeps * g->tiles[ti].nvm * g->S * LANES;
    }
    g->trace_words = words;
    BT_HIP(hipMemcpy(g->d_tiles, g->tiles.data(), (size_t)g->ntiles * sizeof(TileDesc), hipMemcpyHostToDevice));
    if (max_sweeps) {
        BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_trace), std::max<uint64_t>(words, 1) * 4));
        BT_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_trace_counter), (size_t)g->ntiles * LANES * 4));
        BT_HIP(hipMemset(g->d_trace, 0xFF, std::max<uint64_t>(words, 1) * 4));
        BT_HIP(hipMemset(g->d_trace_counter, 0, (size_t)g->ntiles * LANES * 4));
    }
    return BT_OK;
}

// h_trace: groups in batch order; group gi contributes [max_sweeps][nvert(gi)][S] words
int bt_gibbs_trace_fetch(bt_gibbs *g, uint32_t *h_trace, uint64_t max_words, uint64_t *num_sweeps_recorded) {
    if (!g || !h_trace) return fail("bt_gibbs_trace_fetch: null argument");
    if (chain_in_flight(g)) return fail("bt_gibbs_trace_fetch: a resident noise chain is in progress (bt_gibbs_noise_chain_end): the call would wait behind its launch");
    if (!g->d_trace) return fail("bt_gibbs_trace_fetch: tracing is off");
    uint64_t need = 0;
    for (uint32_t gi = 0; gi < g->G; ++gi) need += (uint64_t)g->trace_sweeps * g->group_nvert[gi] * g->S;
    if (max_words < need) return fail("bt_gibbs_trace_fetch: buffer too small");
    BT_HIP(hipSetDevice(g->ctx->device));
    BT_HIP(hipStreamSynchronize(g->ctx->stream));
    std::vector<uint32_t> raw(g->trace_words);
    BT_HIP(hipMemcpy(raw.data(), g->d_trace, g->trace_words * 4, hipMemcpyDeviceToHost));
    uint64_t w = 0;
    for (uint32_t gi = 0; gi < g->G; ++gi) {
        const TileDesc &d = g->tiles[g->group_tile[gi]];
        const uint32_t l = g->group_lane[gi], nv = g->group_nvert[gi];
        for (uint32_t sw = 0; sw < g->trace_sweeps; ++sw)
            for (uint32_t v = 0; v < nv; ++v)
                for (uint32_t s = 0; s < g->S; ++s) h_trace[w++] = raw[d.trace_base + (((uint64_t)sw * d.nvm + v) * g->S + s) * LANES + l];
    }
    if (num_sweeps_recorded) {
        uint32_t n0 = 0;
        BT_HIP(hipMemcpy(&n0, g->d_trace_counter + (size_t)g->group_tile[0] * LANES + g->group_lane[0], 4, hipMemcpyDeviceToHost));
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
    USet s{sptr1(hdr.data()), sptr1(bkt.data()), sptr1(next.data()), (uint32_t)bkt.size()};
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

#ifdef BT_PROF
int bt_diag_prof(unsigned long long *h_out16, int reset) {
    BT_HIP(hipDeviceSynchronize());
    BT_HIP(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_bt_prof), 32 * 8));   // 32 slots
    if (reset) {
        unsigned long long z[32] = {0};
        BT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bt_prof), z, 32 * 8));
    }
    unsigned long long more[32];   // + the simple kernel's translation unit
    BT_HIP(bt::simple_prof_read(more, reset));
    for (int i = 0; i < 32; ++i) h_out16[i] += more[i];
    BT_HIP(bt::hot_prof_read(more, reset));   // ... and gibbs_hot_kernel's
    for (int i = 0; i < 32; ++i) h_out16[i] += more[i];
    BT_HIP(bt::single_prof_read(more, reset));   // ... and gibbs_single_kernel's
    for (int i = 0; i < 32; ++i) h_out16[i] += more[i];
    return BT_OK;
}
#endif

int bt_diag_rng(uint32_t seed, int kind, const double *a, const double *b, uint64_t n, double *h_out) {
    if (!h_out) return fail("bt_diag_rng: null argument");
    std::vector<uint32_t> stv(MT_WORDS);
    mt_seed(stv.data(), seed);
    Mt st = mt_open(stv.data());
    double saved = 0;
    uint32_t avail = 0;
    NormalState nd{&saved, &avail};
    switch (kind) {
        case 0:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = (double)mt_next(st);
            break;
        case 1:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = rng_canonical(st);
            break;
        case 2:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = rng_gamma(st, nd, a[i], b[i]);
            break;
        case 3:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = (double)rng_uniform_int(st, (uint32_t)a[i] + 1u);
            break;
        case 4:
            for (uint64_t i = 0; i < n; ++i) h_out[i] = rng_bernoulli(st, (double)(float)a[0]) ? 1.0 : 0.0;
            break;
        case 5: {
            std::vector<uint32_t> v((size_t)a[0]);
            for (size_t i = 0; i < v.size(); ++i) v[i] = (uint32_t)i;
            rng_shuffle_u32(st, sptr1(v.data()), (uint32_t)v.size());
            for (size_t i = 0; i < v.size(); ++i) h_out[i] = v[i];
            break;
        }
        default:
            return fail("bt_diag_rng: unknown kind");
    }
    return BT_OK;
}

}  // extern "C"
