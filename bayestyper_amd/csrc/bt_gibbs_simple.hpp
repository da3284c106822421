// Sweeps of TWO-HAPLOTYPE single-cluster groups (a biallelic SNV / indel: reference and alternative haplotype) — nine out of ten
// variant-cluster groups of a genome.  Same state, same draw stream and the same decisions as the general sampler in bt_gibbs_tile.hpp
// (which still constructs, resets and drains these clusters); what differs is HOW a sweep is executed (round 4):
//
//  * the cluster's sampler state lives in REGISTERS for a whole chain — the two frequencies, the observation counts, the two emulated
//    unordered_set<uint> as two-entry lists, the polar method's saved variate — and a sample's state in ONE packed LDS word (diplotype, the
//    pending run of collected sweeps, its length, the log fill, ploidy, flags), read and written once per sample and sweep;
//  * the diplotype draw of a sample (at most three candidates) is taken in the LINEAR domain from weights that are fixed for a chain: the
//    unique-k-mer sums of the candidates enter as exp(sum - max), computed once per chain start in double precision and kept as two floats per
//    sample; a sweep multiplies them with the current frequencies (f0 f0, 2 f0 f1, f1 f1) and compares U * total with the running sums.  The
//    decision is VERIFIED: whenever U * total lies within 1e-5 * total of a boundary (the weights' own error is below 2e-7 * total) — about
//    once in 10^4 draws — the reference's chain of logAddition calls over the double-precision log-probabilities decides (simple_exact_code),
//    so every sampled diplotype is the reference's.  log(frequency) is therefore only needed on that path and is no longer kept up to date;
//  * the simplex-size draw of the sparse frequency distribution needs one number per chain (the cached probability vector's first entry: the
//    observation total of a cluster is the same in every sweep);
//  * gamma draws read 1 / sqrt(9 (alpha - 1/3)) from a table (alpha = an observation count + 1) and screen Marsaglia-Tsang's second test in
//    single precision (bt_rng_device.hpp);
//  * generator refills in aligned chunks of four words, four memory requests per lane and chunk (MtRingT::chunk).
// LDS per cluster: the two draw-ahead rings + 12 bytes per sample (round 3: 430 bytes at three samples), so that a CU holds twelve and more
// of these wavefronts instead of six.
//
// Reference behaviour (as in bt_gibbs_tile.hpp):
//   VariantClusterGenotyper::sampleDiplotypes / sampleDiplotype / calcDiplotypeLogProb   VariantClusterGenotyper.cpp:597-755
//   HaplotypeFrequencyDistribution::incrementCount / SparseFrequencyDistribution          HaplotypeFrequencyDistribution.cpp:113-138,
//                                                                                         FrequencyDistribution.cpp:75-93,198-303
//   VariantClusterHaplotypes::updateAlleleKmerStats (logged as runs, applied by drain_collected)
#pragma once
#include "bt_gibbs_tile.hpp"

namespace bt {

// lane-interleaved array in this wavefront's LDS block
// (tiles of two-haplotype clusters are always 64 groups wide — TileDesc::simple requires it — so the row stride is a constant and
// element offsets fold into the ds_* instructions)
template <typename T>
struct LdsArr {
    T BT_LAS *p;
    __device__ inline T BT_LAS &operator[](uint32_t i) const { return p[i * LANES]; }
    __device__ inline LdsArr<T> operator+(uint32_t i) const { return LdsArr<T>{p + i * LANES}; }
};
template <typename T>
__device__ inline LdsArr<T> lds_arr(const Tile &t, int arr) {
    return LdsArr<T>{(T BT_LAS *)(bt_lds_raw + t.d->hoff[arr]) + t.lane};
}

// std::unordered_set<uint> over the universe {0, 1}: both elements always sit in buckets of their own (13 buckets from the first
// insert on), so a new element becomes the head of the iteration order and an erased one just leaves it — a list of at most two
struct Set2 {
    uint32_t n, e0, e1;   // iteration order: e0, then e1
    __device__ inline void clear() { n = 0; }
    __device__ inline void insert(uint32_t e) {   // e is not in the set
        e1 = e0;
        e0 = e;
        ++n;
    }
    __device__ inline void erase(uint32_t e) {    // e is in the set
        if (n == 2 && e0 == e) e0 = e1;
        --n;
    }
    __device__ inline uint32_t at(uint32_t i) const { return i == 0 ? e0 : e1; }
};
template <class HS>
__device__ inline Set2 set2_load(HS s) {
    Set2 r{0, 0, 0};
    uint32_t e = uset_begin(s);
    if (e != US_NONE) {
        r.e0 = e;
        r.n = 1;
        e = s.next[e];
        if (e != US_NONE) {
            r.e1 = e;
            r.n = 2;
        }
    }
    return r;
}
template <class HS>
__device__ inline void set2_store(HS s, const Set2 &v) {   // rebuild the general representation: insert in reverse iteration order
    uset_clear(s);
    if (v.n == 2) uset_insert(s, v.e1);
    if (v.n >= 1) uset_insert(s, v.e0);
}

__device__ inline bool tile_is_simple(const TileDesc BT_CAS &d) { return d.simple != 0; }

// ---- a sample's packed state word -------------------------------------------------------------------------------------------------
// diplotype codes of a two-haplotype cluster: 0 (0,0)  1 (0,1)  2 (1,1)  3 (0,-)  4 (1,-)  5 (-,-)
constexpr uint32_t SD_NONE = 5;
__device__ inline uint32_t sd_code(uint16_t h1, uint16_t h2) { return h1 == NOHAP ? SD_NONE : (h2 == NOHAP ? 3u + h1 : (uint32_t)h1 + h2); }
__device__ inline uint32_t sd_h1(uint32_t code) { return code < 3u ? code >> 1 : (code < 5u ? code - 3u : (uint32_t)NOHAP); }
__device__ inline uint32_t sd_h2(uint32_t code) { return code < 3u ? (code + 1u) >> 1 : (uint32_t)NOHAP; }
__device__ inline uint32_t sd_key(uint32_t code) { return sd_h1(code) | (sd_h2(code) << 16); }
// bits 0-2 diplotype, 3-5 the pending run's diplotype, 6-7 which candidate's weight is the maximum (= 1), 8-15 length of the pending run,
// 16-23 runs in the sample's log, 24-25 ploidy, 26 k-mer-stats cache out of date (A_KSCUPD), 27 a run is pending (A_PENDVALID)
constexpr uint32_t SP_PDIP = 3, SP_WMAX = 6, SP_PEND = 8, SP_EVN = 16, SP_PLOIDY = 24, SP_UPD = 26, SP_PVALID = 27;
constexpr uint32_t SB_WORDS = 3;   // per sample: two weights (float bits), the packed word

// The reference's decision for one sample, as round 3's sweep took it for every draw: the chain of logAddition calls over the candidates' log-probabilities
// (log frequencies + the table of unique-k-mer sums), upper_bound of log(U) + total.  Returns the diplotype code.
__device__ static __noinline__ uint32_t simple_exact_code(Env env, uint32_t s, uint32_t pl, uint32_t nzmask, double f0, double f1, double u01) {
    const Vx c = make_vx(make_tile(env), 0);
    const uint32_t Dcm = c.d().Dcm;
    const Vx::UCPtr uc = c.ucache();
    uint32_t nzl[2] = {0, 0}, nnz = 0;
    if (nzmask & 1u) nzl[nnz++] = 0;
    if (nzmask & 2u) nzl[nnz++] = 1;
    const double lf0 = bt_log(nzl[0] ? f1 : f0), lf1 = nnz > 1 ? bt_log(f1) : 0.0;
    const uint32_t total = pl == 2 ? nnz * (nnz + 1) / 2 : (pl == 1 ? nnz : 0u);
    double lp[3];
    uint16_t ca[3], cb[3];
    if (pl == 2) {
        ca[0] = (uint16_t)nzl[0], cb[0] = (uint16_t)nzl[0];
        ca[1] = (uint16_t)nzl[0], cb[1] = (uint16_t)nzl[1];
        ca[2] = (uint16_t)nzl[1], cb[2] = (uint16_t)nzl[1];
        if (nnz == 1) ca[1] = ca[2] = ca[0], cb[1] = cb[2] = cb[0];
    } else {
        ca[0] = (uint16_t)nzl[0], cb[0] = NOHAP;
        ca[1] = (uint16_t)nzl[nnz > 1 ? 1 : 0], cb[1] = NOHAP;
        ca[2] = ca[1], cb[2] = NOHAP;
    }
    double uv[3];
#pragma unroll
    for (uint32_t q = 0; q < 3; ++q) {   // H = 2: dip_index(a, b) = 2a - a(a-1)/2 + (b - a), haploid 3 + a
        const uint32_t a = ca[q], b = cb[q];
        const uint32_t idx = cb[q] == NOHAP ? 3u + a : 2u * a - (a * (a - 1u)) / 2u + (b - a);
        uv[q] = q < total ? (double)uc[s * Dcm + idx] : 0.0;
    }
#pragma unroll
    for (uint32_t q = 0; q < 3; ++q) {
        const double la = ca[q] == nzl[0] ? lf0 : lf1, lb = cb[q] == nzl[0] ? lf0 : lf1;
        double v = 0;
        if (pl != 2) v += la;
        else if (ca[q] == cb[q]) v += 2 * la;
        else v += BT_LN2 + la + lb;
        lp[q] = v + uv[q];
    }
    if (total == 0) return SD_NONE;
    uint32_t pick = 0;
    double cum1 = 0, cum2 = 0;
    double run = lp[0];
    if (total > 1) {
        run = log_addition(lp[1], run);
        cum1 = run;
    }
    if (total > 2) {
        run = log_addition(lp[2], run);
        cum2 = run;
    }
    const double u = bt_log(u01) + run;
    // upper_bound(cum, u): first index with u < cum[i]; past the end -> last
    if (u < lp[0]) pick = 0;
    else if (total > 1 && u < cum1) pick = 1;
    else if (total > 2 && u < cum2) pick = 2;
    else pick = total - 1;
    return sd_code(ca[pick], cb[pick]);
}

// the log of a sample is full: apply it (rare: a sample that changed its diplotype EV_CAP times within a chain)
__device__ static __noinline__ void simple_apply_full_log(Env env, uint32_t s) {
    const Vx c = make_vx(make_tile(env), 0);
    const GParams BT_CAS &P = env_params(env);
    c.evn()[s] = (uint8_t)EV_CAP;
    apply_collected_log(c, P, s, c.sc()[SC_NSUB_U]);
}

// A sample's candidate weights from the table of unique-k-mer sums: exp(sum - max) of the two candidates that are not the maximum, in candidate
// order, as floats, and which candidate is the maximum (diploid: (0,0) (0,1) (1,1) at table entries 0..2; haploid: (0) (1) at entries 3, 4).
struct SimpleWeights {
    float ea, eb;
    uint32_t wmax;
};
template <class UC>
__device__ inline SimpleWeights simple_weights(const UC &uc, uint32_t Dcm, uint32_t s, uint32_t pl) {
    const uint32_t i0 = pl == 2 ? 0u : 3u;
    const double v0 = pl ? (double)uc[s * Dcm + i0] : 0.0, v1 = pl ? (double)uc[s * Dcm + i0 + 1u] : 0.0;
    const double ninf = -__builtin_huge_val();
    const double v2 = pl == 2 ? (double)uc[s * Dcm + 2u] : ninf;
    uint32_t wmax = 0;
    double m = v0;
    if (v1 > m) m = v1, wmax = 1;
    if (v2 > m) m = v2, wmax = 2;
    const double ea = bt_exp((wmax == 0 ? v1 : v0) - m), eb = bt_exp((wmax == 2 ? v1 : v2) - m);   // the two that are not the maximum, in candidate order
    return SimpleWeights{(float)ea, (float)eb, wmax};
}

// fill_unique_cache for a two-haplotype cluster: the five entries of a sample — (0,0) (0,1) (1,1) and the haploid (0) (1) — in ONE pass over the k-mer
// subset (the general fill walks the subset once per block of candidates that share their first haplotype: three passes here), four k-mers per step: the
// shared operands are read once, the twenty table lookups of a step are independent.  Every entry is the same sum in the same (subset) order as
// unique_log_prob's, so the values are bit-identical.
__device__ static __noinline__ void simple_fill_unique(Env env) {
    const Tile t = make_tile(env);
    const Vx c = make_vx(t, 0);
    const GParams BT_CAS &P = env_params(env);
    const TileDesc BT_CAS &d = *t.d;
    const uint32_t nsub = c.sc()[SC_NSUB_U], Hm = d.Hm, S = P.S;
    const Vx::RPtr<uint8_t> sm = c.subm();
    TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
    const Vx::UCPtr uc = c.ucache();
    // Round 6 (a resident noise chain rebuilds this table every iteration, and its time was the serial chain of S x nsub / 4 x 2 memory round trips): TWO samples per
    // pass over the subset — the multiplicity rows and intercluster multiplicities of a k-mer are read once for both — and the operands of the NEXT block of four
    // k-mers are requested before the current block's forty table lookups are waited for: about one round trip per block and pair of samples instead of four.
    struct Ops {   // one block of four k-mers, a byte per k-mer: the haplotypes' multiplicities, both intercluster multiplicities, the two samples' counts
        uint32_t m0, m1, i0, i1, ca, cb;   // (packed: the block in flight and the next one would otherwise hold 48 registers)
    };
    auto load_ops = [&](uint32_t i, uint32_t sa, uint32_t sb) {
        uint8_t b[6][4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t j = i + r < nsub ? i + r : (nsub ? nsub - 1u : 0u);   // (the tail block re-reads the last k-mer: its lanes are masked below)
            b[0][r] = sm[j * Hm];
            b[1][r] = sm[j * Hm + 1];
            b[2][r] = sic[2 * j];
            b[3][r] = sic[2 * j + 1];
            b[4][r] = scn[j * S + sa];
            b[5][r] = scn[j * S + sb];
        }
        uint32_t w[6];
#pragma unroll
        for (uint32_t q = 0; q < 6; ++q) w[q] = (uint32_t)b[q][0] | ((uint32_t)b[q][1] << 8) | ((uint32_t)b[q][2] << 16) | ((uint32_t)b[q][3] << 24);
        return Ops{w[0], w[1], w[2], w[3], w[4], w[5]};
    };
    auto byte_of = [](uint32_t w, uint32_t r) { return (uint8_t)(w >> (8u * r)); };
    for (uint32_t s0 = 0; s0 < S; s0 += 2) {
        const uint32_t sa = s0, sb = s0 + 1 < S ? s0 + 1 : s0;
        const bool two = s0 + 1 < S;
        const bool ga = P.gender[sa] != 0, gb = P.gender[sb] != 0;
        double acc[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
        Ops cur = load_ops(0, sa, sb);
        for (uint32_t i = 0; i < nsub; i += 4) {
            const Ops nxt = load_ops(i + 4 < nsub ? i + 4 : i, sa, sb);   // (requested before this block's lookups: one round trip, not two)
            // two k-mers x two samples x five candidates = twenty lookups in flight per half block (forty would need more registers than three wavefronts per SIMD leave)
#pragma unroll
            for (uint32_t half = 0; half < 2; ++half) {
                uint8_t m[2][2][5];
#pragma unroll
                for (uint32_t rr = 0; rr < 2; ++rr) {
                    const uint32_t r = 2 * half + rr;
                    const uint8_t m0 = byte_of(cur.m0, r), m1 = byte_of(cur.m1, r);
#pragma unroll
                    for (uint32_t w = 0; w < 2; ++w) {
                        const uint8_t icn = (w ? gb : ga) ? byte_of(cur.i1, r) : byte_of(cur.i0, r);
                        m[w][rr][0] = (uint8_t)((uint8_t)(m0 + m0) + icn);
                        m[w][rr][1] = (uint8_t)((uint8_t)(m0 + m1) + icn);
                        m[w][rr][2] = (uint8_t)((uint8_t)(m1 + m1) + icn);
                        m[w][rr][3] = (uint8_t)(m0 + icn);
                        m[w][rr][4] = (uint8_t)(m1 + icn);
                    }
                }
                double lp[2][2][5];
#pragma unroll
                for (uint32_t rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (uint32_t q = 0; q < 5; ++q) {
                        lp[0][rr][q] = count_log_prob(P, sa, m[0][rr][q], byte_of(cur.ca, 2 * half + rr));
                        lp[1][rr][q] = count_log_prob(P, sb, m[1][rr][q], byte_of(cur.cb, 2 * half + rr));
                    }
                // every entry's sum in subset order, as unique_log_prob's (the k-mers past the end of a tail block add nothing)
#pragma unroll
                for (uint32_t rr = 0; rr < 2; ++rr)
                    if (i + 2 * half + rr < nsub) {
#pragma unroll
                        for (uint32_t q = 0; q < 5; ++q) {
                            acc[0][q] += lp[0][rr][q];
                            acc[1][q] += lp[1][rr][q];
                        }
                    }
                bt_sched_fence();   // (the second half's lookups are not started before the first half's registers are free)
            }
            cur = nxt;
        }
#pragma unroll
        for (uint32_t q = 0; q < 5; ++q) {
            uc[sa * d.Dcm + q] = acc[0][q];
            if (two) uc[sb * d.Dcm + q] = acc[1][q];
        }
    }
}

// Chain entry: the per-sample LDS words from the general arrays, the candidates' weights from the table of unique-k-mer sums (rebuilt when
// the chain start / clearCache marked it), the sparse distribution's simplex-size probability.  Returns P(simplex size = |plus|) for |plus| = 1.
__device__ static __noinline__ double simple_enter(Env env, uint32_t blk_off) {
    const Tile t = make_tile(env);
    const Vx c = make_vx(t, 0);
    const GParams BT_CAS &P = env_params(env);
    const TileDesc BT_CAS &d = *t.d;
    const uint32_t S = P.S, Dcm = d.Dcm;
    SPtrF<uint32_t, LANES> sc = c.sc();
    if (sc[SC_UC_DIRTY]) {   // chain start / clearCache: the dense table of unique-k-mer sums is rebuilt as a whole
        simple_fill_unique(env);
        sc[SC_UC_DIRTY] = 0;
    }
    LdsArr<uint32_t> blk{(uint32_t BT_LAS *)(bt_lds_raw + blk_off) + t.lane};
    const Vx::UCPtr uc = c.ucache();
    TPtr<uint8_t> gp = t.arr<uint8_t>(A_PLOIDY);
    SPtrF<uint16_t, LANES> dip = c.dip(), pdip = c.pend_dip();
    SPtrF<uint32_t, LANES> pend = c.pend();
    SPtrF<uint8_t, LANES> pvalid = c.pend_valid(), upd = c.ksc_upd(), evn = c.evn(), npl = c.nest_ploidy(), nn = c.nest_n();
    uint32_t n_obs = 0;
    for (uint32_t s = 0; s < S; ++s) {
        const uint32_t pl = gp[s];
        // the group's ploidy per sample is the cluster's (a root cluster without nesting: VariantClusterGroup.cpp:225-231)
        npl[s] = (uint8_t)pl;
        nn[s] = 0;
        n_obs += pl == 2 ? 2u : (pl == 1 ? 1u : 0u);
        const SimpleWeights w = simple_weights(uc, Dcm, s, pl);
        const uint32_t wmax = w.wmax;
        blk[SB_WORDS * s] = __float_as_uint(w.ea);
        blk[SB_WORDS * s + 1] = __float_as_uint(w.eb);
        uint32_t pr = pend[s], pv = pvalid[s];
        uint32_t pcode = sd_code(pdip[2 * s], pdip[2 * s + 1]);
        if (pv && pr > 255u) {   // (not produced by this kernel, which drains at the end of every launch that collects)
            log_collected_run(c, P, s, sd_key(pcode), pr, sc[SC_NSUB_U]);
            pr = 0;
            pv = 0;
        }
        blk[SB_WORDS * s + 2] = sd_code(dip[2 * s], dip[2 * s + 1]) | (pcode << SP_PDIP) | (wmax << SP_WMAX) | (pr << SP_PEND) | ((uint32_t)evn[s] << SP_EVN) | (pl << SP_PLOIDY) |
                                ((upd[s] ? 1u : 0u) << SP_UPD) | ((pv ? 1u : 0u) << SP_PVALID);
    }
    double p1 = 1.0;
    if (sc[SC_IS_SPARSE] && n_obs > 0) {
        // cached simplex-size distribution (FrequencyDistribution.cpp:143-196,211-229) for |plus| = 1 (for |plus| = 2 the size is 2)
        TPtr<double> vec = c.simplex();
        const bool cached = d.scache_n && n_obs <= 2 * d.S && 1u <= d.scache_p;
        const uint32_t ci = cached ? (n_obs - 1) * d.scache_p : 0u;
        if (cached) vec = c.scache() + ci * d.scache_len;
        uint32_t len = cached ? (uint32_t)c.sclen()[ci] : 0u;
        if (len == 0) {
            len = simplex_prob_vector(c, P, vec, n_obs, 1u);
            if (cached) c.sclen()[ci] = len;
        }
        p1 = vec[0];
    }
    return p1;
}

// chain exit: back to the general representation
__device__ static __noinline__ void simple_leave(Env env, uint32_t blk_off, double f0, double f1, uint32_t nzmask) {
    const Tile t = make_tile(env);
    const Vx c = make_vx(t, 0);
    const GParams BT_CAS &P = env_params(env);
    LdsArr<uint32_t> blk{(uint32_t BT_LAS *)(bt_lds_raw + blk_off) + t.lane};
    SPtrF<uint16_t, LANES> dip = c.dip(), pdip = c.pend_dip();
    SPtrF<uint32_t, LANES> pend = c.pend();
    SPtrF<uint8_t, LANES> pvalid = c.pend_valid(), upd = c.ksc_upd(), evn = c.evn();
    for (uint32_t s = 0; s < P.S; ++s) {
        const uint32_t pk = blk[SB_WORDS * s + 2];
        const uint32_t code = pk & 7u, pcode = (pk >> SP_PDIP) & 7u;
        dip[2 * s] = (uint16_t)sd_h1(code);
        dip[2 * s + 1] = (uint16_t)sd_h2(code);
        pdip[2 * s] = (uint16_t)sd_h1(pcode);
        pdip[2 * s + 1] = (uint16_t)sd_h2(pcode);
        pend[s] = (pk >> SP_PEND) & 255u;
        evn[s] = (uint8_t)((pk >> SP_EVN) & 255u);
        upd[s] = (uint8_t)((pk >> SP_UPD) & 1u);
        pvalid[s] = (uint8_t)((pk >> SP_PVALID) & 1u);
    }
    // log(frequency) of the non-zero haplotypes: what the general sampler keeps beside the frequencies
    SPtrF<double, LANES> freq = c.freq(), logf = c.logf();
    SPtrF<uint8_t, LANES> nz = c.nz();
    freq[0] = f0;
    freq[1] = f1;
    nz[0] = (uint8_t)(nzmask & 1u);
    nz[1] = (uint8_t)((nzmask >> 1) & 1u);
    if (nzmask & 1u) logf[0] = bt_log(f0);
    if (nzmask & 2u) logf[1] = bt_log(f1);
}

// ---- inside a resident chain of a noise driver (bt_noise_chain.hpp) ----
// A new noise table has arrived (clearGenotyperCache + the next visit's cache fill): the table of unique-k-mer sums is rebuilt and the samples' weights with it.
__device__ static __noinline__ void simple_reweight(Env env, uint32_t blk_off) {
    const Tile t = make_tile(env);
    const Vx c = make_vx(t, 0);
    const GParams BT_CAS &P = env_params(env);
    simple_fill_unique(env);
    c.sc()[SC_UC_DIRTY] = 0;
    LdsArr<uint32_t> blk{(uint32_t BT_LAS *)(bt_lds_raw + blk_off) + t.lane};
    const Vx::UCPtr uc = c.ucache();
    const uint32_t Dcm = t.d->Dcm;
    for (uint32_t s = 0; s < P.S; ++s) {
        const uint32_t pk = blk[SB_WORDS * s + 2];
        const SimpleWeights w = simple_weights(uc, Dcm, s, (pk >> SP_PLOIDY) & 3u);
        blk[SB_WORDS * s] = __float_as_uint(w.ea);
        blk[SB_WORDS * s + 1] = __float_as_uint(w.eb);
        blk[SB_WORDS * s + 2] = (pk & ~(3u << SP_WMAX)) | (w.wmax << SP_WMAX);
    }
}
// VariantClusterGenotyper::getNoiseCounts (:757-779) for the lane's cluster from the packed sample words, then clearCache (:131-138: the table is
// rebuilt by simple_reweight when the next table arrives)
__device__ static __noinline__ void simple_noise_tally(Env env, uint32_t blk_off, const NoiseChainCtl *nc) {
    const Tile t = make_tile(env);
    const Vx c = make_vx(t, 0);
    const GParams BT_CAS &P = env_params(env);
    LdsArr<uint32_t> blk{(uint32_t BT_LAS *)(bt_lds_raw + blk_off) + t.lane};
    auto *bins = nc_bins(nc);
    // from the compact copies of the subset (sample_kmer_subset: rows of the two haplotypes' multiplicities, the intercluster multiplicities and the counts,
    // both zero for a k-mer without counts — what unique_mult / the count test of getNoiseCounts read through the subset's index list), four k-mers per step
    const uint32_t nsu = c.sc()[SC_NSUB_U], Hm = t.d->Hm, S = P.S;
    const Vx::RPtr<uint8_t> sm = c.subm();
    TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
    for (uint32_t i = 0; i < nsu; i += 4) {
        uint8_t m0[4], m1[4], ic0[4], ic1[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t j = i + r < nsu ? i + r : nsu - 1u;
            m0[r] = sm[j * Hm];
            m1[r] = sm[j * Hm + 1];
            ic0[r] = sic[2 * j];
            ic1[r] = sic[2 * j + 1];
        }
        // four samples at a time: their sixteen counts are requested together (round 6: one memory round trip per four samples and block instead of one per sample)
        for (uint32_t s0 = 0; s0 < S; s0 += 4) {
            uint8_t cn[4][4];
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t s = s0 + w < S ? s0 + w : S - 1u;
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r) cn[w][r] = scn[(i + r < nsu ? i + r : nsu - 1u) * S + s];
            }
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t s = s0 + w;
                if (s >= S) continue;
                const uint32_t code = blk[SB_WORDS * s + 2] & 7u, h1 = sd_h1(code), h2 = sd_h2(code);
                const bool g1 = P.gender[s] != 0;
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r) {
                    uint8_t m = 0;
                    if (h1 != (uint32_t)NOHAP) m = (uint8_t)(m + (h1 ? m1[r] : m0[r]));
                    if (h2 != (uint32_t)NOHAP) m = (uint8_t)(m + (h2 ? m1[r] : m0[r]));
                    m = (uint8_t)(m + (g1 ? ic1[r] : ic0[r]));
                    if (i + r < nsu && m == 0) nc_tally(nc, bins, s, cn[w][r]);
                }
            }
        }
    }
    c.sc()[SC_UC_DIRTY] = 1;
}

// n_burn sweeps without and then n_collect sweeps with collection of the tile's clusters (one per lane).
// nc != nullptr: the sweeps are the iterations of a noise driver's chain — before every sweep but the first the workgroup waits for the new noise table
// and rebuilds its weights; after every sweep it tallies its noise counts and takes part in the exchange with the host.
// (One call site per kernel.)
__device__ inline void simple_sweeps(const Env &env, const Tile &t, const GParams BT_CAS &P, uint32_t n_burn, uint32_t n_collect, uint32_t *trace_counter, uint32_t *trace_buf,
                                     uint32_t trace_max, uint32_t tile, const NoiseChainCtl *nc = nullptr) {
    const uint32_t n_sweeps = n_burn + n_collect;
    const TileDesc BT_CAS &d = *t.d;
    const Vx c = make_vx(t, 0);   // (general accessors for the arrays that stay in HBM and for the slow paths)
    const uint32_t S = P.S;
    const uint32_t blk_off = d.sblk;
    const double p_simplex1 = simple_enter(env, blk_off);
    LdsArr<uint32_t> blk{(uint32_t BT_LAS *)(bt_lds_raw + blk_off) + t.lane};
    SPtrF<uint32_t, LANES> sc = c.sc();
    const bool is_sparse = sc[SC_IS_SPARSE] != 0;
    Set2 zero{0, 0, 0}, plus{0, 0, 0};
    if (is_sparse) {
        zero = set2_load(c.zero_set());
        plus = set2_load(c.plus_set());
    }
    double fnd_saved = c.fnd_saved();
    uint32_t fnd_avail = sc[SC_FND_AVAIL];
    double f0, f1;
    uint32_t nzmask, obs0, obs1;
    {
        SPtrF<double, LANES> freq = c.freq();
        SPtrF<uint8_t, LANES> nz = c.nz();
        SPtrF<uint32_t, LANES> obs = c.obs();
        nzmask = (nz[0] ? 1u : 0u) | (nz[1] ? 2u : 0u);
        f0 = (nzmask & 1u) ? (double)freq[0] : 0.0;
        f1 = (nzmask & 2u) ? (double)freq[1] : 0.0;
        obs0 = obs[0];
        obs1 = obs[1];
    }
    typedef MtRingT<LdsArr<uint32_t>> Ring;
    const LdsArr<uint32_t> ring0 = lds_arr<uint32_t>(t, A_RING), ring1 = ring0 + (d.ring_cap[0] + MT_RING_HDR);
    Ring r0 = mt_ring_open_as(c.mt(0), ring0, d.ring_cap[0]), r1 = mt_ring_open_as(c.mt(1), ring1, d.ring_cap[1]);
    const double BT_GAS *a2tab = P.gamma_a2;
    const uint32_t a2n = P.gamma_n;

    const uint32_t sweep0 = nc ? nc->it_begin : 0u;
    for (uint32_t sweep = sweep0; sweep < n_sweeps; ++sweep) {
        const bool collect = sweep >= n_burn;
        if (nc && sweep > sweep0) {
            if (!nc_wait_table(nc, sweep)) break;
            if (nc->help_units) noise_help(env, nc);   // (the other tiles' large tables: bt_noise_help.hpp)
            simple_reweight(env, blk_off);
        }
        // ---- trace row of this sweep ----
        bool tracing = false;
        TPtr<uint32_t> trace_row{(uint32_t BT_GAS *)trace_buf, 0u, 6u};
        if (trace_max) {
            uint32_t *cnt = &trace_counter[(size_t)tile * LANES + t.lane];
            const uint32_t n = *cnt;
            if (n < trace_max) {
                *cnt = n + 1;
                trace_row.off = (uint32_t)(d.trace_base + (size_t)n * d.nvm * S * LANES) + t.lane;
                tracing = true;
            }
        }
        PROF_DECL;
        {   // the visit's words: exactly two per sample from the diplotype generator, the frequency generator's ring full
            const uint32_t want = 2u * S < r0.cap - 3u ? 2u * S : r0.cap - 3u;   // (more samples than the ring holds words for: the draws top up on the way)
            if (bt_wave_any(r0.avail < want)) r0.topup();   // (a ring that holds two visits' words is filled every other visit)
            r1.topup();
        }
        PROF(11);
        // ---- sampleDiplotypes ----
        const double ff00 = f0 * f0, ff01 = 2.0 * f0 * f1, ff11 = f1 * f1;
        // a frequency so small that a candidate's weight could be lost against another's rounding is left to the exact chain
        const bool tiny = ((nzmask & 1u) && f0 < 1e-12) || ((nzmask & 2u) && f1 < 1e-12);
        for (uint32_t s = 0; s < S; ++s) {
            uint32_t pk = blk[SB_WORDS * s + 2];
            const float wa = __uint_as_float(blk[SB_WORDS * s]), wb = __uint_as_float(blk[SB_WORDS * s + 1]);
            // LogDiscreteSampler: the draw happens even for a single outcome (DiscreteSampler.cpp:120-125)
            const double u01 = rng_canonical(r0);
            const uint32_t pl = (pk >> SP_PLOIDY) & 3u, wmax = (pk >> SP_WMAX) & 3u, pcode = pk & 7u;
            uint32_t code = SD_NONE;
            if (pl != 0) {
                const double e0 = wmax == 0 ? 1.0 : (double)wa, e1 = wmax == 0 ? (double)wa : (wmax == 1 ? 1.0 : (double)wb), e2 = wmax == 2 ? 1.0 : (double)wb;
                // candidates in the reference's order over the non-zero haplotypes: diploid (0,0) (0,1) (1,1), haploid (0) (1); a zero frequency
                // gives a zero weight, i.e. the candidate is absent
                const double w0 = (pl == 2 ? ff00 : f0) * e0, w1 = (pl == 2 ? ff01 : f1) * e1, w2 = pl == 2 ? ff11 * e2 : 0.0;
                const double c1 = w0 + w1, tot = c1 + w2, thr = u01 * tot, mg = 1e-5 * tot;
                const uint32_t pick = thr < w0 ? 0u : (thr < c1 ? 1u : 2u);
                const double lo = pick == 1 ? w0 : c1, hi = pick == 0 ? w0 : (pick == 1 ? c1 : tot);
                const bool safe = !tiny && tot > 1e-280 && hi - thr > mg && (pick == 0 || thr - lo > mg);
                code = pl == 2 ? pick : 3u + pick;
                if (!safe) {
                    PROF_CNT(19, 1);
                    code = simple_exact_code(env, s, pl, nzmask, f0, f1, u01);
                }
            }
            // HaplotypeFrequencyDistribution::incrementCount for the drawn haplotypes, in order
            {
                const uint32_t h1 = sd_h1(code), h2 = sd_h2(code);
#pragma unroll
                for (uint32_t w = 0; w < 2; ++w) {
                    const uint32_t h = w ? h2 : h1;
                    if (h == (uint32_t)NOHAP) continue;
                    const uint32_t o = h ? obs1 : obs0;
                    if (is_sparse && o == 0) {
                        zero.erase(h);
                        plus.insert(h);
                    }
                    if (h) obs1 = o + 1;
                    else obs0 = o + 1;
                }
            }
            pk = (pk & ~7u) | code;
            if (code != pcode) pk |= 1u << SP_UPD;   // update_multicluster_multiplicities without multicluster k-mers
            if (tracing) trace_row[s] = sd_key(code);
            // ---- collected sweep: diplotype_sampling_frequencies + updateAlleleKmerStats, logged as runs of identical sweeps ----
            if (collect) {
                const uint32_t pdc = (pk >> SP_PDIP) & 7u, run = (pk >> SP_PEND) & 255u;
                const bool pv = (pk >> SP_PVALID) & 1u;
                if (pv && pdc == code && run < 255u) pk += 1u << SP_PEND;
                else {
                    if (pv && run) {
                        uint32_t n = (pk >> SP_EVN) & 255u;
                        if (n == EV_CAP) {
                            simple_apply_full_log(env, s);
                            n = 0;
                        }
                        TPtr<uint32_t> lg = c.evlog(s);
                        lg[1 + 2 * n] = sd_key(pdc);   // (stores only: nothing waits for HBM while sampling)
                        lg[2 + 2 * n] = run;
                        pk = (pk & ~(255u << SP_EVN)) | ((n + 1u) << SP_EVN);
                    }
                    pk = (pk & ~((7u << SP_PDIP) | (255u << SP_PEND))) | (code << SP_PDIP) | (1u << SP_PEND) | (1u << SP_PVALID);
                }
            }
            blk[SB_WORDS * s + 2] = pk;
        }
        PROF(2);
        // ---- sampleHaplotypeFrequencies ----
        const uint32_t n_obs = obs0 + obs1;
        if (n_obs > 0) {
            double saved = fnd_saved;
            uint32_t avail = fnd_avail;
            const NormalState nd{&saved, &avail};
            double g0 = 0, g1 = 0, norm = 0;
            bool sel0 = !is_sparse, sel1 = !is_sparse;
            uint32_t n_first = 2, n_steps = 2;   // non-sparse: a gamma draw per haplotype, in index order
            Set2 order0 = plus;
            if (is_sparse) {
                // simplex size = |plus| + upper_bound(cached vector, U): with two haplotypes the vector has one entry that is not 1
                const double u = rng_canonical(r1);
                n_first = plus.n;
                n_steps = plus.n + ((plus.n == 1 && !(u < p_simplex1)) ? 1u : 0u);
            }
            // one gamma draw per step: the observed haplotypes in the plus set's iteration order, then unobserved ones drawn from the zero set
            for (uint32_t i = 0; bt_wave_any(i < n_steps); ++i) {
                PROF_CNT(20, 1);
                if (i < n_steps) {
                    uint32_t e, o = 0;
                    if (!is_sparse) e = i;
                    else if (i < n_first) e = order0.at(i);
                    else {
                        const uint32_t pos = rng_uniform_int(r1, zero.n);   // uniform_int(0, |zero| - 1)
                        e = zero.at(pos);
                        zero.erase(e);
                        plus.insert(e);
                    }
                    if (i < n_first) o = e ? obs1 : obs0;
                    const uint32_t ai = o + 1u;
                    const double alpha = (double)ai, a1 = alpha - 1.0 / 3.0;
                    const double a2 = ai < a2n ? (double)a2tab[ai] : 1.0 / sqrt(9.0 * a1);
                    const double v = rng_gamma_v(r1, nd, a1, a2);
                    const double x = a1 * v * 1.0;
                    if (e) g1 = x, sel1 = true;
                    else g0 = x, sel0 = true;
                    norm += x;
                }
            }
            // "for z in zero: freq = 0"; "for p in plus: freq /= norm; zero.insert(p)" then plus.clear()
            f0 = sel0 ? g0 / norm : 0.0;
            f1 = sel1 ? g1 / norm : 0.0;
            nzmask = (sel0 ? 1u : 0u) | (sel1 ? 2u : 0u);
            if (is_sparse) {
                const Set2 order = plus;
                plus.clear();
                for (uint32_t i = 0; i < 2; ++i)
                    if (i < order.n) zero.insert(order.at(i));
            }
            obs0 = obs1 = 0;
            fnd_saved = saved;
            fnd_avail = avail;
        }
        PROF(7);
        if (nc) {
            simple_noise_tally(env, blk_off, nc);
            if (!nc_iteration_end(nc, sweep)) break;
        }
    }
    // ---- back to the general representation ----
    mt_close(r0);
    mt_close(r1);
    if (is_sparse) {
        set2_store(c.zero_set(), zero);
        set2_store(c.plus_set(), plus);
    }
    c.fnd_saved() = fnd_saved;
    sc[SC_FND_AVAIL] = fnd_avail;
    sc[SC_HAP_COUNT] = obs0 + obs1;
    sc[SC_USE_MULTI] = 0;
    {
        SPtrF<uint32_t, LANES> obs = c.obs();
        obs[0] = obs0;
        obs[1] = obs1;
    }
    simple_leave(env, blk_off, f0, f1, nzmask);
}

}  // namespace bt
