// Sweeps of TWO-HAPLOTYPE single-cluster groups (a biallelic SNV / indel: reference and alternative haplotype) — nine out of ten
// variant-cluster groups of a genome.  Same state, same arrays, same draw stream and arithmetic as the general sampler in
// bt_gibbs_tile.hpp (which still constructs, resets and flushes these clusters and runs the rare slow paths); what differs is HOW a
// sweep is executed: one straight-line routine with the cluster's dimensions as constants (H = 2: at most three diplotype
// candidates, so always the reference's chain of logAddition calls), the hot arrays addressed as LDS (ds_* instructions instead of
// generic pointers), the two emulated unordered_set<uint> kept as two-entry lists in registers, and no calls on the common path.
//
// Reference behaviour (as in bt_gibbs_tile.hpp):
//   VariantClusterGenotyper::sampleDiplotypes / sampleDiplotype / calcDiplotypeLogProb   VariantClusterGenotyper.cpp:597-755
//   HaplotypeFrequencyDistribution::incrementCount / SparseFrequencyDistribution          HaplotypeFrequencyDistribution.cpp:113-138,
//                                                                                         FrequencyDistribution.cpp:75-93,198-303
//   VariantClusterHaplotypes::updateAlleleKmerStats (deferred as in update_allele_kmer_stats)
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

struct SimpleState {   // what a run of sweeps keeps in registers between sweeps
    Set2 zero, plus;
    double fnd_saved;
    uint32_t fnd_avail, is_sparse, hap_count;
};

__device__ inline bool tile_is_simple(const TileDesc BT_CAS &d) { return d.simple != 0; }

// n_burn sweeps without and then n_collect sweeps with collection of the tile's clusters (one per lane).  The hot arrays of vertex 0 are
// resident in LDS (RESIDENT_ALL).  (One call site per kernel: the routine is inlined, and every further site would be another 24 000 instructions.)
__device__ inline void simple_sweeps(const Env &env, const Tile &t, const GParams BT_CAS &P, uint32_t n_burn, uint32_t n_collect, uint32_t *trace_counter, uint32_t *trace_buf,
                                     uint32_t trace_max, uint32_t tile) {
    const uint32_t n_sweeps = n_burn + n_collect;
    const TileDesc BT_CAS &d = *t.d;
    const Vx c = make_vx(t, 0);   // (general accessors for the arrays that stay in HBM and for the slow paths)
    const uint32_t S = P.S, Dcm = d.Dcm;
    LdsArr<uint32_t> sc = lds_arr<uint32_t>(t, A_SC), obs = lds_arr<uint32_t>(t, A_OBS), pend = lds_arr<uint32_t>(t, A_PEND);
    LdsArr<uint16_t> dip = lds_arr<uint16_t>(t, A_DIP), pdip = lds_arr<uint16_t>(t, A_PENDDIP);
    LdsArr<uint8_t> nz = lds_arr<uint8_t>(t, A_NZ), upd = lds_arr<uint8_t>(t, A_KSCUPD), pvalid = lds_arr<uint8_t>(t, A_PENDVALID), ploidy = lds_arr<uint8_t>(t, A_NESTPL),
                    nest_n = lds_arr<uint8_t>(t, A_NESTN);
    LdsArr<double> freq = lds_arr<double>(t, A_FREQ), logf = lds_arr<double>(t, A_LOGF);
    const Vx::UCPtr uc = c.ucache();   // LDS when the table is small, else HBM
    // the group's ploidy per sample is the cluster's (a root cluster without nesting: VariantClusterGroup.cpp:225-231)
    {
        TPtr<uint8_t> gp = t.arr<uint8_t>(A_PLOIDY);
        for (uint32_t s = 0; s < S; ++s) {
            ploidy[s] = gp[s];
            nest_n[s] = 0;
        }
    }
    SimpleState st;
    st.is_sparse = sc[SC_IS_SPARSE];
    st.zero = Set2{0, 0, 0};
    st.plus = Set2{0, 0, 0};
    if (st.is_sparse) {
        st.zero = set2_load(c.zero_set());
        st.plus = set2_load(c.plus_set());
    }
    st.fnd_saved = c.fnd_saved();
    st.fnd_avail = sc[SC_FND_AVAIL];
    st.hap_count = sc[SC_HAP_COUNT];
    typedef MtRingT<LdsArr<uint32_t>> Ring;
    const LdsArr<uint32_t> ring0 = lds_arr<uint32_t>(t, A_RING), ring1 = ring0 + (d.ring_cap[0] + MT_RING_HDR);
    Ring r0 = mt_ring_open_as(c.mt(0), ring0, d.ring_cap[0]), r1 = mt_ring_open_as(c.mt(1), ring1, d.ring_cap[1]);

    for (uint32_t sweep = 0; sweep < n_sweeps; ++sweep) {
        const bool collect = sweep >= n_burn;
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
        if (sc[SC_UC_DIRTY]) {   // chain start / clearCache: the dense table of unique-k-mer sums is rebuilt as a whole
            fill_unique_cache(env, 0);
            sc[SC_UC_DIRTY] = 0;
        }
        PROF_DECL;
        r0.topup();
        r1.topup();
        PROF(11);
        // ---- sampleDiplotypes ----
        uint32_t nzl[2] = {0, 0}, nnz = 0;
        if (nz[0]) nzl[nnz++] = 0;
        if (nz[1]) nzl[nnz++] = 1;
        const double lf0 = logf[nzl[0]], lf1 = nnz > 1 ? (double)logf[nzl[1]] : 0.0;
        for (uint32_t s = 0; s < S; ++s) {
            const uint16_t p1 = dip[2 * s], p2 = dip[2 * s + 1];
            const uint32_t pl = ploidy[s];
            const uint32_t total = pl == 2 ? nnz * (nnz + 1) / 2 : (pl == 1 ? nnz : 0u);
            // candidates in the reference's order: diploid (a, b >= a) over the non-zero haplotypes, haploid (a)
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
            // LogDiscreteSampler: the draw happens even for a single outcome (DiscreteSampler.cpp:120-125)
            const double u01 = rng_canonical(r0);
            uint32_t pick = 0;
            if (total == 0) (void)bt_log(u01);
            else {
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
            }
            uint16_t h1 = NOHAP, h2 = NOHAP;
            if (total != 0) {
                h1 = ca[pick];
                h2 = cb[pick];
            }
            dip[2 * s] = h1;
            dip[2 * s + 1] = h2;
            // HaplotypeFrequencyDistribution::incrementCount (x2)
#pragma unroll
            for (uint32_t w = 0; w < 2; ++w) {
                const uint16_t h = w ? h2 : h1;
                if (h == NOHAP) continue;
                st.hap_count += 1;
                const uint32_t o = obs[h];
                if (st.is_sparse && o == 0) {
                    st.zero.erase(h);
                    st.plus.insert(h);
                }
                obs[h] = o + 1;
            }
            if (h1 != p1 || h2 != p2) upd[s] = 1;   // update_multicluster_multiplicities without multicluster k-mers
            if (tracing) trace_row[s] = (uint32_t)h1 | ((uint32_t)h2 << 16);
        }
        PROF(2);
        // ---- collected sweep: diplotype_sampling_frequencies + updateAlleleKmerStats, deferred while nothing changes ----
        if (collect) {
            for (uint32_t s = 0; s < S; ++s) {
                if (pvalid[s] && pdip[2 * s] == dip[2 * s] && pdip[2 * s + 1] == dip[2 * s + 1]) {
                    pend[s] += 1;
                    continue;
                }
                if (pvalid[s] && pend[s]) log_collected_run(c, P, s, (uint32_t)pdip[2 * s] | ((uint32_t)pdip[2 * s + 1] << 16), pend[s], sc[SC_NSUB_U]);
                pdip[2 * s] = dip[2 * s];
                pdip[2 * s + 1] = dip[2 * s + 1];
                pend[s] = 1;
                pvalid[s] = 1;
            }
        }
        PROF(6);
        // ---- sampleHaplotypeFrequencies ----
        if (st.hap_count > 0) {
            double saved = st.fnd_saved;
            uint32_t avail = st.fnd_avail;
            const NormalState nd{&saved, &avail};
            if (!st.is_sparse) {
                double f[2], norm = 0;
#pragma unroll
                for (uint32_t h = 0; h < 2; ++h) {
                    f[h] = rng_gamma(r1, nd, (double)(obs[h] + 1u), 1.0);
                    norm += f[h];
                    obs[h] = 0;
                }
#pragma unroll
                for (uint32_t h = 0; h < 2; ++h) {
                    const double x = f[h] / norm;
                    freq[h] = x;
                    logf[h] = bt_log(x);
                }
            } else {
                const uint32_t n_obs = st.hap_count, plus_size = st.plus.n;
                // cached simplex-size distribution (FrequencyDistribution.cpp:143-196,211-229): at most 3 - plus_size entries
                TPtr<double> vec = c.simplex();
                const bool cached = d.scache_n && n_obs <= 2 * d.S && plus_size <= d.scache_p;
                const uint32_t ci = cached ? (n_obs - 1) * d.scache_p + (plus_size - 1) : 0u;
                if (cached) vec = c.scache() + ci * d.scache_len;
                double head[2];
                head[0] = cached ? (double)vec[0] : 0.0;
                head[1] = cached ? (double)vec[1] : 0.0;
                uint32_t len = cached ? (uint32_t)c.sclen()[ci] : 0u;
                if (len == 0) {
                    len = simplex_prob_vector(c, P, vec, n_obs, plus_size);
                    if (cached) c.sclen()[ci] = len;
                    head[0] = vec[0];
                    head[1] = len > 1 ? (double)vec[1] : 0.0;
                }
                const double u = rng_canonical(r1);
                uint32_t ub = 0;
                if (!(u < head[0])) {
                    ub = 1;
                    if (len > 1 && !(u < head[1])) ub = 2;
                }
                const uint32_t simplex_size = ub + plus_size;
                double f[2] = {0, 0}, norm = 0;
                bool sel[2] = {false, false};
                for (uint32_t i = 0; i < plus_size; ++i) {   // plus in iteration order
                    const uint32_t e = st.plus.at(i);
                    const double x = rng_gamma(r1, nd, (double)obs[e] + 1.0, 1.0);
                    f[e & 1u] = x;
                    norm += x;
                    sel[e & 1u] = true;
                }
                while (st.plus.n < simplex_size) {
                    const uint32_t pos = rng_uniform_int(r1, st.zero.n);   // uniform_int(0, |zero| - 1)
                    const uint32_t e = st.zero.at(pos);
                    const double x = rng_gamma(r1, nd, 1.0, 1.0);
                    f[e & 1u] = x;
                    norm += x;
                    sel[e & 1u] = true;
                    st.zero.erase(e);
                    st.plus.insert(e);
                }
#pragma unroll
                for (uint32_t h = 0; h < 2; ++h) {
                    if (sel[h]) nz[h] = 1;
                    else {
                        freq[h] = 0;
                        nz[h] = 0;
                        obs[h] = 0;
                    }
                }
                // "for p in plus: freq /= norm; zero.insert(p); obs = 0", then plus.clear()
                const Set2 order = st.plus;
                st.plus.clear();
                for (uint32_t i = 0; i < order.n; ++i) {
                    const uint32_t e = order.at(i);
                    const double x = f[e & 1u] / norm;
                    freq[e] = x;
                    logf[e] = bt_log(x);
                    st.zero.insert(e);
                    obs[e] = 0;
                }
            }
            st.fnd_saved = saved;
            st.fnd_avail = avail;
        }
        st.hap_count = 0;
        PROF(7);
    }
    // ---- back to the general representation ----
    mt_close(r0);
    mt_close(r1);
    if (st.is_sparse) {
        set2_store(c.zero_set(), st.zero);
        set2_store(c.plus_set(), st.plus);
    }
    c.fnd_saved() = st.fnd_saved;
    sc[SC_FND_AVAIL] = st.fnd_avail;
    sc[SC_HAP_COUNT] = st.hap_count;
    sc[SC_USE_MULTI] = 0;
}

}  // namespace bt
