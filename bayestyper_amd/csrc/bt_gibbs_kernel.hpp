// Device code of the Gibbs kernels (the group-level schedule around bt_gibbs_tile.hpp / bt_gibbs_simple.hpp) — included by the two
// translation units that instantiate it: bt_gibbs.hip (gibbs_kernel: every tile, every operation) and bt_gibbs_simple_kernel.hip
// (gibbs_simple_kernel: tiles of two-haplotype clusters, the three sampling operations; compiled for half the registers).
#pragma once
#include "bt_gibbs_tile.hpp"
#include "bt_noise_chain.hpp"
#include "bt_noise_help.hpp"
#include "bt_gibbs_simple.hpp"

namespace bt {

enum GibbsOp { OP_RUN = 0, OP_INIT_CHAIN = 1, OP_SWEEP = 2, OP_NOISE = 3, OP_RESET = 4, OP_SETUP = 5, OP_NOISE_CHAIN = 6 };

struct TraceCfg {
    uint32_t max_sweeps;   // 0 = off
    uint32_t *counter;     // [ntiles*64] sweeps recorded per group
    uint32_t *buf;         // tile blocks of [max_sweeps][nvm][S][64]
};

__device__ BT_NOINLINE void group_init_chain(Env env, uint32_t chain, uint32_t nvert, uint32_t nsrc, uint32_t gindex) {
    const Tile t = make_tile(env);
    const GParams BT_CAS &P = env_params(env);
    const uint32_t gseed = P.noise_seeding ? P.seed + (gindex + 1u) * (chain + 1u) : P.seed + (gindex + 1u);
    for (uint32_t v = 0; v < nvert; ++v) {
        const Vx c = make_vx(t, v);
        Env ev = env;
        const bool swap = env.resident == 0xFFFFFFFFu && t.d->hot_bytes != 0;
        if (swap) {
            hot_swap(env, v, true);
            ev.resident = v;
        }
        PROF_DECL;
        if (!make_vx(make_tile(ev), v).sc()[SC_CONSTRUCTED]) genotyper_construct(ev, v, gseed + c.cid());   // VariantClusterGroup.cpp:179-182
        genotyper_reset(ev, v);
        if (swap) hot_swap(env, v, false);
    }
    // shuffleBranchOrdering (VariantClusterGroup.cpp:208-218)
    if (nvert == 1 && nsrc == 1) return;   // nothing to shuffle (a fresh generator is seeded per call, no state carries over)
    uint32_t *bst = (uint32_t *)(t.base + t.d->off[A_BRNG]) + (size_t)t.plane * MT_PAD;
    mt_seed(bst, P.seed + (gindex + 1u) * (chain + 1u));
    Mt brng = mt_open(bst);
    rng_shuffle_u32(brng, t.arr<uint32_t>(A_SOURCES), nsrc);
    for (uint32_t v = 0; v < nvert; ++v) {
        const Vx c = make_vx(t, v);
        rng_shuffle_u32(brng, c.edges(), vx_ne(c));
    }
}

// VariantClusterGenotyper::updateNestedVariantClusterInfo (VariantClusterGenotyper.cpp:140-206): child's info := parent's info, updated
__device__ BT_NOINLINE void prepare_nested(Env env, uint32_t v_parent, uint32_t v_child) {
    if (env.resident != RESIDENT_ALL && env.resident != RESIDENT_NEVER) env.resident = 0xFFFFFFFFu;   // swap mode: between visits every vertex's hot arrays are in HBM
    const Tile t = make_tile(env);
    const GParams BT_CAS &P = env_params(env);
    const Vx c = make_vx(t, v_parent), cc = make_vx(t, v_child);
    const uint32_t nd_n = vx_nd(c), cc_cid = cc.cid();
    const TileDesc BT_CAS &d = c.d();
    TPtr<uint32_t> ndcl = c.a<uint32_t>(A_NDCL, d.NDm > 1 ? d.NDm : 1), ndvo = c.a<uint32_t>(A_NDVOFF, d.NDm + 1);
    TPtr<uint16_t> ndv = c.a<uint16_t>(A_NDVAR, d.NDVm > 1 ? d.NDVm : 1);
    TPtr<uint32_t> pver = c.nver(), cver = cc.nver();
    for (uint32_t s = 0; s < P.S; ++s) {
        // The child's nested info of a sample is a function of the parent's diplotype, k-mer-stats cache and own nested info of that sample:
        // nothing to do while the parent's version of them is the one this info was prepared from.
        const uint32_t pv = pver[s];
        if (cver[P.S + s] == pv) continue;
        cver[P.S + s] = pv;
        cver[s] += 1;
        uint8_t ploidy = c.nest_ploidy()[s];
        uint32_t n = c.nest_n()[s];
        for (uint32_t j = 0; j < n; ++j)
            for (int q = 0; q < 4; ++q) cc.nest_stats(s, j)[q] = (double)c.nest_stats(s, j)[q];
        for (uint32_t which = 0; which < 2; ++which) {
            const uint16_t h = c.dip()[2 * s + which];
            if (h == NOHAP) continue;
            // binary_search(nested_variant_cluster_indices of h, child cluster idx)
            bool found = false;
            uint32_t lo = c.hn_off(h), hi = c.hn_off(h + 1);
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t val = c.hn_idx(mid);
                if (val == cc_cid) {
                    found = true;
                    break;
                }
                if (val < cc_cid) lo = mid + 1;
                else hi = mid;
            }
            if (found) continue;
            ploidy = ploidy == 2 ? 1 : 0;   // updateNestedPloidy
            uint32_t variant_idx = 0xFFFFFFFFu;
            for (uint32_t dd = 0; dd < nd_n; ++dd) {
                if (ndcl[dd] != cc_cid) continue;
                for (uint32_t i = ndvo[dd], i1 = ndvo[dd + 1]; i < i1; ++i) {
                    const uint32_t nv = ndv[i];
                    const uint32_t a = c.hap_allele(h, nv);
                    if (!is_missing(c, nv, a)) {
                        variant_idx = nv;
                        break;
                    }
                }
                break;
            }
            if (variant_idx != 0xFFFFFFFFu && n < 2) {
                for (int q = 0; q < 4; ++q) cc.nest_stats(s, n)[q] = (double)c.ksc(s, which, variant_idx)[q];
                ++n;
            }
        }
        cc.nest_ploidy()[s] = ploidy;
        cc.nest_n()[s] = (uint8_t)n;
    }
}

__device__ __forceinline__ void visit_vertex(const Env &env_in, const Tile &t, const GParams BT_CAS &P, uint32_t v, bool collect, TPtr<uint32_t> trace_row, bool tracing) {
    Env env = env_in;
    const bool swap = env.resident == 0xFFFFFFFFu && t.d->hot_bytes != 0;
    PROF_DECL;
    if (swap) {
        hot_swap(env, v, true);
        env.resident = v;
    }
    PROF(13);
    rng_topup(env, v);
    PROF(11);
    sample_diplotypes(env, v, collect, trace_row.off + v * P.S * LANES, tracing, (uint32_t *)trace_row.base);
    sample_haplotype_frequencies(env, v);
    PROF_DECL2;
    if (swap) hot_swap(env, v, false);
    PROF(13);
}

// VariantClusterGroup::estimateGenotypes + runGibbsSample (VariantClusterGroup.cpp:220-250), recursion unrolled on an explicit stack.
// ONE call site of visit_vertex (and one of group_sweep in the kernel): the sweep functions are part of the kernel body (BT_SWEEP_INLINE),
// tens of thousands of instructions that must exist once.
__device__ __forceinline__ void group_sweep(const Env &env, const Tile &t, const GParams BT_CAS &P, bool collect, uint32_t nvert, uint32_t nsrc, TPtr<uint32_t> trace_row, bool tracing) {
    if (tracing)
        for (uint32_t i = 0; i < t.d->nvm * P.S; ++i) trace_row[i] = 0xFFFFFFFFu;
    TPtr<uint32_t> sources = t.arr<uint32_t>(A_SOURCES), stack = t.arr<uint32_t>(A_STACK);
    TPtr<uint8_t> gploidy = t.arr<uint8_t>(A_PLOIDY);
    for (uint32_t si = 0; si < nsrc; ++si) {
        const uint32_t sv = sources[si];
        {
            const Vx root = make_vx(t, sv);
            for (uint32_t s = 0; s < P.S; ++s) {
                root.nest_ploidy()[s] = gploidy[s];
                root.nest_n()[s] = 0;
            }
        }
        // depth-first from the source: visit a vertex, push it, then descend into the next unvisited child of the top of the stack
        uint32_t next = sv, sp = 0;
        bool have = true;
        while (have) {
            visit_vertex(env, t, P, next, collect, trace_row, tracing);
            if (nvert == 1) break;
            stack[2 * sp] = next;
            stack[2 * sp + 1] = 0;
            ++sp;
            have = false;
            while (sp > 0 && !have) {
                const uint32_t v = stack[2 * (sp - 1)];
                const uint32_t i = stack[2 * (sp - 1) + 1];
                const Vx c = make_vx(t, v);
                if (i < vx_ne(c)) {
                    stack[2 * (sp - 1) + 1] = i + 1;
                    const uint32_t tv = c.edges()[i];
                    {
                        PROF_DECL;
                        prepare_nested(env, v, tv);
                        PROF(14);
                    }
                    next = tv;
                    have = true;
                } else
                    --sp;
            }
        }
    }
}

// VariantClusterGenotyper::getNoiseCounts (:757-779) + clearCache (:131-138) for every vertex of the lane's group, inside a resident chain of a noise
// driver (bt_noise_chain.hpp): the counts go to the workgroup's LDS bins.  The hot arrays are read where they are (LDS for resident groups).
__device__ BT_NOINLINE void noise_tally_group(Env env, uint32_t nvert, const NoiseChainCtl *nc) {
    if (env.resident != RESIDENT_ALL && env.resident != RESIDENT_NEVER) env.resident = 0xFFFFFFFFu;   // swap mode: between visits every vertex's hot arrays are in HBM
    const Tile t = make_tile(env);
    const GParams BT_CAS &P = env_params(env);
    auto *bins = nc_bins(nc);
    for (uint32_t v = 0; v < nvert; ++v) {
        const Vx c = make_vx(t, v);
        // from the compact copies of the subset (sample_kmer_subset: multiplicity rows, intercluster multiplicities and counts — both zero for a k-mer
        // without counts: what unique_mult and the count test of getNoiseCounts read through the subset's index list)
        const uint32_t nsu = c.sc()[SC_NSUB_U], Hm = t.d->Hm;
        const Vx::RPtr<uint8_t> sm = c.subm();
        TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
        for (uint32_t s = 0; s < P.S; ++s) {
            const uint16_t h1 = c.dip()[2 * s], h2 = c.dip()[2 * s + 1];
            const uint8_t gender = P.gender[s];
            for (uint32_t i = t.part; i < nsu; i += t.copies) {   // the copies of a narrow tile's group share the k-mers of the subset (a tally: any order)
                uint8_t m = sic[2 * i + gender];
                if (h1 != NOHAP) m = (uint8_t)(m + sm[i * Hm + h1]);
                if (h2 != NOHAP) m = (uint8_t)(m + sm[i * Hm + h2]);
                if (m == 0) nc_tally(nc, bins, s, scn[i * P.S + s]);
            }
        }
        if (t.part == 0) cache_clear(c, P, false, false);   // (the other copies read nothing this touches before the next sweep)
        if (nc->help_units && help_table(t.d->cache_mode, t.d->simple, t.d->cache_entries, t.d->hoff[A_UCACHE], t.d->Dcm)) {   // the next sweep's sums are computed by every workgroup of the chain (bt_noise_help.hpp)
            if (t.copies > 1u) copies_sync();
            noise_help_publish(t, c);
        }
    }
    if (t.copies > 1u) copies_sync();
    if (nc->help_units && help_table(t.d->cache_mode, t.d->simple, t.d->cache_entries, t.d->hoff[A_UCACHE], t.d->Dcm)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the invalidated tables and the published state, before the arrival
}

struct TraceRow {
    TPtr<uint32_t> row;
    bool on;
};
__device__ inline TraceRow trace_row_for(const Tile &t, const GParams BT_CAS &P, const TraceCfg &tr, uint32_t tile) {
    TraceRow r{TPtr<uint32_t>{(uint32_t BT_GAS *)tr.buf, 0u, 6u}, false};
    if (!tr.max_sweeps) return r;
    uint32_t *cnt = &tr.counter[(size_t)tile * LANES + t.lane];
    const uint32_t n = *cnt;
    if (n >= tr.max_sweeps) return r;
    *cnt = n + 1;
    r.row = TPtr<uint32_t>{(uint32_t BT_GAS *)tr.buf, (uint32_t)(t.d->trace_base + (size_t)n * t.d->nvm * P.S * LANES) + t.lane, 6u};
    r.on = true;
    return r;
}

#ifndef GIBBS_WAVES
#define GIBBS_WAVES 1
#endif
// SIMPLE_ONLY: the instantiation for launch classes made of simple tiles only (two-haplotype clusters, the bulk of a batch) and the three
// sampling operations (bt_gibbs_simple_kernel.hip).
template <bool SIMPLE_ONLY>
__device__ __forceinline__ void gibbs_body(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg, int op, uint32_t arg0, uint32_t arg1,
                                           unsigned long long *__restrict__ hist, TraceCfg tr, const uint32_t *__restrict__ tile_list) {
    const uint32_t tile = kPacked ? ((const uint32_t BT_CAS *)tile_list)[2u * pack_slot()] : (tile_list ? tile_list[blockIdx.x] : blockIdx.x);
    if (kPacked && tile == 0xFFFFFFFFu) return;   // a slot of the last workgroups without a tile
    Env env{tiles, pool, Pg, tile_list, 0xFFFFFFFFu};
    const GParams BT_CAS &P = *(const GParams BT_CAS *)Pg;
    Tile t;
    t.d = (const TileDesc BT_CAS *)&tiles[tile];
    t.base = (uint8_t BT_GAS *)(pool + t.d->base);
    if (!tile_thread_active(tile_split(t.d), t.d->copies)) return;
    t.lane = tile_lane(tile_split(t.d), t.d->copies);
    t.plane = t.lane + t.d->pool_lane0;
    t.wsh = t.d->wsh;
    t.part = tile_part(t.d->copies);
    t.copies = t.d->copies;
    // the extra copies of a narrow tile's groups only take part in the sampling operations (the others tally with atomics)
    if (t.part != 0 && !(op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN || op == OP_NOISE || op == OP_NOISE_CHAIN)) return;
    t.hot = nullptr;
    t.lds0 = kPacked ? ((const uint32_t BT_CAS *)tile_list)[2u * pack_slot() + 1u] : 0u;
    t.resident = 0xFFFFFFFFu;
    // Narrow tiles (few groups + lockstep copies) are the launch's critical path: a handful of long sequential programs.  They take
    // issue priority over the 64-group tiles they share a SIMD with, which have plenty of peers to fill the gaps.
    if (t.d->prio) __builtin_amdgcn_s_setprio(3);
    TPtr<uint32_t> gd = t.arr<uint32_t>(A_GDIMS);
    if (!gd[3]) return;   // padding lane of the last tile
    const uint32_t nvert = kSingle ? 1u : (uint32_t)gd[0], nsrc = kSingle ? 1u : (uint32_t)gd[1], gindex = gd[2];
    // OP_NOISE_CHAIN: a whole chain of a noise driver, the tiles resident (bt_noise_chain.hpp); `hist` carries the launch class's control block.
    // (Lane 0 of a tile is always a group, so thread 0 and wavefront 0 of the workgroup are still here.)
#ifdef BT_NO_NOISE_CHAIN   // translation units whose kernel is never launched for a resident chain (gibbs_hot_kernel, gibbs_single_kernel: bt_gibbs.hip launch())
    const bool is_nc = false;
#else
    const bool is_nc = op == OP_NOISE_CHAIN;
#endif
    const NoiseChainCtl *nc = is_nc ? (const NoiseChainCtl *)hist : nullptr;
    // groups of ONE cluster keep that cluster's hot arrays in LDS for the whole launch; larger groups swap per vertex visit
    // (and so do multi-cluster groups of narrow tiles, whose LDS rows are interleaved over fewer lanes: TileDesc::lds_all)
    // (a resident chain charges every workgroup the same LDS: tiles above its cap keep their hot arrays in HBM for the launch — RESIDENT_NEVER)
    const bool no_lds = is_nc && (t.d->lds_all ? t.d->hot_bytes * t.d->nvm : t.d->hot_bytes) > nc->lds_cap;
    if (no_lds) env.resident = t.resident = RESIDENT_NEVER;
    const bool whole = !no_lds && (t.d->nvm == 1 || t.d->lds_all) && t.d->hot_bytes != 0 && (op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN || is_nc);
    if (whole) {
        env.resident = RESIDENT_ALL;
        for (uint32_t v = 0; v < nvert; ++v) hot_swap(env, v, true);
        t.resident = RESIDENT_ALL;
        t.hot = lds_block() + t.lds0;
    }
    // tiles of two-haplotype clusters run their straight-line sweep in gibbs_simple_kernel only; launched through the general kernel
    // (BT_GIBBS_NO_SIMPLE_KERNEL) they take the general path
    const bool simple = SIMPLE_ONLY;
    if (op == OP_RUN || op == OP_SWEEP || is_nc) {
        // OP_RUN: every chain = init + burn-in + collected sweeps; OP_SWEEP: arg0 sweeps of the chain in progress; OP_NOISE_CHAIN: the iterations of a
        // noise driver's chain, each sweep followed by the noise tally and the exchange with the host.  One loop for all, so that the sweep code
        // (inlined) exists once
        const bool is_run = op == OP_RUN;
        const uint32_t nchains = is_run ? P.num_chains : 1u;
        uint32_t n_burn = is_run ? P.burn_in : (arg1 != 0 ? 0u : arg0), n_collect = is_run ? P.num_iterations : (arg1 != 0 ? arg0 : 0u);
        if (is_nc) {
            n_burn = nc->first_collect < nc->n_iterations ? nc->first_collect : nc->n_iterations;
            n_collect = nc->n_iterations - n_burn;
            if (!nc_begin(nc)) n_burn = n_collect = 0;   // the launch's workgroups are not resident together: no sweep, the state as it was (bt_noise_chain.hpp: roll call)
        }
        for (uint32_t chain = 0; chain < nchains; ++chain) {
            if (is_run) {
                PROF_DECL;
                group_init_chain(env, chain, nvert, nsrc, gindex);
                PROF(15);
            }
            if constexpr (SIMPLE_ONLY) {
                simple_sweeps(env, t, P, n_burn, n_collect, tr.counter, tr.buf, tr.max_sweeps, tile, nc);
                if (is_run || n_collect) drain_collected(env);   // the chain's collected sweeps -> statistics (the next chain has another k-mer subset)
            } else {
                const uint32_t i0 = is_nc ? nc->it_begin : 0u;
                for (uint32_t i = i0; i < n_burn + n_collect; ++i) {
                    if (is_nc && i > i0) {
                        nc_phase(nc, 0x10000000u | i);
                        if (!nc_wait_table(nc, i)) break;
                        if (nc->help_units) {
                            noise_help(env, nc);
                            if (!noise_help_wait(nc, tile, i)) break;
                        }
                    }
                    if (is_nc) nc_phase(nc, 0x40000000u | i);
                    const TraceRow r = trace_row_for(t, P, tr, tile);
                    if constexpr (kSingle) {   // VariantClusterGroup::estimateGenotypes of a one-cluster group: the one visit
                        if (r.on)
                            for (uint32_t q = 0; q < P.S; ++q) r.row[q] = 0xFFFFFFFFu;
                        const Vx root = make_vx(t, 0);
                        TPtr<uint8_t> gploidy = t.arr<uint8_t>(A_PLOIDY);
                        for (uint32_t q = 0; q < P.S; ++q) {
                            root.nest_ploidy()[q] = gploidy[q];
                            root.nest_n()[q] = 0;
                        }
                        visit_vertex(env, t, P, 0u, i >= n_burn, r.row, r.on);
                    } else
                        group_sweep(env, t, P, i >= n_burn, nvert, nsrc, r.row, r.on);
                    if (is_nc) {
                        nc_phase(nc, 0x50000000u | i);
                        noise_tally_group(env, nvert, nc);
                        nc_phase(nc, 0x60000000u | i);
                        if (!nc_iteration_end(nc, i)) break;
                    }
                }
                if (is_run && t.d->logged) drain_collected(env);   // the chain's collected sweeps -> statistics (the next chain has another k-mer subset)
            }
        }
        if (!simple && (is_run || n_collect))
            for (uint32_t v = 0; v < nvert; ++v) flush_vertex(env, v);   // hot arrays: LDS when resident, else HBM (both valid)
    } else if (op == OP_INIT_CHAIN) {
        group_init_chain(env, arg0, nvert, nsrc, gindex);
        // stepwise driving (the noise drivers, tests): the launch is followed by nan_fill_kernel, and the first sweep by ucache_prefill_kernel — at a
        // chain start every haplotype has a non-zero frequency, so the first sweep asks for the whole table: the whole GPU computes it instead of the
        // tile's own lanes
        if (!SIMPLE_ONLY && arg1 != 0 && wide_table(*t.d))
            for (uint32_t v = 0; v < nvert; ++v) make_vx(t, v).sc()[SC_UC_DIRTY] = 0;
    } else if (SIMPLE_ONLY) {
        // (the other operations always go through the general kernel)
    } else if (op == OP_NOISE) {
        // VariantClusterGenotyper::getNoiseCounts (:757-779) for every vertex, then clearCache
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
                        atomicAdd(&hist[s * 256u + cnt], 1ULL);
                    }
                }
            }
            if (t.part == 0) cache_clear(c, P, false, arg0 != 0);   // (the other copies read nothing this touches)
        }
    } else if (op == OP_RESET) {
        // VariantClusterGroup::resetGroup: genotypers are deleted; the shared KmerCounts multiplicities are NOT reset
        for (uint32_t v = 0; v < nvert; ++v) make_vx(t, v).sc()[SC_CONSTRUCTED] = 0;
    } else if (op == OP_SETUP) {
        // mutable copies of the group structure (shuffled in place chain after chain, never restored)
        TPtr<uint32_t> s0 = t.arr<uint32_t>(A_SOURCES0), s1 = t.arr<uint32_t>(A_SOURCES);
        for (uint32_t i = 0; i < nsrc; ++i) s1[i] = s0[i];
        for (uint32_t v = 0; v < nvert; ++v) {
            const Vx c = make_vx(t, v);
            TPtr<uint32_t> e0 = c.a<uint32_t>(A_EDGES0, t.d->NEm > 1 ? t.d->NEm : 1), e1 = c.edges();
            for (uint32_t i = 0, n = vx_ne(c); i < n; ++i) e1[i] = e0[i];
            TPtr<uint32_t> dm = t.arr<uint32_t>(A_VDIMS, v * 8);   // dimensions the samplers read at every call: next to the state scalars
            SPtrF<uint32_t, LANES> sc = c.sc();
            sc[SC_H] = dm[0];
            sc[SC_V] = dm[1];
            sc[SC_NM] = dm[4];
        }
    }
    if (whole)
        for (uint32_t v = 0; v < nvert; ++v) hot_swap(env, v, false);
}
}  // namespace bt
