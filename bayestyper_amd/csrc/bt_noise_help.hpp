// The help phase of a resident noise chain (bt_noise_chain.hpp): the sums a sweep will ask of the large dense tables, computed by EVERY workgroup of the chain.
//
// clearGenotyperCache empties the tables of unique-k-mer sums after every iteration (InferenceEngine.cpp:92), so each sweep evaluates its candidates' sums over
// the k-mer subset again (VariantClusterGenotyper.cpp:619-643).  Inside a sweep that work is bound to the lanes of the cluster's own tile, and the chain's
// iteration lasts as long as its slowest tile: a 14-candidate cluster at ten samples recomputes a thousand sums per iteration — 2.7 ms by its own sixteen
// lanes — while the 1 100 workgroups of the two-haplotype tiles, done after 0.18 ms, wait for it.  The candidates of a visit are the pairs of the haplotypes
// with a non-zero frequency, and a cluster's frequencies only change at the end of its own visit: the set is known when the iteration starts.  So
//   * at the end of an iteration the owner of a large table invalidates it (NaN = "not computed"), publishes what a helper needs of its LDS-resident state
//     (the non-zero flags, the state scalars) to the arrays' HBM home, makes both visible (one agent-scope release) and arrives;
//   * when the next table has arrived, every workgroup takes work units — (cluster, sample) — off one device-wide counter and computes the unit's sums
//     (unique_log_prob_block, the same function the sweep would call: same values), storing them write-through; a finished unit is counted per tile;
//   * the owner waits until its tile's units are counted, acquires, and sweeps: every sum it asks for is there.
// (The ordinary launches do the same between launches with ucache_prefill_kernel; the first resident iteration of a chain finds the tables filled by it.)
#pragma once
#include "bt_gibbs_tile.hpp"
#include "bt_noise_chain.hpp"

namespace bt {

// Which tables are worth sharing out: large dense tables (wide_table) of clusters with seven candidates and more (36 and more pairs per sample).  A work unit
// costs a few microseconds whatever it computes (two device-wide atomics, the owner's state scalars and non-zero flags from HBM), and at ten samples every
// three-candidate cluster has a "large" table (90 entries): 400 000 units per iteration of a chr20-sized chain kept every workgroup busy for 1.6 ms.
constexpr uint32_t NC_HELP_MIN_D = 32;
__host__ __device__ inline bool help_table(uint32_t cache_mode, uint32_t simple, uint32_t cache_entries, uint32_t ucache_hoff, uint32_t Dcm) {
    return cache_mode == 0 && !simple && cache_entries > BT_UC_INVALIDATE_MIN && ucache_hoff == NOHOT && Dcm >= NC_HELP_MIN_D;
}
struct HelpItem {
    uint32_t tile, lane, v, pad;
};
constexpr uint32_t NC_HELP_MAXH = 256;   // non-zero haplotypes a unit lists in LDS (clusters with more: computed by their own tile on demand)

// LDS words behind the bins, the flag word and the profiling stamp: [0] the unit a workgroup took, [1] number of listed haplotypes, [2..) the list (u16)
__device__ inline uint32_t BT_LAS *nc_help_words(const NoiseChainCtl *ctl) { return (uint32_t BT_LAS *)(bt_lds_raw + ctl->bins_off) + ((ctl->S * NC_BINS + 4u) & ~1u); }

// owner side, end of an iteration: vertex v of the lane's group has a large table
__device__ inline void noise_help_publish(const Tile &t, const Vx &c) {
    // invalidate now (the helpers test for NaN), all copies sharing the work
    const Vx::UCPtr uc = c.ucache();
    const double nan = __builtin_nan("");
    for (uint32_t i = t.part, n = c.d().cache_entries; i < n; i += t.copies) uc[i] = nan;
    c.sc()[SC_UC_DIRTY] = 0;
    // what a helper reads of the state that lives in LDS while the chain is resident: the state scalars and the non-zero flags, to their HBM home
    if (t.hot != nullptr && t.part == 0) {
        const TileDesc BT_CAS &d = c.d();
        if (d.hoff[A_SC] != NOHOT) {
            SPtrF<uint32_t, LANES> sc = c.sc();
            TPtr<uint32_t> home = t.arr<uint32_t>(A_SC, c.v * SC_COUNT);
            for (uint32_t i = 0; i < SC_COUNT; ++i) home[i] = sc[i];
        }
        if (d.hoff[A_NZ] != NOHOT) {
            SPtrF<uint8_t, LANES> nz = c.nz();
            TPtr<uint8_t> home = t.arr<uint8_t>(A_NZ, c.v * d.Hm);
            for (uint32_t h = 0; h < c.H; ++h) home[h] = nz[h];
        }
    }
}

// one work unit: the sums of sample s over the pairs of cluster (tile, lane, v)'s non-zero haplotypes that are not in its table yet
__device__ inline void noise_help_unit(const TileDesc *tiles, uint8_t *pool, const GParams BT_CAS &P, const NoiseChainCtl *nc, const HelpItem it, uint32_t s, const NcLanes &L) {
    uint32_t BT_LAS *hw = nc_help_words(nc);
    uint16_t BT_LAS *nzl = (uint16_t BT_LAS *)(hw + 2);
    Tile t;
    t.d = (const TileDesc BT_CAS *)&tiles[it.tile];
    t.base = (uint8_t BT_GAS *)(pool + t.d->base);
    t.lane = it.lane;
    t.plane = it.lane + t.d->pool_lane0;
    t.wsh = t.d->wsh;
    t.part = 0;
    t.copies = t.d->copies;
    t.hot = nullptr;   // (the owner's LDS is not ours: everything from HBM)
    t.lds0 = 0;
    t.resident = 0xFFFFFFFFu;
    const Vx c = make_vx(t, it.v);
    SPtrF<uint32_t, LANES> sc = c.sc();
    const bool skip = !sc[SC_CONSTRUCTED] || sc[SC_UC_DIRTY] || c.H > NC_HELP_MAXH;
    // the list of the non-zero haplotypes, in index order: wavefront 0's lanes take H in strides, a ballot compacts each stride
    uint32_t nnz = 0;
    __syncthreads();   // (the previous unit's readers of the list are done)
    if (!skip && L.on) {
        SPtrF<uint8_t, LANES> nz = c.nz();
        for (uint32_t h0 = 0; h0 < c.H; h0 += L.count) {
            const uint32_t h = h0 + L.rank;
            const bool on = h < c.H && nz[h] != 0;
            const unsigned long long m = __ballot(on);
            if (on) nzl[nnz + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull))] = (uint16_t)h;
            nnz += (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    if (skip || !L.on) return;
    const uint32_t pairs = nnz * (nnz + 1) / 2, total = pairs + (t.d->nvm > 1 ? nnz : 0u);
    const uint32_t nsub_u = sc[SC_NSUB_U];
    const TileDesc BT_CAS &d = c.d();
    const Vx::UCPtr uc = c.ucache();
    for (uint32_t base = EVB * L.rank; base < total; base += EVB * L.count) {
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
            unique_log_prob_block(c, P, s, ha, hb, need, nsub_u, out, /*store=*/false);
#pragma unroll
            for (uint32_t q = 0; q < EVB; ++q)   // written through: the owner is on another CU, possibly another XCD
                if (need[q])   // (as a 64-bit integer: an atomic store of a double may become a compare-and-swap loop, which never ends on the NaN it replaces)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&uc[(uint32_t)s * d.Dcm + dip_index(c, ha[q], hb[q])]), (unsigned long long)__double_as_longlong(out[q]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// every workgroup of the chain, after the table of iteration `it` has arrived: take units until there are none left.  Returns false when the chain was aborted.
__device__ static __noinline__ void noise_help(Env env, const NoiseChainCtl *nc) {
    const TileDesc *tiles = uniform_ptr(env.tiles);
    uint8_t *pool = uniform_ptr(env.pool);
    const GParams BT_CAS &P = env_params(env);
    const HelpItem *items = (const HelpItem *)nc->help_items;
    uint32_t BT_LAS *hw = nc_help_words(nc);
    const NcLanes L = nc_lanes();
    const uint32_t S = nc->S, n_units = nc->help_units;
    // (The wave barriers: in gibbs_chain_kernel a workgroup is ONE wavefront and the compiler drops __syncthreads' s_barrier — and with it the only
    // convergent operation of this loop; it then gave thread 0 (which fetches) and the other lanes (which only read what it fetched) loops of their own,
    // and the other lanes' loop, run first, never ended: they re-read the unit thread 0 was never let to replace.  A wave barrier emits no instruction but
    // keeps the lanes of a wavefront together across it.)
    while (true) {
        if (threadIdx.x == 0) hw[0] = __hip_atomic_fetch_add(nc->help_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_wave_barrier();
        __syncthreads();
        const uint32_t w = __builtin_amdgcn_readfirstlane(hw[0]);   // (one value per wavefront, whatever the lanes' view)
        __syncthreads();
        nc_phase(nc, 0x20000000u | w);
        if (w >= n_units) break;
        const HelpItem it = items[w / S];
        noise_help_unit(tiles, pool, P, nc, it, w % S, L);
        nc_wait_vm();      // the unit's sums are where every XCD reads from
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&nc->help_done[it.tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_wave_barrier();
    }
}

// the owner of a tile with large tables, before the sweep of iteration `it`: every unit of its tile has been computed.  Returns false when the chain was aborted.
__device__ static __noinline__ bool noise_help_wait(const NoiseChainCtl *nc, uint32_t tile, uint32_t it) {
    const uint32_t units = nc->tile_units[tile];
    if (units == 0) return true;
    nc_phase(nc, 0x30000000u | it);
    uint32_t BT_LAS *flag = nc_bins(nc) + nc->S * NC_BINS;
    if (threadIdx.x == 0) {
        const uint32_t want = units * (it - nc->it_begin);
        const unsigned long long t0 = wall_clock64();
        uint32_t v = __hip_atomic_load(&nc->help_done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        while (v < want && ok) {
            __builtin_amdgcn_s_sleep(4);
            v = __hip_atomic_load(&nc->help_done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < want && (unsigned long long)wall_clock64() - t0 > nc->timeout_ticks) {
                nc_abort(nc);
                ok = false;
            }
        }
        *flag = ok ? 0u : 1u;
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    const bool ok = *flag == 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the helpers' sums
    __syncthreads();
    return ok;
}

}  // namespace bt
