// A whole chain of a noise driver in ONE resident launch (round 5).
//
//   InferenceEngine::sampleGenotypesCallback + sampleNoiseParameters, one iteration   src/bayesTyper/InferenceEngine.cpp:77-98
//   VariantClusterGenotyper::getNoiseCounts / clearCache                              src/bayesTyper/VariantClusterGenotyper.cpp:131-138,757-779
//
// An iteration of the noise drivers is: one sweep of every group, the noise-count histogram of all groups (S x 256 bins), the caches
// cleared, ONE gamma draw per sample on the host (libstdc++'s generator and glibc's log / exp / lgamma: bit for bit the reference's), the
// rebuilt Poisson table back to the groups.  Launched per iteration (round 4) that was a sweep kernel + a tally kernel + a memset + two
// copies + a stream synchronisation: 0.62 ms per iteration at chr20 size for 44 us of sweep, because every launch reloads and stores the
// groups' generator state for ONE sweep.  Here the tiles of a chain stay resident — one workgroup per tile, sampler state in registers /
// LDS across iterations — and the per-iteration exchange goes through a mailbox in host-visible (pinned, fine-grained) memory:
//
//   every workgroup:   [wait for table `it`]  sweep  ->  tally its groups' noise counts (LDS bins for counts < 16, the rest straight to the
//                      device histogram)  ->  clear caches  ->  add its bins to the device histogram  ->  arrive (one device-scope atomic)
//   the LAST arriver:  device histogram -> host mailbox, zeroed; hist_seq = it + 1 (system scope);  [the host reduces over the ranks, draws the
//                      rates, rebuilds the table, writes it to the mailbox, table_seq = it + 1];  mailbox table -> the sampler's device table
//                      (write-through stores, acknowledged); table_seq copies on the device = it + 1
//   everyone else:     polls its copy of table_seq (one of NC_SEQ_COPIES words, a cache line apart), then an agent-scope acquire
//
// All workgroups of all launch classes must be resident at the same time (they wait for each other): the host checks that before it takes
// this path (bt_gibbs_noise_chain_begin) and every wait has a deadline after which the chain is aborted with an error instead of hanging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bt {

constexpr uint32_t NC_BINS = 16;          // counts below this are tallied in LDS
constexpr uint32_t NC_SEQ_COPIES = 32;    // copies of the device-side table sequence word ...
constexpr uint32_t NC_SEQ_STRIDE = 64;    // ... this many 32-bit words apart (256 bytes: separate channels)
constexpr uint32_t NC_ABORT = 0xFFFFFFFFu;

struct NoiseChainCtl {   // device memory, one per launch class (the pointers are shared by the classes of a sampler)
    unsigned long long *hist;          // device [S * 256]
    uint32_t *arrived;                 // device: workgroups that finished an iteration, counted over the whole chain
    uint32_t *table_seq;               // device [NC_SEQ_COPIES * NC_SEQ_STRIDE]: tables 1 .. table_seq have been handed to the device
    uint32_t *abort_flag;              // device: a deadline passed somewhere (1), or the roll call at the start of the launch was incomplete (2: no sampler state was touched)
    uint32_t *rollcall;                // device: workgroups of the launch that have started (nc_begin)
    unsigned long long rollcall_ticks; // wall_clock64() ticks the roll call may last
    double *lut_n;                     // device [S * 256]: the sampler's noise table (GParams::lut_n)
    unsigned long long *h_hist;        // host mailbox [S * 256]
    double *h_table;                   // host mailbox [S * 256]
    uint32_t *h_hist_seq;              // host mailbox: histograms 1 .. h_hist_seq have been published (NC_ABORT: aborted)
    uint32_t *h_table_seq;             // host mailbox: tables 1 .. h_table_seq are / were in h_table (NC_ABORT: the host gave up)
    uint32_t total_wgs;                // workgroups of all classes
    uint32_t bins_off;                 // LDS byte offset of this class's bins: [S * NC_BINS] u32 + one flag word
    uint32_t n_iterations, first_collect;
    uint32_t S;
    uint32_t lds_cap;                  // tiles whose hot arrays need more LDS than this keep them in HBM for the chain
    unsigned long long timeout_ticks;  // wall_clock64() ticks (100 MHz) a single wait may last
    uint32_t debug_flags;
    // the help phase (bt_noise_help.hpp): work units = help_items x S, taken off help_next (reset by the workgroup that hands a table over), counted per tile in help_done
    const void *help_items;
    uint32_t *help_next, *help_done;
    const uint32_t *tile_units;        // per tile: units of one iteration (0: no large table)
    uint32_t help_units;
    uint32_t num_tiles;                // workgroups [num_tiles, total_wgs) of the launch have no tile: they only take part in the help phase

    uint32_t it_begin;                 // first iteration of the chain that runs in the resident launch (1: iteration 0 ran as ordinary launches, bt_gibbs_noise_chain_step)
    //        // experiments (BT_NOISE_CHAIN_DEBUG_FLAGS): 1 = no acquire after the wait (wrong results: timing only)
    uint32_t *h_phase;                 // debugging (BT_NOISE_CHAIN_DEBUG_FLAGS & 2): per workgroup, where it is (host-visible), or null
    unsigned long long *busy;          // profiling (BT_NOISE_CHAIN_PROF): per workgroup, ticks between the end of its wait and its arrival, summed over the iterations; or null
};

#if defined(__HIP_DEVICE_COMPILE__)
#define NC_LAS __attribute__((address_space(3)))
#else
#define NC_LAS
#endif
extern __shared__ __attribute__((aligned(16))) uint8_t bt_lds_raw[];

__device__ inline uint32_t NC_LAS *nc_bins(const NoiseChainCtl *ctl) { return (uint32_t NC_LAS *)(bt_lds_raw + ctl->bins_off); }

__device__ inline void nc_wait_vm() {   // every memory operation of this lane issued so far has been acknowledged (gfx9: loads, stores and atomics count in vmcnt)
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// The lanes of a workgroup that share out its element-wise loops: the lanes of wavefront 0 that are still in the kernel (lanes of a tile without a
// group leave the kernel at once, and so do the wavefronts a narrow tile does not use; wavefront 0 always holds the tile's first group).
struct NcLanes {
    uint32_t rank, count;
    bool on;
};
__device__ inline NcLanes nc_lanes() {
    NcLanes r;
    const unsigned long long m = __ballot(1);
    r.on = threadIdx.x < 64u;
    r.count = (uint32_t)__popcll(m);
    r.rank = (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
    return r;
}

__device__ inline void nc_phase(const NoiseChainCtl *ctl, uint32_t code) {
    if (ctl->h_phase) __hip_atomic_store(&ctl->h_phase[blockIdx.x * 64u + (threadIdx.x & 63u)], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one noise count of sample s
__device__ inline void nc_tally(const NoiseChainCtl *ctl, uint32_t NC_LAS *bins, uint32_t s, uint32_t cnt) {
    if (cnt < NC_BINS) __hip_atomic_fetch_add(&bins[s * NC_BINS + cnt], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&ctl->hist[s * 256u + cnt], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void nc_abort(const NoiseChainCtl *ctl, uint32_t code = 1u) {
    __hip_atomic_store(ctl->abort_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t c = 0; c < NC_SEQ_COPIES; ++c) __hip_atomic_store(&ctl->table_seq[c * NC_SEQ_STRIDE], NC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctl->h_hist_seq, NC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Start of the workgroup's part of a chain: every thread of the workgroup calls it (no thread has left the kernel).
// ROLL CALL.  The workgroups of a chain wait for each other every iteration, so all of them must be resident at once.  The host checks that against the
// whole GPU's capacity (bt_gibbs_noise_chain_begin) — but a CU mask, another process or a second rank on the same GPU take slots the check cannot see.  So
// before any sampler state is touched every workgroup reports in and waits until all have: a launch whose workgroups are not co-resident fails HERE, within
// `rollcall_ticks` (a resident launch is complete within milliseconds), with abort code 2, and the host runs the chain launch by launch instead
// (bt_gibbs_noise_chain_step).  Returns false when the chain was aborted: the caller runs no sweep.
__device__ static __noinline__ bool nc_begin(const NoiseChainCtl *ctl) {
    uint32_t NC_LAS *bins = nc_bins(ctl);
    uint32_t NC_LAS *flag = bins + ctl->S * NC_BINS;
    const NcLanes L = nc_lanes();
    if (L.on)
        for (uint32_t i = L.rank; i <= ctl->S * NC_BINS; i += L.count) bins[i] = 0;
    if (ctl->busy && threadIdx.x == 0) *(unsigned long long NC_LAS *)(bins + ((ctl->S * NC_BINS + 2u) & ~1u)) = (unsigned long long)wall_clock64();
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctl->rollcall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        uint32_t bad = 0;
        while (__hip_atomic_load(ctl->rollcall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ctl->total_wgs) {
            if (__hip_atomic_load(ctl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                bad = 1;
                break;
            }
            if (wall_clock64() - t0 > ctl->rollcall_ticks) {
                nc_abort(ctl, 2u);
                bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(20);
        }
        if (!bad && __hip_atomic_load(ctl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) bad = 1;   // (someone gave up while the last ones reported in)
        *flag = bad;
    }
    __builtin_amdgcn_wave_barrier();   // (keeps a one-wavefront workgroup's lanes together where its s_barrier is dropped, see bt_noise_help.hpp)
    __syncthreads();
    const bool ok = *flag == 0;
    __syncthreads();
    if (threadIdx.x == 0) *flag = 0;
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    return ok;
}

// Before iteration `it` >= 1: wait until the table drawn from iteration it - 1's counts is the sampler's.  Every thread of the workgroup calls it.
// Returns false when the chain was aborted.
__device__ static __noinline__ bool nc_wait_table(const NoiseChainCtl *ctl, uint32_t it) {
    uint32_t NC_LAS *bins = nc_bins(ctl);
    uint32_t NC_LAS *flag = bins + ctl->S * NC_BINS;
    if (threadIdx.x == 0) {
        const uint32_t *w = &ctl->table_seq[(blockIdx.x % NC_SEQ_COPIES) * NC_SEQ_STRIDE];
        const unsigned long long t0 = wall_clock64();
        uint32_t v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (v < it) {
            __builtin_amdgcn_s_sleep(20);
            v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < it && wall_clock64() - t0 > ctl->timeout_ticks) {
                nc_abort(ctl);
                v = NC_ABORT;
            }
        }
        *flag = v == NC_ABORT ? 1u : 0u;
        if (ctl->busy) *(unsigned long long NC_LAS *)(bins + ((ctl->S * NC_BINS + 2u) & ~1u)) = (unsigned long long)wall_clock64();
    }
    __builtin_amdgcn_wave_barrier();   // (keeps a one-wavefront workgroup's lanes together where its s_barrier is dropped, see bt_noise_help.hpp)
    __syncthreads();
    const bool ok = *flag == 0;
    if (!(ctl->debug_flags & 1u)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the table (and nothing stale of it in this CU's L1 / this XCD's L2)
    __syncthreads();                                     // (the flag word is rewritten by the next call)
    return ok;
}

// End of iteration `it`: the workgroup's bins join the device histogram; the last workgroup of the chain to arrive hands the histogram to the host
// and fetches the next table.  Every thread of the workgroup calls it.  Returns false when the chain was aborted.
__device__ static __noinline__ bool nc_iteration_end(const NoiseChainCtl *ctl, uint32_t it) {
    uint32_t NC_LAS *bins = nc_bins(ctl);
    uint32_t NC_LAS *flag = bins + ctl->S * NC_BINS;
    const uint32_t S = ctl->S;
    const NcLanes L = nc_lanes();
    nc_wait_vm();      // this lane's direct adds to the device histogram
    __syncthreads();   // ... and every lane's LDS tallies
    if (L.on)
        for (uint32_t i = L.rank; i < S * NC_BINS; i += L.count) {
            const uint32_t n = bins[i];
            if (n) {
                __hip_atomic_fetch_add(&ctl->hist[(i / NC_BINS) * 256u + (i % NC_BINS)], (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bins[i] = 0;
            }
        }
    nc_wait_vm();      // the adds have been performed at the device's coherence point
    __syncthreads();
    if (threadIdx.x == 0) {
        if (ctl->busy) ctl->busy[blockIdx.x] += (unsigned long long)wall_clock64() - *(unsigned long long NC_LAS *)(bins + ((S * NC_BINS + 2u) & ~1u));
        const uint32_t old = __hip_atomic_fetch_add(ctl->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == (it + 1u - ctl->it_begin) * ctl->total_wgs - 1u ? 1u : 0u;
    }
    __builtin_amdgcn_wave_barrier();   // (keeps a one-wavefront workgroup's lanes together where its s_barrier is dropped, see bt_noise_help.hpp)
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();
    if (!last) return true;
    // ---- the last arriver: every workgroup's counts of this iteration are in ctl->hist ----
    const uint32_t nb = S * 256u;
    if (L.on)
        for (uint32_t i = L.rank; i < nb; i += L.count) {
            const unsigned long long v = __hip_atomic_load(&ctl->hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->h_hist[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v) __hip_atomic_store(&ctl->hist[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the histogram is in host memory before the sequence word
    nc_wait_vm();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ctl->h_hist_seq, it + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (it + 1u >= ctl->n_iterations) return true;   // the chain's last iteration: no table follows
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        uint32_t v = __hip_atomic_load(ctl->h_table_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        while (v < it + 1u) {
            __builtin_amdgcn_s_sleep(8);
            v = __hip_atomic_load(ctl->h_table_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v < it + 1u && wall_clock64() - t0 > ctl->timeout_ticks) v = NC_ABORT;
        }
        if (v == NC_ABORT) nc_abort(ctl);
        *flag = v == NC_ABORT ? 1u : 0u;
    }
    __builtin_amdgcn_wave_barrier();   // (keeps a one-wavefront workgroup's lanes together where its s_barrier is dropped, see bt_noise_help.hpp)
    __syncthreads();
    const bool ok = *flag == 0;
    __syncthreads();
    if (!ok) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: the table behind the sequence word
    if (L.on)
        for (uint32_t i = L.rank; i < nb; i += L.count) {
            // (as 64-bit integers: what an atomic access to a double compiles to is the compiler's choice)
            const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&ctl->h_table[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(&ctl->lut_n[i]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written through
        }
    if (threadIdx.x == 0 && ctl->help_units) __hip_atomic_store(ctl->help_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nc_wait_vm();      // acknowledged: the table is where every XCD reads from
    __syncthreads();
    if (L.on)
        for (uint32_t c = L.rank; c < NC_SEQ_COPIES; c += L.count)
            __hip_atomic_store(&ctl->table_seq[c * NC_SEQ_STRIDE], it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
#undef NC_LAS

}  // namespace bt
