// libbtgpu: k-mer count table in HBM + the KMC count-table scan ("k-mer matches/sec").
//
//   bt_table_*            <- KmerCountsHash / ObservedKmerCountsHash<N> (KmerHash.hpp:73-87,
//                            KmerHash.cpp:202-254) over KmerCounts (KmerCounts.cpp:40-223)
//   bt_table_count_intercluster <- KmerCounter::countInterclusterKmersCallback (KmerCounter.cpp:291-338)
//   bt_table_classify_batch     <- VariantClusterGraph::classifyPathKmers, table half (VariantClusterGraph.cpp:902-938)
//   bt_kmc_scan_*         <- KmerCounter::parseSampleKmers(+CallBack) (KmerCounter.cpp:388-524),
//                            record layout of CKMCFile::ReadNextKmer (kmc_file.cpp:428-494)
//
// The reference keeps a two-level map (16.7 M std::vector leaves + a mutex each).  Here the table is
// one open-addressing array in HBM (structure of arrays, linear probing, power-of-two capacity):
// per-key contents are identical, iteration order is free.
#include "bt_internal.hpp"

#include <algorithm>
#include <chrono>
#include <fcntl.h>
#include <unistd.h>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "bt_rng_device.hpp"

using namespace bt;

namespace {

constexpr unsigned BLOCK = 256;
constexpr uint32_t ST_EMPTY = 0, ST_BUSY = 1, ST_READY = 2;

__device__ inline uint64_t mix64(uint64_t x) {   // murmur3 finaliser
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__device__ inline uint64_t table_home(Kmer a, const TableView &t) { return mix64(a.lo ^ mix64(a.hi + 0x9e3779b97f4a7c15ULL)) & t.mask; }

// State and key of one slot in one burst: the (state, meta) word pair first, then the key words, issued back to back.  A slot's key words
// are written exactly once (EMPTY -> BUSY -> READY, no deletion; a cleared / fresh table holds zeros), each with one 64-bit store, and the
// writer stores READY only after both key stores have been ACKNOWLEDGED (s_waitcnt vmcnt(0) between them, publish_slot below).  The three
// loads of a look are issued in order but — a 48-byte slot's words may straddle a 64 / 128 / 256-byte boundary, i.e. sit in different
// channels — are not necessarily SERVED in order, so a look that reads READY may still carry a stale key word, and a stale word is
// always zero.  Hence: "READY and equal" is taken as a match from the burst only when neither word of the wanted key is zero (a zero
// word that compares equal could be the stale one: the all-A k-mer, or a k-mer ending in 23 A's); every other READY case — different
// key, or equal with a zero word — is decided by a second look at the key behind an acquire fence (now certainly served after the
// state was seen READY).  So a key is never missed, never matched to another key's slot, and never inserted twice.
struct SlotLook {
    uint32_t st, cw;
    uint64_t lo, hi;
};
constexpr uint32_t NO_COUNT_WORD = 0xFFFFFFFFu;
// cword: index of a count word of the slot to fetch in the same burst (the caller's saturating add then starts from it instead of loading it), or NO_COUNT_WORD
__device__ inline SlotLook slot_look(const TableView &t, uint64_t idx, uint32_t cword = NO_COUNT_WORD) {
    uint32_t *s = t.slot(idx);
    SlotLook p;
    const uint64_t sm = __hip_atomic_load(reinterpret_cast<uint64_t *>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    p.lo = __hip_atomic_load(reinterpret_cast<uint64_t *>(s + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.hi = __hip_atomic_load(reinterpret_cast<uint64_t *>(s + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.cw = cword != NO_COUNT_WORD ? __hip_atomic_load(s + 6 + cword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    p.st = (uint32_t)sm;
    return p;
}
__device__ inline bool burst_match(const SlotLook &p, Kmer a) {   // p.st == ST_READY: is the burst alone proof that the slot holds a?
    return p.lo == a.lo && p.hi == a.hi && a.lo != 0 && a.hi != 0;
}
__device__ inline bool slot_holds_other_key(const TableView &t, uint64_t idx, const SlotLook &p, Kmer a) {   // p.st == ST_READY, !burst_match(p, a)
    (void)p;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // s_waitcnt vmcnt(0) + buffer_inv sc1: the second look is served after the READY observation
    const uint64_t lo = __hip_atomic_load(t.key_lo(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(t.key_hi(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return lo != a.lo || hi != a.hi;
}

// The lane that won a slot (state BUSY) writes the key and publishes it.  The key words are agent-scope atomic stores (global_store ... sc1):
// written through to the point the other XCDs read from.  READY must not overtake them, and stores to different channels are not ordered
// by issue order, so the lane WAITS until both key stores have been acknowledged — an explicit `s_waitcnt vmcnt(0)`: on gfx9 stores count
// in vmcnt, and a workgroup-scope release fence (round 4) emits no instruction at all — and only then stores the state the same way.
// A full agent-scope release store (flags & 1, BT_TABLE_RELEASE_PUBLISH) would in addition write back every dirty line of this XCD's
// L2 (buffer_wbl2) once per inserted key, which nothing here needs: the slot's words are the only data the reader relies on.
__device__ inline void publish_slot(const TableView &t, uint64_t idx, Kmer a) {
    __hip_atomic_store(t.key_lo(idx), a.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(t.key_hi(idx), a.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t.flags & 1u) {
        __hip_atomic_store(t.state(idx), ST_READY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) expcnt(7) lgkmcnt(15): gfx9 encoding, vmcnt = bits [3:0] + [15:14]
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __hip_atomic_store(t.state(idx), ST_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// findKmer: slot or -1.  A BUSY slot is polled again — the loop has a single exit, no lane leaves it from inside.
__device__ inline int64_t table_find(const TableView &t, Kmer a) {
    uint64_t idx = table_home(a, t);
    int64_t result = -1;
    bool done = false;
    uint64_t probes = 0;
    while (!done) {
        const SlotLook p = slot_look(t, idx);
        if (p.st == ST_EMPTY) {
            done = true;
        } else if (p.st == ST_READY) {
            if (burst_match(p, a) || !slot_holds_other_key(t, idx, p, a)) {
                result = (int64_t)idx;
                done = true;
            } else {
                idx = (idx + 1) & t.mask;
                done = ++probes > t.mask;
            }
        }
        // ST_BUSY: another lane is publishing this slot; poll it again (the writer never waits)
    }
    return result;
}

// addKmer: slot of the (possibly new) key, -1 if the table is full.  The lane that wins a slot publishes it INSIDE the loop body
// (key words, then the READY state with release semantics) before the loop's exit condition is evaluated, so lanes of the same
// wavefront that poll that slot always see it published on a later iteration.
// cword / cw_seen: a count word to fetch along (slot_look) and its value as seen — 0 for a key this lane has just inserted; a hint for sat_add_byte_from.
__device__ inline int64_t table_find_or_insert(const TableView &t, Kmer a, uint32_t cword = NO_COUNT_WORD, uint32_t *cw_seen = nullptr) {
    uint64_t idx = table_home(a, t);
    int64_t result = -1;
    bool done = false;
    uint64_t probes = 0;
    uint32_t seen = 0;
    while (!done) {
        const SlotLook p = slot_look(t, idx, cword);
        if (p.st == ST_EMPTY) {
            if (atomicCAS(t.state(idx), ST_EMPTY, ST_BUSY) == ST_EMPTY) {
                publish_slot(t, idx, a);
                result = (int64_t)idx;
                done = true;
            }
            // lost the race: the slot is BUSY or READY now; look at it again
        } else if (p.st == ST_READY) {
            if (burst_match(p, a) || !slot_holds_other_key(t, idx, p, a)) {
                result = (int64_t)idx;
                seen = p.cw;
                done = true;
            } else {
                idx = (idx + 1) & t.mask;
                if (++probes > t.mask) {
                    atomicExch(t.overflow, 1u);
                    done = true;
                }
            }
        }
        // ST_BUSY: poll again
    }
    // (no key counter is kept: millions of inserts would serialise on that one word; bt_table_status counts the READY slots)
    if (cw_seen) *cw_seen = seen;
    return result;
}

// saturating u8 add on byte `byte_idx` of a word array (ObservedKmerCounts::addSampleCount,
// KmerCounts.cpp:161-171; KmerCounts::updateMultiplicity :178-188)
__device__ inline void sat_add_byte(uint32_t *words, uint64_t byte_idx, uint32_t add) {
    uint32_t *w = &words[byte_idx >> 2];
    const unsigned sh = (unsigned)(byte_idx & 3u) * 8u;
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        uint32_t cur = (old >> sh) & 0xFFu;
        uint32_t nv = cur + add;
        if (nv > 255u) nv = 255u;
        uint32_t desired = (old & ~(0xFFu << sh)) | (nv << sh);
        if (desired == old) return;
        uint32_t prev = atomicCAS(w, old, desired);
        if (prev == old) return;
        old = prev;
    }
}

// the same, starting from a value of the word the caller has already seen (a stale value only costs one more round of the loop)
__device__ inline void sat_add_byte_from(uint32_t *words, uint64_t byte_idx, uint32_t add, uint32_t old) {
    uint32_t *w = &words[byte_idx >> 2];
    const unsigned sh = (unsigned)(byte_idx & 3u) * 8u;
    while (true) {
        uint32_t cur = (old >> sh) & 0xFFu;
        uint32_t nv = cur + add;
        if (nv > 255u) nv = 255u;
        uint32_t desired = (old & ~(0xFFu << sh)) | (nv << sh);
        if (desired == old) {   // nothing to add to this value: make sure the value is current before leaving
            const uint32_t now = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (now == old) return;
            old = now;
            continue;
        }
        uint32_t prev = atomicCAS(w, old, desired);
        if (prev == old) return;
        old = prev;
    }
}

// read-modify-write of one slot's meta word with a pure function of the old value
template <typename F>
__device__ inline uint32_t meta_update(uint32_t *w, F f) {
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        uint32_t desired = f(old);
        if (desired == old) return desired;
        uint32_t prev = atomicCAS(w, old, desired);
        if (prev == old) return desired;
        old = prev;
    }
}

__device__ inline uint32_t sat8(uint32_t cur, uint32_t add) {
    uint32_t s = cur + add;
    return s > 255u ? 255u : s;
}

// KmerCounts::addInterclusterMultiplicity (KmerCounts.cpp:98-118)
__device__ inline uint32_t meta_add_intercluster(uint32_t m, bool is_decoy, uint32_t fem, uint32_t male) {
    uint32_t flags = m & 0xFFu, maxhap = (m >> 8) & 0xFFu, f = (m >> 16) & 0xFFu, ml = (m >> 24) & 0xFFu;
    maxhap = sat8(maxhap, 1);
    if (maxhap > 127u) flags |= BT_KC_MAX_MULTIPLICITY;
    if (is_decoy) flags |= BT_KC_DECOY_OCC;
    else {
        f = sat8(f, fem);
        ml = sat8(ml, male);
    }
    return flags | (maxhap << 8) | (f << 16) | (ml << 24);
}

// KmerCounts::addClusterMultiplicity (KmerCounts.cpp:137-159)
__device__ inline uint32_t meta_add_cluster(uint32_t m, uint32_t mult, bool is_multigroup) {
    uint32_t flags = m & 0xFFu, maxhap = (m >> 8) & 0xFFu;
    if (flags & BT_KC_CLUSTER_OCC) flags |= BT_KC_MULTICLUSTER_OCC;
    flags |= BT_KC_CLUSTER_OCC;
    if (is_multigroup) flags |= BT_KC_MULTIGROUP_OCC;
    maxhap = sat8(maxhap, mult);
    if (maxhap > 127u) flags |= BT_KC_MAX_MULTIPLICITY;
    return (m & 0xFFFF0000u) | flags | (maxhap << 8);
}

__global__ __launch_bounds__(BLOCK) void table_insert_kernel(TableView t, const uint64_t *__restrict__ kmers, uint64_t n, int mark_parameter) {
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        Kmer a{kmers[2 * i], kmers[2 * i + 1]};
        int64_t slot = table_find_or_insert(t, a);
        if (slot >= 0 && mark_parameter) atomicOr(t.meta(slot), (uint32_t)BT_KC_PARAMETER);
    }
}

__global__ __launch_bounds__(BLOCK) void table_find_kernel(TableView t, const uint64_t *__restrict__ kmers, uint64_t n, int64_t *__restrict__ slots) {
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        Kmer a{kmers[2 * i], kmers[2 * i + 1]};
        slots[i] = table_find(t, a);
    }
}

// bt_table_status: number of stored keys = READY slots
__global__ __launch_bounds__(BLOCK) void table_count_kernel(TableView t, uint64_t capacity, unsigned long long *__restrict__ out) {
    unsigned long long mine = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK) mine += *t.state(i) == ST_READY ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(out, mine);
}

// bt_table_reserve: every stored record of `src` moves to its slot in the (larger, empty) table `dst`
__global__ __launch_bounds__(BLOCK) void table_rehash_kernel(TableView src, uint64_t src_capacity, TableView dst) {
    const uint32_t words = src.spad / 4u;
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < src_capacity; i += (uint64_t)gridDim.x * BLOCK) {
        if (*src.state(i) != ST_READY) continue;
        const int64_t slot = table_find_or_insert(dst, Kmer{*src.key_lo(i), *src.key_hi(i)});
        if (slot < 0) continue;   // cannot happen: dst is larger than src
        *dst.meta(slot) = *src.meta(i);
        for (uint32_t w = 0; w < words; ++w) dst.counts(slot)[w] = src.counts(i)[w];
    }
}

// ---------------------------------------------------------------------------------------------
// intercluster scan: sequence tile in LDS -> canonical k-mer per position -> path-Bloom test ->
// table insert + addInterclusterMultiplicity.
// ---------------------------------------------------------------------------------------------
constexpr unsigned SEQ_TILE = 1024;

// ---- countInterclusterParameterKmers (KmerCounter.cpp:161-250) ----
struct ParamRegion {
    uint64_t start, len;     // relative to the span handed to the kernels
    uint32_t seed;
    uint32_t is_decoy;
};
__device__ inline int64_t region_of(const ParamRegion *__restrict__ regs, uint32_t R, uint64_t pos) {   // regions sorted by start, disjoint
    uint32_t lo = 0, hi = R;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (regs[mid].start <= pos) lo = mid + 1;
        else hi = mid;
    }
    if (lo == 0) return -1;
    return pos - regs[lo - 1].start < regs[lo - 1].len ? (int64_t)(lo - 1) : -1;
}
// cand[pos] = 1 + region index parity is not needed: 1 when the window ending at pos lies inside one region and misses the path Bloom
__global__ __launch_bounds__(BLOCK) void param_cand_kernel(BloomView bloom, const ParamRegion *__restrict__ regs, uint32_t R, const uint64_t *__restrict__ kmers,
                                                            const uint8_t *__restrict__ valid, uint64_t n, uint32_t k, uint8_t *__restrict__ cand) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < n; pos += (uint64_t)gridDim.x * BLOCK) {
        uint8_t c = 0;
        if (valid[pos]) {
            const int64_t r = region_of(regs, R, pos);
            if (r >= 0 && pos - regs[r].start + 1 >= k) {   // kmer_pair.reset() at the region start
                const Kmer a{kmers[2 * pos], kmers[2 * pos + 1]};
                c = bloom_contains(nthash64(a, bloom.k), bloom) ? 0 : 1;
            }
        }
        cand[pos] = c;
    }
}
// one lane per non-decoy region: the region's Bernoulli draws, in window order (the only sequential part)
__global__ __launch_bounds__(64) void param_draw_kernel(const ParamRegion *__restrict__ regs, uint32_t r0, uint32_t r1, uint32_t *__restrict__ mt_states, double p,
                                                         uint8_t *__restrict__ cand) {
    const uint32_t r = r0 + blockIdx.x * 64 + threadIdx.x;
    if (r >= r1 || regs[r].is_decoy) return;
    uint32_t *st = mt_states + (size_t)(r - r0) * MT_WORDS;
    mt_seed(st, regs[r].seed);
    Mt rng = mt_open(st);
    for (uint64_t pos = regs[r].start, e = regs[r].start + regs[r].len; pos < e; ++pos)
        if (cand[pos]) cand[pos] = rng_bernoulli(rng, p) ? 2 : 0;   // 2 = accepted
}
__global__ __launch_bounds__(BLOCK) void param_insert_kernel(TableView t, const ParamRegion *__restrict__ regs, uint32_t R, const uint64_t *__restrict__ kmers,
                                                              const uint8_t *__restrict__ cand, uint64_t n) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < n; pos += (uint64_t)gridDim.x * BLOCK) {
        const uint8_t c = cand[pos];
        if (!c) continue;
        const int64_t r = region_of(regs, R, pos);
        const bool decoy = regs[r].is_decoy != 0;
        if (!decoy && c != 2) continue;
        const int64_t slot = table_find_or_insert(t, Kmer{kmers[2 * pos], kmers[2 * pos + 1]});
        if (slot >= 0) atomicOr(t.meta(slot), decoy ? (uint32_t)BT_KC_DECOY_OCC : (uint32_t)BT_KC_PARAMETER);
    }
}

// calculateKmerStats (KmerHash.cpp:256-340): class tallies + exact integer moments of the parameter k-mers' counts per
// (sample, intercluster multiplicity).  Per-workgroup LDS accumulation of the tallies; the (s, m) bins go straight to global
// atomics (parameter k-mers are a ~1e-3 fraction of the table and concentrate on a handful of bins per sample).
__global__ __launch_bounds__(BLOCK) void kmer_stats_kernel(TableView t, uint64_t capacity, uint32_t S, uint32_t gender_mask, unsigned long long *__restrict__ acc) {
    __shared__ unsigned int tally[7];
    if (threadIdx.x < 7) tally[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long *n = acc + 7, *sum = n + (size_t)S * 256, *sumsq = sum + (size_t)S * 256, *nonzero = sumsq + (size_t)S * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK) {
        if (*t.state(i) != ST_READY) continue;
        const uint32_t meta = *t.meta(i);
        const uint32_t flags = meta & 0xffu;
        atomicAdd(&tally[0], 1u);
        if (flags & BT_KC_CLUSTER_OCC) {
            if (flags & (BT_KC_DECOY_OCC | BT_KC_MAX_MULTIPLICITY | BT_KC_MULTIGROUP_OCC)) {   // isExcluded (KmerCounts.cpp:93-96)
                if (flags & BT_KC_DECOY_OCC) atomicAdd(&tally[3], 1u);
                else if (flags & BT_KC_MAX_MULTIPLICITY) atomicAdd(&tally[4], 1u);
                else atomicAdd(&tally[5], 1u);
            } else if (flags & BT_KC_MULTICLUSTER_OCC) atomicAdd(&tally[2], 1u);
            else atomicAdd(&tally[1], 1u);
        } else {
            atomicAdd(&tally[6], 1u);
            if (flags & BT_KC_PARAMETER) {
                const uint8_t *cnt = t.count_bytes(i);
                for (uint32_t s = 0; s < S; ++s) {
                    const uint32_t m = (gender_mask >> s) & 1u ? (meta >> 24) & 0xffu : (meta >> 16) & 0xffu;   // male : female
                    const unsigned long long c = cnt[s];
                    atomicAdd(&n[s * 256 + m], 1ull);
                    atomicAdd(&sum[s * 256 + m], c);
                    atomicAdd(&sumsq[s * 256 + m], c * c);
                    if (c) atomicAdd(&nonzero[s * 256 + m], 1ull);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 7 && tally[threadIdx.x]) atomicAdd(&acc[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

__global__ __launch_bounds__(BLOCK) void intercluster_kernel(TableView t, BloomView bloom, const char *__restrict__ seq, uint64_t len,
                                                             int is_decoy, uint32_t fem, uint32_t male) {
    __shared__ uint8_t codes[SEQ_TILE + 64];
    const unsigned k = t.k;
    const uint64_t num_tiles = (len + SEQ_TILE - 1) / SEQ_TILE;
    for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const uint64_t tile_start = tile * SEQ_TILE;
        const uint64_t halo = k - 1;
        for (unsigned j = threadIdx.x; j < SEQ_TILE + halo; j += BLOCK) {
            int64_t pos = (int64_t)tile_start - (int64_t)halo + (int64_t)j;
            uint8_t c = 0xFF;
            if (pos >= 0 && (uint64_t)pos < len) {
                int code = nt_code(seq[pos]);
                c = code < 0 ? 0xFF : (uint8_t)code;
            }
            codes[j] = c;
        }
        __syncthreads();
        for (unsigned j = threadIdx.x; j < SEQ_TILE; j += BLOCK) {
            uint64_t pos = tile_start + j;
            if (pos >= len) break;
            Kmer fw{0, 0};
            bool ok = true;
            for (unsigned i = 0; i < k; ++i) {
                uint8_t c = codes[j + i];
                ok = ok && (c != 0xFF);
                uint64_t v = (uint64_t)(c & 3u);
                if (i < 32u) fw.lo |= v << (2u * i);
                else fw.hi |= v << (2u * (i - 32u));
            }
            if (!ok) continue;
            Kmer can = kmer_canonical(fw, k);
            if (!bloom_contains(nthash64(can, k), bloom)) continue;
            int64_t slot = table_find_or_insert(t, can);
            if (slot < 0) continue;
            meta_update(t.meta(slot), [&](uint32_t m) { return meta_add_intercluster(m, is_decoy != 0, fem, male); });
        }
        __syncthreads();
    }
}

// the same over a list of tiles of many regions of one sequence: a tile is SEQ_TILE start positions of one region (positions absolute in `seq`)
struct RegionTile {
    uint64_t region_start, region_end, tile_start;   // [region_start, region_end): the region; tile_start: first k-mer start position of the tile
    uint32_t flags, pad;                              // is_decoy | female ploidy << 8 | male ploidy << 16
};
__global__ __launch_bounds__(BLOCK) void intercluster_regions_kernel(TableView t, BloomView bloom, const char *__restrict__ seq, const RegionTile *__restrict__ tiles, uint64_t num_tiles) {
    __shared__ uint8_t codes[SEQ_TILE + 64];
    const unsigned k = t.k;
    for (uint64_t ti = blockIdx.x; ti < num_tiles; ti += gridDim.x) {
        const RegionTile rt = tiles[ti];
        const bool is_decoy = (rt.flags & 0xFFu) != 0;
        const uint32_t fem = (rt.flags >> 8) & 0xFFu, male = (rt.flags >> 16) & 0xFFu;
        // the k-mers that START in [tile_start, tile_start + SEQ_TILE) and lie inside the region: nucleotides [tile_start, tile_start + SEQ_TILE + k - 1)
        for (unsigned j = threadIdx.x; j < SEQ_TILE + k - 1; j += BLOCK) {
            const uint64_t pos = rt.tile_start + j;
            uint8_t c = 0xFF;
            if (pos < rt.region_end) {
                const int code = nt_code(seq[pos]);
                c = code < 0 ? 0xFF : (uint8_t)code;
            }
            codes[j] = c;
        }
        __syncthreads();
        for (unsigned j = threadIdx.x; j < SEQ_TILE; j += BLOCK) {
            if (rt.tile_start + j + k > rt.region_end) break;
            Kmer fw{0, 0};
            bool ok = true;
            for (unsigned i = 0; i < k; ++i) {
                const uint8_t c = codes[j + i];
                ok = ok && (c != 0xFF);
                const uint64_t v = (uint64_t)(c & 3u);
                if (i < 32u) fw.lo |= v << (2u * i);
                else fw.hi |= v << (2u * (i - 32u));
            }
            if (!ok) continue;
            const Kmer can = kmer_canonical(fw, k);
            if (!bloom_contains(nthash64(can, k), bloom)) continue;
            const int64_t slot = table_find_or_insert(t, can);
            if (slot < 0) continue;
            meta_update(t.meta(slot), [&](uint32_t m) { return meta_add_intercluster(m, is_decoy, fem, male); });
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void classify_kernel(TableView t, BloomView mg_bloom, const uint64_t *__restrict__ kmers,
                                                         const uint8_t *__restrict__ mult, uint64_t n, uint8_t *__restrict__ excluded) {
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        Kmer a{kmers[2 * i], kmers[2 * i + 1]};
        const uint32_t m = mult[i];
        int64_t slot = (m > 127u) ? table_find_or_insert(t, a) : table_find(t, a);
        uint8_t ex = 0;
        if (slot >= 0) {
            const bool is_mg = bloom_contains(nthash64(a, t.k), mg_bloom);
            uint32_t nm = meta_update(t.meta(slot), [&](uint32_t old) { return meta_add_cluster(old, m, is_mg); });
            ex = (nm & (BT_KC_DECOY_OCC | BT_KC_MAX_MULTIPLICITY | BT_KC_MULTIGROUP_OCC)) ? 1 : 0;   // isExcluded, KmerCounts.cpp:93-96
        }
        if (excluded) excluded[i] = ex;
    }
}

// ---------------------------------------------------------------------------------------------
// KMC scan.  One record per lane.  A workgroup handles RECS consecutive records whose raw bytes
// it first copies to LDS with 16-byte loads (the 13-byte records are not naturally aligned).
// k-mer of record n = prefix(n) (p symbols, from the LUT) ++ suffix bytes (4 symbols per byte,
// first symbol in the top two bits) — kmc_file.cpp:437-474.
// ---------------------------------------------------------------------------------------------
constexpr unsigned KMC_HINT_SHIFT = 12;     // one hint per 4096 records
struct KmcView {
    const uint64_t *lut;     // 4^p + 1 entries
    const uint32_t *hint;    // hint[j] = prefix of record j << KMC_HINT_SHIFT (+ a terminal entry): a record's prefix is searched between two neighbouring hints
    uint64_t lut_entries;
    uint32_t k, p, counter_size, suffix_bytes, rec_size;
    uint32_t min_count, max_count;   // records whose counter lies outside are skipped (CKMCFile::ReadNextKmer, kmc_file.cpp:496-511)
};

constexpr unsigned KMC_RECS = 256;          // records per workgroup iteration (= BLOCK)
constexpr unsigned KMC_MAX_REC = 24;        // max record size in bytes (k<=64: 16 suffix + 4 counter)

// prefix of record n: largest j with lut[j] <= n  (skips empty prefixes exactly like
// ReadNextKmer's "while (buf[idx] == buf[idx+1]) idx++", kmc_file.cpp:439-445)
__device__ inline uint64_t kmc_prefix_of(const KmcView &v, uint64_t n) {
    // (a prefix holds thousands of records: between two hints there are one or two candidates, so the search below is zero or one step
    // instead of log2(4^p) dependent loads)
    uint64_t lo = v.hint[n >> KMC_HINT_SHIFT], hi = (uint64_t)v.hint[(n >> KMC_HINT_SHIFT) + 1] + 1;   // invariant: lut[lo] <= n < lut[hi]
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if (v.lut[mid] <= n) lo = mid;
        else hi = mid;
    }
    return lo;
}

// Prefix of record n of a block of consecutive records whose first and last prefixes are known (computed once per block): records are
// sorted by prefix and a prefix holds thousands of records, so nearly every block lies inside one prefix and no lane searches at all
__device__ inline uint64_t kmc_prefix_in(const KmcView &v, uint64_t n, uint64_t p_first, uint64_t p_last) {
    uint64_t lo = p_first, hi = p_last + 1;   // invariant: lut[lo] <= n < lut[hi]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (v.lut[mid] <= n) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ inline void kmc_decode(const KmcView &v, uint64_t prefix, const uint8_t *rec, Kmer &out, uint32_t &count) {
    // assemble the k symbols MSB-first into a 128-bit big number, then reverse the group order
    uint64_t bhi = 0, blo = 0;   // symbols s_0 .. s_{k-1}, s_0 most significant, right-aligned at bit 0
    auto push = [&](uint64_t bits, unsigned nbits) {
        bhi = (bhi << nbits) | (blo >> (64u - nbits));
        blo = (blo << nbits) | bits;
    };
    if (v.p) {
        const unsigned pb = 2u * v.p;   // <= 30 bits
        push(prefix & ((1ULL << pb) - 1ULL), pb);
    }
    for (unsigned b = 0; b < v.suffix_bytes; ++b) push((uint64_t)rec[b], 8u);
    // now symbol i (0-based from the left) sits at group (k-1-i); our packing wants symbol i at group i:
    // reverse all 64 groups, then shift down by (64-k) groups
    uint64_t rlo = rev2bit64(bhi), rhi = rev2bit64(blo);
    unsigned sh = 2u * (64u - v.k);
    if (sh == 0) { out.lo = rlo; out.hi = rhi; }
    else if (sh < 64u) { out.lo = (rlo >> sh) | (rhi << (64u - sh)); out.hi = rhi >> sh; }
    else if (sh == 64u) { out.lo = rhi; out.hi = 0; }
    else { out.lo = rhi >> (sh - 64u); out.hi = 0; }
    count = 0;
    for (unsigned b = 0; b < v.counter_size; ++b) count |= (uint32_t)rec[v.suffix_bytes + b] << (8u * b);
}

template <bool DECODE_ONLY>
__global__ __launch_bounds__(BLOCK) void kmc_scan_kernel(KmcView v, BloomView bloom, TableView t, uint32_t sample_idx,
                                                         const uint8_t *__restrict__ records, uint64_t first_record, uint64_t n,
                                                         unsigned long long *__restrict__ hit_count, uint64_t *__restrict__ out_kmers,
                                                         uint32_t *__restrict__ out_counts) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[KMC_RECS * KMC_MAX_REC + 32];
    __shared__ unsigned block_hits;
    __shared__ uint64_t block_prefix[2];
    const uint64_t num_chunks = (n + KMC_RECS - 1) / KMC_RECS;
    for (uint64_t chunk = blockIdx.x; chunk < num_chunks; chunk += gridDim.x) {
        const uint64_t rec0 = chunk * KMC_RECS;
        const unsigned nrec = (unsigned)((n - rec0) < KMC_RECS ? (n - rec0) : KMC_RECS);
        const uint64_t byte0 = rec0 * v.rec_size;
        const unsigned nbytes = nrec * v.rec_size;
        if (threadIdx.x == 0) block_hits = 0;
        if (threadIdx.x < 2) block_prefix[threadIdx.x] = kmc_prefix_of(v, first_record + rec0 + (threadIdx.x ? nrec - 1 : 0));
        // 16-byte aligned window that covers [byte0, byte0 + nbytes)
        const uint64_t a0 = byte0 & ~15ULL;
        const unsigned lead = (unsigned)(byte0 - a0);
        const unsigned nvec = (lead + nbytes + 15u) / 16u;
        const uint64_t total_bytes = n * (uint64_t)v.rec_size;
        for (unsigned j = threadIdx.x; j < nvec; j += BLOCK) {
            const uint64_t off = a0 + (uint64_t)j * 16u;
            if (off + 16u <= total_bytes) {
                *reinterpret_cast<uint4 *>(&stage[j * 16u]) = *reinterpret_cast<const uint4 *>(records + off);
            } else {
                for (unsigned q = 0; q < 16u; ++q) stage[j * 16u + q] = (off + q < total_bytes) ? records[off + q] : 0;
            }
        }
        __syncthreads();
        unsigned my_hit = 0;
        if (threadIdx.x < nrec) {
            const uint64_t ridx = rec0 + threadIdx.x;            // index inside this call
            const uint64_t gidx = first_record + ridx;           // index inside the database
            const uint8_t *rec = &stage[lead + threadIdx.x * v.rec_size];
            Kmer a;
            uint32_t count;
            kmc_decode(v, kmc_prefix_in(v, gidx, block_prefix[0], block_prefix[1]), rec, a, count);
            if (DECODE_ONLY) {
                out_kmers[2 * ridx] = a.lo;
                out_kmers[2 * ridx + 1] = a.hi;
                out_counts[ridx] = count;
            } else if (count >= v.min_count && count <= v.max_count && bloom_contains(nthash64(a, v.k), bloom)) {     // KmerCounter.cpp:412
                my_hit = 1;
                int64_t slot = table_find_or_insert(t, a);            // addKmer(kmer, false), :416
                if (slot >= 0) sat_add_byte(t.counts(slot), sample_idx, count > 255u ? 255u : count);   // :419
            }
        }
        if (!DECODE_ONLY && hit_count) {
            // one atomic per wavefront, then one per workgroup
            unsigned long long ballot = __ballot(my_hit);
            if ((threadIdx.x & 63u) == 0 && ballot) atomicAdd(&block_hits, (unsigned)__popcll(ballot));
            __syncthreads();
            if (threadIdx.x == 0 && block_hits) atomicAdd(hit_count, (unsigned long long)block_hits);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Scan against a ThreadedKmerBloom (65 536 sub-filters of ~1-40 KB each).  The direct kernel above pays two random 64-byte sectors of the
// (cache-exceeding) filter per record.  Here the records of a chunk are first grouped by the sub-filters they route to, so that the filter
// bytes a group of records probes stay in cache.  Rounds 2-3 also had a sorted form for sub-filters above 4 KB (route keys, a rocPRIM radix
// sort on the 16 route bits, one workgroup per sub-filter probing from LDS: 102 B of traffic per record); the partitioned form below now
// takes every sub-filter size — at 36 KB per sub-filter (a ten-sample path filter) the 256 sub-filters of a bucket no longer fit an XCD's L2
// but the Infinity Cache holds the eight buckets in flight: 2.08 against 1.66 x 10^10 records/s — and the library sort is gone.
// Same decisions as the direct kernel, so the table contents are identical.
// ---------------------------------------------------------------------------------------------
struct RouteRec {
    uint32_t h_lo, h_hi, idx;   // ntHash of the record's k-mer, record index inside the chunk
};

// ---------------------------------------------------------------------------------------------
// Partitioned scan.  ONE kernel hashes and writes every route record once, grouped by the upper 8 route bits, and the probe reads it once:
//   kmc_partition_kernel  — a workgroup takes slabs of 4096 records: every lane fetches its eight records' dwords (one contiguous run per wavefront); the
//                           ntHash is computed straight from the raw record bytes (a KMC suffix byte holds four symbols, first symbol in the top bits: one 256-entry
//                           LDS table per four symbols; the prefix's part once per slab) — the k-mer itself is never assembled here —, an LDS
//                           histogram over the 256 buckets, one reservation per bucket and slab in the bucket's region (atomicAdd on its cursor),
//                           a counting sort of the slab in the same LDS, then the runs (about 16 records = 192 bytes per bucket and slab) are
//                           written out coalesced.  Two workgroups per CU: one loads while the other sorts.  Round 4: a bucket's region is cut into
//                           eight stripes, stripe = workgroup index mod 8 = the XCD the workgroup runs on: the 16 384 reservations a chunk makes on
//                           a bucket's cursor (returning atomics on one address, served one after the other) become 2 048 per address, and a
//                           stripe's frontier lines are only ever written from one XCD's L2.  A stripe holds its expected share + 3 % + 512
//                           records; a record beyond it (hash routes are uniform: it does not happen outside the test that forces it) is probed
//                           against the filter in HBM on the spot.
//   kmc_probe_bucket_kernel — persistent workgroups; the ones resident on an XCD walk through the same buckets in the same order (bucket = 8 g + XCD),
//                           sharing a bucket's records between them: the 256 sub-filters of a bucket (0.5 MB at the WGS shape) stay in that XCD's L2
//                           while its records stream through.  Hits are queued in LDS and moved to the chunk's hit list with one global atomic
//                           per workgroup and chunk (round 3: one per short-lived workgroup, 32 768 on one address per chunk).
// 13 B record + 12 B written + 12 B read per record (+ the filter once per chunk).  Same decisions as the direct kernel.
// ---------------------------------------------------------------------------------------------
#ifndef BT_KMC_PSLAB
#define BT_KMC_PSLAB 4096
#endif
constexpr unsigned PBLOCK = 512, PSLAB = BT_KMC_PSLAB, PRPT = PSLAB / PBLOCK;
constexpr unsigned PSTRIPES = 8;
#ifndef BT_KMC_PU
#define BT_KMC_PU 4
#endif
constexpr unsigned PU = BT_KMC_PU;   // records per lane and iteration of the probe kernel

__device__ inline uint32_t route_bucket(uint64_t h, uint32_t bloom_k) { return (uint32_t)((nthash64_seeded(h, bloom_k, BT_ROUTE_SEED) & (uint64_t)(BT_NUM_SUB_BLOOMS - 1u)) >> 8); }

// dynamic LDS: the slab's route records in bucket order; then one byte per sorted record: its bucket  (the raw record bytes no longer pass through
// LDS: every lane fetches its records' dwords from global memory)
static size_t partition_main_bytes(uint32_t /*rec_size*/) { return ((size_t)PSLAB * sizeof(RouteRec) + 15) & ~(size_t)15; }
static size_t partition_lds_bytes(uint32_t rec_size) { return partition_main_bytes(rec_size) + PSLAB; }

// ntHash state after the p prefix symbols (most significant symbol first)
__device__ inline uint64_t kmc_prefix_hash(uint64_t prefix, uint32_t p) {
    uint64_t h = 0;
    for (uint32_t j = 0; j < p; ++j) h = rol64(h, 1) ^ nt_seed((unsigned)((prefix >> (2u * (p - 1u - j))) & 3u));
    return h;
}

__global__ __launch_bounds__(PBLOCK) void kmc_partition_kernel(KmcView v, BloomView bloom, const uint8_t *__restrict__ records, uint64_t first_record, uint64_t rec_offset, uint64_t n,
                                                               uint64_t n_total, uint32_t main_bytes, RouteRec *__restrict__ part, uint32_t cap, unsigned int *__restrict__ cursor,
                                                               uint32_t *__restrict__ hits, unsigned int *__restrict__ num_hits) {
    extern __shared__ __attribute__((aligned(16))) uint8_t part_lds[];
    RouteRec *sorted = reinterpret_cast<RouteRec *>(part_lds);   // [PSLAB], after the slab has been hashed
    uint8_t *sbucket = part_lds + main_bytes;                   // [PSLAB] bucket of sorted[i]
    __shared__ uint64_t block_prefix[2], block_phash[2];
    __shared__ uint64_t tab[256];
    __shared__ uint32_t hist[256], lofs[256], gbase[256], lcur[256], wave_tot[4];
    // four symbols of a suffix byte at a time: tab[b] = the Horner contribution of the symbols c0 c1 c2 c3 packed in byte b with c0 in the TOP two
    // bits (kmc_file.cpp:437-474), c0 hashed first, so that h <- rol(h, 4) ^ tab[b] equals four steps of h <- rol(h, 1) ^ seed[c]
    for (unsigned b = threadIdx.x; b < 256u; b += PBLOCK)
        tab[b] = rol64(nt_seed((b >> 6) & 3u), 3) ^ rol64(nt_seed((b >> 4) & 3u), 2) ^ rol64(nt_seed((b >> 2) & 3u), 1) ^ nt_seed(b & 3u);
    const uint32_t stripe = blockIdx.x % PSTRIPES;
    unsigned int *my_cursor = cursor + stripe * 256u;
    const uint64_t num_slabs = (n + PSLAB - 1) / PSLAB;
    const uint64_t total_bytes = n_total * (uint64_t)v.rec_size;
    for (uint64_t slab = blockIdx.x; slab < num_slabs; slab += gridDim.x) {
        const uint64_t slab0 = slab * PSLAB;
        const uint32_t slab_n = (uint32_t)((n - slab0) < PSLAB ? (n - slab0) : PSLAB);
        if (threadIdx.x < 256u) hist[threadIdx.x] = 0;
        if (threadIdx.x < 2) {
            const uint64_t pf = kmc_prefix_of(v, first_record + rec_offset + slab0 + (threadIdx.x ? slab_n - 1 : 0));
            block_prefix[threadIdx.x] = pf;
            block_phash[threadIdx.x] = kmc_prefix_hash(pf, v.p);
        }
        const uint64_t byte0 = (rec_offset + slab0) * v.rec_size;
        __syncthreads();
        const uint64_t p_first = block_prefix[0], p_last = block_prefix[1];
        // Every lane fetches ITS records straight from global memory (round 5; before, the slab's bytes went through LDS and every suffix byte was a
        // ds_read_u8 at a 13-byte lane stride: four-way bank conflicts on 13 of the 28 LDS instructions per record): the dwords from the 4-byte aligned
        // address below the record, PRPT records in flight per lane — a wavefront's 64 windows overlap into one contiguous 832-byte run —, then
        // v_alignbyte shifts the suffix to the front of the window.  (`records` is 16-byte aligned: bt_kmc_scan_run checks it.)
        constexpr unsigned NW = 4;   // suffix words (k - p <= 64 symbols = 16 bytes)
        uint32_t win[PRPT][NW + 1];
        const unsigned sb = v.suffix_bytes, nw = (sb + 3u) / 4u;
        // (only the slab at the very end of the buffer can hold a dword that reaches past it: that slab reads byte by byte)
        const bool at_end = byte0 + (uint64_t)slab_n * v.rec_size + 8u > total_bytes;
        if (!at_end) {
#pragma unroll
            for (unsigned j = 0; j < PRPT; ++j) {
                const uint32_t r = j * PBLOCK + threadIdx.x;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(records + ((byte0 + (uint64_t)r * v.rec_size) & ~3ULL));
#pragma unroll
                for (unsigned i = 0; i <= NW; ++i) win[j][i] = (r < slab_n && i <= nw) ? q[i] : 0u;
            }
        } else {
#pragma unroll
            for (unsigned j = 0; j < PRPT; ++j) {
                const uint32_t r = j * PBLOCK + threadIdx.x;
                const uint64_t w0 = (byte0 + (uint64_t)r * v.rec_size) & ~3ULL;
#pragma unroll
                for (unsigned i = 0; i <= NW; ++i) {
                    uint32_t x = 0;
                    for (unsigned z = 0; z < 4u; ++z)
                        if (r < slab_n && w0 + 4u * i + z < total_bytes) x |= (uint32_t)records[w0 + 4u * i + z] << (8u * z);
                    win[j][i] = x;
                }
            }
        }
        uint64_t hh[PRPT];
        uint32_t bk[PRPT / 4];
        uint32_t valid = 0;
#pragma unroll
        for (unsigned j = 0; j < PRPT / 4; ++j) bk[j] = 0;
#pragma unroll
        for (unsigned j = 0; j < PRPT; ++j) {
            hh[j] = 0;
            const uint32_t r = j * PBLOCK + threadIdx.x;
            if (r >= slab_n) continue;
            // the prefix part: nearly every slab lies inside one prefix or two
            uint64_t h;
            if (p_first == p_last) h = block_phash[0];
            else {
                const uint64_t pf = kmc_prefix_in(v, first_record + rec_offset + slab0 + r, p_first, p_last);
                h = pf == p_first ? block_phash[0] : (pf == p_last ? block_phash[1] : kmc_prefix_hash(pf, v.p));
            }
            const uint32_t sh = (uint32_t)((byte0 + (uint64_t)r * v.rec_size) & 3ULL);
            uint32_t sw[NW];
#pragma unroll
            for (unsigned i = 0; i < NW; ++i) sw[i] = __builtin_amdgcn_alignbyte(win[j][i + 1], win[j][i], sh);
#pragma unroll
            for (unsigned b = 0; b < 4u * NW; ++b)   // (k - p is a multiple of four: bt_kmc_scan_create)
                if (b < sb) h = rol64(h, 4) ^ tab[(sw[b >> 2] >> (8u * (b & 3u))) & 0xFFu];
            hh[j] = h;
            valid |= 1u << j;
            const uint32_t bucket = route_bucket(h, bloom.k);
            bk[j / 4] |= bucket << (8u * (j & 3u));
            atomicAdd(&hist[bucket], 1u);
        }
        __syncthreads();
        // exclusive scan of the 256 counts (four wavefronts of 64 bins), one reservation per bucket in its stripe's region
        if (threadIdx.x < 256u) {
            const uint32_t c = hist[threadIdx.x];
            uint32_t incl = c;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((int)(threadIdx.x & 63u) >= d) incl += o;
            }
            lofs[threadIdx.x] = incl - c;
            if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
            gbase[threadIdx.x] = c ? atomicAdd(&my_cursor[threadIdx.x], c) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < 256u) {
            uint32_t add = 0;
            for (unsigned wv = 0; wv < (threadIdx.x >> 6); ++wv) add += wave_tot[wv];
            lofs[threadIdx.x] += add;
            lcur[threadIdx.x] = lofs[threadIdx.x];
        }
        __syncthreads();
#pragma unroll
        for (unsigned j = 0; j < PRPT; ++j)
            if (valid & (1u << j)) {
                const uint64_t h = hh[j];
                const uint32_t bucket = (bk[j / 4] >> (8u * (j & 3u))) & 0xFFu;
                const uint32_t pos = atomicAdd(&lcur[bucket], 1u);
                sorted[pos] = RouteRec{(uint32_t)h, (uint32_t)(h >> 32), (uint32_t)(slab0 + j * PBLOCK + threadIdx.x)};
                sbucket[pos] = (uint8_t)bucket;
            }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < slab_n; i += PBLOCK) {
            const RouteRec r = sorted[i];
            const uint32_t b = sbucket[i];
            const uint32_t dest = gbase[b] + (i - lofs[b]);
            if (dest < cap) part[((uint64_t)b * PSTRIPES + stripe) * cap + dest] = r;
            else if (bloom_contains((uint64_t)r.h_lo | ((uint64_t)r.h_hi << 32), bloom)) hits[atomicAdd(num_hits, 1u)] = r.idx;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void kmc_probe_bucket_kernel(BloomView bloom, const RouteRec *__restrict__ part, uint32_t cap, const unsigned int *__restrict__ cursor, uint32_t blocks_per_bucket,
                                                                 uint32_t *__restrict__ hits, unsigned int *__restrict__ num_hits) {
    // workgroups go to the XCDs round-robin: x % 8 picks the XCD.
    //   blocks_per_bucket == 0: persistent workgroups — the gridDim.x / 8 workgroups of an XCD share the records of bucket 8 g + (x % 8), g = 0, 1, ...,
    //                           walking the buckets in the same order;
    //   blocks_per_bucket > 0:  all blocks_per_bucket workgroups of bucket 8 g + (x % 8) are neighbours in dispatch order.
    // Either way an XCD works its way through one bucket (or a few) at a time.
    const uint32_t xcd = blockIdx.x & 7u, y = blockIdx.x >> 3;
    const uint32_t first_bucket = blocks_per_bucket ? (y / blocks_per_bucket) * 8u + xcd : xcd, bucket_step = blocks_per_bucket ? 256u : 8u;
    const uint32_t sub = blocks_per_bucket ? y % blocks_per_bucket : y, step = (blocks_per_bucket ? blocks_per_bucket : (gridDim.x >> 3)) * BLOCK;
    // hits (a few per cent of the records) are collected in an LDS queue and moved to the chunk's hit list with one global atomic per workgroup; a
    // hit that finds the queue full goes to the list directly
    constexpr uint32_t QCAP = 4 * BLOCK;
    __shared__ uint32_t queue[QCAP];
    __shared__ uint32_t qn, qbase;
    // SURVIVOR COMPACTION.  A clear bit ends a record's probes (BloomFilter::containsF), and at 40 % filled only 0.4^q of the records reach round q — but
    // a wavefront that keeps every record in its lane until the last one has failed executes every round for all 64 lanes (round 4's first form: 596
    // VALU instructions per record, half of the kernel's wave cycles waiting for issue: profiles/r04_sq_kmc_before_compaction.txt).  So every record gets
    // PDENSE rounds in its lane; the few that are still alive go to an LDS list, and the workgroup works that list off densely, one survivor per lane.
#ifndef BT_KMC_PDENSE
#define BT_KMC_PDENSE 2
#endif
    constexpr uint32_t SCAP = 3 * BLOCK, PDENSE = BT_KMC_PDENSE;
    __shared__ uint32_t surv_lo[SCAP], surv_hi[SCAP], surv_idx[SCAP];
    __shared__ uint32_t sn;
    if (threadIdx.x == 0) {
        qn = 0;
        sn = 0;
    }
    __syncthreads();
    const uint8_t *filter = reinterpret_cast<const uint8_t *>(bloom.words);
    auto push_hit = [&](uint32_t idx) {
        const uint32_t p = atomicAdd(&qn, 1u);
        if (p < QCAP) queue[p] = idx;
        else hits[atomicAdd(num_hits, 1u)] = idx;
    };
    auto sub_filter = [&](uint64_t h) { return filter + (nthash64_seeded(h, bloom.k, BT_ROUTE_SEED) & (uint64_t)(BT_NUM_SUB_BLOOMS - 1u)) * bloom.stride; };
    // rounds first .. num_hashes - 1 of one record, stopping at the first clear bit
    auto finish = [&](uint64_t h, const uint8_t *bytes, unsigned first) {
        bool in = true;
        for (unsigned q = first; q < bloom.num_hashes && in; ++q) {
            const uint64_t pos = bloom_probe_pos(h, q, bloom);
            in = (bytes[pos >> 3] & (1u << (7u - (unsigned)(pos & 7u)))) != 0;
        }
        return in;
    };
    for (uint32_t bucket = first_bucket; bucket < 256u; bucket += bucket_step) {
        // the bucket's records: eight stripes, seen as one sequence
        uint32_t ofs[PSTRIPES + 1];
        ofs[0] = 0;
#pragma unroll
        for (uint32_t s = 0; s < PSTRIPES; ++s) {
            const uint32_t filled = cursor[s * 256u + bucket];
            ofs[s + 1] = ofs[s] + (filled < cap ? filled : cap);
        }
        const uint32_t n = ofs[PSTRIPES];
        const RouteRec *src = part + (uint64_t)bucket * PSTRIPES * cap;
        // PU records per lane at a time, the probes of one round of all of them issued together.  (The loop variable is the same in every thread of the
        // workgroup: the loop body holds barriers.)
        for (uint32_t base = sub * BLOCK; base < n; base += PU * step) {
            RouteRec r[PU];
            bool in[PU];
#pragma unroll
            for (uint32_t u = 0; u < PU; ++u) {
                const uint32_t i = base + threadIdx.x + u * step;
                in[u] = i < n;
                uint32_t s = 0, o = 0;
#pragma unroll
                for (uint32_t q = 1; q < PSTRIPES; ++q)
                    if (i >= ofs[q]) {
                        s = q;
                        o = ofs[q];
                    }
                r[u] = in[u] ? src[(uint64_t)s * cap + (i - o)] : RouteRec{0u, 0u, 0u};
            }
            uint64_t h[PU];
            const uint8_t *bytes[PU];
#pragma unroll
            for (uint32_t u = 0; u < PU; ++u) {
                h[u] = (uint64_t)r[u].h_lo | ((uint64_t)r[u].h_hi << 32);
                bytes[u] = sub_filter(h[u]);
            }
            const unsigned dense = bloom.num_hashes < PDENSE ? bloom.num_hashes : PDENSE;
            for (unsigned q = 0; q < dense; ++q) {
                uint32_t byte[PU], bit[PU];
#pragma unroll
                for (uint32_t u = 0; u < PU; ++u) {
                    byte[u] = bit[u] = 0;
                    if (in[u]) {
                        const uint64_t pos = bloom_probe_pos(h[u], q, bloom);
                        bit[u] = 1u << (7u - (unsigned)(pos & 7u));
                        byte[u] = (uint32_t)bytes[u][pos >> 3];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < PU; ++u) in[u] = in[u] && (byte[u] & bit[u]) != 0;
            }
#pragma unroll
            for (uint32_t u = 0; u < PU; ++u)
                if (in[u]) {
                    if (dense == bloom.num_hashes) push_hit(r[u].idx);
                    else {
                        const uint32_t p = atomicAdd(&sn, 1u);
                        if (p < SCAP) {
                            surv_lo[p] = r[u].h_lo;
                            surv_hi[p] = r[u].h_hi;
                            surv_idx[p] = r[u].idx;
                        } else if (finish(h[u], bytes[u], dense)) push_hit(r[u].idx);   // (a list that is full: the record is finished in its lane)
                    }
                }
            __syncthreads();
            const uint32_t ns = sn < SCAP ? sn : SCAP;
            for (uint32_t e = threadIdx.x; e < ns; e += BLOCK) {
                const uint64_t hs = (uint64_t)surv_lo[e] | ((uint64_t)surv_hi[e] << 32);
                if (finish(hs, sub_filter(hs), dense)) push_hit(surv_idx[e]);
            }
            __syncthreads();
            if (threadIdx.x == 0) sn = 0;
            __syncthreads();
        }
    }
    __syncthreads();
    const uint32_t n_q = qn < QCAP ? qn : QCAP;
    if (threadIdx.x == 0 && n_q) qbase = atomicAdd(num_hits, n_q);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < n_q; j += BLOCK) hits[qbase + j] = queue[j];
}

// A record's bytes from HBM as words: the aligned words that cover its (at most 20) bytes, shifted so that the record starts at bit 0 of w[0]
// (13 single-byte loads of 64 different lines per wavefront before).  `records` is 16-byte aligned.
__device__ inline void kmc_load_record_words(const uint8_t *__restrict__ records, uint64_t byte_off, uint32_t rec_size, uint64_t total_bytes, uint32_t (&w)[5]) {
    const uint64_t a0 = byte_off & ~3ULL;
    const uint32_t *base = reinterpret_cast<const uint32_t *>(records + a0);
    const uint32_t sh = (uint32_t)(byte_off & 3ULL) * 8u;
    const uint32_t nw = (rec_size + 3u) / 4u;   // words of the shifted record (uniform)
    uint32_t x[6];
#pragma unroll
    for (uint32_t i = 0; i < 6u; ++i) x[i] = (i <= nw && a0 + 4u * i < total_bytes) ? base[i] : 0u;   // (a word that starts inside the buffer: the last one may reach up to 3 bytes into the allocation's padding)
#pragma unroll
    for (uint32_t i = 0; i < 5u; ++i) w[i] = sh ? ((x[i] >> sh) | (x[i + 1] << (32u - sh))) : x[i];
}
__device__ inline uint32_t kmc_word_at(const uint32_t (&w)[5], uint32_t i) { return i == 0 ? w[0] : i == 1 ? w[1] : i == 2 ? w[2] : i == 3 ? w[3] : w[4]; }
__device__ inline uint32_t kmc_byte_at(const uint32_t (&w)[5], uint32_t b) { return (kmc_word_at(w, b >> 2) >> ((b & 3u) * 8u)) & 0xFFu; }

// kmc_decode on the shifted words of a record
__device__ inline void kmc_decode_words(const KmcView &v, uint64_t prefix, const uint32_t (&w)[5], Kmer &out, uint32_t &count) {
    uint64_t bhi = 0, blo = 0;   // symbols s_0 .. s_{k-1}, s_0 most significant, right-aligned at bit 0
    auto push = [&](uint64_t bits, unsigned nbits) {   // 0 < nbits < 64
        bhi = (bhi << nbits) | (blo >> (64u - nbits));
        blo = (blo << nbits) | bits;
    };
    if (v.p) {
        const unsigned pb = 2u * v.p;   // <= 30 bits
        push(prefix & ((1ULL << pb) - 1ULL), pb);
    }
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {   // suffix bytes are a big-endian number: a little-endian word of four of them, byte-swapped, is its next 32 bits
        if (4u * i >= v.suffix_bytes) break;
        const uint32_t nb = v.suffix_bytes - 4u * i < 4u ? v.suffix_bytes - 4u * i : 4u;
        const uint32_t be = __builtin_bswap32(w[i]);
        push(nb == 4u ? (uint64_t)be : (uint64_t)(be >> (32u - 8u * nb)), 8u * nb);
    }
    uint64_t rlo = rev2bit64(bhi), rhi = rev2bit64(blo);
    unsigned sh = 2u * (64u - v.k);
    if (sh == 0) { out.lo = rlo; out.hi = rhi; }
    else if (sh < 64u) { out.lo = (rlo >> sh) | (rhi << (64u - sh)); out.hi = rhi >> sh; }
    else if (sh == 64u) { out.lo = rhi; out.hi = 0; }
    else { out.lo = rhi >> (sh - 64u); out.hi = 0; }
    count = 0;
    for (unsigned b = 0; b < v.counter_size; ++b) count |= kmc_byte_at(w, v.suffix_bytes + b) << (8u * b);
}

// pass 3: the hits of a chunk, one per lane: decode the record again, add its count to the table (KmerCounter.cpp:414-419)
__global__ __launch_bounds__(BLOCK) void kmc_apply_kernel(KmcView v, TableView t, uint32_t sample_idx, const uint8_t *__restrict__ records, uint64_t first_record, uint64_t rec_offset,
                                                          uint64_t n_total, const uint32_t *__restrict__ hits, const unsigned int *__restrict__ num_hits, unsigned long long *__restrict__ hit_count) {
    __shared__ unsigned block_hits;
    if (threadIdx.x == 0) block_hits = 0;
    __syncthreads();
    const uint32_t nh = *num_hits;
    unsigned my_hits = 0;
    for (uint32_t j = blockIdx.x * BLOCK + threadIdx.x; j < nh; j += gridDim.x * BLOCK) {
        const uint64_t ridx = rec_offset + hits[j];
        uint32_t w[5];
        kmc_load_record_words(records, ridx * v.rec_size, v.rec_size, n_total * (uint64_t)v.rec_size, w);
        Kmer a;
        uint32_t count;
        kmc_decode_words(v, kmc_prefix_of(v, first_record + ridx), w, a, count);
        if (count < v.min_count || count > v.max_count) continue;
        my_hits += 1;
        uint32_t seen;
        const int64_t slot = table_find_or_insert(t, a, sample_idx >> 2, &seen);
        if (slot >= 0) sat_add_byte_from(t.counts(slot), sample_idx, count > 255u ? 255u : count, seen);
    }
    if (hit_count) {   // one atomic per workgroup
        for (int off = 32; off > 0; off >>= 1) my_hits += __shfl_down(my_hits, off);
        if ((threadIdx.x & 63u) == 0 && my_hits) atomicAdd(&block_hits, my_hits);
        __syncthreads();
        if (threadIdx.x == 0 && block_hits) atomicAdd(hit_count, (unsigned long long)block_hits);
    }
}

// count rows (bt_table_export_count_rows / bt_table_merge_count_rows): 16 key bytes + spad count bytes per record with a non-zero count
__global__ __launch_bounds__(BLOCK) void export_count_rows_kernel(TableView t, uint64_t cap, uint8_t *__restrict__ rows, uint64_t capacity_rows, unsigned long long *__restrict__ num_rows) {
    const uint32_t words = t.spad / 4u, row_words = 4u + words;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * BLOCK) {
        if (*t.state(i) != ST_READY) continue;
        uint32_t any = 0;
        for (uint32_t w = 0; w < words; ++w) any |= t.counts(i)[w];
        if (!any) continue;
        const unsigned long long at = atomicAdd(num_rows, 1ULL);
        if (at >= capacity_rows) continue;   // (sizing pass, or an undersized buffer: the host compares the counter with the capacity)
        uint32_t *out = reinterpret_cast<uint32_t *>(rows) + at * row_words;
        const uint64_t lo = *t.key_lo(i), hi = *t.key_hi(i);
        out[0] = (uint32_t)lo;
        out[1] = (uint32_t)(lo >> 32);
        out[2] = (uint32_t)hi;
        out[3] = (uint32_t)(hi >> 32);
        for (uint32_t w = 0; w < words; ++w) out[4 + w] = t.counts(i)[w];
    }
}
__global__ __launch_bounds__(BLOCK) void merge_count_rows_kernel(TableView t, const uint8_t *__restrict__ rows, uint64_t n) {
    const uint32_t words = t.spad / 4u, row_words = 4u + words;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t *in = reinterpret_cast<const uint32_t *>(rows) + r * row_words;
        Kmer a;
        a.lo = (uint64_t)in[0] | ((uint64_t)in[1] << 32);
        a.hi = (uint64_t)in[2] | ((uint64_t)in[3] << 32);
        const int64_t slot = table_find_or_insert(t, a);
        if (slot < 0) continue;
        for (uint32_t w = 0; w < words; ++w) {
            const uint32_t c = in[4 + w];
            for (uint32_t b = 0; b < 4u; ++b) {
                const uint32_t add = (c >> (8u * b)) & 0xFFu;
                if (add) sat_add_byte(t.counts(slot), 4u * w + b, add);
            }
        }
    }
}

// makeBloom (src/bayesTyperTools/MakeBloom.cpp:200-295): every record's k-mer goes into the sample's KmerBloom.  Same staging as
// kmc_scan_kernel; the insert is an atomicOr per probe (order-independent, so the filter bytes equal the reference's).
__global__ __launch_bounds__(BLOCK) void kmc_make_bloom_kernel(KmcView v, BloomView bloom, const uint8_t *__restrict__ records, uint64_t first_record, uint64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[KMC_RECS * KMC_MAX_REC + 32];
    __shared__ uint64_t block_prefix[2];
    const uint64_t num_chunks = (n + KMC_RECS - 1) / KMC_RECS;
    for (uint64_t chunk = blockIdx.x; chunk < num_chunks; chunk += gridDim.x) {
        const uint64_t rec0 = chunk * KMC_RECS;
        const unsigned nrec = (unsigned)((n - rec0) < KMC_RECS ? (n - rec0) : KMC_RECS);
        if (threadIdx.x < 2) block_prefix[threadIdx.x] = kmc_prefix_of(v, first_record + rec0 + (threadIdx.x ? nrec - 1 : 0));
        const uint64_t byte0 = rec0 * v.rec_size;
        const unsigned nbytes = nrec * v.rec_size;
        const uint64_t a0 = byte0 & ~15ULL;
        const unsigned lead = (unsigned)(byte0 - a0);
        const unsigned nvec = (lead + nbytes + 15u) / 16u;
        const uint64_t total_bytes = n * (uint64_t)v.rec_size;
        for (unsigned j = threadIdx.x; j < nvec; j += BLOCK) {
            const uint64_t off = a0 + (uint64_t)j * 16u;
            if (off + 16u <= total_bytes) *reinterpret_cast<uint4 *>(&stage[j * 16u]) = *reinterpret_cast<const uint4 *>(records + off);
            else
                for (unsigned q = 0; q < 16u; ++q) stage[j * 16u + q] = (off + q < total_bytes) ? records[off + q] : 0;
        }
        __syncthreads();
        if (threadIdx.x < nrec) {
            Kmer a;
            uint32_t count;
            kmc_decode(v, kmc_prefix_in(v, first_record + rec0 + threadIdx.x, block_prefix[0], block_prefix[1]), &stage[lead + threadIdx.x * v.rec_size], a, count);
            if (count >= v.min_count && count <= v.max_count) bloom_insert(nthash64(a, v.k), bloom);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

static void table_free_arrays(bt::TableView &v) {
    (void)hipFree(v.slots);
    (void)hipFree(v.num_keys);
    (void)hipFree(v.overflow);
    v.slots = v.overflow = nullptr;
    v.num_keys = nullptr;
}

// allocate and zero the slots of a table of `cap` slots; everything allocated so far is released on failure
static int table_alloc_arrays(bt_ctx *ctx, uint64_t cap, uint32_t spad, uint32_t k, bt::TableView &v) {
    v = bt::TableView{};
    v.mask = cap - 1;
    v.spad = spad;
    v.slot_words = bt::TableView::slot_words_for(spad);
    v.k = k;
    if (const char *e = getenv("BT_TABLE_RELEASE_PUBLISH")) v.flags = atoi(e) ? 1u : 0u;
    const uint64_t bytes = cap * (uint64_t)v.slot_words * 4u;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&v.slots), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&v.num_keys), 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&v.overflow), 4);
    if (e == hipSuccess) e = hipMemsetAsync(v.slots, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(v.num_keys, 0, 8, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(v.overflow, 0, 4, ctx->stream);
    if (e != hipSuccess) {
        table_free_arrays(v);
        return fail(std::string("bt_table: allocating ") + std::to_string(cap) + " slots: " + hipGetErrorString(e));
    }
    return BT_OK;
}

int bt_table_create(bt_ctx *ctx, uint64_t expected_size, uint32_t num_samples, uint32_t k, bt_table **out) {
    if (!ctx || !out) return fail("bt_table_create: null argument");
    if (num_samples < 1 || num_samples > 30) return fail("bt_table_create: number of samples must be in 1..30");   // main.cpp:72
    if (k < 1 || k > 64) return fail("bt_table_create: k must be in 1..64");
    uint64_t cap = 1024;
    while (cap < expected_size * 2) cap <<= 1;
    BT_HIP(hipSetDevice(ctx->device));
    bt_table *t = new bt_table();
    t->ctx = ctx;
    t->capacity = cap;
    t->num_samples = num_samples;
    t->spad = (num_samples + 3u) & ~3u;
    t->k = k;
    if (table_alloc_arrays(ctx, cap, t->spad, k, t->v) != BT_OK) {
        delete t;
        return BT_ERR;
    }
    *out = t;
    return BT_OK;
}

int bt_table_clear(bt_table *t) {
    if (!t) return fail("bt_table_clear: null table");
    BT_HIP(hipSetDevice(t->ctx->device));
    BT_HIP(hipMemsetAsync(t->v.slots, 0, t->capacity * (uint64_t)t->v.slot_words * 4u, t->ctx->stream));
    BT_HIP(hipMemsetAsync(t->v.num_keys, 0, 8, t->ctx->stream));
    BT_HIP(hipMemsetAsync(t->v.overflow, 0, 4, t->ctx->stream));
    return BT_OK;
}

int bt_table_reserve(bt_table *t, uint64_t expected_size) {
    if (!t) return fail("bt_table_reserve: null table");
    uint64_t cap = t->capacity;
    while (cap < expected_size * 2) cap <<= 1;
    if (cap == t->capacity) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    int ov = 0;
    if (bt_table_status(t, nullptr, nullptr, &ov) != BT_OK) return BT_ERR;
    if (ov) return fail("bt_table_reserve: the table has already overflowed (records were dropped)");
    bt::TableView nv;
    if (table_alloc_arrays(t->ctx, cap, t->spad, t->k, nv) != BT_OK) return BT_ERR;
    hipLaunchKernelGGL(table_rehash_kernel, dim3(grid_for(t->capacity, BLOCK, t->ctx->num_cu * 16)), dim3(BLOCK), 0, t->ctx->stream, t->v, t->capacity, nv);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(t->ctx->stream);
    if (e != hipSuccess) {
        table_free_arrays(nv);
        return fail(std::string("bt_table_reserve: ") + hipGetErrorString(e));
    }
    table_free_arrays(t->v);
    t->v = nv;
    t->capacity = cap;
    return BT_OK;
}

int bt_table_destroy(bt_table *t) {
    if (!t) return BT_OK;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    table_free_arrays(t->v);
    delete t;
    return BT_OK;
}

int bt_table_status(bt_table *t, uint64_t *num_keys, uint64_t *capacity, int *overflowed) {
    if (!t) return fail("bt_table_status: null table");
    BT_HIP(hipSetDevice(t->ctx->device));
    unsigned long long nk = 0;
    uint32_t ov = 0;
    if (num_keys) {
        BT_HIP(hipMemsetAsync(t->v.num_keys, 0, 8, t->ctx->stream));
        hipLaunchKernelGGL(table_count_kernel, dim3(grid_for(t->capacity, BLOCK, t->ctx->num_cu * 8)), dim3(BLOCK), 0, t->ctx->stream, t->v, t->capacity, t->v.num_keys);
        BT_CHECK_LAUNCH();
    }
    BT_HIP(hipMemcpyAsync(&nk, t->v.num_keys, 8, hipMemcpyDeviceToHost, t->ctx->stream));
    BT_HIP(hipMemcpyAsync(&ov, t->v.overflow, 4, hipMemcpyDeviceToHost, t->ctx->stream));
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    if (num_keys) *num_keys = nk;
    if (capacity) *capacity = t->capacity;
    if (overflowed) *overflowed = (int)ov;
    return BT_OK;
}

int bt_table_insert_batch(bt_table *t, const uint64_t *d_kmers, uint64_t n, int mark_parameter) {
    if (!t) return fail("bt_table_insert_batch: null table");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    hipLaunchKernelGGL(table_insert_kernel, dim3(grid_for(n, BLOCK, t->ctx->num_cu * 16)), dim3(BLOCK), 0, t->ctx->stream, t->v, d_kmers, n,
                       mark_parameter);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_table_find_batch(bt_table *t, const uint64_t *d_kmers, uint64_t n, int64_t *d_slots) {
    if (!t) return fail("bt_table_find_batch: null table");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    hipLaunchKernelGGL(table_find_kernel, dim3(grid_for(n, BLOCK, t->ctx->num_cu * 16)), dim3(BLOCK), 0, t->ctx->stream, t->v, d_kmers, n,
                       d_slots);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_table_read_slots(bt_table *t, const int64_t *h_slots, uint64_t n, uint8_t *h_counts, uint8_t *h_meta) {
    if (!t || !h_slots) return fail("bt_table_read_slots: null argument");
    BT_HIP(hipSetDevice(t->ctx->device));
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    std::vector<uint8_t> cbuf(t->spad);
    for (uint64_t i = 0; i < n; ++i) {
        int64_t s = h_slots[i];
        if (s < 0 || (uint64_t)s >= t->capacity) {
            if (h_counts) for (uint32_t j = 0; j < t->num_samples; ++j) h_counts[i * t->num_samples + j] = 0;
            if (h_meta) for (int j = 0; j < 4; ++j) h_meta[i * 4 + j] = 0;
            continue;
        }
        if (h_counts) {
            BT_HIP(hipMemcpy(cbuf.data(), t->v.counts((uint64_t)s), t->spad, hipMemcpyDeviceToHost));
            for (uint32_t j = 0; j < t->num_samples; ++j) h_counts[i * t->num_samples + j] = cbuf[j];
        }
        if (h_meta) BT_HIP(hipMemcpy(h_meta + i * 4, t->v.meta((uint64_t)s), 4, hipMemcpyDeviceToHost));
    }
    return BT_OK;
}

int bt_table_export(bt_table *t, uint64_t *h_kmers, uint8_t *h_counts, uint8_t *h_meta, uint64_t max_records, uint64_t *num_written) {
    if (!t || !num_written) return fail("bt_table_export: null argument");
    BT_HIP(hipSetDevice(t->ctx->device));
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    const uint64_t cap = t->capacity;
    // the slots come over in blocks of 2^20 (a whole table can be tens of gigabytes)
    const uint32_t sw = t->v.slot_words;
    const uint64_t block = 1ull << 20;
    std::vector<uint32_t> buf(std::min(cap, block) * sw);
    uint64_t w = 0;
    for (uint64_t i0 = 0; i0 < cap; i0 += block) {
        const uint64_t m = std::min(block, cap - i0);
        BT_HIP(hipMemcpy(buf.data(), t->v.slot(i0), m * sw * 4u, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; ++i) {
            const uint32_t *sl = &buf[i * sw];
            if (sl[0] != ST_READY) continue;
            if (w >= max_records) return fail("bt_table_export: output arrays too small");
            if (h_kmers) std::memcpy(h_kmers + 2 * w, sl + 2, 16);
            if (h_counts) std::memcpy(h_counts + w * t->num_samples, sl + 6, t->num_samples);
            if (h_meta) std::memcpy(h_meta + w * 4, sl + 1, 4);
            ++w;
        }
    }
    *num_written = w;
    return BT_OK;
}

int bt_table_count_row_bytes(bt_table *t, uint32_t *row_bytes) {
    if (!t || !row_bytes) return fail("bt_table_count_row_bytes: null argument");
    *row_bytes = 16u + t->spad;
    return BT_OK;
}

int bt_table_export_count_rows(bt_table *t, uint8_t *d_rows, uint64_t capacity_rows, uint64_t *h_num_rows) {
    if (!t || !h_num_rows || (capacity_rows && !d_rows)) return fail("bt_table_export_count_rows: null argument");
    BT_HIP(hipSetDevice(t->ctx->device));
    unsigned long long *d_n = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_n), 8));
    hipError_t e = hipMemsetAsync(d_n, 0, 8, t->ctx->stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<uint64_t>((t->capacity + BLOCK - 1) / BLOCK, 1u << 16);
        hipLaunchKernelGGL(export_count_rows_kernel, dim3(grid), dim3(BLOCK), 0, t->ctx->stream, t->v, t->capacity, d_rows, capacity_rows, d_n);
        e = hipGetLastError();
    }
    unsigned long long n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, t->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->ctx->stream);
    (void)hipFree(d_n);
    if (e != hipSuccess) return fail(std::string("bt_table_export_count_rows: ") + hipGetErrorString(e));
    *h_num_rows = n;
    if (capacity_rows && n > capacity_rows) return fail("bt_table_export_count_rows: output buffer too small");
    return BT_OK;
}

int bt_table_merge_count_rows(bt_table *t, const uint8_t *d_rows, uint64_t num_rows) {
    if (!t || (num_rows && !d_rows)) return fail("bt_table_merge_count_rows: null argument");
    if (num_rows == 0) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    const unsigned grid = (unsigned)std::min<uint64_t>((num_rows + BLOCK - 1) / BLOCK, 1u << 16);
    hipLaunchKernelGGL(merge_count_rows_kernel, dim3(grid), dim3(BLOCK), 0, t->ctx->stream, t->v, d_rows, num_rows);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_table_count_parameter_kmers(bt_table *t, bt_bloom *path_bloom, const char *d_seq, uint32_t num_regions, const uint64_t *h_start, const uint64_t *h_len,
                                   const uint8_t *h_is_decoy, const uint32_t *h_seed, float fraction) {
    if (!t || !path_bloom || !d_seq || !h_start || !h_len || !h_is_decoy || !h_seed) return fail("bt_table_count_parameter_kmers: null argument");
    if (path_bloom->k != t->k) return fail("bt_table_count_parameter_kmers: k mismatch between table and bloom");
    if (!(fraction > 0) || fraction > 1) return fail("bt_table_count_parameter_kmers: fraction must be in (0, 1]");   // KmerCounter.cpp:168-169
    if (num_regions == 0) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    std::vector<uint32_t> order(num_regions);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h_start[a] < h_start[b]; });
    const uint64_t span0 = h_start[order.front()];
    uint64_t span1 = span0;
    std::vector<ParamRegion> regs(num_regions);
    for (uint32_t i = 0; i < num_regions; ++i) {
        const uint32_t r = order[i];
        if (i && h_start[r] < h_start[order[i - 1]] + h_len[order[i - 1]]) return fail("bt_table_count_parameter_kmers: regions overlap");
        regs[i] = ParamRegion{h_start[r] - span0, h_len[r], h_seed[r], h_is_decoy[r] ? 1u : 0u};
        span1 = std::max(span1, h_start[r] + h_len[r]);
    }
    const uint64_t n = span1 - span0;
    if (n == 0) return BT_OK;
    hipStream_t st = t->ctx->stream;
    ParamRegion *d_regs = nullptr;
    uint64_t *d_kmers = nullptr;
    uint8_t *d_valid = nullptr, *d_cand = nullptr;
    uint32_t *d_mt = nullptr;
    const uint32_t batch = 1u << 16;   // regions drawing concurrently (625 state words each)
    auto release = [&]() {
        (void)hipFree(d_regs);
        (void)hipFree(d_kmers);
        (void)hipFree(d_valid);
        (void)hipFree(d_cand);
        (void)hipFree(d_mt);
    };
#define PK(call)                                                                                        \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            release();                                                                                  \
            return fail(std::string("bt_table_count_parameter_kmers: ") + hipGetErrorString(_e));       \
        }                                                                                               \
    } while (0)
    PK(hipMalloc(reinterpret_cast<void **>(&d_regs), (size_t)num_regions * sizeof(ParamRegion)));
    PK(hipMalloc(reinterpret_cast<void **>(&d_kmers), n * 16));
    PK(hipMalloc(reinterpret_cast<void **>(&d_valid), n));
    PK(hipMalloc(reinterpret_cast<void **>(&d_cand), n));
    PK(hipMalloc(reinterpret_cast<void **>(&d_mt), (size_t)std::min(batch, num_regions) * MT_WORDS * 4));
    PK(hipMemcpyAsync(d_regs, regs.data(), (size_t)num_regions * sizeof(ParamRegion), hipMemcpyHostToDevice, st));
    if (bt_kmers_from_sequence(t->ctx, d_seq + span0, n, t->k, d_kmers, d_valid) != BT_OK) {
        release();
        return BT_ERR;
    }
    const unsigned maxb = t->ctx->num_cu * 16;
    hipLaunchKernelGGL(param_cand_kernel, dim3(grid_for(n, BLOCK, maxb)), dim3(BLOCK), 0, st, path_bloom->view(), d_regs, num_regions, d_kmers, d_valid, n, t->k, d_cand);
    for (uint32_t r0 = 0; r0 < num_regions; r0 += batch) {
        const uint32_t r1 = std::min(num_regions, r0 + batch);
        hipLaunchKernelGGL(param_draw_kernel, dim3((r1 - r0 + 63) / 64), dim3(64), 0, st, d_regs, r0, r1, d_mt, (double)fraction, d_cand);
    }
    hipLaunchKernelGGL(param_insert_kernel, dim3(grid_for(n, BLOCK, maxb)), dim3(BLOCK), 0, st, t->v, d_regs, num_regions, d_kmers, d_cand, n);
    PK(hipGetLastError());
    PK(hipStreamSynchronize(st));
#undef PK
    release();
    return BT_OK;
}

int bt_table_kmer_stats(bt_table *t, const uint8_t *h_gender, uint64_t *h_class_counts, uint64_t *h_n, uint64_t *h_nonzero, uint64_t *h_sum, uint64_t *h_sumsq) {
    if (!t || !h_gender || !h_class_counts || !h_n || !h_nonzero || !h_sum || !h_sumsq) return fail("bt_table_kmer_stats: null argument");
    BT_HIP(hipSetDevice(t->ctx->device));
    const uint32_t S = t->num_samples;
    const size_t words = 7 + (size_t)4 * S * 256;
    unsigned long long *d_acc = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_acc), words * 8));
    BT_HIP(hipMemsetAsync(d_acc, 0, words * 8, t->ctx->stream));
    uint32_t gmask = 0;
    for (uint32_t s = 0; s < S; ++s) gmask |= (uint32_t)(h_gender[s] & 1u) << s;
    const unsigned grid = grid_for(t->capacity, BLOCK, t->ctx->num_cu * 8);
    hipLaunchKernelGGL(kmer_stats_kernel, dim3(grid), dim3(BLOCK), 0, t->ctx->stream, t->v, t->capacity, S, gmask, d_acc);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(t->ctx->stream);
    std::vector<unsigned long long> h(words);
    if (e == hipSuccess) e = hipMemcpy(h.data(), d_acc, words * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_acc);
    if (e != hipSuccess) return fail(std::string("bt_table_kmer_stats: ") + hipGetErrorString(e));
    for (int i = 0; i < 7; ++i) h_class_counts[i] = h[i];
    for (size_t i = 0; i < (size_t)S * 256; ++i) {
        h_n[i] = h[7 + i];
        h_sum[i] = h[7 + (size_t)S * 256 + i];
        h_sumsq[i] = h[7 + (size_t)2 * S * 256 + i];
        h_nonzero[i] = h[7 + (size_t)3 * S * 256 + i];
    }
    return BT_OK;
}

int bt_table_count_intercluster(bt_table *t, bt_bloom *path_bloom, const char *d_seq, uint64_t len, int is_decoy, uint32_t female_ploidy,
                                uint32_t male_ploidy) {
    if (!t || !path_bloom) return fail("bt_table_count_intercluster: null argument");
    if (path_bloom->k != t->k) return fail("bt_table_count_intercluster: k mismatch between table and bloom");
    if (len == 0) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    unsigned grid = grid_for((len + SEQ_TILE - 1) / SEQ_TILE, 1, t->ctx->num_cu * 8);
    hipLaunchKernelGGL(intercluster_kernel, dim3(grid), dim3(BLOCK), 0, t->ctx->stream, t->v, path_bloom->view(), d_seq, len, is_decoy,
                       female_ploidy, male_ploidy);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

// all regions of one chromosome in ONE launch (a call per region is a kernel launch per region: 175 000 launches for a chr20-sized unit)
int bt_table_count_intercluster_regions(bt_table *t, bt_bloom *path_bloom, const char *d_seq, uint32_t num_regions, const uint64_t *h_start, const uint64_t *h_len,
                                        const uint8_t *h_is_decoy, const uint8_t *h_female_ploidy, const uint8_t *h_male_ploidy) {
    if (!t || !path_bloom || (num_regions && (!h_start || !h_len || !h_is_decoy || !h_female_ploidy || !h_male_ploidy))) return fail("bt_table_count_intercluster_regions: null argument");
    if (path_bloom->k != t->k) return fail("bt_table_count_intercluster_regions: k mismatch between table and bloom");
    std::vector<RegionTile> tiles;
    for (uint32_t r = 0; r < num_regions; ++r)
        for (uint64_t a = 0; a < h_len[r]; a += SEQ_TILE)
            tiles.push_back(RegionTile{h_start[r], h_start[r] + h_len[r], h_start[r] + a, (uint32_t)h_is_decoy[r] | ((uint32_t)h_female_ploidy[r] << 8) | ((uint32_t)h_male_ploidy[r] << 16), 0u});
    if (tiles.empty()) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    RegionTile *d_tiles = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_tiles), tiles.size() * sizeof(RegionTile)));
    hipError_t e = hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(RegionTile), hipMemcpyHostToDevice, t->ctx->stream);
    if (e == hipSuccess) {
        const unsigned grid = grid_for(tiles.size(), 1, t->ctx->num_cu * 8);
        hipLaunchKernelGGL(intercluster_regions_kernel, dim3(grid), dim3(BLOCK), 0, t->ctx->stream, t->v, path_bloom->view(), d_seq, (const RegionTile *)d_tiles, (uint64_t)tiles.size());
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(t->ctx->stream);   // (the tile list is a host vector and a temporary device buffer)
    (void)hipFree(d_tiles);
    if (e != hipSuccess || e2 != hipSuccess) return fail(std::string("bt_table_count_intercluster_regions: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return BT_OK;
}

int bt_table_classify_batch(bt_table *t, bt_bloom *multigroup_bloom, const uint64_t *d_kmers, const uint8_t *d_mult, uint64_t n,
                            uint8_t *d_excluded) {
    if (!t || !multigroup_bloom) return fail("bt_table_classify_batch: null argument");
    if (multigroup_bloom->k != t->k) return fail("bt_table_classify_batch: k mismatch between table and bloom");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    hipLaunchKernelGGL(classify_kernel, dim3(grid_for(n, BLOCK, t->ctx->num_cu * 16)), dim3(BLOCK), 0, t->ctx->stream, t->v,
                       multigroup_bloom->view(), d_kmers, d_mult, n, d_excluded);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_kmc_scan_create(bt_ctx *ctx, uint32_t k, uint32_t lut_prefix_len, uint32_t counter_size, uint64_t total_records,
                       const uint64_t *h_prefix_lut, bt_kmc_scan **out) {
    if (lut_prefix_len > 15) return fail("bt_kmc_scan_create: lut_prefix_len <= 15");
    return bt_kmc_scan_create_bins(ctx, k, lut_prefix_len, counter_size, total_records, h_prefix_lut, (1ULL << (2 * lut_prefix_len)) + 1, out);
}

int bt_kmc_scan_create_bins(bt_ctx *ctx, uint32_t k, uint32_t lut_prefix_len, uint32_t counter_size, uint64_t total_records,
                            const uint64_t *h_prefix_lut, uint64_t num_lut_entries, bt_kmc_scan **out) {
    if (!ctx || !out || !h_prefix_lut) return fail("bt_kmc_scan_create: null argument");
    if (k < 1 || k > 64) return fail("bt_kmc_scan_create: k must be in 1..64");
    if (lut_prefix_len > k || lut_prefix_len > 15 || ((k - lut_prefix_len) % 4) != 0)
        return fail("bt_kmc_scan_create: (k - lut_prefix_len) must be a non-negative multiple of 4 and lut_prefix_len <= 15");
    if (counter_size < 1 || counter_size > 4) return fail("bt_kmc_scan_create: counter_size must be in 1..4");
    bt_kmc_scan *s = new bt_kmc_scan();
    s->ctx = ctx;
    s->k = k;
    s->p = lut_prefix_len;
    s->counter_size = counter_size;
    s->suffix_bytes = (k - lut_prefix_len) / 4;
    s->rec_size = s->suffix_bytes + counter_size;
    s->total = total_records;
    if (num_lut_entries < 2 || ((num_lut_entries - 1) % (1ULL << (2 * lut_prefix_len))) != 0) {
        delete s;
        return fail("bt_kmc_scan_create: the prefix LUT must hold a multiple of 4^p entries plus the terminal one");
    }
    s->lut_entries = num_lut_entries;
    if (h_prefix_lut[s->lut_entries - 1] != total_records || h_prefix_lut[0] != 0) {
        delete s;
        return fail("bt_kmc_scan_create: prefix LUT must start at 0 and end at total_records");
    }
    hipError_t e = hipSetDevice(ctx->device);
    // hint[j] = prefix of record j * 4096 = largest i with lut[i] <= j * 4096 (one pass over the table); the terminal entry is the last prefix
    const uint64_t num_hints = (total_records >> KMC_HINT_SHIFT) + 2;
    std::vector<uint32_t> hint(num_hints);
    {
        uint64_t i = 0;
        for (uint64_t j = 0; j + 1 < num_hints; ++j) {
            const uint64_t n = j << KMC_HINT_SHIFT;
            while (i + 2 < s->lut_entries && h_prefix_lut[i + 1] <= n) ++i;
            hint[j] = (uint32_t)i;
        }
        hint[num_hints - 1] = (uint32_t)(s->lut_entries - 2);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_lut), s->lut_entries * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_hint), num_hints * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_lut, h_prefix_lut, s->lut_entries * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_hint, hint.data(), num_hints * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        if (s->d_lut) (void)hipFree(s->d_lut);
        if (s->d_hint) (void)hipFree(s->d_hint);
        delete s;
        return fail(std::string("bt_kmc_scan_create: ") + hipGetErrorString(e));
    }
    *out = s;
    return BT_OK;
}

int bt_kmc_scan_set_count_range(bt_kmc_scan *s, uint32_t min_count, uint64_t max_count) {
    if (!s) return fail("bt_kmc_scan_set_count_range: null handle");
    s->min_count = min_count;
    s->max_count = max_count > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)max_count;   // counters are at most 4 bytes wide
    return BT_OK;
}

static void free_host_staging(bt_kmc_scan *s);
static void free_routed(bt_kmc_scan *s);
int bt_kmc_scan_destroy(bt_kmc_scan *s) {
    if (!s) return BT_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->d_lut) (void)hipFree(s->d_lut);
    if (s->d_hint) (void)hipFree(s->d_hint);
    free_host_staging(s);
    free_routed(s);
    delete s;
    return BT_OK;
}

static KmcView make_kmc_view(const bt_kmc_scan *s) {
    KmcView v;
    v.lut = s->d_lut;
    v.hint = s->d_hint;
    v.lut_entries = s->lut_entries;
    v.k = s->k;
    v.p = s->p;
    v.counter_size = s->counter_size;
    v.suffix_bytes = s->suffix_bytes;
    v.rec_size = s->rec_size;
    v.min_count = s->min_count;
    v.max_count = s->max_count;
    return v;
}

static void free_routed(bt_kmc_scan *s) {
    for (int b = 0; b < 2; ++b) {
        if (s->d_route_vals[b]) (void)hipFree(s->d_route_vals[b]);
        s->d_route_vals[b] = nullptr;
    }
    if (s->d_num_hits) (void)hipFree(s->d_num_hits);
    if (s->d_part_cursor) (void)hipFree(s->d_part_cursor);
    for (int b = 0; b < 2; ++b) {
        if (s->d_route_vals2[b]) (void)hipFree(s->d_route_vals2[b]);
        s->d_route_vals2[b] = nullptr;
        if (s->part_done[b]) (void)hipEventDestroy(s->part_done[b]);
        if (s->part_free[b]) (void)hipEventDestroy(s->part_free[b]);
        s->part_done[b] = s->part_free[b] = nullptr;
    }
    if (s->scan_begin) (void)hipEventDestroy(s->scan_begin);
    s->scan_begin = nullptr;
    if (s->d_num_hits2) (void)hipFree(s->d_num_hits2);
    if (s->d_part_cursor2) (void)hipFree(s->d_part_cursor2);
    s->d_part_cursor2 = s->d_num_hits2 = nullptr;
    s->d_part_cursor = nullptr;
    s->part_cap = 0;
    s->d_num_hits = nullptr;
    s->routed_cap = 0;
}

// buffers of the partitioned scan for chunks of up to `cap` records (kept with the handle): the 256 x 8 stripe regions (a stripe's expected
// share + 3 % + 512 records) and the chunk's hit list
static int ensure_routed(bt_kmc_scan *s, uint64_t cap) {
    if (s->routed_cap >= cap) return BT_OK;
    free_routed(s);
    const uint64_t part_cap = cap / (256 * PSTRIPES) + cap / (256 * PSTRIPES) * 3 / 100 + 512;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&s->d_route_vals[0]), 256 * PSTRIPES * part_cap * sizeof(RouteRec));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_route_vals[1]), cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_part_cursor), 256 * PSTRIPES * sizeof(unsigned int));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(kmc_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)partition_lds_bytes(KMC_MAX_REC));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_num_hits), 4);
    if (e != hipSuccess) {
        free_routed(s);
        return fail(std::string("bt_kmc_scan: buffers of the partitioned scan: ") + hipGetErrorString(e));
    }
    s->routed_cap = cap;
    s->part_cap = (uint32_t)part_cap;
    // the second set (a scan of several chunks overlaps the partition of the next chunk with the probe / apply of the current one): BT_KMC_OVERLAP=1.  Measured in
    // round 6 (profiles/r06_kmc_overlap.txt): 25.6 -> 25.4 ms per 10^9 records at the WGS filter shape, 46.4 -> 46.0 ms at 36 KB sub-filters — the three kernels
    // stress different units of a CU but share the memory pipeline, and 1.1 GB of HBM for 1 % is not a default
    if (getenv("BT_KMC_OVERLAP")) {
        hipError_t e2 = hipMalloc(reinterpret_cast<void **>(&s->d_route_vals2[0]), 256 * PSTRIPES * part_cap * sizeof(RouteRec));
        if (e2 == hipSuccess) e2 = hipMalloc(reinterpret_cast<void **>(&s->d_route_vals2[1]), cap * sizeof(uint32_t));
        if (e2 == hipSuccess) e2 = hipMalloc(reinterpret_cast<void **>(&s->d_part_cursor2), 256 * PSTRIPES * sizeof(unsigned int));
        if (e2 == hipSuccess) e2 = hipMalloc(reinterpret_cast<void **>(&s->d_num_hits2), 4);
        for (int b = 0; b < 2 && e2 == hipSuccess; ++b) {
            e2 = hipEventCreateWithFlags(&s->part_done[b], hipEventDisableTiming);
            if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&s->part_free[b], hipEventDisableTiming);
        }
        if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&s->scan_begin, hipEventDisableTiming);
        if (e2 != hipSuccess) {   // (not enough memory for two sets: the scan runs single-buffered)
            (void)hipGetLastError();
            for (int b = 0; b < 2; ++b) {
                if (s->d_route_vals2[b]) (void)hipFree(s->d_route_vals2[b]);
                s->d_route_vals2[b] = nullptr;
            }
            if (s->d_part_cursor2) (void)hipFree(s->d_part_cursor2);
            if (s->d_num_hits2) (void)hipFree(s->d_num_hits2);
            s->d_part_cursor2 = s->d_num_hits2 = nullptr;
        }
    }
    return BT_OK;
}

constexpr uint64_t kRoutedChunk = 1ull << 26;      // records per chunk (12 bytes of route record + 4 of hit list each)
constexpr uint64_t kRoutedMinRecords = 1ull << 22;  // below this the direct kernel is as fast

int bt_kmc_scan_run(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const uint8_t *d_records,
                    uint64_t first_record, uint64_t n, uint64_t *d_hit_count) {
    if (!s || !path_bloom || !table) return fail("bt_kmc_scan_run: null argument");
    if (path_bloom->k != s->k || table->k != s->k) return fail("bt_kmc_scan_run: k mismatch");
    if (sample_idx >= table->num_samples) return fail("bt_kmc_scan_run: sample index out of range");
    if (first_record + n > s->total) return fail("bt_kmc_scan_run: record range exceeds the database");
    if ((reinterpret_cast<uintptr_t>(d_records) & 15u) != 0) return fail("bt_kmc_scan_run: d_records must be 16-byte aligned");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(s->ctx->device));
    // a ThreadedKmerBloom and enough records: the partitioned form.  BT_KMC_ROUTED=0 / 1 forces the direct / partitioned form.
    const char *force = getenv("BT_KMC_ROUTED");
    bool routed = path_bloom->num_sub == BT_NUM_SUB_BLOOMS && n >= kRoutedMinRecords;
    if (force) routed = path_bloom->num_sub == BT_NUM_SUB_BLOOMS && atoi(force) != 0;
    if (!routed) {
        unsigned grid = grid_for((n + KMC_RECS - 1) / KMC_RECS, 1, s->ctx->num_cu * 8);
        hipLaunchKernelGGL(kmc_scan_kernel<false>, dim3(grid), dim3(BLOCK), 0, s->ctx->stream, make_kmc_view(s), path_bloom->view(), table->v,
                           sample_idx, d_records, first_record, n, reinterpret_cast<unsigned long long *>(d_hit_count), (uint64_t *)nullptr,
                           (uint32_t *)nullptr);
        BT_CHECK_LAUNCH();
        return BT_OK;
    }
    uint64_t chunk = std::min<uint64_t>(n, kRoutedChunk);
    if (const char *e = getenv("BT_KMC_ROUTED_CHUNK")) chunk = std::max<uint64_t>(1024, std::min<uint64_t>(chunk, strtoull(e, nullptr, 0)));   // tests: several chunks at small sizes
    if (ensure_routed(s, chunk) != BT_OK) return BT_ERR;
    const KmcView kv = make_kmc_view(s);
    uint32_t part_cap = s->part_cap;
    if (const char *e = getenv("BT_KMC_PART_CAP")) part_cap = std::max<uint32_t>(1, std::min<uint32_t>(part_cap, (uint32_t)strtoul(e, nullptr, 0)));   // tests: records beyond a stripe's region
    // Two buffer sets: chunk i's partition — LDS-bound (the hash table's random reads, the counting sort) — is enqueued on a class stream of the context (probed
    // to run concurrently with the context's stream, bt_ctx.hip) and the chunk's probe — VALU-bound — and apply — latency-bound — follow on the context's stream,
    // so that chunk i + 1 is partitioned while chunk i is probed and applied; a set is reused when the apply kernel that read it has ended (part_free).
    const bool overlap = s->d_route_vals2[0] != nullptr && n > chunk && getenv("BT_KMC_OVERLAP") != nullptr;
    hipStream_t main_st = s->ctx->stream, part_st = main_st;
    if (overlap) {
        int prio_lo = 0, prio_hi = 0;
        BT_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        hipStream_t cs[1] = {nullptr};
        BT_HIP(ctx_class_streams(s->ctx, 1, getenv("BT_GIBBS_NO_PRIO") ? prio_lo : prio_hi, cs));   // (the context's first class stream: the priority the samplers ask for)
        part_st = cs[0];
        BT_HIP(hipEventRecord(s->scan_begin, main_st));   // (what precedes the scan on the context's stream — the table's clear, the previous sample's scan)
        BT_HIP(hipStreamWaitEvent(part_st, s->scan_begin, 0));
    }
    uint64_t ci = 0;
    for (uint64_t off = 0; off < n; off += chunk, ++ci) {
        const uint64_t m = std::min<uint64_t>(chunk, n - off);
        const int b = overlap ? (int)(ci & 1u) : 0;
        RouteRec *part = (RouteRec *)(b ? s->d_route_vals2[0] : s->d_route_vals[0]);
        uint32_t *hit_list = reinterpret_cast<uint32_t *>(b ? s->d_route_vals2[1] : s->d_route_vals[1]);
        unsigned int *num_hits = b ? s->d_num_hits2 : s->d_num_hits, *cursor = b ? s->d_part_cursor2 : s->d_part_cursor;
        if (overlap && ci >= 2) BT_HIP(hipStreamWaitEvent(part_st, s->part_free[b], 0));
        BT_HIP(hipMemsetAsync(num_hits, 0, 4, part_st));
        BT_HIP(hipMemsetAsync(cursor, 0, 256 * PSTRIPES * sizeof(unsigned int), part_st));
        const unsigned pgrid = grid_for((m + PSLAB - 1) / PSLAB, 1, s->ctx->num_cu * 4);
        hipLaunchKernelGGL(kmc_partition_kernel, dim3(pgrid), dim3(PBLOCK), partition_lds_bytes(s->rec_size), part_st, kv, path_bloom->view(), d_records, first_record, off, m, n,
                           (uint32_t)partition_main_bytes(s->rec_size), part, part_cap, cursor, hit_list, num_hits);
        BT_CHECK_LAUNCH();
        if (overlap) {
            BT_HIP(hipEventRecord(s->part_done[b], part_st));
            BT_HIP(hipStreamWaitEvent(main_st, s->part_done[b], 0));
        }
        // blocks per bucket: two rounds of PU records per lane; BT_KMC_PROBE_BPB=0 -> persistent workgroups (eight per CU): measured 2.3x slower (71 against
        // 31 ms per 10^9 records: every wavefront of the chip in the same phase of the same bucket at the same time)
        uint32_t bpb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(256, (m / 256 + BLOCK * 2 * PU - 1) / (BLOCK * 2 * PU)));
        if (const char *e = getenv("BT_KMC_PROBE_BPB")) bpb = (uint32_t)std::min<uint64_t>(256, strtoul(e, nullptr, 0));
        const uint32_t pwgs = bpb ? 256u * bpb : 8u * (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(s->ctx->num_cu, (m / 256 + BLOCK * 16 - 1) / (BLOCK * 16)));
        hipLaunchKernelGGL(kmc_probe_bucket_kernel, dim3(pwgs), dim3(BLOCK), 0, main_st, path_bloom->view(), (const RouteRec *)part, part_cap, (const unsigned int *)cursor, bpb, hit_list, num_hits);
        BT_CHECK_LAUNCH();
        hipLaunchKernelGGL(kmc_apply_kernel, dim3(s->ctx->num_cu * 8), dim3(BLOCK), 0, main_st, kv, table->v, sample_idx, d_records, first_record, off, n, (const uint32_t *)hit_list,
                           (const unsigned int *)num_hits, reinterpret_cast<unsigned long long *>(d_hit_count));
        BT_CHECK_LAUNCH();
        if (overlap) BT_HIP(hipEventRecord(s->part_free[b], main_st));
    }
    return BT_OK;
}

static void free_host_staging(bt_kmc_scan *s) {
    // the buffers go back to the context (the next sample's scan takes them); what the context held before — smaller ones — is released
    const bool keep = s->h_pin[0] && s->h_pin[1] && s->d_stage[0] && s->d_stage[1] && s->stage_bytes >= s->ctx->kmc_stage_bytes;
    for (int b = 0; b < 2; ++b) {
        if (keep) {
            if (s->ctx->kmc_pin[b]) (void)hipHostFree(s->ctx->kmc_pin[b]);
            if (s->ctx->kmc_dev[b]) (void)hipFree(s->ctx->kmc_dev[b]);
            s->ctx->kmc_pin[b] = s->h_pin[b];
            s->ctx->kmc_dev[b] = s->d_stage[b];
        } else {
            if (s->h_pin[b]) (void)hipHostFree(s->h_pin[b]);
            if (s->d_stage[b]) (void)hipFree(s->d_stage[b]);
        }
        if (s->copied[b]) (void)hipEventDestroy(s->copied[b]);
        if (s->scanned[b]) (void)hipEventDestroy(s->scanned[b]);
        s->h_pin[b] = s->d_stage[b] = nullptr;
        s->copied[b] = s->scanned[b] = nullptr;
    }
    if (keep) s->ctx->kmc_stage_bytes = s->stage_bytes;
    if (s->d_host_hits) (void)hipFree(s->d_host_hits);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    s->d_host_hits = nullptr;
    s->copy_stream = nullptr;
    s->stage_bytes = 0;
}

// host range -> pinned buffer with several threads: one core's memcpy (~12 GB/s measured on the MI355X host) is well below the link
static void parallel_copy(uint8_t *dst, const uint8_t *src, size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t threads = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(hw ? hw : 1), bytes >> 22}));
    if (threads == 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t part = (bytes / threads + 4095) / 4096 * 4096;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < threads; ++t) {
        const size_t lo = std::min(bytes, t * part), hi = std::min(bytes, lo + part);
        if (hi > lo) pool.emplace_back([=]() { std::memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &th : pool) th.join();
}

// several threads pread() one range of a file into a (pinned) buffer: no page faults (a memory-mapped file costs one per 4 KB — or per fault-around window — of a
// first pass), the kernel copies from the page cache or reads the device, each thread its own stripe
static int parallel_pread(int fd, uint8_t *dst, uint64_t offset, size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    size_t threads = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)(hw ? hw / 2 : 1), bytes >> 21}));
    if (const char *e = getenv("BT_KMC_READ_THREADS")) threads = std::max<size_t>(1, std::min<size_t>((size_t)atoi(e), bytes >> 16));   // tuning
    const size_t part = (bytes / threads + 4095) / 4096 * 4096;
    std::vector<int> failed(threads, 0);
    auto work = [&](size_t t) {
        size_t lo = std::min(bytes, t * part);
        const size_t hi = std::min(bytes, lo + part);
        while (lo < hi) {
            const ssize_t got = ::pread(fd, dst + lo, hi - lo, (off_t)(offset + lo));
            if (got <= 0) {
                failed[t] = 1;
                return;
            }
            lo += (size_t)got;
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (int f : failed)
        if (f) return 1;
    return 0;
}

static int kmc_scan_stream(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const uint8_t *h_records, int fd, uint64_t file_offset, uint64_t first_record,
                           uint64_t n, uint64_t chunk_records, uint64_t *h_hit_count);

int bt_kmc_scan_run_host(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const uint8_t *h_records, uint64_t first_record, uint64_t n,
                         uint64_t chunk_records, uint64_t *h_hit_count) {
    if (!s || !path_bloom || !table || !h_records) return fail("bt_kmc_scan_run_host: null argument");
    return kmc_scan_stream(s, path_bloom, table, sample_idx, h_records, -1, 0, first_record, n, chunk_records, h_hit_count);
}

int bt_kmc_scan_run_file(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const char *suf_path, uint64_t payload_offset, uint64_t first_record, uint64_t n,
                         uint64_t chunk_records, uint64_t *h_hit_count) {
    if (!s || !path_bloom || !table || !suf_path) return fail("bt_kmc_scan_run_file: null argument");
    const int fd = ::open(suf_path, O_RDONLY);
    if (fd < 0) return fail(std::string("bt_kmc_scan_run_file: cannot open ") + suf_path);
    const int rc = kmc_scan_stream(s, path_bloom, table, sample_idx, nullptr, fd, payload_offset + first_record * s->rec_size, first_record, n, chunk_records, h_hit_count);
    ::close(fd);
    return rc;
}

static int kmc_scan_stream(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const uint8_t *h_records, int fd, uint64_t file_offset, uint64_t first_record,
                           uint64_t n, uint64_t chunk_records, uint64_t *h_hit_count) {
    if (first_record + n > s->total) return fail("bt_kmc_scan_run_host: record range exceeds the database");
    if (n == 0) {
        if (h_hit_count) *h_hit_count = 0;
        return BT_OK;
    }
    BT_HIP(hipSetDevice(s->ctx->device));
    const auto t_enter = std::chrono::steady_clock::now();
    const uint64_t rec = s->rec_size;
    chunk_records = std::max<uint64_t>(16, std::min<uint64_t>(chunk_records ? chunk_records : (1ull << 23), n + 15) / 16 * 16);   // chunk starts stay 16-byte aligned
    const size_t chunk_bytes = chunk_records * rec;
    // two staging slots: pinned host buffer -> device buffer on a copy stream, scan on the context's stream, events both ways
    hipError_t e = hipSuccess;
    if (s->stage_bytes < chunk_bytes) {
        free_host_staging(s);
        e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
        const bool reuse = s->ctx->kmc_stage_bytes >= chunk_bytes && s->ctx->kmc_pin[0] && s->ctx->kmc_pin[1];   // the previous sample's scan left them with the context
        for (int b = 0; b < 2 && e == hipSuccess; ++b) {
            if (reuse) {
                s->h_pin[b] = s->ctx->kmc_pin[b];
                s->d_stage[b] = s->ctx->kmc_dev[b];
                s->ctx->kmc_pin[b] = s->ctx->kmc_dev[b] = nullptr;
            } else {
                e = hipHostMalloc(reinterpret_cast<void **>(&s->h_pin[b]), chunk_bytes, hipHostMallocDefault);
                if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_stage[b]), chunk_bytes + 16);
            }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->copied[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->scanned[b], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->d_host_hits), 8);
        if (e != hipSuccess) {
            free_host_staging(s);
            return fail(std::string("bt_kmc_scan_run_host: staging buffers: ") + hipGetErrorString(e));
        }
        s->stage_bytes = reuse ? s->ctx->kmc_stage_bytes : chunk_bytes;
        if (reuse) s->ctx->kmc_stage_bytes = 0;
    }
    const bool timing = getenv("BT_STAGE_TIMES") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    double read_s = 0, wait_s = 0;
    e = hipMemsetAsync(s->d_host_hits, 0, 8, s->ctx->stream);
    int rc = BT_OK;
    uint64_t done = 0;
    for (uint64_t i = 0; e == hipSuccess && rc == BT_OK && done < n; ++i) {
        const int b = (int)(i & 1);
        const uint64_t m = std::min(chunk_records, n - done);
        const auto t_w = std::chrono::steady_clock::now();
        if (i >= 2) e = hipEventSynchronize(s->copied[b]);   // the pinned buffer of this slot has been read by its previous copy
        if (e != hipSuccess) break;
        const auto t_r = std::chrono::steady_clock::now();
        wait_s += std::chrono::duration<double>(t_r - t_w).count();
        if (fd >= 0) {
            if (parallel_pread(fd, s->h_pin[b], file_offset + done * rec, m * rec) != 0) {
                rc = fail("bt_kmc_scan_run_file: reading the records failed");
                break;
            }
        } else
            parallel_copy(s->h_pin[b], h_records + done * rec, m * rec);
        read_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_r).count();
        if (i >= 2) e = hipStreamWaitEvent(s->copy_stream, s->scanned[b], 0);   // the device buffer of this slot has been scanned
        if (e == hipSuccess) e = hipMemcpyAsync(s->d_stage[b], s->h_pin[b], m * rec, hipMemcpyHostToDevice, s->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(s->copied[b], s->copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s->ctx->stream, s->copied[b], 0);
        if (e != hipSuccess) break;
        rc = bt_kmc_scan_run(s, path_bloom, table, sample_idx, s->d_stage[b], first_record + done, m, reinterpret_cast<uint64_t *>(s->d_host_hits));
        if (rc == BT_OK) e = hipEventRecord(s->scanned[b], s->ctx->stream);
        done += m;
    }
    unsigned long long hits = 0;
    if (e == hipSuccess && rc == BT_OK) e = hipMemcpyAsync(&hits, s->d_host_hits, 8, hipMemcpyDeviceToHost, s->ctx->stream);
    const auto t_loop = std::chrono::steady_clock::now();
    hipError_t e2 = hipStreamSynchronize(s->ctx->stream);
    (void)hipStreamSynchronize(s->copy_stream);
    if (timing)
        fprintf(stderr, "  kmc stream: %.3f s enqueueing (%.3f s reading into the pinned slots, %.3f s waiting for a slot), %.3f s draining; staging set-up before that %.3f s\n",
                std::chrono::duration<double>(t_loop - t_begin).count(), read_s, wait_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count(),
                std::chrono::duration<double>(t_begin - t_enter).count());
    if (rc != BT_OK) return rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail(std::string("bt_kmc_scan_run_host: ") + hipGetErrorString(e));
    int overflowed = 0;
    if (bt_table_status(table, nullptr, nullptr, &overflowed) != BT_OK) return BT_ERR;
    if (overflowed) return fail("bt_kmc_scan_run_host: the count table is full, matched k-mers were dropped (bt_table_reserve before the scan)");
    if (h_hit_count) *h_hit_count = hits;
    return BT_OK;
}

int bt_kmc_scan_make_bloom(bt_kmc_scan *s, bt_bloom *sample_bloom, const uint8_t *d_records, uint64_t first_record, uint64_t n) {
    if (!s || !sample_bloom) return fail("bt_kmc_scan_make_bloom: null argument");
    if (sample_bloom->k != s->k) return fail("bt_kmc_scan_make_bloom: k mismatch");
    if (first_record + n > s->total) return fail("bt_kmc_scan_make_bloom: record range exceeds the database");
    if ((reinterpret_cast<uintptr_t>(d_records) & 15u) != 0) return fail("bt_kmc_scan_make_bloom: d_records must be 16-byte aligned");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(s->ctx->device));
    unsigned grid = grid_for((n + KMC_RECS - 1) / KMC_RECS, 1, s->ctx->num_cu * 8);
    hipLaunchKernelGGL(kmc_make_bloom_kernel, dim3(grid), dim3(BLOCK), 0, s->ctx->stream, make_kmc_view(s), sample_bloom->view(), d_records, first_record, n);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_kmc_scan_decode(bt_kmc_scan *s, const uint8_t *d_records, uint64_t first_record, uint64_t n, uint64_t *d_kmers, uint32_t *d_counts) {
    if (!s || !d_kmers || !d_counts) return fail("bt_kmc_scan_decode: null argument");
    if (first_record + n > s->total) return fail("bt_kmc_scan_decode: record range exceeds the database");
    if ((reinterpret_cast<uintptr_t>(d_records) & 15u) != 0) return fail("bt_kmc_scan_decode: d_records must be 16-byte aligned");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(s->ctx->device));
    unsigned grid = grid_for((n + KMC_RECS - 1) / KMC_RECS, 1, s->ctx->num_cu * 8);
    BloomView nob{};
    TableView not_{};
    hipLaunchKernelGGL(kmc_scan_kernel<true>, dim3(grid), dim3(BLOCK), 0, s->ctx->stream, make_kmc_view(s), nob, not_, 0u, d_records,
                       first_record, n, (unsigned long long *)nullptr, d_kmers, d_counts);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

}  // extern "C"
