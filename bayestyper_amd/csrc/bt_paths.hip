// libbtgpu — path k-mer enumeration over variant-cluster graphs.
//   bt_paths_count_kmers  <- VariantClusterGraph::countPathKmers (VariantClusterGraph.cpp:800-846) + KmerCounter.cpp:252-289
//   bt_paths_classify     <- VariantClusterGraph::classifyPathKmers (:848-939)
//   bt_paths_candidates   <- VariantClusterGraph::getHaplotypeCandidates + updateVariantPathIndices (:941-1184)
//
// The reference walks every best path nucleotide by nucleotide three times, filling per-cluster unordered_maps.  Here every
// best path of every cluster is laid out once as a TEXT in HBM (segments of vertex sequences; a separator before every
// disconnected vertex and after every path so that no k-mer window spans them), the canonical k-mer of every window is
// enumerated once by the sequence kernel, and the maps become two open-addressing indexes built with atomics:
//   A: (cluster, k-mer)       -> first text position (atomicMin), max-over-paths multiplicity
//   B: (k-mer, path)          -> multiplicity on that path
// "first-seen order" of the reference's row numbering is the order of first text positions, recovered with a prefix sum.
// Integer / byte work, HBM random access; no MFMA.
#include "bt_internal.hpp"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <exception>
#include <cstring>
#include <map>
#include <numeric>
#include <thread>

using namespace bt;

namespace {

constexpr unsigned BLOCK = 256;
constexpr uint32_t ST_EMPTY = 0, ST_BUSY = 1, ST_READY = 2;
constexpr uint32_t NOPATH = 0xFFFFFFFFu;

struct Seg {          // one included vertex of one path
    uint64_t dst;     // first text position of its nucleotides
    uint64_t src;     // first nucleotide in the vertex sequence array
    uint32_t len;
    uint32_t nt0;     // num_nucleotides of the path before this vertex
    uint32_t gpath;   // global path id
    uint32_t pad;
};

struct IndexA {       // (cluster, k-mer) -> first position, max multiplicity, list id
    uint64_t *lo, *hi;
    uint32_t *cluster, *state, *maxmult, *list_id;
    unsigned long long *first;
    uint64_t mask;
};
struct IndexB {       // (k-mer, global path) -> count
    uint64_t *lo, *hi;
    uint32_t *gpath, *state, *count, *slot_a;
    uint64_t mask;
};

__device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

// find-or-insert: returns the slot of (lo, hi, tag); tag = cluster (A) or global path (B)
__device__ inline uint64_t index_insert(uint64_t *klo, uint64_t *khi, uint32_t *ktag, uint32_t *state, uint64_t mask, uint64_t lo, uint64_t hi, uint32_t tag) {
    uint64_t idx = mix64(lo ^ mix64(hi + 0x9e3779b97f4a7c15ULL * (tag + 1u))) & mask;
    while (true) {
        uint32_t st = __hip_atomic_load(&state[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (st == ST_EMPTY) {
            const uint32_t prev = atomicCAS(&state[idx], ST_EMPTY, ST_BUSY);
            if (prev == ST_EMPTY) {
                __hip_atomic_store(&klo[idx], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&khi[idx], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ktag[idx], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&state[idx], ST_READY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return idx;
            }
            st = prev;
        }
        if (st == ST_READY) {
            if (__hip_atomic_load(&klo[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lo &&
                __hip_atomic_load(&khi[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == hi &&
                __hip_atomic_load(&ktag[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag)
                return idx;
            idx = (idx + 1) & mask;
        }
        // ST_BUSY: the writer publishes without waiting; poll again
    }
}

// ---- text layout: every position inside a segment gets its nucleotide (ASCII), its global path and its nucleotide index ----
__global__ __launch_bounds__(BLOCK) void text_kernel(const Seg *__restrict__ segs, uint64_t nseg, const uint8_t *__restrict__ seq, uint64_t L,
                                                      char *__restrict__ text, uint32_t *__restrict__ pos_path, uint32_t *__restrict__ pos_nt) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK) {
        // last segment with dst <= pos
        uint64_t lo = 0, hi = nseg;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (segs[mid].dst <= pos) lo = mid + 1;
            else hi = mid;
        }
        char ch = 'N';
        uint32_t gp = NOPATH, nt = 0;
        if (lo > 0) {
            const Seg s = segs[lo - 1];
            if (pos - s.dst < s.len) {
                const uint32_t o = (uint32_t)(pos - s.dst);
                ch = "ACGT"[seq[s.src + o] & 3u];
                gp = s.gpath;
                nt = s.nt0 + o;
            }
        }
        text[pos] = ch;
        pos_path[pos] = gp;
        pos_nt[pos] = nt;
    }
}

__global__ __launch_bounds__(BLOCK) void bloom_insert_valid_kernel(BloomView bloom, const uint64_t *__restrict__ kmers, const uint8_t *__restrict__ valid, uint64_t L) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK)
        if (valid[pos]) bloom_insert(nthash64(Kmer{kmers[2 * pos], kmers[2 * pos + 1]}, bloom.k), bloom);
}

// ---- index build ----
__global__ __launch_bounds__(BLOCK) void index_kernel(IndexA A, IndexB B, const uint64_t *__restrict__ kmers, const uint8_t *__restrict__ valid,
                                                       const uint32_t *__restrict__ pos_path, const uint32_t *__restrict__ path_cluster, uint64_t L,
                                                       uint32_t *__restrict__ pos_slot_a) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK) {
        if (!valid[pos]) continue;
        const uint64_t lo = kmers[2 * pos], hi = kmers[2 * pos + 1];
        const uint32_t gp = pos_path[pos], c = path_cluster[gp];
        const uint64_t a = index_insert(A.lo, A.hi, A.cluster, A.state, A.mask, lo, hi, c);
        atomicMin(&A.first[a], (unsigned long long)pos);
        pos_slot_a[pos] = (uint32_t)a;
        const uint64_t b = index_insert(B.lo, B.hi, B.gpath, B.state, B.mask, lo, hi, gp);
        B.slot_a[b] = (uint32_t)a;   // every inserter of this entry writes the same value
        atomicAdd(&B.count[b], 1u);
    }
}
// max over paths of the saturating per-path multiplicity (VariantClusterGraph.cpp:887-910)
__global__ __launch_bounds__(BLOCK) void maxmult_kernel(IndexA A, IndexB B) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i <= B.mask; i += (uint64_t)gridDim.x * BLOCK) {
        if (B.state[i] != ST_READY) continue;
        const uint32_t cnt = B.count[i] > 255u ? 255u : B.count[i];
        atomicMax(&A.maxmult[B.slot_a[i]], cnt);
    }
}
// dense list of the distinct (cluster, k-mer) entries: order is irrelevant (the table update commutes)
__global__ __launch_bounds__(BLOCK) void list_kernel(IndexA A, uint64_t *__restrict__ list_kmers, uint8_t *__restrict__ list_mult, uint32_t *__restrict__ list_cluster,
                                                      unsigned long long *__restrict__ cursor) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i <= A.mask; i += (uint64_t)gridDim.x * BLOCK) {
        if (A.state[i] != ST_READY) continue;
        const unsigned long long j = atomicAdd(cursor, 1ULL);
        list_kmers[2 * j] = A.lo[i];
        list_kmers[2 * j + 1] = A.hi[i];
        list_mult[j] = (uint8_t)A.maxmult[i];
        list_cluster[j] = A.cluster[i];
        A.list_id[i] = (uint32_t)j;
    }
}
__global__ __launch_bounds__(BLOCK) void classify_tally_kernel(const uint32_t *__restrict__ list_cluster, const uint8_t *__restrict__ excluded, uint64_t n,
                                                                uint32_t *__restrict__ num_path_kmers, uint32_t *__restrict__ has_excluded) {
    for (uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (uint64_t)gridDim.x * BLOCK) {
        atomicAdd(&num_path_kmers[list_cluster[j]], 1u);
        if (excluded[j]) atomicOr(&has_excluded[list_cluster[j]], 1u);
    }
}

// ---- countPathMultigroupKmers: D counts distinct (group, k-mer) pairs, E tracks per k-mer the first group seen and "another group too" ----
// ---- countPathMultigroupKmers in the reference's single-thread order (KmerCounter.cpp:105-145) ----
// The reference walks the groups in index order; a group's distinct path k-mers sit in ONE std::unordered_set<std::bitset<2k>> that
// is clear()ed between groups (its bucket count survives) and are visited in that container's iteration order.  A k-mer the path
// filter reports at that moment goes into the multigroup table — because an earlier group (or unit) holds it, or as a false positive
// of what has been inserted SO FAR — and is otherwise inserted.  Reproduced exactly:
//   1. every distinct (group, k-mer) gets its time t = (k-mers of earlier groups) + (rank in its group's iteration order): the
//      container (libstdc++ _Hashtable, unique keys, std::hash<bitset> = _Hash_bytes, prime rehash policy) is replayed per group by
//      one lane from the k-mers' first occurrences in path order;
//   2. a k-mer is reported at its first time t iff every one of its filter bits was set before t.  A k-mer that is reported is not
//      inserted, but all its bits are set already, so "first time bit b is set" = min t over ALL k-mers having b (0 for bits set by
//      earlier units): no fixed point is needed;
//   3. only k-mers whose bits are ALL either set by earlier units or shared with another k-mer of the unit (a second bit array filled
//      while a scratch copy of the filter takes the unit's k-mers) can qualify — a few per 10^4 —, so the per-bit times are kept in
//      a small hash table over just those k-mers' bits.
constexpr uint32_t SQ_NONE = 0xFFFFFFFFu, SQ_BEFORE = 0xFFFFFFFEu;
// std::hash<std::bitset<2k>>: _Hash_bytes (libstdc++-v3/libsupc++/hash_bytes.cc, 64-bit) over the bitset's first (2k + 7) / 8 bytes, seed 0xc70f6907
__host__ __device__ inline uint64_t std_hash_bitset(uint64_t lo, uint64_t hi, unsigned k) {
    const uint64_t mul = (0xc6a4a793ULL << 32) + 0x5bd1e995ULL;
    const unsigned len = (2 * k + 7) / 8, full = len / 8;
    uint64_t hash = 0xc70f6907ULL ^ (len * mul);
    const uint64_t w[3] = {lo, hi, 0};
    for (unsigned i = 0; i < full; ++i) {
        uint64_t data = w[i] * mul;
        data ^= data >> 47;
        data *= mul;
        hash ^= data;
        hash *= mul;
    }
    if (len & 7u) {
        hash ^= w[full] & ((1ULL << (8u * (len & 7u))) - 1ULL);
        hash *= mul;
    }
    hash ^= hash >> 47;
    hash *= mul;
    hash ^= hash >> 47;
    return hash;
}
// libstdc++'s bucket counts when a container grows one element at a time: 1 -> 13 -> _M_next_bkt(2 B) -> ...
__host__ __device__ inline uint64_t std_next_bucket_count(uint64_t b) {
    const uint64_t chain[31] = {13ull, 29ull, 59ull, 127ull, 257ull, 541ull, 1109ull, 2357ull, 5087ull, 10273ull, 20753ull, 42043ull, 85229ull, 172933ull, 351061ull, 712697ull,
                                1447153ull, 2938679ull, 5967347ull, 12117689ull, 24607243ull, 49969847ull, 101473717ull, 206062531ull, 418451333ull, 849749479ull,
                                1725587117ull, 3504151727ull, 8589934583ull, 25769803693ull, 68719476731ull};
    for (int i = 0; i < 31; ++i)
        if (chain[i] > b) return chain[i];
    return chain[30];
}

// first pass: the (k-mer, group) index D with the first text position of every entry, the k-mer index E with "seen in two groups"
__global__ __launch_bounds__(BLOCK) void multigroup_kernel(uint64_t *__restrict__ dlo, uint64_t *__restrict__ dhi, uint32_t *__restrict__ dtag, uint32_t *__restrict__ dstate,
                                                            uint32_t *__restrict__ dfirst, uint32_t *__restrict__ pos_d, uint64_t *__restrict__ elo, uint64_t *__restrict__ ehi,
                                                            uint32_t *__restrict__ etag, uint32_t *__restrict__ estate, uint32_t *__restrict__ egroup,
                                                            uint32_t *__restrict__ emulti, uint64_t mask, const uint64_t *__restrict__ kmers, const uint8_t *__restrict__ valid,
                                                            const uint32_t *__restrict__ pos_path, const uint32_t *__restrict__ path_cluster,
                                                            const uint32_t *__restrict__ cluster_group, uint64_t L) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK) {
        if (!valid[pos]) continue;
        const uint64_t lo = kmers[2 * pos], hi = kmers[2 * pos + 1];
        const uint32_t g = cluster_group[path_cluster[pos_path[pos]]];
        const uint64_t d = index_insert(dlo, dhi, dtag, dstate, mask, lo, hi, g);
        atomicMin(&dfirst[d], (uint32_t)pos);
        pos_d[pos] = (uint32_t)d;
        const uint64_t e = index_insert(elo, ehi, etag, estate, mask, lo, hi, 0u);
        const uint32_t prev = atomicCAS(&egroup[e], 0xFFFFFFFFu, g);
        if (prev != 0xFFFFFFFFu && prev != g) emulti[e] = 1u;
    }
}
__global__ __launch_bounds__(BLOCK) void mg_flag_kernel(const uint8_t *__restrict__ valid, const uint32_t *__restrict__ pos_d, const uint32_t *__restrict__ dfirst, uint64_t L,
                                                         uint32_t *__restrict__ flag) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK)
        flag[pos] = (valid[pos] && dfirst[pos_d[pos]] == (uint32_t)pos) ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void mg_seq_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ rowpos, uint64_t L, uint32_t *__restrict__ seq_pos) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK)
        if (flag[pos]) seq_pos[rowpos[pos]] = (uint32_t)pos;
}
// One lane per group: replay the inserts of the group's distinct k-mers (first occurrences, path order) into an unordered_set that
// starts with `b_init[g]` buckets (what earlier groups left behind), then number the nodes in iteration order.
// libstdc++ _Hashtable: a node enters at the front of its bucket (behind the bucket's "before" node) or, for an empty bucket, at the
// front of the whole list; a rehash re-threads the list front to back the same way (hashtable.h: _M_insert_bucket_begin, _M_rehash_aux).
// The bucket array itself is not materialised (a small group may inherit millions of buckets from a large earlier one): the buckets in
// use live in a per-group open-addressing map bucket -> "before" node with room for twice the group's k-mers.
// key(node, lo, hi) hands out the node's k-mer; rank_out(node, rank) receives its position in iteration order.  Shared by the kernel and
// by bt_diag_kmer_set_order (the same code run on the host against the real container in the CPU tests).
template <typename KeyFn, typename RankFn>
__host__ __device__ inline void replay_kmer_set(uint32_t n, uint64_t b_init, unsigned k, uint32_t *nx, uint32_t *mk, uint32_t *mv, uint32_t mmask, KeyFn key, RankFn rank_out) {
    uint64_t B = b_init, next_resize = B > 1 ? B : 0;
    uint32_t head = SQ_NONE;
    auto bucket_of = [&](uint32_t node) {
        uint64_t lo, hi;
        key(node, lo, hi);
        return (uint32_t)(std_hash_bitset(lo, hi, k) % B);
    };
    auto map_clear = [&]() {
        for (uint32_t i = 0; i <= mmask; ++i) mk[i] = SQ_NONE;
    };
    auto map_slot = [&](uint32_t b) {   // slot of bucket b (inserted empty when absent)
        uint32_t i = (b * 2654435761u) & mmask;
        while (mk[i] != b) {
            if (mk[i] == SQ_NONE) {
                mk[i] = b;
                mv[i] = SQ_NONE;
                break;
            }
            i = (i + 1) & mmask;
        }
        return i;
    };
    map_clear();
    for (uint32_t e = 0; e < n; ++e) {
        if ((uint64_t)e + 1 > next_resize) {   // _Prime_rehash_policy::_M_need_rehash(B, e, 1), max_load_factor 1
            uint64_t min_bkts = (uint64_t)e + 1;
            if (next_resize == 0 && min_bkts < 11) min_bkts = 11;
            if (min_bkts >= B) {
                B = std_next_bucket_count(B);   // _M_next_bkt(max(min_bkts + 1, 2 B)) for one-at-a-time growth
                next_resize = B;
                map_clear();
                uint32_t p = head, bbegin = SQ_NONE;   // bbegin: map slot of the bucket that currently begins the list
                head = SQ_NONE;
                while (p != SQ_NONE) {
                    const uint32_t nxt = nx[p], sl = map_slot(bucket_of(p));
                    if (mv[sl] == SQ_NONE) {
                        nx[p] = head;
                        head = p;
                        mv[sl] = SQ_BEFORE;
                        if (nx[p] != SQ_NONE) mv[bbegin] = p;
                        bbegin = sl;
                    } else {
                        const uint32_t prev = mv[sl];
                        if (prev == SQ_BEFORE) {
                            nx[p] = head;
                            head = p;
                        } else {
                            nx[p] = nx[prev];
                            nx[prev] = p;
                        }
                    }
                    p = nxt;
                }
            } else
                next_resize = B;
        }
        const uint32_t sl = map_slot(bucket_of(e));
        if (mv[sl] != SQ_NONE) {
            const uint32_t prev = mv[sl];
            if (prev == SQ_BEFORE) {
                nx[e] = head;
                head = e;
            } else {
                nx[e] = nx[prev];
                nx[prev] = e;
            }
        } else {
            nx[e] = head;
            head = e;
            if (nx[e] != SQ_NONE) mv[map_slot(bucket_of(nx[e]))] = e;
            mv[sl] = SQ_BEFORE;
        }
    }
    uint32_t rank = 0;
    for (uint32_t p = head; p != SQ_NONE; p = nx[p]) rank_out(p, rank++);
}
__global__ __launch_bounds__(BLOCK) void mg_order_kernel(const uint64_t *__restrict__ kmers, const uint32_t *__restrict__ seq_pos, const uint32_t *__restrict__ goff,
                                                          const uint64_t *__restrict__ moff, const uint64_t *__restrict__ b_init, uint32_t *__restrict__ next,
                                                          uint32_t *__restrict__ map_key, uint32_t *__restrict__ map_val, uint32_t *__restrict__ time_of_seq, uint32_t G,
                                                          unsigned k) {
    const uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
    if (g >= G) return;
    const uint32_t i0 = goff[g], n = goff[g + 1] - i0;
    if (n == 0) return;
#ifdef BT_MG_INSERTION_ORDER   // (test of the tests: with insertion order instead of the container's order the parity test must fail)
    for (uint32_t p = 0; p < n; ++p) time_of_seq[i0 + p] = i0 + p + 1u;
    return;
#endif
    replay_kmer_set(
        n, b_init[g], k, next + i0, map_key + moff[g], map_val + moff[g], (uint32_t)(moff[g + 1] - moff[g]) - 1u,
        [&](uint32_t node, uint64_t &lo, uint64_t &hi) {
            const uint64_t pos = seq_pos[i0 + node];
            lo = kmers[2 * pos];
            hi = kmers[2 * pos + 1];
        },
        [&](uint32_t node, uint32_t rank) { time_of_seq[i0 + node] = i0 + rank + 1u; });   // times start at 1: 0 = "set by an earlier unit"
}
// first time of every distinct k-mer of the unit
__global__ __launch_bounds__(BLOCK) void mg_etime_kernel(const uint64_t *__restrict__ kmers, const uint32_t *__restrict__ seq_pos, const uint32_t *__restrict__ time_of_seq,
                                                          uint64_t n, uint64_t *__restrict__ elo, uint64_t *__restrict__ ehi, uint32_t *__restrict__ etag,
                                                          uint32_t *__restrict__ estate, uint64_t mask, uint32_t *__restrict__ etime) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t pos = seq_pos[i];
        const uint64_t e = index_insert(elo, ehi, etag, estate, mask, kmers[2 * pos], kmers[2 * pos + 1], 0u);
        atomicMin(&etime[e], time_of_seq[i]);
    }
}
__device__ inline void bloom_word_bit(uint64_t h, unsigned i, uint64_t base, const BloomView &b, uint64_t &word, uint32_t &m, uint64_t &bit_id) {
    const uint64_t pos = bloom_probe_pos(h, i, b), byte = base + (pos >> 3);
    word = byte >> 2;
    m = 1u << (uint32_t)((byte & 3u) * 8u + (7u - (unsigned)(pos & 7u)));
    bit_id = base * 8u + pos;
}
// scratch filter f1 takes the unit's distinct k-mers; f2 = bits that were set already when a k-mer set them (shared with another k-mer)
__global__ __launch_bounds__(BLOCK) void mg_shared_bits_kernel(BloomView bv, const uint64_t *__restrict__ elo, const uint64_t *__restrict__ ehi, const uint32_t *__restrict__ estate,
                                                                uint64_t mask, uint32_t *__restrict__ f1, uint32_t *__restrict__ f2) {
    for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e <= mask; e += (uint64_t)gridDim.x * BLOCK) {
        if (estate[e] != ST_READY) continue;
        const uint64_t h = nthash64(Kmer{elo[e], ehi[e]}, bv.k), base = bloom_sub_base(h, bv);
        for (unsigned i = 0; i < bv.num_hashes; ++i) {
            uint64_t w, id;
            uint32_t m;
            bloom_word_bit(h, i, base, bv, w, m, id);
            if (atomicOr(&f1[w], m) & m) atomicOr(&f2[w], m);
        }
    }
}
// candidates: not yet multigroup, every bit set by an earlier unit or shared inside the unit.  out == nullptr: count only.
__global__ __launch_bounds__(BLOCK) void mg_candidates_kernel(BloomView bv, const uint64_t *__restrict__ elo, const uint64_t *__restrict__ ehi, const uint32_t *__restrict__ estate,
                                                               const uint32_t *__restrict__ emulti, uint64_t mask, const uint32_t *__restrict__ f2,
                                                               unsigned long long *__restrict__ cursor, uint64_t *__restrict__ out) {
    for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e <= mask; e += (uint64_t)gridDim.x * BLOCK) {
        if (estate[e] != ST_READY || emulti[e]) continue;
        const uint64_t h = nthash64(Kmer{elo[e], ehi[e]}, bv.k), base = bloom_sub_base(h, bv);
        bool all = true;
        for (unsigned i = 0; i < bv.num_hashes && all; ++i) {
            uint64_t w, id;
            uint32_t m;
            bloom_word_bit(h, i, base, bv, w, m, id);
            all = ((bv.words[w] | f2[w]) & m) != 0;
        }
        if (all) {
            const unsigned long long j = atomicAdd(cursor, 1ULL);
            if (out) out[j] = e;
        }
    }
}
__device__ inline uint64_t cb_find_or_insert(unsigned long long *keys, uint64_t mask, uint64_t id, bool insert) {   // returns the slot, or ~0 if absent
    uint64_t i = mix64(id) & mask;
    while (true) {
        unsigned long long cur = __hip_atomic_load(&keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == ~0ULL) {
            if (!insert) return ~0ULL;
            cur = atomicCAS(&keys[i], ~0ULL, (unsigned long long)id);
            if (cur == ~0ULL) return i;
        }
        if (cur == id) return i;
        i = (i + 1) & mask;
    }
}
__global__ __launch_bounds__(BLOCK) void mg_cand_bits_kernel(BloomView bv, const uint64_t *__restrict__ elo, const uint64_t *__restrict__ ehi, const uint64_t *__restrict__ cand,
                                                              uint64_t ncand, unsigned long long *__restrict__ cb_keys, uint32_t *__restrict__ cb_time, uint64_t cb_mask) {
    for (uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; j < ncand; j += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t e = cand[j], h = nthash64(Kmer{elo[e], ehi[e]}, bv.k), base = bloom_sub_base(h, bv);
        for (unsigned i = 0; i < bv.num_hashes; ++i) {
            uint64_t w, id;
            uint32_t m;
            bloom_word_bit(h, i, base, bv, w, m, id);
            const uint64_t slot = cb_find_or_insert(cb_keys, cb_mask, id, true);
            if (bv.words[w] & m) atomicMin(&cb_time[slot], 0u);   // set by an earlier unit
        }
    }
}
__global__ __launch_bounds__(BLOCK) void mg_bit_times_kernel(BloomView bv, const uint64_t *__restrict__ elo, const uint64_t *__restrict__ ehi, const uint32_t *__restrict__ estate,
                                                              const uint32_t *__restrict__ etime, uint64_t mask, unsigned long long *__restrict__ cb_keys,
                                                              uint32_t *__restrict__ cb_time, uint64_t cb_mask) {
    for (uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; e <= mask; e += (uint64_t)gridDim.x * BLOCK) {
        if (estate[e] != ST_READY) continue;
        const uint64_t h = nthash64(Kmer{elo[e], ehi[e]}, bv.k), base = bloom_sub_base(h, bv);
        for (unsigned i = 0; i < bv.num_hashes; ++i) {
            uint64_t w, id;
            uint32_t m;
            bloom_word_bit(h, i, base, bv, w, m, id);
            const uint64_t slot = cb_find_or_insert(cb_keys, cb_mask, id, false);
            if (slot != ~0ULL) atomicMin(&cb_time[slot], etime[e]);
        }
    }
}
__global__ __launch_bounds__(BLOCK) void mg_reported_kernel(BloomView bv, const uint64_t *__restrict__ elo, const uint64_t *__restrict__ ehi, const uint32_t *__restrict__ etime,
                                                             const uint64_t *__restrict__ cand, uint64_t ncand, unsigned long long *__restrict__ cb_keys,
                                                             const uint32_t *__restrict__ cb_time, uint64_t cb_mask, uint32_t *__restrict__ emulti) {
    for (uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; j < ncand; j += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t e = cand[j], h = nthash64(Kmer{elo[e], ehi[e]}, bv.k), base = bloom_sub_base(h, bv);
        uint32_t latest = 0;
        for (unsigned i = 0; i < bv.num_hashes; ++i) {
            uint64_t w, id;
            uint32_t m;
            bloom_word_bit(h, i, base, bv, w, m, id);
            const uint32_t t = cb_time[cb_find_or_insert(cb_keys, cb_mask, id, false)];
            latest = t > latest ? t : latest;
        }
        if (latest < etime[e]) emulti[e] = 1u;   // every bit was set before the k-mer's first turn: the filter reports it
    }
}
__global__ __launch_bounds__(BLOCK) void multigroup_list_kernel(const uint64_t *__restrict__ elo, const uint64_t *__restrict__ ehi, const uint32_t *__restrict__ estate,
                                                                 const uint32_t *__restrict__ emulti, const uint32_t *__restrict__ dstate, uint64_t mask,
                                                                 uint64_t *__restrict__ out, unsigned long long *__restrict__ counters /* [0] multigroup, [1] (group,kmer) pairs */) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i <= mask; i += (uint64_t)gridDim.x * BLOCK) {
        if (dstate[i] == ST_READY) atomicAdd(&counters[1], 1ULL);
        if (estate[i] == ST_READY && emulti[i]) {
            const unsigned long long j = atomicAdd(&counters[0], 1ULL);
            out[2 * j] = elo[i];
            out[2 * j + 1] = ehi[i];
        }
    }
}

// ---- candidates ----
// per distinct (cluster, k-mer): table record -> excluded?, multicluster?; list_flags[j]: bit0 in table, bit1 excluded, bit2 multicluster
__global__ __launch_bounds__(BLOCK) void record_kernel(TableView t, const int64_t *__restrict__ slots, uint64_t n, uint8_t *__restrict__ list_flags) {
    for (uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (uint64_t)gridDim.x * BLOCK) {
        uint8_t f = 0;
        if (slots[j] >= 0) {
            const uint32_t flags = *t.meta((uint64_t)slots[j]) & 0xffu;
            f = 1;
            if (flags & (BT_KC_DECOY_OCC | BT_KC_MAX_MULTIPLICITY | BT_KC_MULTIGROUP_OCC)) f |= 2;   // isExcluded (KmerCounts.cpp:93-96)
            if (flags & BT_KC_MULTICLUSTER_OCC) f |= 4;
        }
        list_flags[j] = f;
    }
}
// 1 at the first text position of every non-excluded distinct (cluster, k-mer): its prefix sum is the reference's row numbering
__global__ __launch_bounds__(BLOCK) void first_flag_kernel(IndexA A, const uint8_t *__restrict__ valid, const uint32_t *__restrict__ pos_slot_a,
                                                            const uint8_t *__restrict__ list_flags, uint64_t L, uint32_t *__restrict__ flag) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK) {
        uint32_t f = 0;
        if (valid[pos]) {
            const uint32_t a = pos_slot_a[pos];
            f = (A.first[a] == pos && !(list_flags[A.list_id[a]] & 2)) ? 1u : 0u;
        }
        flag[pos] = f;
    }
}
// block-wise exclusive scan: 1024 elements per workgroup (4 per lane); block totals to `sums`
__global__ __launch_bounds__(BLOCK) void scan_block_kernel(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out, uint32_t *__restrict__ sums) {
    __shared__ uint32_t part[BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * (BLOCK * 4) + (uint64_t)threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = base + q < n ? in[base + q] : 0u;
        s += v[q];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (unsigned off = 1; off < BLOCK; off <<= 1) {   // Hillis-Steele over the 256 lane totals
        const uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (base + q < n) out[base + q] = run;
        run += v[q];
    }
    if (threadIdx.x == BLOCK - 1) sums[blockIdx.x] = part[BLOCK - 1];
}
__global__ __launch_bounds__(BLOCK) void scan_add_kernel(uint32_t *__restrict__ out, uint64_t n, const uint32_t *__restrict__ block_off) {
    const uint64_t base = (uint64_t)blockIdx.x * (BLOCK * 4) + (uint64_t)threadIdx.x * 4;
    const uint32_t add = block_off[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (base + q < n) out[base + q] += add;
}
// per row: key, table record
__global__ __launch_bounds__(BLOCK) void rows_kernel(IndexA A, TableView t, const uint8_t *__restrict__ valid, const uint32_t *__restrict__ pos_slot_a,
                                                      const uint32_t *__restrict__ flag, const uint32_t *__restrict__ row_of_pos, const int64_t *__restrict__ slots,
                                                      const uint8_t *__restrict__ list_flags, uint64_t L, uint32_t S, uint32_t *__restrict__ a_row,
                                                      uint64_t *__restrict__ row_key, uint8_t *__restrict__ row_flags, uint8_t *__restrict__ row_counts,
                                                      uint8_t *__restrict__ row_ic) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK) {
        if (!flag[pos]) continue;
        const uint32_t a = pos_slot_a[pos], row = row_of_pos[pos], j = A.list_id[a];
        a_row[a] = row;
        row_key[2 * (uint64_t)row] = A.lo[a];
        row_key[2 * (uint64_t)row + 1] = A.hi[a];
        row_flags[row] = list_flags[j];
        const int64_t slot = slots[j];
        for (uint32_t s = 0; s < S; ++s) row_counts[(uint64_t)row * S + s] = slot >= 0 ? t.count_bytes((uint64_t)slot)[s] : (uint8_t)0;
        const uint32_t meta = slot >= 0 ? *t.meta((uint64_t)slot) : 0u;
        row_ic[2 * (uint64_t)row] = (uint8_t)((meta >> 16) & 0xffu);
        row_ic[2 * (uint64_t)row + 1] = (uint8_t)((meta >> 24) & 0xffu);
    }
}
// haplotype_kmer_multiplicities(row, path) = multiplicity of the k-mer on that path (one writer per cell)
__global__ __launch_bounds__(BLOCK) void mult_kernel(IndexA A, IndexB B, const uint8_t *__restrict__ list_flags, const uint32_t *__restrict__ a_row,
                                                      const uint32_t *__restrict__ path_cluster, const uint32_t *__restrict__ path_local,
                                                      const uint32_t *__restrict__ cluster_row0, const uint64_t *__restrict__ cluster_mult0,
                                                      const uint32_t *__restrict__ cluster_h, uint8_t *__restrict__ mult, uint32_t *__restrict__ over127) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i <= B.mask; i += (uint64_t)gridDim.x * BLOCK) {
        if (B.state[i] != ST_READY) continue;
        const uint32_t a = B.slot_a[i];
        if (list_flags[A.list_id[a]] & 2) continue;
        const uint32_t gp = B.gpath[i], c = path_cluster[gp];
        const uint32_t local_row = a_row[a] - cluster_row0[c];
        if (B.count[i] > 127u) atomicOr(over127, 1u);   // the reference asserts <= 127 (VariantClusterGraph.cpp:1060)
        mult[cluster_mult0[c] + (uint64_t)local_row * cluster_h[c] + path_local[gp]] = (uint8_t)B.count[i];
    }
}
// updateVariantPathIndices: one (row, variant, path) triple per window and running variant that covers its last nucleotide
struct Interval {
    uint32_t first, second;
    uint16_t variant, pad;
};
__global__ __launch_bounds__(BLOCK) void triples_kernel(IndexA A, const uint8_t *__restrict__ valid, const uint32_t *__restrict__ pos_slot_a,
                                                         const uint8_t *__restrict__ list_flags, const uint32_t *__restrict__ a_row, const uint32_t *__restrict__ pos_path,
                                                         const uint32_t *__restrict__ pos_nt, const uint32_t *__restrict__ path_local,
                                                         const uint32_t *__restrict__ iv_off, const Interval *__restrict__ iv, uint64_t L,
                                                         unsigned long long *__restrict__ cursor, uint64_t *__restrict__ out /* null: count only */) {
    for (uint64_t pos = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; pos < L; pos += (uint64_t)gridDim.x * BLOCK) {
        if (!valid[pos]) continue;
        const uint32_t a = pos_slot_a[pos];
        if (list_flags[A.list_id[a]] & 2) continue;
        const uint32_t gp = pos_path[pos], nt = pos_nt[pos];
        for (uint32_t e = iv_off[gp]; e < iv_off[gp + 1]; ++e) {
            if (iv[e].first <= nt && nt < iv[e].second) {
                const unsigned long long j = atomicAdd(cursor, 1ULL);
                if (out) out[j] = ((uint64_t)a_row[a] << 32) | ((uint64_t)iv[e].variant << 16) | path_local[gp];
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void gather_u32_kernel(const uint32_t *__restrict__ in, const uint64_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}

template <typename T>
int dev_alloc(T **p, uint64_t n, std::vector<void *> &owned) {
    *p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), std::max<uint64_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return fail(std::string("bt_paths: device allocation of ") + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e));
    owned.push_back(*p);
    return BT_OK;
}

uint64_t pow2_at_least(uint64_t n) {
    uint64_t c = 16;
    while (c < n) c <<= 1;
    return c;
}

}  // namespace

struct bt_paths {
    bt_ctx *ctx = nullptr;
    uint32_t k = 0, C = 0;
    // host copies of what the host-side half needs
    std::vector<uint32_t> vertex_off, num_paths, vertex_nested, refvar_off, var_off;
    std::vector<uint64_t> seq_off, path_off;
    std::vector<uint16_t> vertex_variant, vertex_allele, refvar, var_num_alleles;
    std::vector<uint8_t> vertex_flags, path_vertices, var_has_dependency;
    std::vector<uint32_t> path_cluster, path_local, cluster_path0;   // global path ids
    uint64_t num_gpaths = 0, L = 0, num_valid = 0;
    std::vector<void *> owned;
    // device
    uint8_t *d_seq = nullptr;
    char *d_text = nullptr;
    uint32_t *d_pos_path = nullptr, *d_pos_nt = nullptr, *d_pos_slot_a = nullptr;
    uint64_t *d_kmers = nullptr;
    uint8_t *d_valid = nullptr;
    uint32_t *d_path_cluster = nullptr, *d_path_local = nullptr, *d_iv_off = nullptr;
    Interval *d_iv = nullptr;
    IndexA A{};
    IndexB B{};
    bool indexed = false;
    uint64_t n_list = 0;
    uint64_t *d_list_kmers = nullptr;
    uint8_t *d_list_mult = nullptr, *d_list_excluded = nullptr, *d_list_flags = nullptr;
    uint32_t *d_list_cluster = nullptr;
    int64_t *d_list_slots = nullptr;
    std::vector<uint64_t> cluster_text0;   // first text position of each cluster (+ L at the end)
    // candidates result (host)
    bool have_candidates = false;
    uint32_t S = 0;
    std::vector<uint32_t> kmer_off, kv_off, unique_off, unique_idx, multi_off, multi_idx, hapnest_off, hapnest_idx, nestdep_off, nestdep_cluster, nestdep_var_off, kv_bits;
    std::vector<uint8_t> mult, has_counts, counts, ic;
    std::vector<uint64_t> key;
    std::vector<uint16_t> kv_var, hap_allele, nestdep_var;
};

namespace {

// host threads of the assembly steps: BT_HOST_THREADS (the executable sets it from -p), else up to 16; never more than one per 256 items
unsigned host_threads(uint64_t items) {
    unsigned t = 0;
    if (const char *e = getenv("BT_HOST_THREADS")) t = (unsigned)std::max(1, atoi(e));
    else t = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(t, items / 256 + 1));
}
// fn(t) for t in [0, T) on T threads (the caller's included); the first exception of any of them is rethrown after all have been joined
template <typename F>
void run_on_threads(unsigned T, F &&fn) {
    if (T <= 1) {
        fn(0u);
        return;
    }
    std::vector<std::exception_ptr> err(T);
    auto guarded = [&](unsigned t) {
        try {
            fn(t);
        } catch (...) {
            err[t] = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(guarded, t);
    } catch (...) {   // (a thread could not be started: the ranges of the missing threads are run here)
        for (unsigned t = (unsigned)pool.size() + 1; t < T; ++t) guarded(t);
    }
    guarded(0u);
    for (auto &th : pool) th.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

// Host half of getHaplotypeCandidates for one path: running-variant intervals in path-nucleotide coordinates
// (VariantClusterGraph.cpp:984-1011), haplotype allele indices (:990-996,1095-1102), nested cluster list (:1013-1017,1093)
void walk_path_host(const bt_paths &p, uint32_t c, uint32_t lp, std::vector<Interval> &iv, std::vector<uint16_t> &alleles, std::vector<uint32_t> &nested) {
    const uint32_t v0 = p.vertex_off[c], nv = p.vertex_off[c + 1] - v0, V = p.var_off[c + 1] - p.var_off[c];
    const uint8_t *row = p.path_vertices.data() + p.path_off[c] + (uint64_t)lp * nv;
    alleles.assign(V, 0xFFFF);
    nested.clear();
    std::map<std::pair<uint16_t, uint16_t>, size_t> open;   // (variant, allele) -> index into iv of its current interval
    uint32_t nn = 0;
    for (uint32_t vi = 0; vi < nv; ++vi) {
        if (!row[vi]) continue;
        const uint32_t v = v0 + vi;
        const uint32_t len = (uint32_t)(p.seq_off[v + 1] - p.seq_off[v]);
        if (p.vertex_variant[v] != 0xFFFF) {
            if (!(p.vertex_flags[v] & 1)) alleles[p.vertex_variant[v]] = p.vertex_allele[v];
            const auto key = std::make_pair(p.vertex_variant[v], p.vertex_allele[v]);
            auto it = open.find(key);
            // the reference's emplace keeps an existing entry only while it is still alive, i.e. ends exactly at nn + k - 1
            if (it == open.end() || iv[it->second].second != nn + p.k - 1) {
                iv.push_back(Interval{nn + ((p.vertex_flags[v] >> 1) & 1u), nn + p.k - 1, p.vertex_variant[v], 0});
                open[key] = iv.size() - 1;
                it = open.find(key);
            }
            iv[it->second].second += len;
        }
        for (uint32_t r = p.refvar_off[v]; r < p.refvar_off[v + 1]; ++r) {
            auto it = open.find(std::make_pair(p.refvar[r], (uint16_t)0));
            if (it != open.end() && iv[it->second].second == nn + p.k - 1) iv[it->second].second += len;
        }
        if (p.vertex_nested[v] != 0xFFFFFFFFu) nested.push_back(p.vertex_nested[v]);
        nn += len;
    }
    std::sort(nested.begin(), nested.end());
    for (uint32_t var = 0; var < V; ++var)
        if (alleles[var] == 0xFFFF) alleles[var] = (uint16_t)(p.var_num_alleles[p.var_off[c] + var] - 1);
}

int build_index(bt_paths *p) {
    if (p->indexed) return BT_OK;
    const uint64_t cap = pow2_at_least(2 * std::max<uint64_t>(p->num_valid, 8));
    IndexA &A = p->A;
    IndexB &B = p->B;
#define TRY(x)                         \
    do {                               \
        const int _rc = (x);           \
        if (_rc != BT_OK) return _rc;  \
    } while (0)
    TRY(dev_alloc(&A.lo, cap, p->owned));
    TRY(dev_alloc(&A.hi, cap, p->owned));
    TRY(dev_alloc(&A.cluster, cap, p->owned));
    TRY(dev_alloc(&A.state, cap, p->owned));
    TRY(dev_alloc(&A.maxmult, cap, p->owned));
    TRY(dev_alloc(&A.list_id, cap, p->owned));
    TRY(dev_alloc(&A.first, cap, p->owned));
    A.mask = cap - 1;
    TRY(dev_alloc(&B.lo, cap, p->owned));
    TRY(dev_alloc(&B.hi, cap, p->owned));
    TRY(dev_alloc(&B.gpath, cap, p->owned));
    TRY(dev_alloc(&B.state, cap, p->owned));
    TRY(dev_alloc(&B.count, cap, p->owned));
    TRY(dev_alloc(&B.slot_a, cap, p->owned));
    B.mask = cap - 1;
    TRY(dev_alloc(&p->d_pos_slot_a, p->L, p->owned));
    hipStream_t st = p->ctx->stream;
    BT_HIP(hipMemsetAsync(A.state, 0, cap * 4, st));
    BT_HIP(hipMemsetAsync(A.maxmult, 0, cap * 4, st));
    BT_HIP(hipMemsetAsync(A.first, 0xFF, cap * 8, st));
    BT_HIP(hipMemsetAsync(B.state, 0, cap * 4, st));
    BT_HIP(hipMemsetAsync(B.count, 0, cap * 4, st));
    const unsigned maxb = p->ctx->num_cu * 16;
    hipLaunchKernelGGL(index_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, A, B, p->d_kmers, p->d_valid, p->d_pos_path, p->d_path_cluster, p->L,
                       p->d_pos_slot_a);
    BT_CHECK_LAUNCH();
    hipLaunchKernelGGL(maxmult_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, A, B);
    BT_CHECK_LAUNCH();
    // dense list of the distinct (cluster, k-mer) entries
    unsigned long long *d_cursor = nullptr;
    TRY(dev_alloc(&d_cursor, 1, p->owned));
    BT_HIP(hipMemsetAsync(d_cursor, 0, 8, st));
    TRY(dev_alloc(&p->d_list_kmers, 2 * p->num_valid, p->owned));
    TRY(dev_alloc(&p->d_list_mult, p->num_valid, p->owned));
    TRY(dev_alloc(&p->d_list_cluster, p->num_valid, p->owned));
    TRY(dev_alloc(&p->d_list_excluded, p->num_valid, p->owned));
    TRY(dev_alloc(&p->d_list_flags, p->num_valid, p->owned));
    TRY(dev_alloc(&p->d_list_slots, p->num_valid, p->owned));
    hipLaunchKernelGGL(list_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, A, p->d_list_kmers, p->d_list_mult, p->d_list_cluster, d_cursor);
    BT_CHECK_LAUNCH();
    unsigned long long n = 0;
    BT_HIP(hipMemcpyAsync(&n, d_cursor, 8, hipMemcpyDeviceToHost, st));
    BT_HIP(hipStreamSynchronize(st));
    p->n_list = n;
    p->indexed = true;
    return BT_OK;
#undef TRY
}

}  // namespace

extern "C" {

int bt_paths_create(bt_ctx *ctx, const bt_paths_batch *b, uint32_t k, bt_paths **out, uint64_t *h_num_kmer_occurrences) {
    if (!ctx || !b || !out) return fail("bt_paths_create: null argument");
    if (k < 1 || k > 64) return fail("bt_paths_create: k must be in 1..64");
    if (b->num_clusters == 0) return fail("bt_paths_create: empty batch");
    BT_HIP(hipSetDevice(ctx->device));
    bt_paths *p = new bt_paths();
    p->ctx = ctx;
    p->k = k;
    p->C = b->num_clusters;
    const uint32_t C = p->C, NV = b->vertex_off[C];
    p->vertex_off.assign(b->vertex_off, b->vertex_off + C + 1);
    p->num_paths.assign(b->num_paths, b->num_paths + C);
    p->seq_off.assign(b->seq_off, b->seq_off + NV + 1);
    p->vertex_variant.assign(b->vertex_variant, b->vertex_variant + NV);
    p->vertex_allele.assign(b->vertex_allele, b->vertex_allele + NV);
    p->vertex_flags.assign(b->vertex_flags, b->vertex_flags + NV);
    p->vertex_nested.assign(b->vertex_nested, b->vertex_nested + NV);
    p->refvar_off.assign(b->refvar_off, b->refvar_off + NV + 1);
    p->refvar.assign(b->refvar, b->refvar + p->refvar_off[NV]);
    p->path_off.assign(b->path_off, b->path_off + C + 1);
    p->path_vertices.assign(b->path_vertices, b->path_vertices + p->path_off[C]);
    p->var_off.assign(b->var_off, b->var_off + C + 1);
    p->var_num_alleles.assign(b->var_num_alleles, b->var_num_alleles + p->var_off[C]);
    p->var_has_dependency.assign(b->var_has_dependency, b->var_has_dependency + p->var_off[C]);
    // ---- segments: text layout of every path ----
    std::vector<Seg> segs;
    std::vector<uint32_t> iv_off(1, 0);
    std::vector<Interval> iv;
    uint64_t at = 0;
    p->cluster_path0.assign(C + 1, 0);
    std::vector<uint16_t> alleles;
    std::vector<uint32_t> nested;
    for (uint32_t c = 0; c < C; ++c) {
        const uint32_t v0 = p->vertex_off[c], nv = p->vertex_off[c + 1] - v0;
        if (p->path_off[c + 1] - p->path_off[c] != (uint64_t)p->num_paths[c] * nv) {
            delete p;
            return fail("bt_paths_create: path_off does not match num_paths x vertices");
        }
        p->cluster_text0.push_back(at);
        p->cluster_path0[c] = (uint32_t)p->path_cluster.size();
        for (uint32_t lp = 0; lp < p->num_paths[c]; ++lp) {
            const uint32_t gp = (uint32_t)p->path_cluster.size();
            p->path_cluster.push_back(c);
            p->path_local.push_back(lp);
            const uint8_t *row = p->path_vertices.data() + p->path_off[c] + (uint64_t)lp * nv;
            uint32_t nn = 0;
            for (uint32_t vi = 0; vi < nv; ++vi) {
                if (!row[vi]) continue;
                const uint32_t v = v0 + vi;
                const uint32_t len = (uint32_t)(p->seq_off[v + 1] - p->seq_off[v]);
                if (p->vertex_flags[v] & 1) ++at;   // kmer_pair.reset(): one separator position
                if (len) segs.push_back(Seg{at, p->seq_off[v], len, nn, gp, 0});
                at += len;
                nn += len;
            }
            ++at;   // separator between paths
            walk_path_host(*p, c, lp, iv, alleles, nested);
            iv_off.push_back((uint32_t)iv.size());
        }
    }
    p->cluster_path0[C] = (uint32_t)p->path_cluster.size();
    p->cluster_text0.push_back(at);
    p->num_gpaths = p->path_cluster.size();
    p->L = at;
    if (p->L >= (1ull << 32)) {
        delete p;
        return fail("bt_paths_create: more than 2^32 path nucleotides in one batch (split the unit)");
    }
    // ---- upload, build the text, enumerate the k-mers ----
    Seg *d_segs = nullptr;
    int rc = BT_OK;
    auto up = [&](auto **dst, const auto &vec) {
        if (rc != BT_OK) return;
        rc = dev_alloc(dst, vec.size(), p->owned);
        if (rc == BT_OK && !vec.empty() && hipMemcpy(*dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice) != hipSuccess) rc = fail("bt_paths_create: upload failed");
    };
    std::vector<uint8_t> seq(b->seq, b->seq + p->seq_off[NV]);
    up(&p->d_seq, seq);
    up(&d_segs, segs);
    up(&p->d_path_cluster, p->path_cluster);
    up(&p->d_path_local, p->path_local);
    up(&p->d_iv_off, iv_off);
    up(&p->d_iv, iv);
    if (rc == BT_OK) rc = dev_alloc(&p->d_text, p->L, p->owned);
    if (rc == BT_OK) rc = dev_alloc(&p->d_pos_path, p->L, p->owned);
    if (rc == BT_OK) rc = dev_alloc(&p->d_pos_nt, p->L, p->owned);
    if (rc == BT_OK) rc = dev_alloc(&p->d_kmers, 2 * p->L, p->owned);
    if (rc == BT_OK) rc = dev_alloc(&p->d_valid, p->L, p->owned);
    if (rc != BT_OK) {
        bt_paths_destroy(p);
        return rc;
    }
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(text_kernel, dim3(grid_for(p->L, BLOCK, ctx->num_cu * 16)), dim3(BLOCK), 0, st, d_segs, (uint64_t)segs.size(), p->d_seq, p->L, p->d_text,
                       p->d_pos_path, p->d_pos_nt);
    if (hipGetLastError() != hipSuccess || bt_kmers_from_sequence(ctx, p->d_text, p->L, k, p->d_kmers, p->d_valid) != BT_OK) {
        bt_paths_destroy(p);
        return fail("bt_paths_create: k-mer enumeration failed");
    }
    // number of windows
    std::vector<uint8_t> hv(p->L);
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(hv.data(), p->d_valid, p->L, hipMemcpyDeviceToHost) != hipSuccess) {
        bt_paths_destroy(p);
        return fail("bt_paths_create: device error");
    }
    p->num_valid = std::accumulate(hv.begin(), hv.end(), (uint64_t)0);
    if (h_num_kmer_occurrences) *h_num_kmer_occurrences = p->num_valid;
    *out = p;
    return BT_OK;
}

int bt_diag_kmer_set_order(const uint64_t *h_kmers, uint32_t n, uint64_t initial_buckets, unsigned k, uint32_t *h_rank, uint64_t *h_final_buckets) {
    if (!h_kmers || !h_rank) return fail("bt_diag_kmer_set_order: null argument");
    if (initial_buckets == 0) initial_buckets = 1;
    uint64_t cap = 4;
    while (cap < 2ull * n) cap <<= 1;
    std::vector<uint32_t> nx(std::max<uint32_t>(n, 1)), mk(cap), mv(cap);
    replay_kmer_set(
        n, initial_buckets, k, nx.data(), mk.data(), mv.data(), (uint32_t)cap - 1u,
        [&](uint32_t node, uint64_t &lo, uint64_t &hi) {
            lo = h_kmers[2 * (size_t)node];
            hi = h_kmers[2 * (size_t)node + 1];
        },
        [&](uint32_t node, uint32_t rank) { h_rank[node] = rank; });
    if (h_final_buckets) {
        uint64_t B = initial_buckets;
        while (B < n) B = std_next_bucket_count(B);
        *h_final_buckets = B;
    }
    return BT_OK;
}

int bt_paths_destroy(bt_paths *p) {
    if (!p) return BT_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (void *q : p->owned) (void)hipFree(q);
    delete p;
    return BT_OK;
}

int bt_paths_count_kmers(bt_paths *p, bt_bloom *path_bloom) {
    if (!p || !path_bloom) return fail("bt_paths_count_kmers: null argument");
    if (path_bloom->k != p->k) return fail("bt_paths_count_kmers: k mismatch");
    BT_HIP(hipSetDevice(p->ctx->device));
    hipLaunchKernelGGL(bloom_insert_valid_kernel, dim3(grid_for(p->L, BLOCK, p->ctx->num_cu * 16)), dim3(BLOCK), 0, p->ctx->stream, path_bloom->view(), p->d_kmers,
                       p->d_valid, p->L);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_paths_count_multigroup(bt_paths *p, const uint32_t *h_cluster_group, bt_bloom *path_bloom, bt_table *multigroup_table, uint64_t *h_num_path_kmers) {
    if (!p || !h_cluster_group || !path_bloom || !multigroup_table) return fail("bt_paths_count_multigroup: null argument");
    if (path_bloom->k != p->k || multigroup_table->k != p->k) return fail("bt_paths_count_multigroup: k mismatch");
    for (uint32_t c = 1; c < p->C; ++c)
        if (h_cluster_group[c] < h_cluster_group[c - 1]) return fail("bt_paths_count_multigroup: clusters must be listed group by group, groups in index order");
    if (p->L >= 0xFFFFFFFEull) return fail("bt_paths_count_multigroup: more than 2^32 path positions in one unit");
    BT_HIP(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    const uint64_t cap = pow2_at_least(2 * std::max<uint64_t>(p->num_valid, 8));
    if (cap > (1ull << 32)) return fail("bt_paths_count_multigroup: more than 2^31 k-mer occurrences in one unit");
    const uint32_t G = p->C ? h_cluster_group[p->C - 1] + 1u : 0u;
    std::vector<void *> tmp;
    auto cleanup = [&]() {
        for (void *q : tmp) (void)hipFree(q);
    };
    uint64_t *dlo, *dhi, *elo, *ehi, *d_out;
    uint32_t *dtag, *dstate, *dfirst, *pos_d, *etag, *estate, *egroup, *emulti, *etime, *d_cg, *d_flag, *d_rowpos, *d_sums;
    unsigned long long *d_counters;
    const uint64_t nblk = (p->L + BLOCK * 4 - 1) / (BLOCK * 4);
    int rc = BT_OK;
    auto A = [&](auto **q, uint64_t n) {
        if (rc == BT_OK) rc = dev_alloc(q, n, tmp);
    };
    A(&dlo, cap); A(&dhi, cap); A(&dtag, cap); A(&dstate, cap); A(&dfirst, cap); A(&pos_d, p->L);
    A(&elo, cap); A(&ehi, cap); A(&etag, cap); A(&estate, cap); A(&egroup, cap); A(&emulti, cap); A(&etime, cap);
    A(&d_out, 2 * p->num_valid); A(&d_cg, p->C); A(&d_counters, 4);
    A(&d_flag, p->L); A(&d_rowpos, p->L + 1); A(&d_sums, nblk);
    if (rc != BT_OK) {
        cleanup();
        return rc;
    }
#define MGH(call)                                                                             \
    do {                                                                                      \
        const hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                               \
            cleanup();                                                                        \
            return fail(std::string("bt_paths_count_multigroup: ") + hipGetErrorString(_e));  \
        }                                                                                     \
    } while (0)
#define MGR(call)            \
    do {                     \
        const int _r = (call); \
        if (_r != BT_OK) {   \
            cleanup();       \
            return _r;       \
        }                    \
    } while (0)
    MGH(hipMemsetAsync(dstate, 0, cap * 4, st));
    MGH(hipMemsetAsync(estate, 0, cap * 4, st));
    MGH(hipMemsetAsync(dfirst, 0xFF, cap * 4, st));
    MGH(hipMemsetAsync(egroup, 0xFF, cap * 4, st));
    MGH(hipMemsetAsync(etime, 0xFF, cap * 4, st));
    MGH(hipMemsetAsync(emulti, 0, cap * 4, st));
    MGH(hipMemsetAsync(d_counters, 0, 32, st));
    MGH(hipMemcpyAsync(d_cg, h_cluster_group, (size_t)p->C * 4, hipMemcpyHostToDevice, st));
    const unsigned maxb = p->ctx->num_cu * 16;
    const bt::BloomView bv = path_bloom->view();
    // 1. indexes; first occurrences in path order
    hipLaunchKernelGGL(multigroup_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, dlo, dhi, dtag, dstate, dfirst, pos_d, elo, ehi, etag, estate, egroup, emulti,
                       cap - 1, p->d_kmers, p->d_valid, p->d_pos_path, p->d_path_cluster, d_cg, p->L);
    hipLaunchKernelGGL(mg_flag_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, p->d_valid, pos_d, dfirst, p->L, d_flag);
    uint64_t n_total = 0;
    if (p->L) {
        hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, d_flag, p->L, d_rowpos, d_sums);
        std::vector<uint32_t> sums(nblk);
        MGH(hipMemcpyAsync(sums.data(), d_sums, nblk * 4, hipMemcpyDeviceToHost, st));
        MGH(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < nblk; ++i) {
            const uint32_t v = sums[i];
            sums[i] = (uint32_t)n_total;
            n_total += v;
        }
        MGH(hipMemcpyAsync(d_sums, sums.data(), nblk * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, d_rowpos, p->L, d_sums);
        MGH(hipStreamSynchronize(st));   // (sums goes out of scope)
    }
    const uint32_t total32 = (uint32_t)n_total;
    MGH(hipMemcpyAsync(d_rowpos + p->L, &total32, 4, hipMemcpyHostToDevice, st));
    // 2. per group: its range of first occurrences, the bucket count it inherits, room for its bucket map
    std::vector<uint64_t> gstart(G + 1, p->L);
    for (uint32_t c = p->C; c-- > 0;) gstart[h_cluster_group[c]] = p->cluster_text0[c];
    for (uint32_t g = G; g-- > 0;)
        if (gstart[g] == p->L && g + 1 <= G) gstart[g] = gstart[g + 1];   // a group index without clusters
    uint64_t *d_gstart = nullptr, *d_moff = nullptr, *d_binit = nullptr;
    uint32_t *d_goff = nullptr, *d_seq = nullptr, *d_time = nullptr, *d_next = nullptr, *d_mk = nullptr, *d_mv = nullptr;
    A(&d_gstart, G + 1); A(&d_goff, G + 1); A(&d_seq, n_total); A(&d_time, n_total); A(&d_next, n_total); A(&d_moff, G + 1); A(&d_binit, G + 1);
    MGR(rc);
    MGH(hipMemcpyAsync(d_gstart, gstart.data(), (size_t)(G + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_u32_kernel, dim3((G + 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d_rowpos, d_gstart, G + 1, d_goff);
    hipLaunchKernelGGL(mg_seq_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, d_flag, d_rowpos, p->L, d_seq);
    std::vector<uint32_t> goff(G + 1);
    MGH(hipMemcpyAsync(goff.data(), d_goff, (size_t)(G + 1) * 4, hipMemcpyDeviceToHost, st));
    MGH(hipStreamSynchronize(st));
    std::vector<uint64_t> moff(G + 1, 0), binit(G + 1, 1);
    {
        uint64_t B = 1;
        for (uint32_t g = 0; g < G; ++g) {
            const uint64_t n = goff[g + 1] - goff[g];
            binit[g] = B;
            while (B < n) B = std_next_bucket_count(B);   // the set keeps its bucket count across clear()
            if (B >= 0xFFFFFFFEull) {
                cleanup();
                return fail("bt_paths_count_multigroup: a group with more than 3.5e9 distinct path k-mers");
            }
            moff[g + 1] = moff[g] + (n ? pow2_at_least(2 * n) : 0);
        }
    }
    A(&d_mk, moff[G]); A(&d_mv, moff[G]);
    MGR(rc);
    MGH(hipMemcpyAsync(d_moff, moff.data(), (size_t)(G + 1) * 8, hipMemcpyHostToDevice, st));
    MGH(hipMemcpyAsync(d_binit, binit.data(), (size_t)(G + 1) * 8, hipMemcpyHostToDevice, st));
    // 3. times
    if (G) hipLaunchKernelGGL(mg_order_kernel, dim3((G + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, p->d_kmers, d_seq, d_goff, d_moff, d_binit, d_next, d_mk, d_mv, d_time, G, p->k);
    hipLaunchKernelGGL(mg_etime_kernel, dim3(grid_for(n_total, BLOCK, maxb)), dim3(BLOCK), 0, st, p->d_kmers, d_seq, d_time, n_total, elo, ehi, etag, estate, cap - 1, etime);
    // 4. k-mers the filter reports at their first turn
    uint32_t *f1 = nullptr, *f2 = nullptr;
    const uint64_t fwords = (path_bloom->bytes + 3) / 4;
    A(&f1, fwords); A(&f2, fwords);
    MGR(rc);
    MGH(hipMemsetAsync(f1, 0, fwords * 4, st));
    MGH(hipMemsetAsync(f2, 0, fwords * 4, st));
    hipLaunchKernelGGL(mg_shared_bits_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, bv, elo, ehi, estate, cap - 1, f1, f2);
    hipLaunchKernelGGL(mg_candidates_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, bv, elo, ehi, estate, emulti, cap - 1, f2, d_counters + 2,
                       (uint64_t *)nullptr);
    unsigned long long ncand = 0;
    MGH(hipMemcpyAsync(&ncand, d_counters + 2, 8, hipMemcpyDeviceToHost, st));
    MGH(hipStreamSynchronize(st));
    if (ncand) {
        uint64_t *d_cand = nullptr;
        unsigned long long *cb_keys = nullptr;
        uint32_t *cb_time = nullptr;
        const uint64_t cb_cap = pow2_at_least(2 * ncand * path_bloom->num_hashes);
        A(&d_cand, ncand); A(&cb_keys, cb_cap); A(&cb_time, cb_cap);
        MGR(rc);
        MGH(hipMemsetAsync(cb_keys, 0xFF, cb_cap * 8, st));
        MGH(hipMemsetAsync(cb_time, 0xFF, cb_cap * 4, st));
        hipLaunchKernelGGL(mg_candidates_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, bv, elo, ehi, estate, emulti, cap - 1, f2, d_counters + 3, d_cand);
        hipLaunchKernelGGL(mg_cand_bits_kernel, dim3(grid_for(ncand, BLOCK, maxb)), dim3(BLOCK), 0, st, bv, elo, ehi, d_cand, (uint64_t)ncand, cb_keys, cb_time, cb_cap - 1);
        hipLaunchKernelGGL(mg_bit_times_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, bv, elo, ehi, estate, etime, cap - 1, cb_keys, cb_time, cb_cap - 1);
        hipLaunchKernelGGL(mg_reported_kernel, dim3(grid_for(ncand, BLOCK, maxb)), dim3(BLOCK), 0, st, bv, elo, ehi, etime, d_cand, (uint64_t)ncand, cb_keys, cb_time,
                           cb_cap - 1, emulti);
    }
    hipLaunchKernelGGL(multigroup_list_kernel, dim3(grid_for(cap, BLOCK, maxb)), dim3(BLOCK), 0, st, elo, ehi, estate, emulti, dstate, cap - 1, d_out, d_counters);
    unsigned long long counters[2] = {0, 0};
    MGH(hipGetLastError());
    MGH(hipMemcpyAsync(counters, d_counters, 16, hipMemcpyDeviceToHost, st));
    MGH(hipStreamSynchronize(st));
    {   // the reference's KmerHash grows on demand; this table is grown before the unit's multigroup k-mers go in (their number is known here)
        uint64_t keys = 0, capacity = 0;
        int overflowed = 0;
        rc = bt_table_status(multigroup_table, &keys, &capacity, &overflowed);
        if (rc == BT_OK && 2 * (keys + counters[0]) > capacity) rc = bt_table_reserve(multigroup_table, keys + counters[0]);
    }
    if (rc == BT_OK) rc = bt_table_insert_batch(multigroup_table, d_out, counters[0], 0);
    if (rc == BT_OK) rc = bt_paths_count_kmers(p, path_bloom);
    if (rc == BT_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail("bt_paths_count_multigroup: device error");
    cleanup();
#undef MGH
#undef MGR
    if (h_num_path_kmers) *h_num_path_kmers = counters[1];
    return rc;
}

int bt_paths_classify(bt_paths *p, bt_table *table, bt_bloom *multigroup_bloom, uint32_t *h_num_path_kmers, uint8_t *h_has_excluded) {
    if (!p || !table || !multigroup_bloom) return fail("bt_paths_classify: null argument");
    if (table->k != p->k) return fail("bt_paths_classify: k mismatch");
    BT_HIP(hipSetDevice(p->ctx->device));
    int rc = build_index(p);
    if (rc != BT_OK) return rc;
    rc = bt_table_classify_batch(table, multigroup_bloom, p->d_list_kmers, p->d_list_mult, p->n_list, p->d_list_excluded);
    if (rc != BT_OK) return rc;
    uint32_t *d_n = nullptr, *d_ex = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_n), (size_t)p->C * 4));
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_ex), (size_t)p->C * 4));
    hipStream_t st = p->ctx->stream;
    (void)hipMemsetAsync(d_n, 0, (size_t)p->C * 4, st);
    (void)hipMemsetAsync(d_ex, 0, (size_t)p->C * 4, st);
    hipLaunchKernelGGL(classify_tally_kernel, dim3(grid_for(p->n_list, BLOCK, p->ctx->num_cu * 16)), dim3(BLOCK), 0, st, p->d_list_cluster, p->d_list_excluded,
                       p->n_list, d_n, d_ex);
    std::vector<uint32_t> n(p->C), ex(p->C);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(n.data(), d_n, (size_t)p->C * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ex.data(), d_ex, (size_t)p->C * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_n);
    (void)hipFree(d_ex);
    if (e != hipSuccess) return fail(std::string("bt_paths_classify: ") + hipGetErrorString(e));
    for (uint32_t c = 0; c < p->C; ++c) {
        if (h_num_path_kmers) h_num_path_kmers[c] = n[c];
        if (h_has_excluded) h_has_excluded[c] = ex[c] ? 1 : 0;
    }
    return BT_OK;
}

int bt_paths_candidates(bt_paths *p, bt_table *table, bt_paths_candidates_sizes *sizes) {
    if (!p || !table || !sizes) return fail("bt_paths_candidates: null argument");
    if (table->k != p->k) return fail("bt_paths_candidates: k mismatch");
    BT_HIP(hipSetDevice(p->ctx->device));
    int rc = build_index(p);
    if (rc != BT_OK) return rc;
    hipStream_t st = p->ctx->stream;
    const unsigned maxb = p->ctx->num_cu * 16;
    const uint32_t C = p->C, S = table->num_samples;
    p->S = S;
    std::vector<void *> tmp;
    auto cleanup = [&]() {
        for (void *q : tmp) (void)hipFree(q);
    };
#define TRYC(x)                  \
    do {                         \
        const int _rc = (x);     \
        if (_rc != BT_OK) {      \
            cleanup();           \
            return _rc;          \
        }                        \
    } while (0)
#define HIPC(call)                                                                      \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            cleanup();                                                                  \
            return bt::fail(std::string(#call) + ": " + hipGetErrorString(_e));         \
        }                                                                               \
    } while (0)
    // 1. table records of the distinct (cluster, k-mer) entries
    TRYC(bt_table_find_batch(table, p->d_list_kmers, p->n_list, p->d_list_slots));
    hipLaunchKernelGGL(record_kernel, dim3(grid_for(p->n_list, BLOCK, maxb)), dim3(BLOCK), 0, st, table->v, p->d_list_slots, p->n_list, p->d_list_flags);
    // 2. row numbering = prefix sum over first-occurrence flags in text order
    uint32_t *d_flag = nullptr, *d_rowpos = nullptr, *d_sums = nullptr, *d_a_row = nullptr;
    const uint64_t nblk = (p->L + BLOCK * 4 - 1) / (BLOCK * 4);
    TRYC(dev_alloc(&d_flag, p->L, tmp));
    TRYC(dev_alloc(&d_rowpos, p->L + 1, tmp));
    TRYC(dev_alloc(&d_sums, nblk, tmp));
    TRYC(dev_alloc(&d_a_row, p->A.mask + 1, tmp));
    hipLaunchKernelGGL(first_flag_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, p->A, p->d_valid, p->d_pos_slot_a, p->d_list_flags, p->L, d_flag);
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, d_flag, p->L, d_rowpos, d_sums);
    std::vector<uint32_t> sums(nblk);
    HIPC(hipMemcpyAsync(sums.data(), d_sums, nblk * 4, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    uint64_t total_rows = 0;
    for (uint64_t i = 0; i < nblk; ++i) {
        const uint32_t v = sums[i];
        sums[i] = (uint32_t)total_rows;
        total_rows += v;
    }
    if (total_rows >= (1ull << 32)) {
        cleanup();
        return fail("bt_paths_candidates: more than 2^32 k-mer rows in one batch");
    }
    HIPC(hipMemcpyAsync(d_sums, sums.data(), nblk * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblk), dim3(BLOCK), 0, st, d_rowpos, p->L, d_sums);
    // kmer_off[c] = rows before the cluster's first text position
    p->kmer_off.assign(C + 1, 0);
    {
        // rows before the first text position of every cluster (d_rowpos has L + 1 entries: the last one is the total)
        uint64_t *d_idx = nullptr;
        uint32_t *d_out = nullptr;
        TRYC(dev_alloc(&d_idx, C, tmp));
        TRYC(dev_alloc(&d_out, C, tmp));
        HIPC(hipMemcpyAsync(d_rowpos + p->L, &total_rows, 4, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(d_idx, p->cluster_text0.data(), (size_t)C * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_u32_kernel, dim3((C + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d_rowpos, d_idx, C, d_out);
        HIPC(hipMemcpyAsync(p->kmer_off.data(), d_out, (size_t)C * 4, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    p->kmer_off[C] = (uint32_t)total_rows;
    const uint64_t R = total_rows;
    // 3. per row: key + table record
    uint64_t *d_row_key = nullptr;
    uint8_t *d_row_flags = nullptr, *d_row_counts = nullptr, *d_row_ic = nullptr;
    TRYC(dev_alloc(&d_row_key, 2 * R, tmp));
    TRYC(dev_alloc(&d_row_flags, R, tmp));
    TRYC(dev_alloc(&d_row_counts, R * S, tmp));
    TRYC(dev_alloc(&d_row_ic, 2 * R, tmp));
    hipLaunchKernelGGL(rows_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, p->A, table->v, p->d_valid, p->d_pos_slot_a, d_flag, d_rowpos, p->d_list_slots,
                       p->d_list_flags, p->L, S, d_a_row, d_row_key, d_row_flags, d_row_counts, d_row_ic);
    // 4. multiplicity matrix
    std::vector<uint64_t> mult0(C + 1, 0);
    for (uint32_t c = 0; c < C; ++c) mult0[c + 1] = mult0[c] + (uint64_t)(p->kmer_off[c + 1] - p->kmer_off[c]) * p->num_paths[c];
    uint8_t *d_mult = nullptr;
    uint32_t *d_row0 = nullptr, *d_h = nullptr, *d_over = nullptr;
    uint64_t *d_mult0 = nullptr;
    TRYC(dev_alloc(&d_mult, mult0[C], tmp));
    TRYC(dev_alloc(&d_row0, C + 1, tmp));
    TRYC(dev_alloc(&d_h, C, tmp));
    TRYC(dev_alloc(&d_mult0, C + 1, tmp));
    TRYC(dev_alloc(&d_over, 1, tmp));
    HIPC(hipMemsetAsync(d_mult, 0, std::max<uint64_t>(mult0[C], 1), st));
    HIPC(hipMemsetAsync(d_over, 0, 4, st));
    HIPC(hipMemcpyAsync(d_row0, p->kmer_off.data(), (size_t)(C + 1) * 4, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(d_h, p->num_paths.data(), (size_t)C * 4, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(d_mult0, mult0.data(), (size_t)(C + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mult_kernel, dim3(grid_for(p->B.mask + 1, BLOCK, maxb)), dim3(BLOCK), 0, st, p->A, p->B, p->d_list_flags, d_a_row, p->d_path_cluster,
                       p->d_path_local, d_row0, d_mult0, d_h, d_mult, d_over);
    // 5. (row, variant, path) triples
    unsigned long long *d_cursor = nullptr;
    TRYC(dev_alloc(&d_cursor, 1, tmp));
    HIPC(hipMemsetAsync(d_cursor, 0, 8, st));
    hipLaunchKernelGGL(triples_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, p->A, p->d_valid, p->d_pos_slot_a, p->d_list_flags, d_a_row, p->d_pos_path,
                       p->d_pos_nt, p->d_path_local, p->d_iv_off, p->d_iv, p->L, d_cursor, (uint64_t *)nullptr);
    unsigned long long ntrip = 0;
    HIPC(hipMemcpyAsync(&ntrip, d_cursor, 8, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    uint64_t *d_trip = nullptr;
    TRYC(dev_alloc(&d_trip, ntrip, tmp));
    HIPC(hipMemsetAsync(d_cursor, 0, 8, st));
    hipLaunchKernelGGL(triples_kernel, dim3(grid_for(p->L, BLOCK, maxb)), dim3(BLOCK), 0, st, p->A, p->d_valid, p->d_pos_slot_a, p->d_list_flags, d_a_row, p->d_pos_path,
                       p->d_pos_nt, p->d_path_local, p->d_iv_off, p->d_iv, p->L, d_cursor, d_trip);
    HIPC(hipGetLastError());
    // sort the triples by (row, variant, path) on the device (rocPRIM radix sort over the 64-bit keys)
    if (ntrip > 1) {
        uint64_t *d_sorted = nullptr;
        TRYC(dev_alloc(&d_sorted, ntrip, tmp));
        size_t tmp_bytes = 0;
        HIPC(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_trip, d_sorted, (size_t)ntrip, 0, 64, st));
        uint8_t *d_tmp = nullptr;
        TRYC(dev_alloc(&d_tmp, tmp_bytes, tmp));
        HIPC(rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_trip, d_sorted, (size_t)ntrip, 0, 64, st));
        d_trip = d_sorted;
    }
    // 6. fetch and assemble on the host
    std::vector<uint64_t> trip(ntrip);
    std::vector<uint8_t> row_flags(R);
    uint32_t over = 0;
    p->key.resize(2 * R);
    p->counts.resize(R * S);
    p->ic.resize(2 * R);
    p->mult.resize(mult0[C]);
    HIPC(hipStreamSynchronize(st));
    if (ntrip) HIPC(hipMemcpy(trip.data(), d_trip, ntrip * 8, hipMemcpyDeviceToHost));
    if (R) {
        HIPC(hipMemcpy(row_flags.data(), d_row_flags, R, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(p->key.data(), d_row_key, 2 * R * 8, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(p->counts.data(), d_row_counts, R * S, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(p->ic.data(), d_row_ic, 2 * R, hipMemcpyDeviceToHost));
    }
    if (mult0[C]) HIPC(hipMemcpy(p->mult.data(), d_mult, mult0[C], hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(&over, d_over, 4, hipMemcpyDeviceToHost));
    cleanup();
#undef TRYC
#undef HIPC
    if (over) return fail("bt_paths_candidates: a path k-mer occurs more than 127 times on one haplotype (the reference asserts <= 127)");
    // ---- assembly on the host, on several threads: every thread takes a range of clusters (their rows are a contiguous range of rows, their triples a
    // contiguous range of the sorted triples), builds its pieces of the lists with offsets relative to the piece, and the pieces are concatenated in
    // cluster order with the offsets rebased.  (One thread did all of this in rounds 1-3: 0.37 s for 175 000 clusters, more than the unit's sampling launch.)
    struct Piece {
        std::vector<uint32_t> unique_off, unique_idx, multi_off, multi_idx, kv_off, kv_bits, hapnest_off, hapnest_idx, nestdep_off, nestdep_cluster, nestdep_var_off;
        std::vector<uint16_t> kv_var, hap_allele, nestdep_var;
    };
    const unsigned T = host_threads(C);
    std::vector<Piece> piece(T);
    p->has_counts.resize(R);
    auto work = [&](unsigned t) {
        Piece &q = piece[t];
        const uint32_t c0 = (uint32_t)((uint64_t)C * t / T), c1 = (uint32_t)((uint64_t)C * (t + 1) / T);
        const uint64_t r0 = p->kmer_off[c0], r1 = p->kmer_off[c1];
        for (uint64_t r = r0; r < r1; ++r) p->has_counts[r] = row_flags[r] & 1;
        // unique / multicluster row lists, in row (= first-seen) order
        for (uint32_t c = c0; c < c1; ++c) {
            for (uint32_t r = p->kmer_off[c]; r < p->kmer_off[c + 1]; ++r) (row_flags[r] & 4 ? q.multi_idx : q.unique_idx).push_back(r - p->kmer_off[c]);
            q.unique_off.push_back((uint32_t)q.unique_idx.size());
            q.multi_off.push_back((uint32_t)q.multi_idx.size());
        }
        // variant_haplotype_indices: the triples arrive sorted by (row, variant, path); entries of a row ordered by variant
        uint64_t ti = (uint64_t)(std::lower_bound(trip.begin(), trip.end(), r0 << 32) - trip.begin());
        for (uint32_t c = c0; c < c1; ++c) {
            const uint32_t HW = (p->num_paths[c] + 31) / 32;
            for (uint64_t r = p->kmer_off[c]; r < p->kmer_off[c + 1]; ++r) {
                while (ti < ntrip && (trip[ti] >> 32) == r) {
                    const uint16_t var = (uint16_t)((trip[ti] >> 16) & 0xffff);
                    q.kv_var.push_back(var);
                    const size_t w0 = q.kv_bits.size();
                    q.kv_bits.resize(w0 + HW, 0);
                    while (ti < ntrip && (trip[ti] >> 32) == r && (uint16_t)((trip[ti] >> 16) & 0xffff) == var) {
                        const uint32_t path = (uint32_t)(trip[ti] & 0xffff);
                        q.kv_bits[w0 + (path >> 5)] |= 1u << (path & 31);
                        ++ti;
                    }
                }
                q.kv_off.push_back((uint32_t)q.kv_var.size());
            }
        }
        // haplotypes and the nested dependency map (O(paths x vertices))
        std::vector<Interval> iv_dummy;
        std::vector<uint16_t> alleles;
        std::vector<uint32_t> nested;
        for (uint32_t c = c0; c < c1; ++c) {
            for (uint32_t lp = 0; lp < p->num_paths[c]; ++lp) {
                iv_dummy.clear();
                walk_path_host(*p, c, lp, iv_dummy, alleles, nested);
                q.hap_allele.insert(q.hap_allele.end(), alleles.begin(), alleles.end());
                q.hapnest_idx.insert(q.hapnest_idx.end(), nested.begin(), nested.end());
                q.hapnest_off.push_back((uint32_t)q.hapnest_idx.size());
            }
            std::map<uint32_t, std::vector<uint16_t>> dep;   // VariantClusterGraph.cpp:1112-1132
            for (uint32_t v = p->vertex_off[c]; v < p->vertex_off[c + 1]; ++v) {
                if (p->vertex_nested[v] == 0xFFFFFFFFu) continue;
                auto &lst = dep[p->vertex_nested[v]];
                if (p->vertex_variant[v] != 0xFFFF) lst.push_back(p->vertex_variant[v]);
                for (uint32_t r = p->refvar_off[v]; r < p->refvar_off[v + 1]; ++r) lst.push_back(p->refvar[r]);
                std::sort(lst.begin(), lst.end(), std::greater<uint16_t>());
            }
            for (auto &e : dep) {
                q.nestdep_cluster.push_back(e.first);
                q.nestdep_var.insert(q.nestdep_var.end(), e.second.begin(), e.second.end());
                q.nestdep_var_off.push_back((uint32_t)q.nestdep_var.size());
            }
            q.nestdep_off.push_back((uint32_t)q.nestdep_cluster.size());
        }
    };
    try {
        if (C) run_on_threads(T, work);
    } catch (const std::exception &e) {   // (a worker ran out of memory on its pieces: an error of the call, not of the process)
        return fail(std::string("bt_paths_candidates: ") + e.what());
    }
    // concatenation: `idx` lists appended as they are, `off` lists (cumulative ends, relative to the piece) rebased on what precedes the piece
    auto cat = [&](auto &dst, auto Piece::*m) {
        size_t n = 0;
        for (auto &q : piece) n += (q.*m).size();
        dst.clear();
        dst.reserve(n);
        for (auto &q : piece) dst.insert(dst.end(), (q.*m).begin(), (q.*m).end());
    };
    auto cat_off = [&](std::vector<uint32_t> &dst, std::vector<uint32_t> Piece::*off, auto Piece::*idx) {
        size_t n = 1;
        for (auto &q : piece) n += (q.*off).size();
        dst.clear();
        dst.reserve(n);
        dst.push_back(0);
        uint32_t base = 0;
        for (auto &q : piece) {
            for (uint32_t v : q.*off) dst.push_back(base + v);
            base += (uint32_t)(q.*idx).size();
        }
    };
    cat_off(p->unique_off, &Piece::unique_off, &Piece::unique_idx);
    cat_off(p->multi_off, &Piece::multi_off, &Piece::multi_idx);
    cat_off(p->kv_off, &Piece::kv_off, &Piece::kv_var);
    cat_off(p->hapnest_off, &Piece::hapnest_off, &Piece::hapnest_idx);
    cat_off(p->nestdep_off, &Piece::nestdep_off, &Piece::nestdep_cluster);
    cat_off(p->nestdep_var_off, &Piece::nestdep_var_off, &Piece::nestdep_var);
    cat(p->unique_idx, &Piece::unique_idx);
    cat(p->multi_idx, &Piece::multi_idx);
    cat(p->kv_var, &Piece::kv_var);
    cat(p->kv_bits, &Piece::kv_bits);
    cat(p->hap_allele, &Piece::hap_allele);
    cat(p->hapnest_idx, &Piece::hapnest_idx);
    cat(p->nestdep_cluster, &Piece::nestdep_cluster);
    cat(p->nestdep_var, &Piece::nestdep_var);
    piece.clear();
    p->have_candidates = true;
    sizes->rows = R;
    sizes->mult_bytes = p->mult.size();
    sizes->nnz = p->kv_var.size();
    sizes->kv_words = p->kv_bits.size();
    sizes->num_unique = p->unique_idx.size();
    sizes->num_multi = p->multi_idx.size();
    sizes->hap_allele = p->hap_allele.size();
    sizes->num_haplotypes = p->num_gpaths;
    sizes->hapnest = p->hapnest_idx.size();
    sizes->nestdep = p->nestdep_cluster.size();
    sizes->nestdep_var = p->nestdep_var.size();
    return BT_OK;
}

int bt_paths_candidates_fetch(bt_paths *p, bt_paths_candidates_out *o) {
    if (!p || !o) return fail("bt_paths_candidates_fetch: null argument");
    if (!p->have_candidates) return fail("bt_paths_candidates_fetch: bt_paths_candidates has not run");
    auto cp = [](const auto &v, auto *dst) {   // (the large arrays on several threads: one core's memcpy into fresh pages is the slow part)
        if (!dst || v.empty()) return;
        const size_t bytes = v.size() * sizeof(v[0]);
        const unsigned T = host_threads(bytes >> 14);   // at least 4 MB per thread
        const uint8_t *src = reinterpret_cast<const uint8_t *>(v.data());
        uint8_t *out = reinterpret_cast<uint8_t *>(dst);
        try {
            run_on_threads(T, [&](unsigned t) {
                const size_t a = bytes * t / T / 64 * 64, b = t + 1 == T ? bytes : bytes * (t + 1) / T / 64 * 64;
                if (b > a) std::memcpy(out + a, src + a, b - a);
            });
        } catch (...) {   // (no thread could be started: one copy)
            std::memcpy(out, src, bytes);
        }
    };
    cp(p->kmer_off, o->kmer_off);
    cp(p->mult, o->hap_kmer_mult);
    cp(p->key, o->kmer_key);
    cp(p->has_counts, o->kmer_has_counts);
    cp(p->counts, o->kmer_counts);
    cp(p->ic, o->kmer_ic_mult);
    cp(p->kv_off, o->kv_off);
    cp(p->kv_var, o->kv_var);
    cp(p->kv_bits, o->kv_bits);
    cp(p->unique_off, o->unique_off);
    cp(p->unique_idx, o->unique_idx);
    cp(p->multi_off, o->multi_off);
    cp(p->multi_idx, o->multi_idx);
    cp(p->hap_allele, o->hap_allele);
    cp(p->hapnest_off, o->hapnest_off);
    cp(p->hapnest_idx, o->hapnest_idx);
    cp(p->nestdep_off, o->nestdep_off);
    cp(p->nestdep_cluster, o->nestdep_cluster);
    cp(p->nestdep_var_off, o->nestdep_var_off);
    cp(p->nestdep_var, o->nestdep_var);
    return BT_OK;
}

}  // extern "C"
