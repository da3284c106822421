// libbtgpu — the per-sample best-path search: VariantClusterGraph::findSamplePaths / mergePaths / isPathsRedundant / filterPaths /
// addPathIndices (src/bayesTyper/VariantClusterGraph.cpp:389-798) with VariantClusterGraphPath (VariantClusterGraphPath.cpp:38-225).
//
// The search is a sequential dynamic programme over the vertices of ONE cluster (candidate paths of a vertex = merged paths of
// its predecessors, shuffled with the cluster's mt19937, extended by the vertex with one sample-Bloom lookup per completed k-mer,
// then cut to max_sample_haplotypes by a two-pass greedy); clusters are independent.  As for the Gibbs sampler the unit of
// parallelism is therefore the cluster: one lane runs one cluster's search, a launch carries all clusters of a unit, and the
// random accesses into the (GB-sized) sample Bloom filter of thousands of concurrently running searches overlap in HBM.
// Every lane works in a private scratch region (path slots with a free list, per-vertex path lists, the cluster's generator).
// Integer work except the k-mer score ratio (fp64 division, IEEE).  First version: per-cluster scratch is contiguous, not
// lane-interleaved (DESIGN.md §7 lists that as the next step for this kernel).
#include "bt_internal.hpp"

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "bt_rng_device.hpp"

using namespace bt;

namespace {

constexpr uint32_t MIN_OBSERVED_KMERS = 2;      // VariantClusterGraphPath.cpp:36
constexpr uint32_t MIN_NUM_SAMPLE_PATHS = 1;    // VariantClusterGraph.cpp:60
constexpr uint32_t HDR_WORDS = 8;               // slot header: len, score_first, score_second, window count, window lo (2), window hi (2)

struct FindCluster {
    uint32_t v0, nv;
    uint32_t cap_slots, cur_cap;
    uint32_t slot_words;      // HDR_WORDS + nv
    uint32_t best_cap;        // rows the best-path bitmap can hold
    uint64_t scratch;         // word offset of the cluster's scratch region
    uint64_t best;            // byte offset of the cluster's best rows
};

struct FindGraph {
    const uint64_t *seq_off;
    const uint8_t *seq;
    const uint8_t *vflags;
    const uint32_t *in_off, *in_src;   // in_off indexed by global vertex
    const uint32_t *last_use;          // per global vertex: local index of its last successor (or its own index)
};

// view of one cluster's scratch region
struct Work {
    uint32_t *slots;      // cap_slots * slot_words
    uint32_t *free_stack; // cap_slots
    uint32_t *vlist;      // nv * max_haps
    uint32_t *vcount;     // nv
    uint32_t *cur;        // cur_cap
    uint32_t *tmp;        // nv   (vertex list of a best row)
    uint32_t *covered;    // nv   (0/1)
    uint32_t *mt;         // MT_WORDS
    uint32_t free_top;
    uint32_t slot_words, nv, v0, k, max_haps;
    FindGraph g;
    BloomView bloom;
    uint32_t *overflow;   // global flag
};

__device__ inline uint32_t vlen(const Work &w, uint32_t vi) { return (uint32_t)(w.g.seq_off[w.v0 + vi + 1] - w.g.seq_off[w.v0 + vi]); }
__device__ inline uint32_t vnt(const Work &w, uint32_t vi, uint32_t i) { return w.g.seq[w.g.seq_off[w.v0 + vi] + i] & 3u; }
__device__ inline bool vdisc(const Work &w, uint32_t vi) { return w.g.vflags[w.v0 + vi] & 1u; }
__device__ inline bool vredundant(const Work &w, uint32_t vi) { return w.g.vflags[w.v0 + vi] & 2u; }
__device__ inline uint32_t *slot(const Work &w, uint32_t s) { return w.slots + (size_t)s * w.slot_words; }
__device__ inline uint32_t ent_index(uint32_t e) { return e & 0xFFFFFFu; }
__device__ inline uint32_t ent_obs(uint32_t e) { return e >> 24; }

__device__ inline uint32_t slot_alloc(Work &w) {
    if (w.free_top == 0) {
        atomicExch(w.overflow, 1u);
        return 0;
    }
    return w.free_stack[--w.free_top];
}
__device__ inline void slot_free(Work &w, uint32_t s) { w.free_stack[w.free_top++] = s; }
__device__ inline void slot_copy(const Work &w, uint32_t dst, uint32_t src) {
    const uint32_t *a = slot(w, src);
    uint32_t *b = slot(w, dst);
    const uint32_t n = HDR_WORDS + a[0];
    for (uint32_t i = 0; i < n; ++i) b[i] = a[i];
}

// VariantClusterGraphPath::updateScore (VariantClusterGraphPath.cpp:87-129)
__device__ inline void update_score(const Work &w, uint32_t *p, bool observed, uint32_t cur_sequence_length) {
    uint32_t *ent = p + HDR_WORDS;
    if (observed) {
        p[1]++;
        int32_t r = (int32_t)p[0] - 1;
        if (cur_sequence_length > 1 || !vredundant(w, ent_index(ent[r]))) {
            if (ent_obs(ent[r]) < MIN_OBSERVED_KMERS) ent[r] += 1u << 24;
        }
        for (--r; r >= 0; --r) {
            if (w.k <= cur_sequence_length || ent_obs(ent[r]) == MIN_OBSERVED_KMERS) break;
            if (ent_obs(ent[r]) < MIN_OBSERVED_KMERS) ent[r] += 1u << 24;
            cur_sequence_length += vlen(w, ent_index(ent[r]));
        }
    }
    p[2]++;
}
// VariantClusterGraphPath::addVertex (:46-85): the window is the forward k-mer of KmerPair (Kmer.tpp:44-81); count = valid nucleotides since reset
__device__ inline void add_vertex(const Work &w, uint32_t *p, uint32_t vi) {
    uint32_t *ent = p + HDR_WORDS;
    ent[p[0]] = vi;
    p[0]++;
    if (vdisc(w, vi)) {
        if (vlen(w, vi) == 0) ent[p[0] - 1] = vi | (MIN_OBSERVED_KMERS << 24);
        p[3] = 0;
    }
    Kmer fw{(uint64_t)p[4] | ((uint64_t)p[5] << 32), (uint64_t)p[6] | ((uint64_t)p[7] << 32)};
    uint32_t cnt = p[3];
    const uint32_t n = vlen(w, vi), top = 2u * (w.k - 1u);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t c = vnt(w, vi, i);
        // shift one nucleotide out at the bottom, write the new one at position k-1
        fw.lo = (fw.lo >> 2) | (fw.hi << 62);
        fw.hi >>= 2;
        if (top >= 64u) fw.hi |= c << (top - 64u);
        else fw.lo |= c << top;
        fw = kmer_mask(fw, w.k);
        if (cnt < w.k) ++cnt;
        if (cnt == w.k) {
            const Kmer low = kmer_canonical(fw, w.k);
            update_score(w, p, bloom_contains(nthash64(low, w.k), w.bloom), i + 1u);
        }
    }
    p[3] = cnt;
    p[4] = (uint32_t)fw.lo;
    p[5] = (uint32_t)(fw.lo >> 32);
    p[6] = (uint32_t)fw.hi;
    p[7] = (uint32_t)(fw.hi >> 32);
}
__device__ inline double kmer_score(const uint32_t *p) { return p[2] > 0 ? p[1] / static_cast<double>(p[2]) : 1.0; }   // :136-148
__device__ inline uint32_t vertex_score(const Work &w, const uint32_t *p, bool is_complete) {                         // :150-188
    const uint32_t *ent = p + HDR_WORDS;
    uint32_t score = 0;
    for (uint32_t i = 0; i < p[0]; ++i)
        if (ent_obs(ent[i]) == MIN_OBSERVED_KMERS && !w.covered[ent_index(ent[i])]) ++score;
    if (!is_complete) {
        uint32_t cur = 0;
        for (int32_t r = (int32_t)p[0] - 1; r >= 0; --r) {
            if ((w.k - 1u) <= cur || ent_obs(ent[r]) == MIN_OBSERVED_KMERS) break;
            if (!w.covered[ent_index(ent[r])]) ++score;
            cur += vlen(w, ent_index(ent[r]));
        }
    }
    return score;
}
__device__ inline void update_covered(const Work &w, const uint32_t *p, bool is_complete) {   // :190-225
    const uint32_t *ent = p + HDR_WORDS;
    for (uint32_t i = 0; i < p[0]; ++i)
        if (ent_obs(ent[i]) == MIN_OBSERVED_KMERS) w.covered[ent_index(ent[i])] = 1;
    if (!is_complete) {
        uint32_t cur = 0;
        for (int32_t r = (int32_t)p[0] - 1; r >= 0; --r) {
            if ((w.k - 1u) <= cur || ent_obs(ent[r]) == MIN_OBSERVED_KMERS) break;
            w.covered[ent_index(ent[r])] = 1;
            cur += vlen(w, ent_index(ent[r]));
        }
    }
}
// isPathsRedundant (VariantClusterGraph.cpp:525-626): both vertex lists spell the same nucleotides and separators, read backwards
__device__ inline bool paths_redundant(const Work &w, const uint32_t *e1, uint32_t n1, const uint32_t *e2, uint32_t n2) {
    int32_t i1 = (int32_t)n1 - 1, i2 = (int32_t)n2 - 1;
    uint32_t r1 = vlen(w, ent_index(e1[i1])), r2 = vlen(w, ent_index(e2[i2]));
    bool d1 = false, d2 = false;
    while (true) {
        while (r1 == 0) {
            if (vdisc(w, ent_index(e1[i1]))) d1 = true;
            --i1;
            if (i1 >= 0) r1 = vlen(w, ent_index(e1[i1]));
            else break;
        }
        while (r2 == 0) {
            if (vdisc(w, ent_index(e2[i2]))) d2 = true;
            --i2;
            if (i2 >= 0) r2 = vlen(w, ent_index(e2[i2]));
            else break;
        }
        if (d1 != d2) return false;
        d1 = false;
        d2 = false;
        if (i1 < 0 || i2 < 0) break;
        const uint32_t a = ent_index(e1[i1]), b = ent_index(e2[i2]);
        if (a == b && r1 == r2) {   // the same position of the same vertex: the rest of this vertex is shared
            r1 = 0;
            r2 = 0;
        }
        while (r1 != 0 && r2 != 0) {
            if (vnt(w, a, r1 - 1) != vnt(w, b, r2 - 1)) return false;
            --r1;
            --r2;
        }
    }
    return !(i1 >= 0 || i2 >= 0);
}
// mergePaths (:474-523), copy semantics (the reference's move on the last edge is an optimisation)
__device__ inline void merge_paths(Work &w, uint32_t &ncur, const uint32_t *in, uint32_t nin, uint32_t cur_cap) {
    const uint32_t main_size = ncur;
    for (uint32_t j = 0; j < nin; ++j) {
        const uint32_t *ip = slot(w, in[j]);
        bool redundant = false;
        for (uint32_t m = 0; m < main_size; ++m) {
            uint32_t *mp = slot(w, w.cur[m]);
            if (paths_redundant(w, mp + HDR_WORDS, mp[0], ip + HDR_WORDS, ip[0])) {
                if (mp[0] < ip[0]) slot_copy(w, w.cur[m], in[j]);
                redundant = true;
                break;
            }
        }
        if (!redundant) {
            if (ncur >= cur_cap) {
                atomicExch(w.overflow, 1u);
                return;
            }
            const uint32_t s = slot_alloc(w);
            slot_copy(w, s, in[j]);
            w.cur[ncur++] = s;
        }
    }
}
__device__ inline bool double_compare(double a, double b) {   // Utils::doubleCompare (Utils.hpp:81-87)
    const double mn = a < b ? a : b;
    return a == b || fabs(a - b) < fabs(mn) * 2.220446049250313080847263336181640625e-16 * 100;
}
// filterPaths (:628-724)
__device__ inline void filter_paths(Work &w, uint32_t &ncur, uint32_t max_paths, bool is_complete) {
    if (!(ncur > max_paths || (is_complete && ncur > MIN_NUM_SAMPLE_PATHS))) return;
    bool is_first_pass = true;
    for (uint32_t i = 0; i < w.nv; ++i) w.covered[i] = 0;
    uint32_t sorted_end = 0;
    while (sorted_end != ncur) {
        uint32_t best = sorted_end;
        double best_kmer = kmer_score(slot(w, w.cur[best]));
        uint32_t best_vertex = vertex_score(w, slot(w, w.cur[best]), is_complete);
        for (uint32_t it = sorted_end + 1; it < ncur; ++it) {
            const uint32_t *p = slot(w, w.cur[it]);
            const double cur_kmer = kmer_score(p);
            const uint32_t cur_vertex = vertex_score(w, p, is_complete);
            if (is_first_pass) {
                if (cur_vertex > 0) {
                    if ((double_compare(cur_kmer, best_kmer) && cur_vertex > best_vertex) || cur_kmer > best_kmer || best_vertex == 0) {
                        best = it;
                        best_kmer = cur_kmer;
                        best_vertex = cur_vertex;
                    }
                }
            } else if (!is_complete || cur_vertex == p[0]) {
                if (cur_kmer > best_kmer) {
                    best = it;
                    best_kmer = cur_kmer;
                    best_vertex = cur_vertex;
                }
            }
        }
        if (is_first_pass) update_covered(w, slot(w, w.cur[best]), is_complete);
        else if (is_complete && sorted_end >= MIN_NUM_SAMPLE_PATHS && best_vertex < slot(w, w.cur[best])[0]) break;
        if (sorted_end != best) {
            const uint32_t t = w.cur[sorted_end];
            w.cur[sorted_end] = w.cur[best];
            w.cur[best] = t;
        }
        if (is_first_pass && best_vertex == 0) {
            is_first_pass = false;
            for (uint32_t i = 0; i < w.nv; ++i) w.covered[i] = 0;
        } else {
            ++sorted_end;
            if (sorted_end == max_paths) break;
        }
    }
    for (uint32_t i = sorted_end; i < ncur; ++i) slot_free(w, w.cur[i]);
    ncur = sorted_end;
}

__global__ __launch_bounds__(64) void find_paths_kernel(const FindCluster *__restrict__ clusters, uint32_t C, FindGraph g, BloomView bloom, const uint32_t *__restrict__ seeds,
                                                        uint32_t k, uint32_t max_haps, uint32_t *__restrict__ scratch, uint8_t *__restrict__ best_rows,
                                                        uint32_t *__restrict__ best_count, uint32_t *__restrict__ overflow) {
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const FindCluster fc = clusters[c];
    Work w;
    uint32_t *base = scratch + fc.scratch;
    w.slots = base;
    base += (size_t)fc.cap_slots * fc.slot_words;
    w.free_stack = base;
    base += fc.cap_slots;
    w.vlist = base;
    base += (size_t)fc.nv * max_haps;
    w.vcount = base;
    base += fc.nv;
    w.cur = base;
    base += fc.cur_cap;
    w.tmp = base;
    base += fc.nv;
    w.covered = base;
    base += fc.nv;
    w.mt = base;
    w.slot_words = fc.slot_words;
    w.nv = fc.nv;
    w.v0 = fc.v0;
    w.k = k;
    w.max_haps = max_haps;
    w.g = g;
    w.bloom = bloom;
    w.overflow = overflow;
    w.free_top = fc.cap_slots;
    for (uint32_t i = 0; i < fc.cap_slots; ++i) w.free_stack[i] = fc.cap_slots - 1u - i;
    for (uint32_t i = 0; i < fc.nv; ++i) w.vcount[i] = 0;
    mt_seed(w.mt, seeds[c]);
    Mt rng = mt_open(w.mt);
    // ---- findSamplePaths (:389-472) ----
    uint32_t ncur = 0;
    for (uint32_t vi = 0; vi < fc.nv; ++vi) {
        ncur = 0;
        const uint32_t e0 = g.in_off[fc.v0 + vi], e1 = g.in_off[fc.v0 + vi + 1];
        if (e0 == e1) {
            const uint32_t s = slot_alloc(w);
            uint32_t *p = slot(w, s);
            for (uint32_t i = 0; i < HDR_WORDS; ++i) p[i] = 0;
            w.cur[ncur++] = s;
        } else {
            for (uint32_t e = e0; e < e1; ++e) {
                const uint32_t src = g.in_src[e];
                merge_paths(w, ncur, w.vlist + (size_t)src * max_haps, w.vcount[src], fc.cur_cap);
            }
        }
        rng_shuffle_u32(rng, w.cur, ncur);
        for (uint32_t i = 0; i < ncur; ++i) add_vertex(w, slot(w, w.cur[i]), vi);
        filter_paths(w, ncur, max_haps, false);
        for (uint32_t i = 0; i < ncur; ++i) w.vlist[(size_t)vi * max_haps + i] = w.cur[i];
        w.vcount[vi] = ncur;
        for (uint32_t e = e0; e < e1; ++e) {   // predecessors whose last successor this vertex is are no longer needed
            const uint32_t src = g.in_src[e];
            if (g.last_use[fc.v0 + src] == vi && w.vcount[src]) {
                for (uint32_t i = 0; i < w.vcount[src]; ++i) slot_free(w, w.vlist[(size_t)src * max_haps + i]);
                w.vcount[src] = 0;
            }
        }
    }
    mt_close(rng);
    filter_paths(w, ncur, max_haps, true);
    // ---- addPathIndices (:726-798) ----
    uint8_t *rows = best_rows + fc.best;
    uint32_t nrows = best_count[c];
    // redundant flags of the final paths live in the high bit of cur[]
    for (uint32_t r = 0; r < nrows; ++r) {
        uint8_t *row = rows + (size_t)r * fc.nv;
        uint32_t nb = 0;
        for (uint32_t vi = 0; vi < fc.nv; ++vi)
            if (row[vi]) w.tmp[nb++] = vi;
        for (uint32_t pi = 0; pi < ncur; ++pi) {
            if (w.cur[pi] & 0x80000000u) continue;
            const uint32_t *p = slot(w, w.cur[pi]);
            if (paths_redundant(w, w.tmp, nb, p + HDR_WORDS, p[0])) {
                if (nb < p[0]) {
                    for (uint32_t vi = 0; vi < fc.nv; ++vi) row[vi] = 0;
                    for (uint32_t i = 0; i < p[0]; ++i) row[ent_index(p[HDR_WORDS + i])] = 1;
                }
                w.cur[pi] |= 0x80000000u;
                break;
            }
        }
    }
    for (uint32_t pi = 0; pi < ncur; ++pi) {
        if (w.cur[pi] & 0x80000000u) continue;
        if (nrows >= fc.best_cap) {
            atomicExch(overflow, 2u);
            break;
        }
        const uint32_t *p = slot(w, w.cur[pi]);
        uint8_t *row = rows + (size_t)nrows * fc.nv;
        for (uint32_t vi = 0; vi < fc.nv; ++vi) row[vi] = 0;
        for (uint32_t i = 0; i < p[0]; ++i) row[ent_index(p[HDR_WORDS + i])] = 1;
        ++nrows;
    }
    best_count[c] = nrows;
}

}  // namespace

struct bt_find_paths {
    bt_ctx *ctx = nullptr;
    uint32_t k = 0, C = 0, max_haps = 0;
    std::vector<FindCluster> clusters;
    std::vector<uint32_t> nv;
    FindGraph g{};
    FindCluster *d_clusters = nullptr;
    uint32_t *d_scratch = nullptr, *d_best_count = nullptr, *d_overflow = nullptr, *d_seeds = nullptr;
    uint8_t *d_best = nullptr;
    uint64_t best_bytes = 0;
    std::vector<void *> owned;
};

extern "C" {

int bt_find_paths_create(bt_ctx *ctx, const bt_paths_batch *b, uint32_t k, uint32_t max_sample_haplotypes, uint32_t num_samples, bt_find_paths **out) {
    if (!ctx || !b || !out) return fail("bt_find_paths_create: null argument");
    if (!b->in_off || !b->in_src) return fail("bt_find_paths_create: the batch carries no edges (in_off / in_src)");
    if (k < 1 || k > 64) return fail("bt_find_paths_create: k must be in 1..64");
    if (max_sample_haplotypes < 1 || num_samples < 1) return fail("bt_find_paths_create: max_sample_haplotypes and num_samples must be positive");
    BT_HIP(hipSetDevice(ctx->device));
    bt_find_paths *f = new bt_find_paths();
    f->ctx = ctx;
    f->k = k;
    f->C = b->num_clusters;
    f->max_haps = max_sample_haplotypes;
    const uint32_t C = f->C, NV = b->vertex_off[C];
    std::vector<uint32_t> last_use(NV);
    uint64_t scratch_words = 0, best_bytes = 0;
    for (uint32_t c = 0; c < C; ++c) {
        const uint32_t v0 = b->vertex_off[c], nv = b->vertex_off[c + 1] - v0;
        if (nv == 0 || nv >= (1u << 24)) {
            delete f;
            return fail("bt_find_paths_create: a cluster needs between 1 and 2^24 - 1 vertices");
        }
        uint32_t max_indeg = 1;
        for (uint32_t vi = 0; vi < nv; ++vi) last_use[v0 + vi] = vi;
        for (uint32_t vi = 0; vi < nv; ++vi) {
            const uint32_t e0 = b->in_off[v0 + vi], e1 = b->in_off[v0 + vi + 1];
            max_indeg = std::max(max_indeg, e1 - e0);
            for (uint32_t e = e0; e < e1; ++e) {
                if (b->in_src[e] >= vi) {
                    delete f;
                    return fail("bt_find_paths_create: edges must point from lower to higher vertex indices");
                }
                last_use[v0 + b->in_src[e]] = std::max(last_use[v0 + b->in_src[e]], vi);
            }
        }
        // vertices whose path lists are alive at the same time: the list of u lives from the step of u to the step of its last successor
        uint32_t live_max = 1;
        {
            std::vector<int32_t> delta(nv + 1, 0);
            for (uint32_t vi = 0; vi < nv; ++vi) {
                delta[vi] += 1;
                delta[last_use[v0 + vi] + 1] -= 1;
            }
            int32_t live = 0;
            for (uint32_t vi = 0; vi < nv; ++vi) {
                live += delta[vi];
                live_max = std::max<uint32_t>(live_max, (uint32_t)live);
            }
        }
        FindCluster fc{};
        fc.v0 = v0;
        fc.nv = nv;
        fc.cur_cap = max_indeg * max_sample_haplotypes + 1;
        fc.cap_slots = (live_max + max_indeg + 1) * max_sample_haplotypes + 1;
        fc.slot_words = HDR_WORDS + nv;
        fc.best_cap = max_sample_haplotypes * num_samples;
        fc.scratch = scratch_words;
        fc.best = best_bytes;
        scratch_words += (uint64_t)fc.cap_slots * fc.slot_words + fc.cap_slots + (uint64_t)nv * max_sample_haplotypes + nv + fc.cur_cap + nv + nv + MT_WORDS;
        scratch_words = (scratch_words + 3) & ~3ull;
        best_bytes += (uint64_t)fc.best_cap * nv;
        f->clusters.push_back(fc);
        f->nv.push_back(nv);
    }
    f->best_bytes = best_bytes;
    int rc = BT_OK;
    auto up = [&](auto **dst, const auto *src, uint64_t n) {
        if (rc != BT_OK) return;
        using T = std::remove_cv_t<std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>>;
        T *p = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&p), std::max<uint64_t>(n, 1) * sizeof(T)) != hipSuccess) {
            rc = fail("bt_find_paths_create: device allocation failed");
            return;
        }
        f->owned.push_back(p);
        if (src && n && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) rc = fail("bt_find_paths_create: upload failed");
        *dst = p;
    };
    uint64_t *d_seq_off = nullptr;
    uint8_t *d_seq = nullptr, *d_vflags = nullptr;
    uint32_t *d_in_off = nullptr, *d_in_src = nullptr, *d_last = nullptr;
    up(&d_seq_off, b->seq_off, (uint64_t)NV + 1);
    up(&d_seq, b->seq, b->seq_off[NV]);
    up(&d_vflags, b->vertex_flags, NV);
    up(&d_in_off, b->in_off, (uint64_t)NV + 1);
    up(&d_in_src, b->in_src, b->in_off[NV]);
    up(&d_last, last_use.data(), NV);
    up(&f->d_clusters, f->clusters.data(), C);
    up(&f->d_scratch, (const uint32_t *)nullptr, scratch_words);
    up(&f->d_best, (const uint8_t *)nullptr, best_bytes);
    up(&f->d_best_count, (const uint32_t *)nullptr, C);
    up(&f->d_overflow, (const uint32_t *)nullptr, 1);
    up(&f->d_seeds, (const uint32_t *)nullptr, C);
    if (rc == BT_OK && (hipMemset(f->d_best_count, 0, (size_t)C * 4) != hipSuccess || hipMemset(f->d_overflow, 0, 4) != hipSuccess)) rc = fail("bt_find_paths_create: memset failed");
    if (rc != BT_OK) {
        bt_find_paths_destroy(f);
        return rc;
    }
    f->g = FindGraph{d_seq_off, d_seq, d_vflags, d_in_off, d_in_src, d_last};
    *out = f;
    return BT_OK;
}

int bt_find_paths_destroy(bt_find_paths *f) {
    if (!f) return BT_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    for (void *q : f->owned) (void)hipFree(q);
    delete f;
    return BT_OK;
}

int bt_find_paths_sample(bt_find_paths *f, bt_bloom *sample_bloom, const uint32_t *h_seeds) {
    if (!f || !sample_bloom || !h_seeds) return fail("bt_find_paths_sample: null argument");
    if (sample_bloom->k != f->k) return fail("bt_find_paths_sample: k mismatch");
    BT_HIP(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    BT_HIP(hipMemcpyAsync(f->d_seeds, h_seeds, (size_t)f->C * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(find_paths_kernel, dim3((f->C + 63) / 64), dim3(64), 0, st, f->d_clusters, f->C, f->g, sample_bloom->view(), f->d_seeds, f->k, f->max_haps,
                       f->d_scratch, f->d_best, f->d_best_count, f->d_overflow);
    BT_CHECK_LAUNCH();
    uint32_t ov = 0;
    BT_HIP(hipMemcpyAsync(&ov, f->d_overflow, 4, hipMemcpyDeviceToHost, st));
    BT_HIP(hipStreamSynchronize(st));
    if (ov == 1) return fail("bt_find_paths_sample: path scratch exhausted (internal sizing error)");
    if (ov == 2) return fail("bt_find_paths_sample: more best paths than max_sample_haplotypes x num_samples");
    return BT_OK;
}

int bt_find_paths_sizes(bt_find_paths *f, uint32_t *h_num_paths, uint64_t *h_total_bytes) {
    if (!f || !h_num_paths) return fail("bt_find_paths_sizes: null argument");
    BT_HIP(hipSetDevice(f->ctx->device));
    BT_HIP(hipStreamSynchronize(f->ctx->stream));
    BT_HIP(hipMemcpy(h_num_paths, f->d_best_count, (size_t)f->C * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t c = 0; c < f->C; ++c) total += (uint64_t)h_num_paths[c] * f->nv[c];
    if (h_total_bytes) *h_total_bytes = total;
    return BT_OK;
}

int bt_find_paths_fetch(bt_find_paths *f, uint8_t *h_path_vertices) {
    if (!f || !h_path_vertices) return fail("bt_find_paths_fetch: null argument");
    BT_HIP(hipSetDevice(f->ctx->device));
    BT_HIP(hipStreamSynchronize(f->ctx->stream));
    std::vector<uint32_t> n(f->C);
    BT_HIP(hipMemcpy(n.data(), f->d_best_count, (size_t)f->C * 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> all(f->best_bytes);
    if (f->best_bytes) BT_HIP(hipMemcpy(all.data(), f->d_best, f->best_bytes, hipMemcpyDeviceToHost));
    uint8_t *o = h_path_vertices;
    for (uint32_t c = 0; c < f->C; ++c) {
        const uint64_t bytes = (uint64_t)n[c] * f->nv[c];
        if (bytes) std::memcpy(o, all.data() + f->clusters[c].best, bytes);
        o += bytes;
    }
    return BT_OK;
}

}  // extern "C"
