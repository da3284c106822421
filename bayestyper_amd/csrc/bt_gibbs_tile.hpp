// Device-side Gibbs genotyper: one variant-cluster group per lane, 64 groups per wavefront ("tile"), gfx950.
//
// Reference behaviour restated (not copied); every function cites what it mirrors:
//   VariantClusterGenotyper   src/bayesTyper/VariantClusterGenotyper.cpp:59-206,569-785
//   VariantClusterHaplotypes  src/bayesTyper/VariantClusterHaplotypes.cpp:45-372
//   (Sparse)FrequencyDistribution / HaplotypeFrequencyDistribution
//                             src/bayesTyper/FrequencyDistribution.cpp:43-303, HaplotypeFrequencyDistribution.cpp:79-138
//   SparsityEstimator         src/bayesTyper/SparsityEstimator.cpp:41-87
//   LogDiscreteSampler        src/bayesTyper/DiscreteSampler.cpp:100-125, Utils::logAddition Utils.hpp:105-124
//   KmerStats                 src/bayesTyper/KmerStats.cpp:51-121
//   VariantClusterGroup       src/bayesTyper/VariantClusterGroup.cpp:171-250
//
// HBM layout.  The sampler is a chain of dependent random draws per group, so the parallel axis is the group: a lane owns
// a group for a whole launch.  64 groups of similar shape form a TILE handled by one wavefront.  Every array of a tile is
// lane-interleaved: element i of vertex v of lane l sits at  off_X + ((v*LEN_X + i)*64 + l)*sizeof(T), with off_X / LEN_X
// tile-uniform (scalar registers) — a wave-uniform index is ONE coalesced transaction instead of 64 scattered ones, and
// there is no per-lane pointer to chase.  Arrays are padded to the tile's maximum dimensions; each lane keeps its true
// dimensions.  Only the mt19937 states are per-lane contiguous (their position drifts between lanes).
#pragma once
#include "bt_rng_device.hpp"

namespace bt {

constexpr unsigned LANES = 64;
constexpr uint16_t NOHAP = 0xFFFF;
constexpr double BT_LN2 = 0.693147180559945309417232121458176568;
constexpr double BT_DBL_EPS = 2.220446049250313080847263336181640625e-16;
#ifndef BT_LINEAR_DRAW_MIN
#define BT_LINEAR_DRAW_MIN 4u   // candidate sets up to this size always use the reference's chain of logAddition calls
#endif
#ifndef BT_UC_INVALIDATE_MIN
#define BT_UC_INVALIDATE_MIN 64u   // clearGenotyperCache between the sweeps of a chain: tables above this many entries are invalidated and refilled on demand (block-wise), smaller ones rebuilt whole
#endif
#ifndef BT_EVAL_BLOCK
#define BT_EVAL_BLOCK 4u        // candidates evaluated per step of sample_diplotypes' blocked evaluation (their cache words are requested together)
#endif
constexpr uint32_t EVB = BT_EVAL_BLOCK;
// BT_SINGLE: the translation unit of gibbs_single_kernel (bt_gibbs_single_kernel.hip) — every group of its tiles is ONE cluster without multicluster
// k-mers (TileDesc::logged tiles: nvm == 1, NMm == 0).  The nested-group traversal, the multicluster sums and the immediate statistics path are
// compiled out of the sweep, which then fits three wavefronts per SIMD.
#ifdef BT_SINGLE
constexpr bool kSingle = true;
#else
constexpr bool kSingle = false;
#endif
// BT_PACKED: the translation units of gibbs_hot_kernel / gibbs_single_kernel.  A launch has ONE dynamic LDS size, and a schedule of a whole-genome batch is
// LDS-capacity-bound (sum over the tiles of LDS x duration against 160 KB per CU: the model reproduces the measured 3.7 s of the round-5 bench batch to 1 %), so
// charging every tile of a launch class its hungriest tile's need cost a third of the non-simple tiles' LDS.  In a packed launch a workgroup is a SLAB of LDS
// shared by up to blockDim.x / 64 wavefronts that each run ANOTHER tile (one-wavefront tiles only): wavefront w of workgroup b takes slot b * waves + w of the
// launch's slot list — (tile, byte offset of the tile's block in the slab) pairs, 0xFFFFFFFF = no tile — which the host fills by best-fit-decreasing bin
// packing (bt_gibbs.hip: pack_class).  The wavefronts of a workgroup never synchronise with each other.
#ifdef BT_PACKED
constexpr bool kPacked = true;
#else
constexpr bool kPacked = false;
#endif
constexpr unsigned MT_PAD = 640;   // words reserved per generator (625 used)
#ifdef BT_MT_BLOCK
constexpr unsigned MT_RING_PAD = 2 * MT_BUF;   // a cluster's ring generators keep two state buffers (bt_rng_device.hpp: block form)
#else
constexpr unsigned MT_RING_PAD = MT_PAD;
#endif

// scalar slots per vertex (A_SC)
// SC_H .. SC_NM: the cluster's dimensions (copied from A_VDIMS by OP_SETUP, never cleared): with the scalars in LDS a sampler function
// starts without a round trip to HBM
enum { SC_USE_MULTI = 0, SC_NSUB_U, SC_NSUB_M, SC_HAP_COUNT, SC_CONSTRUCTED, SC_DIP_ENTRIES, SC_DIP_OVERFLOW, SC_IS_SPARSE, SC_FND_AVAIL, SC_UC_DIRTY, SC_H, SC_V, SC_NM, SC_COUNT };
constexpr uint32_t SC_STATE_COUNT = SC_H;   // slots a (re)constructed genotyper resets

// arrays of a tile.  [V] = per vertex (index v*LEN + i), [G] = per group
enum TileArr {
    // ---- inputs ----
    A_M = 0,        // u8  [V] Km*Hm      haplotype_kmer_multiplicities, index k*Hm + h
    A_HASC,         // u8  [V] Km
    A_COUNTS,       // u8  [V] Km*S
    A_IC,           // u8  [V] Km*2
    A_SHARED,       // i32 [V] Km
    A_KVOFF,        // u32 [V] Km+1       CSR offsets (0-based within the vertex)
    A_KVVAR,        // u16 [V] NNZm
    A_KVBITS,       // u32 [V] NNZm*HWm
    A_HAPAL,        // u16 [V] Hm*Vm      index h*Vm + v
    A_HAPCELL,      // u32 [V] Hm*Vm      index h*Vm + v: low 16 bits = the variant whose k-mer stats haplotype h contributes to variant v (v itself, or the last
                    //                    variant before v without a missing allele when h's allele of v is the missing one; 0xFFFF: none), high 16 bits =
                    //                    allele cell of (v, allele of h) within a sample's allele statistics (ALBASE[v] + allele)
    A_HNOFF,        // u32 [V] Hm+1
    A_HNIDX,        // u32 [V] HNm
    A_VARNA,        // u16 [V] Vm
    A_VARDEP,       // u8  [V] Vm
    A_ALBASE,       // u32 [V] Vm+1
    A_NDCL,         // u32 [V] NDm
    A_NDVOFF,       // u32 [V] NDm+1
    A_NDVAR,        // u16 [V] NDVm
    A_UNIQ0,        // u32 [V] NUm
    A_MULTI0,       // u32 [V] NMm
    A_EDGES0,       // u32 [V] NEm
    A_VDIMS,        // u32 [V] 8: H, V, K, nu, nm, nd, ne, cid
    A_VDIMS2,       // u32 [V] 2: A (alleles), reserved
    A_GDIMS,        // u32 [G] 4: nvert, nsrc, group index, valid
    A_SOURCES0,     // u32 [G] nvm
    A_PLOIDY,       // u8  [G] S
    // ---- state ----
    A_MT,           // u32 [V] 2 generators, per-lane contiguous (MT_PAD words each)
    A_FNDSAVED,     // f64 [V] 1
    A_SPARSITY,     // f64 [V] 1
    A_UNIQ, A_MULTI, A_USUB, A_MSUB,   // u32 [V]
    A_SMM,          // u8  [V] NMm*S
    A_DIP,          // u16 [V] 2*S
    A_FREQ,         // f64 [V] Hm
    A_OBS,          // u32 [V] Hm
    A_NZ,           // u8  [V] Hm
    A_ZHDR, A_ZBKT, A_PHDR, A_PBKT, A_UNEXT,   // u32 [V] 4, Bcap, 4, Bcap, Hm
    A_HVCOUNT,      // u32 [V] Hm*Vm
    A_UCACHE,       // f64 [V] cache_entries
    A_UCTAG,        // u32 [V] cache_entries (mode 1) or 1
    A_CUM,          // f64 [V] max(D2m,1)
    A_NZLIST,       // u16 [V] Hm
    A_SIMPLEX,      // f64 [V] Hm+1
    A_SCACHE,       // f64 [V] scache_n*scache_len   cached simplex probability vectors
    A_SCLEN,        // u32 [V] scache_n              their lengths (0 = not computed)
    A_KSC,          // f64 [V] S*2*Vm*4
    A_KSCUPD,       // u8  [V] S
    A_DIPKEYS,      // u32 [V] dip_cap
    A_DIPFREQ,      // u32 [V] dip_cap*S
    A_ASTATS,       // f64 [V] S*Am*12
    A_NESTPL, A_NESTN,   // u8 [V] S
    A_NESTST,       // f64 [V] S*2*4
    A_KSCKEY,       // u32 [V] S*(KSC_WAYS+1)   per sample: the diplotypes (h1 | h2 << 16) whose k-mer-stats cache is kept in A_KSCDATA, then the next entry to replace
    A_KSCDATA,      // f64 [V] S*KSC_WAYS*2*Vm*4  those caches ([sample][entry][haplotype slot][variant] KmerStats)
    A_EVLOG,        // u32 [V] S*(2*EV_CAP+1)   tiles of two-haplotype clusters: per sample the runs of collected sweeps not yet applied to the statistics —
                    //                    (diplotype h1 | h2 << 16, run length) pairs in order from [s][1] on (bt_gibbs_simple.hpp: simple_drain)
    A_EVN,          // u8  [V] S          ... and how many of them (kept with the hot arrays)
    A_NVER,         // u32 [V] 2*S        [s]: version of what a child's nested info is derived from (the sample's diplotype, its k-mer-stats cache, the vertex's own
                    //                    nested info); [S + s]: the parent's version this vertex's nested info was last prepared from
    A_PENDNEST,     // f64 [V] S*2*4      the nested sources the pending (deferred) collected sweeps of a sample saw: [s][j][count, fraction, mean], [s][0][3] = how many
    A_SC,           // u32 [V] SC_COUNT
    A_EDGES,        // u32 [V] NEm
    A_COVER,        // u8  [V] Km
    A_MCACHE,       // f64 [V] cache_entries  multicluster log-prob sums per (sample, diplotype)
    A_MCTAG,        // u32 [V] cache_entries  key + 1 (0 = empty)
    A_MCGEN,        // u32 [V] cache_entries  generation of the sample at fill time
    A_MGEN,         // u32 [V] S              current generation per sample (bumped when the other clusters' contribution changes)
    A_OTH,          // u8  [V] NMm*S          other clusters' multiplicity per subset k-mer at the current generation
    A_SUBM,         // u8  [V] NUm*Hm         multiplicity rows of the unique subset k-mers, in subset order (compact copy of A_M rows)
    A_SUBCNT,       // u8  [V] NUm*S          their observed counts (0 when the k-mer has no count record)
    A_SUBIC,        // u8  [V] NUm*2          their intercluster multiplicities (0 when no count record)
    A_SKVOFF,       // u32 [V] NUm+1          variant_haplotype_indices CSR of the unique subset k-mers, in subset order
    A_SKVVAR,       // u16 [V] NNZm
    A_SKVBITS,      // u32 [V] NNZm*HWm
    A_KSCTMP,       // f64 [V] 2*Vm*4         scratch accumulators of one sample's k-mer-stats cache rebuild
    A_LOGF,         // f64 [V] Hm             log(frequency) of the non-zero haplotypes (pure function of A_FREQ, refreshed with it)
    A_PEND,         // u32 [V] S    collected sweeps not yet materialised for the sample (run-length of identical contributions)
    A_PENDDIP,      // u16 [V] 2*S  the diplotype those pending sweeps drew
    A_PENDVALID,    // u8  [V] S
    A_SOURCES,      // u32 [G] nvm
    A_STACK,        // u32 [G] 2*(nvm+1)
    A_BRNG,         // u32 [G] per-lane contiguous MT_PAD
    A_SHMULT,       // u8  [G] NSHm*S
    A_MSUBM,        // u8  [V] NMm*Hm         multiplicity rows of the multicluster subset k-mers, in subset order
    A_MSUBC,        // u8  [V] NMm*S          their observed counts
    A_MSUBIC,       // u8  [V] NMm*2          their intercluster multiplicities
    A_MSUBSH,       // u32 [V] NMm            their slot in the group's shared-multiplicity table
    A_RING,         // u32 [V] ring_len       draw-ahead rings of the cluster's two generators (MtRing blocks: ring_cap[g] + MT_RING_HDR words each)
    A_COUNT
};

struct TileDesc {
    uint32_t S, nvm, Hm, Vm, Km, HWm, NUm, NMm, NNZm, HNm, NDm, NDVm, NEm, NSHm, Am, Bcap, D2m, Dcm, cache_mode, cache_entries, dip_cap, scache_n, scache_len,
        scache_p, num_lanes, first_group;
    uint64_t base;            // byte offset of this tile in the pool
    uint64_t trace_base;      // word offset of this tile's trace block ([sweep][vertex][S][64])
    uint64_t off[A_COUNT];    // byte offsets of the arrays from the tile base
    uint32_t hoff[A_COUNT];   // byte offset inside the wavefront's LDS block of the arrays kept resident there (NOHOT otherwise)
    uint32_t hot_bytes, split;   // hot_bytes: LDS bytes of ONE vertex's hot arrays; split: wavefronts working on this tile (1, 2, 4 or 8), see tile_lane()
    uint32_t copies, copies_pad;    // narrow tiles (split 1): the wavefront's idle lanes run `copies` identical instances of every group, see Tile::part
    uint32_t uc_width;              // 0: the unique-sum table is lane-interleaved like every other array; else it is per-lane contiguous over uc_width lanes
    uint32_t mat_width;             // the same for the two K x H multiplicity matrices (A_M, A_SUBM): 0 interleaved, else the tile's lane count
    uint32_t lds_stride, lds_all;   // lanes a hot-array row is interleaved over in LDS (4 .. 64); lds_all: every vertex has its own LDS block (no swaps)
    uint32_t simple;                // every group of the tile is one two-haplotype cluster without multicluster k-mers: sweeps run in simple_sweeps()
    uint32_t ring_cap[2], ring_len; // draw-ahead words of the diplotype / frequency generator (powers of two); ring_len = both blocks
    uint32_t prio;                  // the tile's wavefronts raise their issue priority (narrow tiles: the launch's critical path)
    uint32_t pool_lane0;            // first lane of this tile in the pool block it shares with the neighbouring narrow tiles (0 for 64-lane tiles)
    uint32_t logged;                // single clusters without multicluster k-mers: collected sweeps are logged as runs and applied at the end of the chain
    uint32_t wsh;                   // log2 of the tile's width W (power of two >= num_lanes): every array of the tile is interleaved over W lanes, in HBM and in LDS
    uint32_t teams;                 // sample_diplotypes: the copies form this many teams that draw as many samples at a time (1: none); A_CUM holds one block per team
    uint32_t sblk;                  // tiles of two-haplotype clusters: LDS byte offset of the per-sample words of simple_sweeps() (bt_gibbs_simple.hpp), after the rings
};
#ifndef BT_EV_CAP
#define BT_EV_CAP 32
#endif
constexpr uint32_t EV_CAP = BT_EV_CAP;    // logged runs per (cluster, sample) before the log is applied early
constexpr uint32_t KSC_WAYS = 4;          // recently rebuilt k-mer-stats caches kept per (cluster, sample), see collect_sample_body
constexpr uint32_t KSC_NOKEY = 0xFFFEFFFEu;   // (haplotype indices are < 0xFFFE)
constexpr uint32_t NOHOT = 0xFFFFFFFFu;
constexpr uint32_t RESIDENT_ALL = 0xFFFFFFFEu;   // Env/Tile::resident: every vertex of the group has its hot arrays in LDS
constexpr uint32_t RESIDENT_NEVER = 0xFFFFFFFDu; // Env/Tile::resident: this launch keeps the tile's hot arrays in HBM (gibbs_chain_kernel: a tile whose LDS block would keep the chain's tiles from being resident together)

struct GParams {
    uint32_t S, seed, num_chains, burn_in, num_iterations, max_hvk, noise_seeding;
    double rate;                // (double)(float)kmer_subsampling_rate, as bernoulli_distribution stores it
    uint8_t gender[32];
    const double BT_GAS *lut_g;        // [S][256][256]
    const double BT_GAS *lut_n;        // [S][256]
    const double BT_GAS *lgamma_int;   // lgamma(n) for integer n in [0, lgamma_n): computed on the host (libm), gathered on the device
    const double BT_GAS *gamma_a2;     // 1 / sqrt(9 (n - 1/3)) for integer n in [1, gamma_n): Marsaglia-Tsang's a2 for alpha = an observation count + 1
    uint32_t lgamma_n, gamma_n;
};

extern __shared__ __attribute__((aligned(16))) uint8_t bt_lds_raw[];
__device__ inline uint8_t *lds_block() { return (uint8_t *)bt_lds_raw; }
// packed launches: this wavefront's slot (wave-uniform)
__device__ inline uint32_t pack_slot() { return blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// The arrays every tile with an LDS block keeps there (bt_gibbs.hip: hot_arrs, minus the ones that depend on the tile's shape).  In the translation
// unit of gibbs_hot_kernel (BT_HOT_ALL: launch classes whose tiles all keep every vertex resident for the whole launch) an access to one of them is an
// LDS access at compile time — a ds instruction counted by lgkmcnt alone — instead of a flat access through a pointer that may be either, which
// the hardware counts on both counters and the compiler has to wait out with both at zero.
__host__ __device__ constexpr bool hot_core(int a) {
    return a == A_SC || a == A_DIP || a == A_NESTPL || a == A_NESTN || a == A_KSCUPD || a == A_PEND || a == A_PENDDIP || a == A_PENDVALID || a == A_EVN || a == A_FREQ || a == A_LOGF ||
           a == A_OBS || a == A_NZ || a == A_NZLIST || a == A_UNEXT || a == A_ZHDR || a == A_ZBKT || a == A_PHDR || a == A_PBKT || a == A_RING || a == A_FNDSAVED;
}

// ---- lane view of a tile / of one vertex of the lane's group ------------------------------------------------------
struct Tile {
    uint8_t BT_GAS *base;
    const TileDesc BT_CAS *d;
    uint32_t lane;      // the group's lane in its tile: LDS rows, trace rows
    uint32_t plane;     // its lane in the pool block the tile shares with its neighbours (TileDesc::pool_lane0 + lane): everything in HBM
    // Narrow tiles leave most lanes of their wavefront idle.  Instead, 64 / width ("copies") threads run the SAME group: identical
    // program, identical loads, identical stores to identical addresses (SIMT lockstep makes every read-modify-write of the copies
    // read before any of them writes) — which costs nothing — and the phases that are data-parallel inside one group (the dense
    // table fill, the compact subset copies) are divided among the copies by `part`.
    uint32_t part, copies;
    uint32_t wsh;        // TileDesc::wsh
    uint8_t *hot;        // this wavefront's LDS block (generic pointer) or nullptr
    uint32_t lds0;       // byte offset of that block in the workgroup's LDS (packed launches; 0 otherwise)
    uint32_t resident;   // vertex whose hot arrays currently live in LDS (0xFFFFFFFF: none)
    // hot-capable array: LDS when the vertex is resident, HBM otherwise; one code path through generic pointers
    template <typename T>
    __device__ inline SPtrF<T, LANES> harr(int a, uint32_t v, uint32_t len) const {
#ifdef BT_HOT_ALL
        if (hot_core(a)) return SPtrF<T, LANES>{(T *)(lds_block() + (kPacked ? lds0 : 0u) + v * d->hot_bytes + d->hoff[a]), lane, wsh};
#endif
        const uint32_t ho = d->hoff[a];
        if (hot != nullptr && ho != NOHOT && (resident == RESIDENT_ALL || v == resident))
            return SPtrF<T, LANES>{(T *)(hot + (resident == RESIDENT_ALL ? v * d->hot_bytes : 0u) + ho), lane, wsh};
        return SPtrF<T, LANES>{(T *)(uint8_t *)(base + d->off[a]), ((v * len) << wsh) + plane, wsh};
    }
    template <typename T>
    __device__ inline TPtr<T> arr(int a, uint32_t first = 0) const {
        return TPtr<T>{(T BT_GAS *)(base + d->off[a]), (first << wsh) + plane, wsh};
    }
};

// the kernel's wave-uniform environment, re-derived inside every non-inlined function (arguments of a call are vector
// registers to the compiler; readfirstlane turns them back into scalars)
struct Env {
    const TileDesc *tiles;
    uint8_t *pool;
    const struct GParams *P;
    const uint32_t *tile_list;   // block -> tile index (nullptr: identity)
    uint32_t resident;           // vertex resident in LDS, 0xFFFFFFFF = none
};

struct Vx {   // vertex context: tile + vertex index + the lane's true dimensions of that vertex
    Tile t;
    uint32_t v, H, V, nm;
    __device__ inline const TileDesc BT_CAS &d() const { return *t.d; }
    // dimensions only the chain start needs stay in HBM
    __device__ inline uint32_t K() const { return t.arr<uint32_t>(A_VDIMS, v * 8)[2]; }
    __device__ inline uint32_t nu() const { return t.arr<uint32_t>(A_VDIMS, v * 8)[3]; }
    __device__ inline uint32_t cid() const { return t.arr<uint32_t>(A_VDIMS, v * 8)[7]; }
    template <typename T>
    __device__ inline TPtr<T> a(int arr, uint32_t len) const { return t.arr<T>(arr, v * len); }
    template <typename T>
    struct RPtr {   // global-memory array with a run-time element stride
        T BT_GAS *base;
        uint32_t off, sh;
        __device__ inline T BT_GAS &operator[](uint32_t i) const { return base[off + (i << sh)]; }
        __device__ inline RPtr<T> operator+(uint32_t i) const { return RPtr<T>{base, off + (i << sh), sh}; }
    };
    template <typename T>
    struct FPtr {   // the same through a generic pointer (LDS or HBM)
        T *base;
        uint32_t off, sh;
        __device__ inline T &operator[](uint32_t i) const { return base[off + (i << sh)]; }
        __device__ inline FPtr<T> operator+(uint32_t i) const { return FPtr<T>{base, off + (i << sh), sh}; }
    };
    typedef FPtr<double> UCPtr;
    // the K x H multiplicity matrices of a narrow tile are per-lane contiguous too (they are the bulk of a large cluster's state)
    __device__ inline RPtr<uint8_t> mat(int arr, uint32_t rows) const {
        uint8_t BT_GAS *b = (uint8_t BT_GAS *)(t.base + d().off[arr]);
        const uint32_t w = d().mat_width, n = rows * d().Hm;
        if (w) return RPtr<uint8_t>{b, (v * w + t.plane) * n, 0u};
        return RPtr<uint8_t>{b, ((v * n) << t.wsh) + t.plane, t.wsh};
    }
    // inputs
    __device__ inline uint8_t M(uint32_t k, uint32_t h) const { return mat(A_M, d().Km)[(uint32_t)k * d().Hm + h]; }
    __device__ inline uint8_t has_counts(uint32_t k) const { return a<uint8_t>(A_HASC, d().Km)[k]; }
    __device__ inline uint8_t count(uint32_t k, uint32_t s) const { return a<uint8_t>(A_COUNTS, (uint32_t)d().Km * d().S)[(uint32_t)k * d().S + s]; }
    __device__ inline uint8_t ic(uint32_t k, uint32_t g) const { return a<uint8_t>(A_IC, (uint32_t)d().Km * 2)[2 * k + g]; }
    __device__ inline int32_t shared_idx(uint32_t k) const { return a<int32_t>(A_SHARED, d().Km)[k]; }
    __device__ inline uint32_t kv_off(uint32_t k) const { return a<uint32_t>(A_KVOFF, d().Km + 1)[k]; }
    __device__ inline uint16_t kv_var(uint32_t e) const { return a<uint16_t>(A_KVVAR, d().NNZm)[e]; }
    __device__ inline bool kv_bit(uint32_t e, uint32_t h) const { return (a<uint32_t>(A_KVBITS, (uint32_t)d().NNZm * d().HWm)[(uint32_t)e * d().HWm + (h >> 5)] >> (h & 31u)) & 1u; }
    __device__ inline uint32_t hap_cell(uint32_t h, uint32_t var) const { return a<uint32_t>(A_HAPCELL, (uint32_t)d().Hm * d().Vm)[(uint32_t)h * d().Vm + var]; }
    __device__ inline TPtr<double> astats_cell(uint32_t s, uint32_t cell) const { return a<double>(A_ASTATS, (uint32_t)d().S * d().Am * 12) + ((uint32_t)s * d().Am + cell) * 12; }
    __device__ inline uint16_t hap_allele(uint32_t h, uint32_t var) const { return a<uint16_t>(A_HAPAL, (uint32_t)d().Hm * d().Vm)[(uint32_t)h * d().Vm + var]; }
    __device__ inline uint32_t hn_off(uint32_t h) const { return a<uint32_t>(A_HNOFF, d().Hm + 1)[h]; }
    __device__ inline uint32_t hn_idx(uint32_t i) const { return a<uint32_t>(A_HNIDX, d().HNm)[i]; }
    __device__ inline uint16_t var_na(uint32_t var) const { return a<uint16_t>(A_VARNA, d().Vm)[var]; }
    __device__ inline uint8_t var_dep(uint32_t var) const { return a<uint8_t>(A_VARDEP, d().Vm)[var]; }
    __device__ inline uint32_t allele_base(uint32_t var) const { return a<uint32_t>(A_ALBASE, d().Vm + 1)[var]; }
    // state
    __device__ inline uint32_t *mt(uint32_t g) const { return (uint32_t *)(t.base + d().off[A_MT]) + ((size_t)((v * 2 + g) << t.wsh) + t.plane) * MT_RING_PAD; }
    __device__ inline SPtrF<uint32_t, LANES> ring(uint32_t g) const { return t.harr<uint32_t>(A_RING, v, d().ring_len) + (g ? d().ring_cap[0] + MT_RING_HDR : 0u); }
    __device__ inline MtRing rng(uint32_t g) const { return mt_ring_open(mt(g), ring(g), d().ring_cap[g]); }
    __device__ inline void rng_seed(uint32_t g, uint32_t seed) const { mt_ring_seed(mt(g), ring(g), d().ring_cap[g], seed); }
    __device__ inline SPtrF<uint32_t, LANES> sc() const { return t.harr<uint32_t>(A_SC, v, SC_COUNT); }
    __device__ inline TPtr<uint32_t> uniq() const { return a<uint32_t>(A_UNIQ, d().NUm); }
    __device__ inline TPtr<uint32_t> multi() const { return a<uint32_t>(A_MULTI, d().NMm); }
    __device__ inline TPtr<uint32_t> usub() const { return a<uint32_t>(A_USUB, d().NUm); }
    __device__ inline TPtr<uint32_t> msub() const { return a<uint32_t>(A_MSUB, d().NMm); }
    __device__ inline TPtr<uint8_t> smm() const { return a<uint8_t>(A_SMM, (uint32_t)d().NMm * d().S); }
    __device__ inline SPtrF<uint16_t, LANES> dip() const { return t.harr<uint16_t>(A_DIP, v, 2 * d().S); }
    __device__ inline SPtrF<double, LANES> freq() const { return t.harr<double>(A_FREQ, v, d().Hm); }
    __device__ inline RPtr<uint8_t> subm() const { return mat(A_SUBM, d().NUm); }
    __device__ inline TPtr<uint8_t> subcnt() const { return a<uint8_t>(A_SUBCNT, d().NUm * d().S); }
    __device__ inline TPtr<uint8_t> subic() const { return a<uint8_t>(A_SUBIC, d().NUm * 2); }
    __device__ inline TPtr<uint32_t> skv_off() const { return a<uint32_t>(A_SKVOFF, d().NUm + 1); }
    __device__ inline TPtr<uint16_t> skv_var() const { return a<uint16_t>(A_SKVVAR, d().NNZm > 1 ? d().NNZm : 1); }
    __device__ inline TPtr<uint32_t> skv_bits() const { return a<uint32_t>(A_SKVBITS, (d().NNZm > 1 ? d().NNZm : 1) * d().HWm); }
    __device__ inline SPtrF<double, LANES> ksc_tmp() const { return t.harr<double>(A_KSCTMP, v, 2 * d().Vm * 4); }
    __device__ inline SPtrF<double, LANES> logf() const { return t.harr<double>(A_LOGF, v, d().Hm); }
    __device__ inline SPtrF<uint32_t, LANES> obs() const { return t.harr<uint32_t>(A_OBS, v, d().Hm); }
    __device__ inline SPtrF<uint8_t, LANES> nz() const { return t.harr<uint8_t>(A_NZ, v, d().Hm); }
    __device__ inline SPtrF<uint32_t, LANES> unext() const { return t.harr<uint32_t>(A_UNEXT, v, d().Hm); }
    typedef USetP<SPtrF<uint32_t, LANES>> HSet;
    __device__ inline HSet zero_set() const { return HSet{t.harr<uint32_t>(A_ZHDR, v, 4), t.harr<uint32_t>(A_ZBKT, v, d().Bcap), unext(), d().Bcap}; }
    __device__ inline HSet plus_set() const { return HSet{t.harr<uint32_t>(A_PHDR, v, 4), t.harr<uint32_t>(A_PBKT, v, d().Bcap), unext(), d().Bcap}; }
    __device__ inline TPtr<uint32_t> hvcount() const { return a<uint32_t>(A_HVCOUNT, (uint32_t)d().Hm * d().Vm); }
    // the [S][D] table of unique-k-mer sums.  Wide tiles: lane-interleaved.  Narrow tiles (whose lanes would leave most of every
    // 64-lane row unused): each lane's table contiguous, so a tile pays for its own lanes only and even 256 candidates x 30 samples
    // (10^6 entries, 8 MB) stay dense
    __device__ inline UCPtr ucache() const {
        const uint32_t ho = d().hoff[A_UCACHE];
        if (t.hot != nullptr && ho != NOHOT && (t.resident == RESIDENT_ALL || v == t.resident))   // a table of a few entries lives in LDS
            return UCPtr{(double *)(t.hot + (t.resident == RESIDENT_ALL ? v * d().hot_bytes : 0u) + ho), t.lane, t.wsh};
        double *b = (double *)(uint8_t *)(t.base + d().off[A_UCACHE]);
        const uint32_t w = d().uc_width;
        if (w) return UCPtr{b, (v * w + t.plane) * d().cache_entries, 0u};
        return UCPtr{b, ((v * d().cache_entries) << t.wsh) + t.plane, t.wsh};
    }
    __device__ inline TPtr<uint32_t> uctag() const { return a<uint32_t>(A_UCTAG, d().cache_mode == 1 ? d().cache_entries : 1); }
    __device__ inline SPtrF<double, LANES> cum() const { return t.harr<double>(A_CUM, v, (d().D2m > 1 ? d().D2m : 1) * (d().teams > 1 ? d().teams : 1)); }
    __device__ inline SPtrF<uint16_t, LANES> nzlist() const { return t.harr<uint16_t>(A_NZLIST, v, d().Hm); }
    __device__ inline TPtr<double> simplex() const { return a<double>(A_SIMPLEX, d().Hm + 1); }
    __device__ inline TPtr<double> scache() const { return a<double>(A_SCACHE, (uint32_t)(d().scache_n ? d().scache_n : 1) * (d().scache_len ? d().scache_len : 1)); }
    __device__ inline TPtr<uint32_t> sclen() const { return a<uint32_t>(A_SCLEN, d().scache_n > 1 ? d().scache_n : 1); }
    __device__ inline TPtr<double> ksc(uint32_t s, uint32_t which, uint32_t var) const {
        return a<double>(A_KSC, (uint32_t)d().S * 2 * d().Vm * 4) + (((uint32_t)s * 2 + which) * d().Vm + var) * 4;
    }
    __device__ inline SPtrF<uint8_t, LANES> ksc_upd() const { return t.harr<uint8_t>(A_KSCUPD, v, d().S); }
    __device__ inline TPtr<uint32_t> dip_keys() const { return a<uint32_t>(A_DIPKEYS, d().dip_cap); }
    __device__ inline TPtr<uint32_t> dip_freq() const { return a<uint32_t>(A_DIPFREQ, (uint32_t)d().dip_cap * d().S); }
    __device__ inline TPtr<double> astats(uint32_t s, uint32_t var, uint32_t al) const {
        return a<double>(A_ASTATS, (uint32_t)d().S * d().Am * 12) + ((uint32_t)s * d().Am + allele_base(var) + al) * 12;
    }
    __device__ inline SPtrF<uint8_t, LANES> nest_ploidy() const { return t.harr<uint8_t>(A_NESTPL, v, d().S); }
    __device__ inline SPtrF<uint8_t, LANES> nest_n() const { return t.harr<uint8_t>(A_NESTN, v, d().S); }
    __device__ inline TPtr<uint32_t> ksc_key(uint32_t s) const { return a<uint32_t>(A_KSCKEY, (uint32_t)d().S * (KSC_WAYS + 1u)) + (uint32_t)s * (KSC_WAYS + 1u); }
    __device__ inline TPtr<double> ksc_data(uint32_t s, uint32_t e) const {
        return a<double>(A_KSCDATA, (uint32_t)d().S * KSC_WAYS * 2 * d().Vm * 4) + ((uint32_t)s * KSC_WAYS + e) * 2 * d().Vm * 4;
    }
    __device__ inline TPtr<uint32_t> evlog(uint32_t s) const { return a<uint32_t>(A_EVLOG, (uint32_t)d().S * (2u * EV_CAP + 1u)) + (uint32_t)s * (2u * EV_CAP + 1u); }
    __device__ inline SPtrF<uint8_t, LANES> evn() const { return t.harr<uint8_t>(A_EVN, v, d().S); }
    __device__ inline TPtr<uint32_t> nver() const { return a<uint32_t>(A_NVER, (uint32_t)d().S * 2); }
    __device__ inline TPtr<double> pend_nest(uint32_t s) const { return a<double>(A_PENDNEST, (uint32_t)d().S * 8) + (uint32_t)s * 8; }
    __device__ inline TPtr<double> nest_stats(uint32_t s, uint32_t j) const { return a<double>(A_NESTST, (uint32_t)d().S * 8) + ((uint32_t)s * 2 + j) * 4; }
    __device__ inline TPtr<uint32_t> edges() const { return a<uint32_t>(A_EDGES, d().NEm > 1 ? d().NEm : 1); }
    __device__ inline TPtr<uint8_t> cover_rows() const { return a<uint8_t>(A_COVER, d().Km); }
    __device__ inline TPtr<double> mcache() const { return a<double>(A_MCACHE, d().cache_entries); }
    __device__ inline TPtr<uint32_t> mctag() const { return a<uint32_t>(A_MCTAG, d().cache_entries); }
    __device__ inline TPtr<uint32_t> mcgen() const { return a<uint32_t>(A_MCGEN, d().cache_entries); }
    __device__ inline SPtrF<uint32_t, LANES> mgen() const { return t.harr<uint32_t>(A_MGEN, v, d().S); }
    __device__ inline TPtr<uint8_t> oth() const { return a<uint8_t>(A_OTH, d().NMm * d().S); }
    __device__ inline TPtr<uint8_t> msubm() const { return a<uint8_t>(A_MSUBM, (d().NMm > 1 ? d().NMm : 1) * d().Hm); }
    __device__ inline TPtr<uint8_t> msubc() const { return a<uint8_t>(A_MSUBC, (d().NMm > 1 ? d().NMm : 1) * d().S); }
    __device__ inline TPtr<uint8_t> msubic() const { return a<uint8_t>(A_MSUBIC, (d().NMm > 1 ? d().NMm : 1) * 2); }
    __device__ inline TPtr<uint32_t> msubsh() const { return a<uint32_t>(A_MSUBSH, d().NMm > 1 ? d().NMm : 1); }
    __device__ inline SPtrF<uint32_t, LANES> pend() const { return t.harr<uint32_t>(A_PEND, v, d().S); }
    __device__ inline SPtrF<uint16_t, LANES> pend_dip() const { return t.harr<uint16_t>(A_PENDDIP, v, 2 * d().S); }
    __device__ inline SPtrF<uint8_t, LANES> pend_valid() const { return t.harr<uint8_t>(A_PENDVALID, v, d().S); }
    __device__ inline double &fnd_saved() const { return t.harr<double>(A_FNDSAVED, v, 1)[0]; }
    __device__ inline double BT_GAS &sparsity() const { return a<double>(A_SPARSITY, 1)[0]; }
    __device__ inline NormalState fnd() const { return NormalState{&fnd_saved(), &sc()[SC_FND_AVAIL]}; }   // `available` may live in LDS
    __device__ inline TPtr<uint8_t> shared_mult() const { return t.arr<uint8_t>(A_SHMULT); }
};

// A tile is worked on by `split` wavefronts (TileDesc::split, chosen per tile by the host): wavefront w owns the 64/split
// consecutive group lanes starting at w * 64/split and runs with only that many active threads; wavefronts of the workgroup
// beyond `split` exit at once.  The memory layout (HBM pool and LDS hot arrays, both interleaved over 64 group lanes) does not
// depend on the split: it only trades SIMD width for more wavefronts that each wait on fewer diverging lanes.
// wavefronts of this launch that work on the tile: the tile's split, or fewer when the workgroup has fewer (gibbs_chain_kernel: ONE wavefront per tile,
// its 64 lanes the tile's 64 groups)
__device__ inline uint32_t tile_split(const TileDesc BT_CAS *d) {
    if (kPacked) return 1u;   // (the wavefronts of a packed workgroup run different tiles)
    const uint32_t waves = blockDim.x >> 6;
    return d->split < waves ? d->split : waves;
}
__device__ inline uint32_t tile_lane(uint32_t split, uint32_t copies) { return (kPacked ? 0u : (threadIdx.x >> 6) * (64u / split)) + ((threadIdx.x & 63u) % (64u / copies)); }
__device__ inline uint32_t tile_part(uint32_t copies) { return (threadIdx.x & 63u) / (64u / copies); }
__device__ inline bool tile_thread_active(uint32_t split, uint32_t copies) {
    if (kPacked) return true;   // (one wavefront per tile, all of its lanes)
    return (threadIdx.x >> 6) < split && (copies > 1u || (threadIdx.x & 63u) < 64u / split);
}
// make the stores of the sibling copies visible before reading what they produced (same wavefront: program order + L1 write-through)
__device__ inline void copies_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ inline Tile make_tile(const Env &e_in) {
    Tile t;
    const TileDesc *tiles = uniform_ptr(e_in.tiles);
    uint8_t *pool = uniform_ptr(e_in.pool);
    const uint32_t *list = uniform_ptr(e_in.tile_list);
    const uint32_t tile = kPacked ? ((const uint32_t BT_CAS *)list)[2u * pack_slot()] : (list ? ((const uint32_t BT_CAS *)list)[blockIdx.x] : blockIdx.x);
    t.lds0 = kPacked ? ((const uint32_t BT_CAS *)list)[2u * pack_slot() + 1u] : 0u;
    t.d = (const TileDesc BT_CAS *)&tiles[tile];
    t.base = (uint8_t BT_GAS *)(pool + t.d->base);
    t.lane = tile_lane(tile_split(t.d), t.d->copies);
    t.plane = t.lane + t.d->pool_lane0;
    t.wsh = t.d->wsh;
    t.part = tile_part(t.d->copies);
    t.copies = t.d->copies;
    t.resident = e_in.resident;   // per lane: lanes of a tile may be at different vertices of their groups
    t.hot = (t.resident != 0xFFFFFFFFu && t.resident != RESIDENT_NEVER && t.d->hot_bytes) ? lds_block() + t.lds0 : nullptr;
    return t;
}
__device__ inline const GParams BT_CAS &env_params(const Env &e) { return *(const GParams BT_CAS *)uniform_ptr(e.P); }

__device__ inline Vx make_vx(const Tile &t, uint32_t v) {
    Vx x;
    x.t = t;
    x.v = v;
    SPtrF<uint32_t, LANES> sc = t.harr<uint32_t>(A_SC, v, SC_COUNT);
    x.H = sc[SC_H];
    x.V = sc[SC_V];
    x.nm = kSingle ? 0u : (uint32_t)sc[SC_NM];
    return x;
}
__device__ inline uint32_t vx_nd(const Vx &c) { return c.t.arr<uint32_t>(A_VDIMS, c.v * 8)[5]; }
__device__ inline uint32_t vx_ne(const Vx &c) { return c.t.arr<uint32_t>(A_VDIMS, c.v * 8)[6]; }

// ---- LDS residency of a vertex's hot arrays ----------------------------------------------------------------------
template <typename T>
__device__ inline void hot_copy(const Tile &t, int arr, uint32_t v, uint32_t len, bool to_lds, uint32_t lds_vertex_off, uint32_t live = 0xFFFFFFFFu) {
    const uint32_t ho = t.d->hoff[arr];
    if (ho == NOHOT) return;
    const uint32_t sh = t.wsh;   // the rows have the tile's width in LDS and in HBM
    T BT_LAS *l = (T BT_LAS *)(bt_lds_raw + (kPacked ? t.lds0 : 0u) + lds_vertex_off + ho) + t.lane;   // explicit LDS pointer: the copies of different arrays can overlap (no aliasing with HBM)
    T BT_GAS *g = (T BT_GAS *)(t.base + t.d->off[arr]) + ((v * len) << sh) + t.plane;
    // eight elements in flight per step (the copy is latency-bound: one wavefront, one memory round trip per step)
    uint32_t i = 0;
    len = live < len ? live : len;   // (the vertex stride above used the full row count)
    constexpr int U = 8;
    if (to_lds) {
        for (; i + U <= len; i += U) {
            T tmp[U];
#pragma unroll
            for (int q = 0; q < U; ++q) tmp[q] = g[(i + q) << sh];
#pragma unroll
            for (int q = 0; q < U; ++q) l[(i + q) << sh] = tmp[q];
        }
        for (; i < len; ++i) l[i << sh] = g[i << sh];
    } else {
        for (; i + U <= len; i += U) {
            T tmp[U];
#pragma unroll
            for (int q = 0; q < U; ++q) tmp[q] = l[(i + q) << sh];
#pragma unroll
            for (int q = 0; q < U; ++q) g[(i + q) << sh] = tmp[q];
        }
        for (; i < len; ++i) g[i << sh] = l[i << sh];
    }
}
__device__ inline uint32_t uniform_tile_hot_bytes(const Env &e) {
    const TileDesc *tiles = uniform_ptr(e.tiles);
    const uint32_t *list = uniform_ptr(e.tile_list);
    const uint32_t tile = kPacked ? ((const uint32_t BT_CAS *)list)[2u * pack_slot()] : (list ? ((const uint32_t BT_CAS *)list)[blockIdx.x] : blockIdx.x);
    return ((const TileDesc BT_CAS *)&tiles[tile])->hot_bytes;
}
// move every hot array of vertex v between HBM and the wavefront's LDS block (lane-wise, coalesced)
__device__ BT_NOINLINE void hot_swap(Env env, uint32_t v, bool to_lds) {
    const uint32_t voff = env.resident == RESIDENT_ALL ? v * uniform_tile_hot_bytes(env) : 0u;
    env.resident = 0xFFFFFFFFu;
    const Tile t = make_tile(env);
    const TileDesc BT_CAS &d = *t.d;
    if (!d.hot_bytes) return;
    hot_copy<uint32_t>(t, A_SC, v, SC_COUNT, to_lds, voff);
    hot_copy<uint16_t>(t, A_DIP, v, 2 * d.S, to_lds, voff);
    hot_copy<uint8_t>(t, A_NESTPL, v, d.S, to_lds, voff);
    hot_copy<uint8_t>(t, A_NESTN, v, d.S, to_lds, voff);
    hot_copy<uint8_t>(t, A_KSCUPD, v, d.S, to_lds, voff);
    hot_copy<uint32_t>(t, A_MGEN, v, d.S, to_lds, voff);
    hot_copy<uint32_t>(t, A_PEND, v, d.S, to_lds, voff);
    hot_copy<uint16_t>(t, A_PENDDIP, v, 2 * d.S, to_lds, voff);
    hot_copy<uint8_t>(t, A_PENDVALID, v, d.S, to_lds, voff);
    hot_copy<uint8_t>(t, A_EVN, v, d.S, to_lds, voff);
    hot_copy<uint32_t>(t, A_RING, v, d.ring_len, to_lds, voff);
    hot_copy<double>(t, A_FNDSAVED, v, 1, to_lds, voff);
    hot_copy<double>(t, A_UCACHE, v, d.cache_entries, to_lds, voff);   // (hot only when the whole table is a few words per lane)
    // per-haplotype arrays: only this lane's H entries are live (rows are Hm apart; the tail is never read)
    const uint32_t H = t.arr<uint32_t>(A_VDIMS, v * 8)[0];
    hot_copy<double>(t, A_FREQ, v, d.Hm, to_lds, voff, H);
    hot_copy<double>(t, A_LOGF, v, d.Hm, to_lds, voff, H);
    hot_copy<uint32_t>(t, A_OBS, v, d.Hm, to_lds, voff, H);
    hot_copy<uint8_t>(t, A_NZ, v, d.Hm, to_lds, voff, H);
    hot_copy<uint32_t>(t, A_UNEXT, v, d.Hm, to_lds, voff, H);
    hot_copy<uint32_t>(t, A_ZHDR, v, 4, to_lds, voff);
    hot_copy<uint32_t>(t, A_ZBKT, v, d.Bcap, to_lds, voff);
    hot_copy<uint32_t>(t, A_PHDR, v, 4, to_lds, voff);
    hot_copy<uint32_t>(t, A_PBKT, v, d.Bcap, to_lds, voff);
    // A_NZLIST and A_CUM are per-call scratch: resident in LDS but never copied
}

// ---- optional per-phase cycle accounting (build with -DBT_PROF; read with bt_diag_prof) ----
#ifdef BT_PROF
static __device__ unsigned long long g_bt_prof[32];   // (one copy per translation unit; bt_diag_prof reads the general kernel's)
#define PROF_DECL unsigned long long _pt = __builtin_readcyclecounter()
#define PROF_DECL2 _pt = __builtin_readcyclecounter()
#define PROF(sec)                                                            \
    do {                                                                     \
        const unsigned long long _now = __builtin_readcyclecounter();        \
        if (threadIdx.x == 0) atomicAdd(&g_bt_prof[sec], _now - _pt);        \
        _pt = _now;                                                          \
    } while (0)
#define PROF_CNT(slot, n)                                                    \
    do {                                                                     \
        if (threadIdx.x == 0) atomicAdd(&g_bt_prof[slot], (unsigned long long)(n)); \
    } while (0)
#else
#define PROF_DECL
#define PROF_DECL2
#define PROF(sec)
#define PROF_CNT(slot, n)
#endif

// ---- Utils::logAddition (Utils.hpp:105-124) ----
__device__ inline double log_addition(double a, double b) {
    if (a < b) return b + bt_log1p(bt_exp(a - b));
    return a + bt_log1p(bt_exp(b - a));
}

// ---- KmerStats (KmerStats.cpp:51-63): ks = {count, fraction, mean, M2} ----
// Values are pulled into registers with independent loads, updated, and written back: one memory round trip per
// KmerStats object instead of a dependent load/store chain per field.
struct KS {
    double c, f, m, m2;
};
template <typename P>
__device__ inline KS ks_load(P p) { return KS{(double)p[0], (double)p[1], (double)p[2], (double)p[3]}; }
template <typename P>
__device__ inline void ks_store(P p, const KS &k) { p[0] = k.c; p[1] = k.f; p[2] = k.m; p[3] = k.m2; }
__device__ inline void ks_add_r(KS &k, double value) {
    const double count = k.c + 1.0;
    k.c = count;
    // !doubleCompare(value, 0): value == 0 <=> equal (Utils.hpp:81-87 with b = 0)
    k.f += ((value == 0.0 ? 0.0 : 1.0) - k.f) / count;
    const double delta = value - k.m;
    k.m += delta / count;
    k.m2 += delta * (value - k.m);
}
template <typename P>
__device__ inline void ks_add(P ks, double value) {
    KS k = ks_load(ks);
    ks_add_r(k, value);
    ks_store(ks, k);
}
__device__ inline double count_log_prob(const GParams BT_CAS &P, uint32_t s, uint8_t mult, uint8_t count) {   // CountDistribution.cpp:255-265
    if (mult == 0) return P.lut_n[s * 256u + count];
    return P.lut_g[((uint32_t)s * 256u + mult) * 256u + count];
}

// ---- VariantClusterHaplotypes multiplicity getters (VariantClusterHaplotypes.cpp:45-108), uchar arithmetic ----
__device__ inline uint8_t dip_mult(const Vx &c, uint32_t k, uint16_t h1, uint16_t h2) {
    uint8_t m = 0;
    if (h1 != NOHAP) m = (uint8_t)(m + c.M(k, h1));
    if (h2 != NOHAP) m = (uint8_t)(m + c.M(k, h2));
    return m;
}
__device__ inline uint8_t unique_mult(const Vx &c, uint32_t k, uint16_t h1, uint16_t h2, uint8_t gender) {
    uint8_t m = dip_mult(c, k, h1, h2);
    if (c.has_counts(k)) m = (uint8_t)(m + c.ic(k, gender));
    return m;
}
__device__ inline uint8_t multi_mult(const Vx &c, const GParams BT_CAS &P, uint32_t k, uint16_t h1, uint16_t h2, uint16_t p1, uint16_t p2, uint32_t s) {
    const uint8_t icm = c.ic(k, P.gender[s]);
    if (c.count(k, s) == 0) return (uint8_t)(dip_mult(c, k, h1, h2) + icm);
    const uint8_t shared = c.shared_mult()[(uint32_t)c.shared_idx(k) * P.S + s];
    return (uint8_t)(shared - dip_mult(c, k, p1, p2) + dip_mult(c, k, h1, h2) + icm);
}

// ---- FrequencyDistribution::reset / SparseFrequencyDistribution::reset (FrequencyDistribution.cpp:49-54,104-115) ----
__device__ inline void freq_reset(const Vx &c) {
    const double f = 1 / (double)c.H;
    const double lf = bt_log(f);
    SPtrF<uint32_t, LANES> obs = c.obs();
    SPtrF<double, LANES> freq = c.freq(), logf = c.logf();
    SPtrF<uint8_t, LANES> nz = c.nz();
    for (uint32_t h = 0; h < c.H; ++h) {
        obs[h] = 0;
        freq[h] = f;
        logf[h] = lf;
        nz[h] = 1;
    }
    if (c.sc()[SC_IS_SPARSE]) {
        uset_clear(c.plus_set());
        Vx::HSet z = c.zero_set();
        uset_clear(z);
        for (uint32_t h = 0; h < c.H; ++h) uset_insert(z, h);
    }
}

// ---- SparsityEstimator::estimateMinimumColumnCover (SparsityEstimator.cpp:41-87), unweighted ----
// returns the cover size; uses `rng` (freshly seeded by the caller), cover_rows, obs (column sums), nzlist
template <class G>
__device__ inline uint32_t sparsity_cover(const Vx &c, G &rng) {
    TPtr<uint8_t> rows = c.cover_rows();
    SPtrF<uint32_t, LANES> obs = c.obs();
    SPtrF<uint16_t, LANES> nzl = c.nzlist();
    const uint32_t Hm = c.d().Hm, H = c.H;
    const Vx::RPtr<uint8_t> M = c.mat(A_M, c.d().Km);
    // The reference recomputes the column sums of the still-uncovered rows in every round; the sums are integers, so keeping
    // them up to date (add every row once, subtract it when it gets covered) gives the same values with one pass per row.
    auto row_apply = [&](uint32_t k, bool add) {
        uint32_t h = 0;
        for (; h + 8 <= H; h += 8) {
            uint32_t m[8], o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = M[k * Hm + h + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = obs[h + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) obs[h + q] = add ? o[q] + m[q] : o[q] - m[q];
        }
        for (; h < H; ++h) {
            const uint32_t m = M[k * Hm + h];
            obs[h] = add ? obs[h] + m : obs[h] - m;
        }
    };
    uint32_t remaining = 0;
    for (uint32_t h = 0; h < H; ++h) obs[h] = 0;
    for (uint32_t k = 0, K = c.K(); k < K; ++k) {
        const uint8_t r = c.has_counts(k) ? 1 : 0;
        rows[k] = r;
        remaining += r;
        if (r) row_apply(k, true);
    }
    uint32_t cover = 0;
    while (remaining > 0) {
        uint32_t best = 0;
        for (uint32_t h = 0; h < H; ++h) {
            const uint32_t o = obs[h];
            best = o > best ? o : best;
        }
        if (best == 0) break;   // the reference asserts max_row_cover > 0
        uint32_t m = 0;
        for (uint32_t h = 0; h < H; ++h)
            if (obs[h] == best) nzl[m++] = (uint16_t)h;
        // DiscreteSampler with outcomes 1,1,...: cum = 1..m; sample = upper_bound(cum, canonical * m) (DiscreteSampler.cpp:61-87)
        const double x = rng_canonical(rng) * (double)m;
        uint32_t pick = 0;
        if (m > 1) {
            while (pick < m && !((double)(pick + 1) > x)) ++pick;
            if (pick >= m) pick = m - 1;
        }
        const uint32_t col = nzl[pick];
        ++cover;
        for (uint32_t k = 0, K = c.K(); k < K; ++k)
            if (rows[k] && M[k * Hm + col] != 0) {
                rows[k] = 0;
                --remaining;
                row_apply(k, false);
            }
    }
    return cover;
}

// ---- VariantClusterGenotyper ctor (VariantClusterGenotyper.cpp:59-106) ----
__device__ BT_NOINLINE void genotyper_construct(Env env, uint32_t vtx, uint32_t prng_seed) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    const TileDesc BT_CAS &d = c.d();
    c.rng_seed(0, prng_seed);
    SPtrF<uint32_t, LANES> sc = c.sc();
    for (uint32_t i = 0; i < SC_STATE_COUNT; ++i) sc[i] = 0;
    // a (re)built genotyper starts from the k-mer index lists in first-seen order (they are shuffled in place per chain)
    {
        TPtr<uint32_t> u0 = c.a<uint32_t>(A_UNIQ0, d.NUm), u = c.uniq(), m0 = c.a<uint32_t>(A_MULTI0, d.NMm), m = c.multi();
        for (uint32_t i = 0, nu = c.nu(); i < nu; ++i) u[i] = u0[i];
        for (uint32_t i = 0; i < c.nm; ++i) m[i] = m0[i];
    }
    SPtrF<uint16_t, LANES> dip = c.dip();
    SPtrF<uint8_t, LANES> upd = c.ksc_upd();
    for (uint32_t s = 0; s < P.S; ++s) {
        dip[2 * s] = NOHAP;
        dip[2 * s + 1] = NOHAP;
        upd[s] = 1;
        c.pend()[s] = 0;
        c.pend_valid()[s] = 0;
        c.nver()[s] = 1;
        c.nver()[P.S + s] = 0xFFFFFFFFu;   // nested info not prepared yet
        c.evn()[s] = 0;
    }
    {
        const uint32_t A = c.t.arr<uint32_t>(A_VDIMS2, c.v * 2)[0];
        TPtr<double> as = c.a<double>(A_ASTATS, (uint32_t)d.S * d.Am * 12);
        for (size_t s = 0; s < P.S; ++s)
            for (size_t i = 0; i < (uint32_t)A * 12; ++i) as[s * d.Am * 12 + i] = 0;
        TPtr<double> k = c.a<double>(A_KSC, (uint32_t)d.S * 2 * d.Vm * 4);
        for (size_t i = 0; i < (uint32_t)P.S * 2 * d.Vm * 4; ++i) k[i] = 0;
        TPtr<uint32_t> dk = c.dip_keys(), df = c.dip_freq();
        for (uint32_t i = 0; i < d.dip_cap; ++i) dk[i] = 0;
        for (size_t i = 0; i < (uint32_t)d.dip_cap * P.S; ++i) df[i] = 0;
        TPtr<uint32_t> sl = c.sclen();
        for (uint32_t i = 0; i < d.scache_n; ++i) sl[i] = 0;
    }
    // SparsityEstimator(prng_seed), then (Sparse)FrequencyDistribution(.., prng_seed) with a fresh generator
    c.rng_seed(1, prng_seed);
    uint32_t cover;
    {
        MtRing m = c.rng(1);
        cover = sparsity_cover(c, m);
    }
    c.rng_seed(1, prng_seed);
    c.fnd_saved() = 0;
    sc[SC_FND_AVAIL] = 0;
    sc[SC_IS_SPARSE] = cover > 0 ? 1u : 0u;   // HaplotypeFrequencyDistribution.cpp:79-89
    if (cover > 0) {
        const double sp = (double)cover / (double)c.H;
        const double cap = 1 - BT_DBL_EPS * 100;
        c.sparsity() = sp < cap ? sp : cap;    // FrequencyDistribution.cpp:98
        uset_init(c.zero_set());
        uset_init(c.plus_set());
    }
    freq_reset(c);
    sc[SC_CONSTRUCTED] = 1;
}

// Slot of a key in the tag-checked direct-mapped tables (cache_mode 1).  The copies of a narrow tile evaluate different candidates at
// the same time, so each copy owns a private 1/copies-th of the table: no two threads ever write the same slot concurrently.  What a
// copy finds in its part only decides whether it recomputes a value, never the value itself.
__device__ inline uint32_t hashed_slot(const Vx &c, uint32_t key) {
    const uint32_t sub = c.d().cache_entries / c.t.copies;   // both powers of two
    return ((key * 2654435761u) & (sub - 1u)) + c.t.part * sub;
}

// all_copies_run: every copy of the group executes this call (sampling operations) and clears its own part; otherwise the calling
// thread clears the whole table
// tiles whose dense tables the host invalidates and refills with the whole GPU between launches (nan_fill_kernel, ucache_prefill_kernel in bt_gibbs.hip)
__device__ inline bool wide_table(const TileDesc BT_CAS &d) { return d.cache_mode == 0 && !d.simple && d.cache_entries > BT_UC_INVALIDATE_MIN && d.hoff[A_UCACHE] == NOHOT; }

// wide_fill: the caller's launch is followed by nan_fill_kernel over this tile's table block (the noise drivers, between two iterations): a
// large table is then already invalidated when the next sweep starts
__device__ inline void cache_clear(const Vx &c, const GParams BT_CAS &P, bool all_copies_run, bool wide_fill = false) {   // VariantClusterGenotyper::clearCache (:131-138)
    const TileDesc BT_CAS &d = c.d();
    if (wide_fill && !all_copies_run && wide_table(d)) {   // (simple tiles rebuild their few entries whole)
        c.sc()[SC_UC_DIRTY] = 0;
    } else if (d.cache_mode == 0) {
        // dense table: rebuilt at the next visit.  As a whole (fill_unique_cache) at a chain start, where the first sweep asks for
        // every entry anyway, and for small tables.  A large table cleared between the sweeps of a chain (clearGenotyperCache of the
        // noise drivers) is only invalidated: the sweeps that follow ask for the pairs of the few non-zero haplotypes, which are
        // then computed on demand like the reference does
        c.sc()[SC_UC_DIRTY] = (!all_copies_run && d.cache_entries > BT_UC_INVALIDATE_MIN) ? 2u : 1u;
    } else if (d.cache_mode == 1) {
        TPtr<uint32_t> tg = c.uctag();
        const uint32_t sub = all_copies_run ? d.cache_entries / c.t.copies : d.cache_entries;
        for (uint32_t i = all_copies_run ? c.t.part * sub : 0u, e = i + sub; i < e; ++i) tg[i] = 0;
    }    // ... and the multicluster sums (the reference clears both maps): their entries are stamped with the sample's generation, so a bump
    // invalidates them.  (Without it a sum cached under the previous noise table survived clearGenotyperCache in the noise drivers.)
    if (!kSingle && d.NMm) {
        SPtrF<uint32_t, LANES> mg = c.mgen();
        for (uint32_t s = 0; s < P.S; ++s) mg[s] += 1;
    }
}

// ---- VariantClusterHaplotypes::sampleKmerSubset (+ isMaxHaplotypeVariantKmer) (VariantClusterHaplotypes.cpp:110-177) ----
__device__ inline bool is_max_hv_kmer(const Vx &c, uint32_t k, uint32_t maxk) {
    bool is_max = true;
    TPtr<uint32_t> hv = c.hvcount();
    const uint32_t Vm = c.d().Vm, HWm = c.d().HWm, HW = (c.H + 31) / 32;
    TPtr<uint32_t> bits = c.a<uint32_t>(A_KVBITS, (uint32_t)c.d().NNZm * HWm);
    for (uint32_t e = c.kv_off(k), e1 = c.kv_off(k + 1); e < e1; ++e) {
        const uint32_t var = c.kv_var(e);
        for (uint32_t w = 0; w < HW; ++w) {
            uint32_t word = bits[e * HWm + w];
            // the counters of distinct haplotypes are independent: four are read before any is written back
            while (word) {
                uint32_t h[4], cnt[4], n = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h[q] = 0;
                    if (word) {
                        h[q] = w * 32 + (uint32_t)__builtin_ctz(word);
                        word &= word - 1;
                        n = q + 1;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) cnt[q] = hv[h[q] * Vm + var];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((uint32_t)q < n && cnt[q] < maxk) {
                        hv[h[q] * Vm + var] = cnt[q] + 1;
                        is_max = false;
                    }
            }
        }
    }
    return is_max;
}
__device__ inline void sample_kmer_subset(const Vx &c, const GParams BT_CAS &P) {
    const uint32_t Vm = c.d().Vm;
    TPtr<uint32_t> hv = c.hvcount();
    for (uint32_t h = 0; h < c.H; ++h)
        for (uint32_t v = 0; v < c.V; ++v) hv[(uint32_t)h * Vm + v] = 0;
    uint32_t nsu = 0, nsm = 0;
    MtRing rng = c.rng(0);
    TPtr<uint32_t> uniq = c.uniq(), usub = c.usub(), multi = c.multi(), msub = c.msub();
    PROF_DECL;
    // The generator is consumed exactly as the reference does (shuffle, one Bernoulli draw per k-mer; unique list, then the
    // multicluster list), but the isMaxHaplotypeVariantKmer filter — which draws nothing — runs afterwards over the selected
    // k-mers only, in the same order: every lane then works on ITS j-th selected k-mer instead of the wave stepping through
    // all k-mers with the ~10 % of lanes that selected that one.
    const uint32_t nu = c.nu();
    rng_shuffle_u32(rng, uniq, nu);
    PROF(8);
    uint32_t nu_sel = 0, nm_sel = 0;
    for (uint32_t i = 0; i < nu; ++i)
        if (rng_bernoulli(rng, P.rate)) usub[nu_sel++] = uniq[i];
    rng_shuffle_u32(rng, multi, c.nm);
    for (uint32_t i = 0; i < c.nm; ++i)
        if (rng_bernoulli(rng, P.rate)) msub[nm_sel++] = multi[i];
    for (uint32_t i = 0; i < nu_sel; ++i) {
        const uint32_t k = usub[i];
        if (!is_max_hv_kmer(c, k, P.max_hvk)) usub[nsu++] = k;
    }
    for (uint32_t i = 0; i < nm_sel; ++i) {
        const uint32_t k = msub[i];
        if (!is_max_hv_kmer(c, k, P.max_hvk)) msub[nsm++] = k;
    }
    PROF(9);
    mt_close(rng);
    {
        // compact, subset-ordered copies of what calcDiplotypeLogProb reads per unique subset k-mer: the per-candidate sum then
        // walks plain arrays (index i, no k-mer indirection), so its loads are independent and coalesce across the wavefront
        const uint32_t Hm = c.d().Hm;
        const Vx::RPtr<uint8_t> sm = c.subm();
        TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
        TPtr<uint32_t> so = c.skv_off(), sb = c.skv_bits();
        TPtr<uint16_t> sv = c.skv_var();
        const uint32_t HWm = c.d().HWm, HW = (c.H + 31) / 32;
        uint32_t ne = 0;
        so[0] = 0;
        for (uint32_t i = 0; i < nsu; ++i) {   // CSR offsets first (sequential, every copy of the group computes the same values)
            const uint32_t k = usub[i];
            ne += c.kv_off(k + 1) - c.kv_off(k);
            so[i + 1] = ne;
        }
        for (uint32_t i = c.t.part; i < nsu; i += c.t.copies) {   // k-mer i is copied by copy i % copies (lockstep: iteration j handles j*copies + part)
            const uint32_t k = usub[i];
            uint32_t ne = so[i];
            for (uint32_t e = c.kv_off(k), e1 = c.kv_off(k + 1); e < e1; ++e, ++ne) {
                sv[ne] = c.kv_var(e);
                for (uint32_t w = 0; w < HW; ++w) sb[ne * HWm + w] = c.a<uint32_t>(A_KVBITS, c.d().NNZm * HWm)[e * HWm + w];
            }
            const bool hc = c.has_counts(k) != 0;
            {
                // row copy, eight multiplicities in flight (loads and stores may alias as far as the compiler knows)
                const Vx::RPtr<uint8_t> Mrow = c.mat(A_M, c.d().Km) + k * Hm;
                uint32_t h = 0;
                for (; h + 8 <= c.H; h += 8) {
                    uint8_t t8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t8[q] = Mrow[h + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) sm[i * Hm + h + q] = t8[q];
                }
                for (; h < c.H; ++h) sm[i * Hm + h] = Mrow[h];
            }
            for (uint32_t ss = 0; ss < P.S; ++ss) scn[i * P.S + ss] = hc ? c.count(k, ss) : (uint8_t)0;
            sic[2 * i] = hc ? c.ic(k, 0) : (uint8_t)0;
            sic[2 * i + 1] = hc ? c.ic(k, 1) : (uint8_t)0;
        }
    }
    {
        // the same for the multicluster subset k-mers (multi_refresh / multi_log_prob / updateMulticlusterDiplotypeLogProb inputs)
        const uint32_t Hm = c.d().Hm;
        TPtr<uint8_t> mm = c.msubm(), mcn = c.msubc(), mic = c.msubic();
        TPtr<uint32_t> msh = c.msubsh();
        for (uint32_t i = c.t.part; i < nsm; i += c.t.copies) {
            const uint32_t k = msub[i];
            for (uint32_t h = 0; h < c.H; ++h) mm[i * Hm + h] = c.M(k, h);
            for (uint32_t ss = 0; ss < P.S; ++ss) mcn[i * P.S + ss] = c.count(k, ss);
            mic[2 * i] = c.ic(k, 0);
            mic[2 * i + 1] = c.ic(k, 1);
            msh[i] = (uint32_t)c.shared_idx(k);
        }
    }
    if (c.t.copies > 1u) copies_sync();
    PROF(10);
    SPtrF<uint32_t, LANES> sc = c.sc();
    sc[SC_NSUB_U] = nsu;
    sc[SC_NSUB_M] = nsm;
    TPtr<uint8_t> smm = c.smm(), oth = c.oth();
    for (uint32_t i = 0; i < nsm * P.S; ++i) {
        smm[i] = 0;
        oth[i] = 0;
    }
    for (uint32_t s = 0; s < P.S; ++s) c.mgen()[s] += 1;   // a new k-mer subset invalidates the multicluster cache
    SPtrF<uint8_t, LANES> upd = c.ksc_upd();
    for (uint32_t s = 0; s < P.S; ++s) {
        upd[s] = 1;
        for (uint32_t e = 0; e < KSC_WAYS; ++e) c.ksc_key(s)[e] = KSC_NOKEY;   // ... and the kept k-mer-stats caches (collect_sample_body)
        c.ksc_key(s)[KSC_WAYS] = 0;
    }
}

// ---- VariantClusterGenotyper::reset (VariantClusterGenotyper.cpp:113-129) ----
__device__ BT_NOINLINE void genotyper_reset(Env env, uint32_t vtx) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    c.sc()[SC_USE_MULTI] = 0;
    sample_kmer_subset(c, P);
    cache_clear(c, P, true);
    freq_reset(c);   // HaplotypeFrequencyDistribution::reset (counts are 0 here, as the reference asserts)
}

__device__ inline uint32_t dip_index(const Vx &c, uint16_t h1, uint16_t h2) {
    if (h2 == NOHAP) return c.H * (c.H + 1) / 2 + h1;
    return (uint32_t)h1 * c.H - ((uint32_t)h1 * ((uint32_t)h1 - 1u)) / 2u + ((uint32_t)h2 - (uint32_t)h1);
}

// unique part of calcDiplotypeLogProb with its per-(sample, diplotype) cache (VariantClusterGenotyper.cpp:619-643).
// The cached value is a pure function of (sample, diplotype, k-mer subset), so a dense table, a direct-mapped table
// or no table at all give bit-identical sums (same summation order).
__device__ inline double unique_log_prob(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint16_t h1, uint16_t h2, uint32_t nsub_u) {
    const TileDesc BT_CAS &d = c.d();
    const uint32_t idx = dip_index(c, h1, h2);
    uint32_t slot = 0;
    const Vx::UCPtr uc = c.ucache();
    if (d.cache_mode == 0) {
        const double v = uc[(uint32_t)s * d.Dcm + idx];
        if (v == v) return v;
    } else if (d.cache_mode == 1) {
        const uint32_t key = s * d.Dcm + idx + 1u;
        slot = hashed_slot(c, key);
        if (c.uctag()[slot] == key) return uc[slot];
    }
    double acc = 0;
    const uint8_t gender = P.gender[s];
    {
        const uint32_t Hm = d.Hm, S = P.S;
        const Vx::RPtr<uint8_t> sm = c.subm();
        TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
        const bool two = h2 != NOHAP;
        uint32_t i = 0;
        // eight k-mers per step: all loads of a step are issued before the first table lookup; the sum itself stays in subset order
        for (; i + 8 <= nsub_u; i += 8) {
            uint8_t m[8], cn[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint8_t mm = sm[(i + q) * Hm + h1];
                if (two) mm = (uint8_t)(mm + sm[(i + q) * Hm + h2]);
                m[q] = (uint8_t)(mm + sic[2 * (i + q) + gender]);
                cn[q] = scn[(i + q) * S + s];
            }
            double lp[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) lp[q] = count_log_prob(P, s, m[q], cn[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += lp[q];
        }
        for (; i < nsub_u; ++i) {
            uint8_t mm = sm[i * Hm + h1];
            if (two) mm = (uint8_t)(mm + sm[i * Hm + h2]);
            const uint8_t m = (uint8_t)(mm + sic[2 * i + gender]);
            acc += count_log_prob(P, s, m, scn[i * S + s]);
        }
    }
    if (d.cache_mode == 0) uc[(uint32_t)s * d.Dcm + idx] = acc;
    else if (d.cache_mode == 1) {
        c.uctag()[slot] = s * d.Dcm + idx + 1u;
        uc[slot] = acc;
    }
    return acc;
}

// The misses of a block of candidates evaluated TOGETHER (the noise drivers clear the caches every iteration, so every sweep asks for the sums
// of its candidates again): one pass over the k-mer subset, four k-mers per step — the shared operands (intercluster multiplicity, count) are
// read once per k-mer, the candidates' multiplicity rows and table lookups are independent loads — instead of one pass per candidate.
// Every sum still runs in subset order, so the values are those of unique_log_prob; they are stored in the table the same way.
__device__ inline void unique_log_prob_block(const Vx &c, const GParams BT_CAS &P, uint32_t s, const uint16_t (&ha)[EVB], const uint16_t (&hb)[EVB], const bool (&need)[EVB],
                                             uint32_t nsub_u, double (&out)[EVB], bool store = true) {
    const TileDesc BT_CAS &d = c.d();
    const uint32_t Hm = d.Hm, S = P.S;
    const uint8_t gender = P.gender[s];
    const Vx::RPtr<uint8_t> sm = c.subm();
    TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
    uint16_t a[EVB], b[EVB];
#pragma unroll
    for (uint32_t q = 0; q < EVB; ++q) {   // safe addresses for the slots that need nothing
        a[q] = need[q] ? ha[q] : (uint16_t)0;
        b[q] = need[q] ? hb[q] : NOHAP;
    }
    double acc[EVB];
#pragma unroll
    for (uint32_t q = 0; q < EVB; ++q) acc[q] = 0;
    uint32_t i = 0;
    for (; i + 4 <= nsub_u; i += 4) {
        uint8_t m[4][EVB], cn[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint8_t icn = sic[2 * (i + r) + gender];
            cn[r] = scn[(i + r) * S + s];
#pragma unroll
            for (uint32_t q = 0; q < EVB; ++q) {
                uint8_t mm = sm[(i + r) * Hm + a[q]];
                if (b[q] != NOHAP) mm = (uint8_t)(mm + sm[(i + r) * Hm + b[q]]);
                m[r][q] = (uint8_t)(mm + icn);
            }
        }
        double lp[4][EVB];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
            for (uint32_t q = 0; q < EVB; ++q) lp[r][q] = count_log_prob(P, s, m[r][q], cn[r]);
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
            for (uint32_t q = 0; q < EVB; ++q) acc[q] += lp[r][q];
    }
    for (; i < nsub_u; ++i) {
        const uint8_t icn = sic[2 * i + gender], cn = scn[i * S + s];
#pragma unroll
        for (uint32_t q = 0; q < EVB; ++q) {
            uint8_t mm = sm[i * Hm + a[q]];
            if (b[q] != NOHAP) mm = (uint8_t)(mm + sm[i * Hm + b[q]]);
            acc[q] += count_log_prob(P, s, (uint8_t)(mm + icn), cn);
        }
    }
    const Vx::UCPtr uc = c.ucache();
#pragma unroll
    for (uint32_t q = 0; q < EVB; ++q) {
        out[q] = acc[q];
        if (!need[q] || !store) continue;
        const uint32_t idx = dip_index(c, ha[q], hb[q]);
        if (d.cache_mode == 0) uc[(uint32_t)s * d.Dcm + idx] = acc[q];
        else if (d.cache_mode == 1) {
            const uint32_t key = s * d.Dcm + idx + 1u, slot = hashed_slot(c, key);
            c.uctag()[slot] = key;
            uc[slot] = acc[q];
        }
    }
}

// Dense-table tiles (cache_mode 0) do not fill the unique-part cache on demand: a miss would stall the 63 other lanes of the
// wave for a whole pass over the k-mer subset, and over a chain nearly every (sample, diplotype) entry gets requested by some
// sweep anyway.  Instead the whole table is computed once per cache epoch (chain start / clearCache) with all lanes busy.
// Every entry is the same sum in the same (subset) order as the on-demand evaluation, so values are bit-identical.
// Four candidates sharing the first haplotype are evaluated per pass: 11 loads per k-mer for 4 sums instead of 20.
__device__ BT_NOINLINE void fill_unique_cache(Env env, uint32_t vtx) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    const TileDesc BT_CAS &d = c.d();
    const uint32_t nsub = c.sc()[SC_NSUB_U];
    const uint32_t Hm = d.Hm, S = P.S, H = c.H;
    const Vx::RPtr<uint8_t> sm = c.subm();
        TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
    const Vx::UCPtr uc = c.ucache();
    for (uint32_t s = 0; s < S; ++s) {
        const uint8_t gender = P.gender[s];
        const uint32_t row = s * d.Dcm;
        // a == H is the haploid row: candidates (b, none)
        // blocks (a, b0..b0+3) in row order; entries are independent sums, so the copies of this group take every copies-th block:
        // copy `part` walks ITS OWN block sequence (the copies stay in lockstep: iteration j handles blocks j*copies + part)
        uint32_t a = 0, b0 = 0;
        auto next_block = [&]() {
            b0 += 4;
            if (b0 >= H) {
                ++a;
                b0 = a == H ? 0u : a;
            }
        };
        for (uint32_t q = 0; q < c.t.part && a <= H; ++q) next_block();
        while (a <= H) {   // (a, b0) advance in next_block
            const bool hap = a == H;
            {
                uint32_t hb[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) hb[q] = b0 + q < H ? b0 + q : H - 1;
                double acc[4] = {0, 0, 0, 0};
                uint32_t i = 0;
                // four k-mers x four candidates per step: all 28 operand loads, then the 16 table lookups, then the additions
                // (each candidate's sum still runs in subset order)
                for (; i + 4 <= nsub; i += 4) {
                    uint8_t m[4][4], cn[4];
#pragma unroll
                    for (uint32_t r = 0; r < 4; ++r) {
                        const uint8_t ma = hap ? (uint8_t)0 : (uint8_t)sm[(i + r) * Hm + a];
                        const uint8_t icn = sic[2 * (i + r) + gender];
                        cn[r] = scn[(i + r) * S + s];
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) m[r][q] = (uint8_t)((uint8_t)(ma + (uint8_t)sm[(i + r) * Hm + hb[q]]) + icn);
                    }
                    double lp[4][4];
#pragma unroll
                    for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) lp[r][q] = count_log_prob(P, s, m[r][q], cn[r]);
#pragma unroll
                    for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q) acc[q] += lp[r][q];
                }
                for (; i < nsub; ++i) {
                    const uint8_t ma = hap ? (uint8_t)0 : (uint8_t)sm[i * Hm + a];
                    const uint8_t icn = sic[2 * i + gender], cn = scn[i * S + s];
                    uint8_t m[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) m[q] = (uint8_t)((uint8_t)(ma + (uint8_t)sm[i * Hm + hb[q]]) + icn);
                    double lp[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) lp[q] = count_log_prob(P, s, m[q], cn);
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) acc[q] += lp[q];
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q)
                    if (b0 + q < H) uc[row + (hap ? H * (H + 1) / 2 + b0 + q : dip_index(c, (uint16_t)a, (uint16_t)(b0 + q)))] = acc[q];
            }
            for (uint32_t q = 0; q < c.t.copies && a <= H; ++q) next_block();
        }
    }
    if (c.t.copies > 1u) copies_sync();
}

// multicluster part (VariantClusterGenotyper.cpp:647-661).  For a candidate d the k-mer multiplicity is
// (shared - M[current diplotype]) + M[d] + intercluster, i.e. "what the OTHER clusters currently contribute" + own candidate.
// The reference caches the sum per (sample, diplotype) and patches it when another cluster moved (:569-595); here the cache
// is generation-stamped per sample: multi_refresh() compares the other clusters' contribution of every subset k-mer with a
// snapshot and bumps the sample's generation when anything moved, which invalidates that sample's entries in O(1).  A hit
// returns exactly the direct sum a miss would compute.
__device__ inline uint8_t msub_dip_mult(TPtr<uint8_t> mm, uint32_t Hm, uint32_t i, uint16_t h1, uint16_t h2) {
    uint8_t m = 0;
    if (h1 != NOHAP) m = (uint8_t)(m + mm[i * Hm + h1]);
    if (h2 != NOHAP) m = (uint8_t)(m + mm[i * Hm + h2]);
    return m;
}
__device__ inline void multi_refresh(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint16_t p1, uint16_t p2, uint32_t nsub_m) {
    TPtr<uint8_t> oth = c.oth(), shm = c.shared_mult(), mm = c.msubm(), mcn = c.msubc();
    TPtr<uint32_t> msh = c.msubsh();
    const uint32_t Hm = c.d().Hm;
    bool moved = false;
    for (uint32_t sub = 0; sub < nsub_m; ++sub) {
        uint8_t o = 0;
        if (mcn[sub * P.S + s] != 0) o = (uint8_t)(shm[(uint32_t)msh[sub] * P.S + s] - msub_dip_mult(mm, Hm, sub, p1, p2));
        if (o != oth[sub * P.S + s]) {
            oth[sub * P.S + s] = o;
            moved = true;
        }
    }
    if (moved) c.mgen()[s] += 1;
}
// For a candidate the k-mer multiplicity getMulticlusterKmerMultiplicity returns (VariantClusterHaplotypes.cpp:82-108) is exactly
// oth + M[h1] + M[h2] + intercluster in uchar arithmetic, with oth as refreshed by multi_refresh for the current generation.
// The misses of a block of 8 candidates, evaluated together: per subset k-mer the shared operands (oth, intercluster, count) are
// read once and the candidates' multiplicity rows and table lookups are independent loads.  Sums stay in subset order.
__device__ inline void multi_log_prob_block(const Vx &c, const GParams BT_CAS &P, uint32_t s, const uint16_t (&ha)[EVB], const uint16_t (&hb)[EVB], const bool (&need)[EVB],
                                            uint32_t nsub_m, double (&out)[EVB]) {
    TPtr<uint8_t> oth = c.oth(), mm = c.msubm(), mcn = c.msubc(), mic = c.msubic();
    const uint32_t Hm = c.d().Hm;
    const uint8_t gender = P.gender[s];
    double acc[EVB];
#pragma unroll
    for (int q = 0; q < (int)EVB; ++q) acc[q] = 0;
    for (uint32_t i = 0; i < nsub_m; ++i) {
        const uint8_t o = (uint8_t)(oth[i * P.S + s]), icn = mic[2 * i + gender], cn = mcn[i * P.S + s];
        uint8_t m[EVB];
#pragma unroll
        for (int q = 0; q < (int)EVB; ++q) {
            const uint16_t a = need[q] ? ha[q] : (uint16_t)0, b = need[q] ? hb[q] : NOHAP;   // safe addresses for unused slots
            m[q] = (uint8_t)((uint8_t)(o + msub_dip_mult(mm, Hm, i, a, b)) + icn);
        }
        double lp[EVB];
#pragma unroll
        for (int q = 0; q < (int)EVB; ++q) lp[q] = count_log_prob(P, s, m[q], cn);
#pragma unroll
        for (int q = 0; q < (int)EVB; ++q) acc[q] += lp[q];
    }
#pragma unroll
    for (int q = 0; q < (int)EVB; ++q) out[q] = acc[q];
}

// ---- HaplotypeFrequencyDistribution::incrementCount (HaplotypeFrequencyDistribution.cpp:113-125) ----
__device__ inline void hfd_increment(const Vx &c, uint16_t h, bool is_sparse, uint32_t &hap_count) {
    if (h == NOHAP) return;   // num_missing_count is only read by an assert in the reference
    hap_count += 1;
    SPtrF<uint32_t, LANES> obs = c.obs();
    const uint32_t o = obs[h];
    if (is_sparse && o == 0) {   // SparseFrequencyDistribution::incrementObservationCount (:198-207)
        // the reference inserts into plus, then erases from zero; the sets share their `next` words here, so leave
        // the zero list first (the resulting containers are identical)
        uset_erase(c.zero_set(), h);
        uset_insert(c.plus_set(), h);
    }
    obs[h] = o + 1;
}

// ---- diplotype_sampling_frequencies (VariantClusterGenotyper.cpp:692-696) as an open-addressing table ----
__device__ inline void dip_table_add(const Vx &c, const GParams BT_CAS &P, uint16_t h1, uint16_t h2, uint32_t s, uint32_t times) {
    const uint32_t key = ((uint32_t)h1 | ((uint32_t)h2 << 16));
    // stored tag: key + 1 (0 = empty slot); the null diplotype (NOHAP, NOHAP) would wrap to 0 and is stored as 0xFFFFFFFF,
    // which no other key + 1 can equal because haplotype indices are < 0xFFFE
    const uint32_t want = key == 0xFFFFFFFFu ? 0xFFFFFFFFu : key + 1u;
    const uint32_t cap = c.d().dip_cap, mask = cap - 1u;
    TPtr<uint32_t> keys = c.dip_keys(), freq = c.dip_freq();
    uint32_t slot = (key * 2654435761u) & mask;
    for (uint32_t probes = 0; probes < cap; ++probes) {
        const uint32_t tag = keys[slot];
        if (tag == 0) {
            keys[slot] = want;
            c.sc()[SC_DIP_ENTRIES] += 1;
            freq[(uint32_t)slot * P.S + s] += times;
            return;
        }
        if (tag == want) {
            freq[(uint32_t)slot * P.S + s] += times;
            return;
        }
        slot = (slot + 1u) & mask;
    }
    c.sc()[SC_DIP_OVERFLOW] = 1;
}

// ---- VariantClusterHaplotypes::updateMulticlusterKmerMultiplicities (VariantClusterHaplotypes.cpp:197-233) ----
__device__ inline void update_multicluster_multiplicities(const Vx &c, const GParams BT_CAS &P, uint16_t h1, uint16_t h2, uint16_t p1, uint16_t p2, uint32_t s, uint32_t nsub_m) {
    if (h1 != p1 || h2 != p2) c.ksc_upd()[s] = 1;
    if (c.nm == 0) return;
    TPtr<uint8_t> shm = c.shared_mult();
    if (h1 != p1 || h2 != p2) {
        TPtr<uint32_t> multi = c.multi();
        for (uint32_t i = 0; i < c.nm; ++i) {
            const uint32_t k = multi[i];
            const uint8_t cur = dip_mult(c, k, h1, h2), pre = dip_mult(c, k, p1, p2);
            if (cur != pre) {
                const size_t at = (uint32_t)c.shared_idx(k) * P.S + s;
                uint8_t m = shm[at];
                m = (uint8_t)(m - pre);
                m = (uint8_t)(m + cur);
                shm[at] = m;
            }
        }
    }
    TPtr<uint8_t> smm = c.smm(), mm = c.msubm(), mcn = c.msubc();
    TPtr<uint32_t> msh = c.msubsh();
    const uint32_t Hm = c.d().Hm;
    bool changed = false;
    for (uint32_t s0 = 0; s0 < nsub_m; s0 += 8) {   // eight subset k-mers per step: all loads of a step first (the stores to smm would serialise them)
        uint32_t shi[8];
        uint8_t sh[8], dm[8], cn[8], old[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) shi[q] = s0 + q < nsub_m ? (uint32_t)msh[s0 + q] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) {
            const uint32_t sub = s0 + q < nsub_m ? s0 + q : s0;
            sh[q] = shm[shi[q] * P.S + s];
            dm[q] = msub_dip_mult(mm, Hm, sub, h1, h2);
            cn[q] = mcn[sub * P.S + s];
            old[q] = smm[sub * P.S + s];
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q)
            if (s0 + q < nsub_m) {
                changed = changed || (dm[q] > 0 && cn[q] > 0 && sh[q] != old[q]);
                if (sh[q] != old[q]) smm[(s0 + q) * P.S + s] = sh[q];
            }
    }
    if (changed) c.ksc_upd()[s] = 1;
}

__device__ inline bool is_missing(const Vx &c, uint32_t var, uint32_t a) { return c.var_dep(var) && a == (uint32_t)c.var_na(var) - 1u; }

// r repetitions of KmerStats::addValue(value) on a register copy.  Once an accumulator has converged onto the value
// (delta == 0 and the non-zero fraction saturated) every further addValue only increments the count, so the remaining
// repetitions collapse into one exact addition.
__device__ inline void ks_add_rep(KS &k, double value, uint32_t r) {
    const double nzv = value == 0.0 ? 0.0 : 1.0;
    for (uint32_t i = 0; i < r; ++i) {
        if (k.m == value && k.f == nzv && k.c > 0.0) {
            k.c += (double)(r - i);
            return;
        }
        ks_add_r(k, value);
    }
}

// Materialise `r` identical collected sweeps of sample s that drew diplotype (h1, h2) while its k-mer-stats cache stayed
// unchanged — diplotype_sampling_frequencies += r and, per allele cell, the reference's exact sequence of addKmerStats calls
// replayed r times — followed by the `nn` addNestedHaplotypeKmerStats contributions of this sweep (:360-372; nn > 0 only with r = 1).
//
// An allele cell holds three independent KmerStats (count, fraction, mean), and the cells of different variants are disjoint, so the
// work is 3 V independent Welford chains.  They are items (variant, statistic) dealt to the copies of the group (a 64-group tile: all
// to the one lane); every chain sees its values in the reference's order: haplotype 1's source, haplotype 2's (alternating when both
// fall into the same cell), then the nested sources.
// r repetitions of the value list L[0..n) on one KmerStats (n <= 4)
__device__ inline void ks_list_rep(KS &k, const double (&L)[4], uint32_t n, uint32_t r) {
    if (n == 0) return;
    bool same = true;
    for (uint32_t i = 1; i < 4; ++i) same = same && (i >= n || L[i] == L[0]);
    if (same) {
        ks_add_rep(k, L[0], n * r);
        return;
    }
    for (uint32_t rep = 0; rep < r; ++rep)
        for (uint32_t i = 0; i < 4; ++i)
            if (i < n) ks_add_r(k, L[i]);
}
// The nested sources come from the sample's pending copy (A_PENDNEST: what the sweeps being replayed saw), nn of them.
__device__ inline void replay_collected(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint16_t h1, uint16_t h2, uint32_t r, uint32_t nn) {
    const uint32_t V = c.V;
    const bool one = h1 != NOHAP, two = one && h2 != NOHAP;
    // nested sources (the same for every variant)
    double ncnt[2] = {0, 0}, nf[2] = {0, 0}, nm[2] = {0, 0};
    {
        TPtr<double> q = c.pend_nest(s);
        for (uint32_t j = 0; j < 2u; ++j)
            if (j < nn) {
                ncnt[j] = q[4 * j];
                nf[j] = q[4 * j + 1];
                nm[j] = q[4 * j + 2];
            }
    }
    // Items in batches of three per copy: all descriptors first, then all sources and cells, then the chains, then the stores — a batch
    // costs three dependent memory round trips instead of that many per item.
    constexpr uint32_t NB = 3;
    for (uint32_t base = c.t.part; base < 3 * V; base += NB * c.t.copies) {
        uint32_t var[NB], st[NB], d1[NB], d2[NB], cn[NB];
        bool live[NB];
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) {
            const uint32_t item = base + b * c.t.copies;
            live[b] = item < 3 * V;
            var[b] = live[b] ? item / 3u : 0u;
            st[b] = live[b] ? item - var[b] * 3u : 0u;
            d1[b] = one ? c.hap_cell(h1, var[b]) : 0xFFFFu;
            d2[b] = two ? c.hap_cell(h2, var[b]) : 0xFFFFu;
            cn[b] = nn ? c.allele_base(var[b]) + (uint32_t)c.var_na(var[b]) - 1u : 0u;   // addNestedHaplotypeKmerStats' cell: the variant's last allele
        }
        double v1[NB], v2[NB];
        bool en1[NB], en2[NB], use2[NB], use3[NB];
        KS k1[NB], k2[NB], k3[NB];   // the (at most three) cells of the item: haplotype 1's, haplotype 2's, the nested one
        uint32_t c1[NB], c2[NB];
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) {
            const bool ok1 = live[b] && (d1[b] & 0xFFFFu) != 0xFFFFu, ok2 = live[b] && (d2[b] & 0xFFFFu) != 0xFFFFu;
            c1[b] = d1[b] >> 16;
            c2[b] = d2[b] >> 16;
            v1[b] = v2[b] = 0;
            en1[b] = en2[b] = false;
            if (ok1) {
                TPtr<double> q = c.ksc(s, 0, d1[b] & 0xFFFFu);
                const double cnt = q[0], val = q[st[b]];
                v1[b] = val;
                en1[b] = st[b] == 0 || cnt != 0.0;
            }
            if (ok2) {
                TPtr<double> q = c.ksc(s, 1, d2[b] & 0xFFFFu);
                const double cnt = q[0], val = q[st[b]];
                v2[b] = val;
                en2[b] = st[b] == 0 || cnt != 0.0;
            }
            bool nested_any = false;
            for (uint32_t j = 0; j < 2u; ++j) nested_any = nested_any || (j < nn && (st[b] == 0 || ncnt[j] != 0.0));
            use2[b] = en2[b] && !(en1[b] && c2[b] == c1[b]);
            use3[b] = live[b] && nested_any && !(en1[b] && cn[b] == c1[b]) && !(use2[b] && cn[b] == c2[b]);
            k1[b] = k2[b] = k3[b] = KS{0, 0, 0, 0};
            if (en1[b]) k1[b] = ks_load(c.astats_cell(s, c1[b]) + 4u * st[b]);
            if (use2[b]) k2[b] = ks_load(c.astats_cell(s, c2[b]) + 4u * st[b]);
            if (use3[b]) k3[b] = ks_load(c.astats_cell(s, cn[b]) + 4u * st[b]);
        }
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) {
            if (!live[b]) continue;
            // per cell the values it receives in one sweep, in the reference's order: haplotype 1's source, haplotype 2's, the nested ones
            double L1[4] = {0, 0, 0, 0}, L2[4] = {0, 0, 0, 0}, L3[4] = {0, 0, 0, 0};
            uint32_t n1 = 0, n2 = 0, n3 = 0;
            if (en1[b]) L1[n1++] = v1[b];
            if (en2[b]) {
                if (use2[b]) L2[n2++] = v2[b];
                else L1[n1++] = v2[b];
            }
            for (uint32_t j = 0; j < 2u; ++j) {
                if (!(j < nn && (st[b] == 0 || ncnt[j] != 0.0))) continue;
                const double val = st[b] == 0 ? ncnt[j] : (st[b] == 1 ? nf[j] : nm[j]);
                if (en1[b] && cn[b] == c1[b]) L1[n1++] = val;
                else if (use2[b] && cn[b] == c2[b]) L2[n2++] = val;
                else L3[n3++] = val;
            }
            ks_list_rep(k1[b], L1, n1, r);
            ks_list_rep(k2[b], L2, n2, r);
            ks_list_rep(k3[b], L3, n3, r);
        }
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) {
            if (!live[b]) continue;
            if (en1[b]) ks_store(c.astats_cell(s, c1[b]) + 4u * st[b], k1[b]);
            if (use2[b]) ks_store(c.astats_cell(s, c2[b]) + 4u * st[b], k2[b]);
            if (use3[b]) ks_store(c.astats_cell(s, cn[b]) + 4u * st[b], k3[b]);
        }
    }
    dip_table_add(c, P, h1, h2, s, r);
    if (c.t.copies > 1u) copies_sync();
}

__device__ inline void flush_sample(const Vx &c, const GParams BT_CAS &P, uint32_t s) {
    const uint32_t r = c.pend()[s];
    if (r == 0) return;
    c.pend()[s] = 0;
    replay_collected(c, P, s, c.pend_dip()[2 * s], c.pend_dip()[2 * s + 1], r, (uint32_t)(double)c.pend_nest(s)[3]);
}

// materialise everything still pending (end of a launch: results may be read next)


// updateKmerStatsCache for one sample (VariantClusterHaplotypes.cpp:247-277): kmer_stats_cache[s] for the diplotype (h1, h2)
__device__ inline void rebuild_kmer_stats_cache(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint16_t h1, uint16_t h2, uint32_t nsub_u, uint32_t nsub_m) {
    {
    // Rebuild kmer_stats_cache[s] (VariantClusterHaplotypes.cpp:247-277).  The cache is 2 x V independent KmerStats accumulators
    // (haplotype slot x variant), each a strictly sequential Welford recurrence over the subset k-mers that lie on its haplotype
    // and overlap its variant, in subset order.  Every accumulator is run as its own pass IN REGISTERS — the copies of the group
    // take different accumulators — over the compact subset arrays read in blocks of eight k-mers (all operand loads of a block
    // first): per accumulator the same values in the same order as the reference's single interleaved pass.
    const uint32_t Hm = c.d().Hm, HWm = c.d().HWm, V = c.V;
    const bool two = h2 != NOHAP;
    const uint8_t g = P.gender[s];
    // The accumulators' state after the unique k-mers of the chain's subset is a pure function of (sample, diplotype) until the next
    // chain: the last few are kept (chains move between a handful of diplotypes), a hit skips the 2 V passes over the unique subset.
    const uint32_t dkey = (uint32_t)h1 | ((uint32_t)h2 << 16);
    TPtr<uint32_t> kk = c.ksc_key(s);
    uint32_t hit_way = KSC_WAYS, victim = 0;
    {
        uint32_t keys[KSC_WAYS];
#pragma unroll
        for (uint32_t e = 0; e < KSC_WAYS; ++e) keys[e] = kk[e];
        victim = kk[KSC_WAYS];
#pragma unroll
        for (uint32_t e = 0; e < KSC_WAYS; ++e)
            if (hit_way == KSC_WAYS && keys[e] == dkey) hit_way = e;
    }
    const Vx::RPtr<uint8_t> sm = c.subm();
    TPtr<uint8_t> scn = c.subcnt(), sic = c.subic();
    TPtr<uint32_t> so = c.skv_off(), sb = c.skv_bits(), msub = c.msub();
    TPtr<uint16_t> sv = c.skv_var();
    for (uint32_t a = c.t.part; a < 2 * V; a += c.t.copies) {
        const uint32_t which = a / V, var = a - which * V;
        KS acc{0, 0, 0, 0};
        const uint16_t h = which ? h2 : h1;
        if (h1 != NOHAP && (which == 0 || two)) {
            const uint32_t hw = h >> 5, hb = h & 31u;
            if (hit_way < KSC_WAYS)
                acc = ks_load(c.ksc_data(s, hit_way) + (which * c.d().Vm + var) * 4u);
            else {
            for (uint32_t i0 = 0; i0 < nsub_u; i0 += 8) {
                const uint32_t nb = nsub_u - i0 < 8u ? nsub_u - i0 : 8u;
                uint32_t eo[9];
                uint8_t dm[8], icn[8], cn[8];
#pragma unroll
                for (uint32_t q = 0; q < 9; ++q) eo[q] = q <= nb ? (uint32_t)so[i0 + q] : 0u;
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) {
                    const uint32_t i = q < nb ? i0 + q : i0;
                    uint8_t m = sm[i * Hm + h1];
                    if (two) m = (uint8_t)(m + sm[i * Hm + h2]);
                    dm[q] = q < nb ? m : (uint8_t)0;
                    icn[q] = sic[2 * i + g];
                    cn[q] = scn[i * P.S + s];
                }
                // first entry of every k-mer of the block (most k-mers overlap one variant)
                uint32_t v0[8], b0[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) {
                    const bool has = q < nb && eo[q + 1] > eo[q];
                    v0[q] = has ? (uint32_t)sv[eo[q]] : 0xFFFFFFFFu;
                    b0[q] = has ? (uint32_t)sb[eo[q] * HWm + hw] : 0u;
                }
#pragma unroll
                for (uint32_t q = 0; q < 8; ++q) {
                    if (q >= nb || dm[q] == 0) continue;
                    bool hit = v0[q] == var && ((b0[q] >> hb) & 1u);
                    for (uint32_t e = eo[q] + 1; e < eo[q + 1]; ++e)    // a k-mer overlaps a variant once: at most one entry matches
                        if (sv[e] == var && ((sb[e * HWm + hw] >> hb) & 1u)) hit = true;
                    if (hit) ks_add_r(acc, cn[q] / (double)(uint8_t)(dm[q] + icn[q]));
                }
            }
            }
        }
        // (the state after the unique k-mers is what is kept: the multicluster k-mers' multiplicities depend on the other clusters)
        if (hit_way == KSC_WAYS) ks_store(c.ksc_data(s, victim) + (which * c.d().Vm + var) * 4u, acc);
        if (h1 != NOHAP && (which == 0 || two)) {
            for (uint32_t i = 0; i < nsub_m; ++i) {
                const uint32_t k = msub[i];
                if (dip_mult(c, k, h1, h2) == 0) continue;
                bool hit = false;
                for (uint32_t e = c.kv_off(k), e1 = c.kv_off(k + 1); e < e1; ++e)
                    if (c.kv_var(e) == var && c.kv_bit(e, h)) hit = true;
                if (!hit) continue;
                const uint8_t mult = multi_mult(c, P, k, h1, h2, h1, h2, s);
                double kmer_count = 0;
                if (c.has_counts(k)) kmer_count = c.count(k, s) / (double)mult;
                ks_add_r(acc, kmer_count);
            }
        }
        ks_store(c.ksc(s, which, var), acc);
    }
    if (hit_way == KSC_WAYS) {
        kk[victim] = dkey;
        kk[KSC_WAYS] = victim + 1u < KSC_WAYS ? victim + 1u : 0u;
    }
    if (c.t.copies > 1u) copies_sync();
    }
}

// One collected sweep of a vertex: diplotype_sampling_frequencies (VariantClusterGenotyper.cpp:692-696) and
// updateAlleleKmerStats (VariantClusterHaplotypes.cpp:235-298).  A sample whose diplotype and k-mer-stats cache are the
// same as in the previous collected sweep contributes exactly the same updates again; those are counted (pend) and
// replayed later instead of being read-modify-written in HBM every sweep.
// the slow path of one sample of a collected sweep: materialise what is pending, rebuild the sample's k-mer-stats cache when its
// diplotype (or a multicluster multiplicity) changed, apply this sweep's contribution
__device__ inline void collect_sample_body(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint32_t nsub_u, uint32_t nsub_m) {
    SPtrF<uint16_t, LANES> dip = c.dip(), pdip = c.pend_dip();
    SPtrF<uint8_t, LANES> upd = c.ksc_upd(), pvalid = c.pend_valid();
    const uint16_t h1 = dip[2 * s], h2 = dip[2 * s + 1];
    const uint8_t u = upd[s];
    const uint32_t nn = c.nest_n()[s];
    PROF_DECL;
    PROF_CNT(19, 1);
    flush_sample(c, P, s);
    PROF(16);
    {
        if (u) {
            PROF_CNT(20, 1);
            upd[s] = 0;
            c.nver()[s] += 1;   // the children's nested info reads this cache
            rebuild_kmer_stats_cache(c, P, s, h1, h2, nsub_u, nsub_m);
            PROF(17);
        }
        {   // the nested sources this sweep sees become the sample's pending copy (what later identical sweeps are compared with)
            TPtr<double> pn = c.pend_nest(s);
            double nv[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t j = 0; j < 2u; ++j)
                if (j < nn) {
                    TPtr<double> q = c.nest_stats(s, j);
                    nv[3 * j] = q[0];
                    nv[3 * j + 1] = q[1];
                    nv[3 * j + 2] = q[2];
                }
            for (uint32_t j = 0; j < 2u; ++j) {
                pn[4 * j] = nv[3 * j];
                pn[4 * j + 1] = nv[3 * j + 1];
                pn[4 * j + 2] = nv[3 * j + 2];
            }
            pn[3] = (double)nn;
        }
        replay_collected(c, P, s, h1, h2, 1, nn);   // this sweep + addNestedHaplotypeKmerStats (:360-372)
        pvalid[s] = 1;
        pdip[2 * s] = h1;
        pdip[2 * s + 1] = h2;
        PROF(18);
    }
}
// ---- collected sweeps of single clusters without multicluster k-mers (TileDesc::logged): logged as runs, applied for all lanes together ----
// A wavefront used to pay the statistics update of a collected sweep — replay of the pending run, rebuild of the k-mer-stats
// cache, this sweep's contribution — whenever ANY of its lanes changed its diplotype, with one or two lanes active.  Here a sample's
// collected sweeps are only logged while sampling: runs (diplotype, length), appended when the diplotype changes.  The log is applied at
// the end of the chain (the cache is a function of the chain's k-mer subset), every lane working through its own entries at the same
// time.  Per sample the entries are applied in order, each exactly as the immediate update would have been (cache of that diplotype,
// then the run's replay), so every statistic sees the same values in the same order.
__device__ inline void apply_collected_log(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint32_t nsub_u) {
    TPtr<uint32_t> lg = c.evlog(s);
    const uint32_t n = c.evn()[s];
    for (uint32_t e = 0; e < n; ++e) {
        const uint32_t key = lg[1 + 2 * e], r = lg[2 + 2 * e];
        const uint16_t h1 = (uint16_t)(key & 0xFFFFu), h2 = (uint16_t)(key >> 16);
        rebuild_kmer_stats_cache(c, P, s, h1, h2, nsub_u, 0);
        replay_collected(c, P, s, h1, h2, r, 0);
    }
    c.evn()[s] = 0;
}
template <bool ROOM_CHECKED = false>   // ROOM_CHECKED: the caller has applied a full log already (apply_full_log, out of line)
__device__ inline void log_collected_run(const Vx &c, const GParams BT_CAS &P, uint32_t s, uint32_t key, uint32_t r, uint32_t nsub_u) {
    TPtr<uint32_t> lg = c.evlog(s);
    uint32_t n = c.evn()[s];
    if (!ROOM_CHECKED && n == EV_CAP) {   // (a sample that keeps changing its diplotype: apply what is logged now)
        apply_collected_log(c, P, s, nsub_u);
        n = 0;
    }
    lg[1 + 2 * n] = key;   // (stores only: nothing waits for HBM while sampling)
    lg[2 + 2 * n] = r;
    c.evn()[s] = (uint8_t)(n + 1);
}
// gibbs_single_kernel: the early application of a full log as a function of its own (once in thousands of sweeps; inlined, the statistics code would
// sit in the middle of the sweep's register allocation)
__device__ BT_NOINLINE void apply_full_log(Env env, uint32_t s) {
    const Vx c = make_vx(make_tile(env), 0);
    apply_collected_log(c, env_params(env), s, c.sc()[SC_NSUB_U]);
}
// end of a chain / of a launch: the open runs join the log, the log is applied
__device__ BT_NOINLINE void drain_collected(Env env) {
    const Vx c = make_vx(make_tile(env), 0);
    const GParams BT_CAS &P = env_params(env);
    const uint32_t nsub_u = c.sc()[SC_NSUB_U];
    SPtrF<uint16_t, LANES> pdip = c.pend_dip();
    SPtrF<uint8_t, LANES> pvalid = c.pend_valid(), upd = c.ksc_upd();
    SPtrF<uint32_t, LANES> pend = c.pend();
    for (uint32_t s = 0; s < P.S; ++s) {
        if (pvalid[s] && pend[s]) log_collected_run(c, P, s, (uint32_t)pdip[2 * s] | ((uint32_t)pdip[2 * s + 1] << 16), pend[s], nsub_u);
        pend[s] = 0;
        pvalid[s] = 0;
        upd[s] = 1;
        apply_collected_log(c, P, s, nsub_u);
    }
}

// materialise everything still pending (end of a launch: results may be read next)
__device__ BT_NOINLINE void flush_vertex(Env env, uint32_t vtx) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    if (c.d().logged) {
        drain_collected(env);
        return;
    }
    for (uint32_t s = 0; s < P.S; ++s) flush_sample(c, P, s);
}
// (BT_COLLECT_OUTLINE: a function of its own — the pending run's replay, the k-mer-stats rebuild and this sweep's contribution are 4 000 instructions that a
// sample needs when its diplotype changed in a collected sweep; out of line they stay out of the sweep's register allocation)
__device__ BT_NOINLINE void collect_sample_slow(Env env, uint32_t vtx, uint32_t s) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    SPtrF<uint32_t, LANES> sc = c.sc();
    collect_sample_body(c, P, s, sc[SC_NSUB_U], sc[SC_NSUB_M]);
}

__device__ BT_SWEEPFN void update_allele_kmer_stats(Env env, uint32_t vtx, uint32_t nsub_u, uint32_t nsub_m) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    SPtrF<uint16_t, LANES> dip = c.dip(), pdip = c.pend_dip();
    SPtrF<uint8_t, LANES> upd = c.ksc_upd(), pvalid = c.pend_valid();
    if (kSingle || c.d().logged) {   // single cluster, no multicluster k-mers: log the run, apply it with everybody else's at the end of the chain
        SPtrF<uint32_t, LANES> pend = c.pend();
        for (uint32_t s = 0; s < P.S; ++s) {
            const uint16_t h1 = dip[2 * s], h2 = dip[2 * s + 1];
            if (pvalid[s] && pdip[2 * s] == h1 && pdip[2 * s + 1] == h2) {
                pend[s] += 1;
                continue;
            }
            if (pvalid[s] && pend[s]) {
                if (kSingle) {
                    if (c.evn()[s] == EV_CAP) apply_full_log(env, s);
                    log_collected_run<true>(c, P, s, (uint32_t)pdip[2 * s] | ((uint32_t)pdip[2 * s + 1] << 16), pend[s], nsub_u);
                } else
                    log_collected_run(c, P, s, (uint32_t)pdip[2 * s] | ((uint32_t)pdip[2 * s + 1] << 16), pend[s], nsub_u);
            }
            pdip[2 * s] = h1;
            pdip[2 * s + 1] = h2;
            pend[s] = 1;
            pvalid[s] = 1;
        }
        return;
    }
    for (uint32_t s = 0; s < P.S; ++s) {
        const uint16_t h1 = dip[2 * s], h2 = dip[2 * s + 1];
#ifndef ABL_NODEFER
#ifdef BT_NODEFER_NARROW
        if (c.t.copies == 1u)
#endif
        if (pvalid[s] && !upd[s] && pdip[2 * s] == h1 && pdip[2 * s + 1] == h2) {
            const uint32_t nn = c.nest_n()[s];
            bool same = true;
            if (nn) {   // nested groups: the parent's contribution of this sweep equals the one the pending sweeps saw
                TPtr<double> pn = c.pend_nest(s);
                same = (double)pn[3] == (double)nn;
                for (uint32_t j = 0; j < 2u; ++j)
                    if (j < nn) {
                        TPtr<double> q = c.nest_stats(s, j);
                        same = same && (double)q[0] == (double)pn[4 * j] && (double)q[1] == (double)pn[4 * j + 1] && (double)q[2] == (double)pn[4 * j + 2];
                    }
            } else
                same = true;
            if (same) {
                c.pend()[s] += 1;
                continue;
            }
        }
#endif
#ifdef BT_COLLECT_OUTLINE
        collect_sample_slow(env, vtx, s);
#else
        collect_sample_body(c, P, s, nsub_u, nsub_m);
#endif
    }
}

// ---- sampleDiplotypes / sampleDiplotype / calcDiplotypeLogProb (VariantClusterGenotyper.cpp:597-755) ----
__device__ BT_SWEEPFN void sample_diplotypes(Env env, uint32_t vtx, bool collect, uint32_t trace_word, bool tracing, uint32_t *trace_buf) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    const TPtr<uint32_t> trace_row{(uint32_t BT_GAS *)uniform_ptr(trace_buf), trace_word, 6u};   // (trace blocks are interleaved over 64 lanes whatever the tile)
    SPtrF<uint32_t, LANES> sc = c.sc();
    const uint32_t nsub_u = sc[SC_NSUB_U], nsub_m = kSingle ? 0u : (uint32_t)sc[SC_NSUB_M];
    const bool use_multi = !kSingle && sc[SC_USE_MULTI] != 0, is_sparse = sc[SC_IS_SPARSE] != 0;
    uint32_t hap_count = sc[SC_HAP_COUNT];
    PROF_DECL;
    if (sc[SC_UC_DIRTY] == 2u) {   // invalidate only: NaN marks "not computed" (unique_log_prob fills on demand)
        const Vx::UCPtr ucl = c.ucache();
        const double nan = __builtin_nan("");
        for (uint32_t i = c.t.part, n = c.d().cache_entries; i < n; i += c.t.copies) ucl[i] = nan;
        if (c.t.copies > 1u) copies_sync();
        sc[SC_UC_DIRTY] = 0;
    } else if (sc[SC_UC_DIRTY]) {
        fill_unique_cache(env, vtx);
        sc[SC_UC_DIRTY] = 0;
    }
    PROF(12);
    MtRing rng = c.rng(0);
    SPtrF<uint16_t, LANES> nzl = c.nzlist();
    SPtrF<double, LANES> logf = c.logf();
    SPtrF<uint16_t, LANES> dip = c.dip();
    uint32_t nnz = 0;
    {
        SPtrF<uint8_t, LANES> nz = c.nz();
        for (uint32_t h = 0; h < c.H; ++h)
            if (nz[h]) nzl[nnz++] = (uint16_t)h;
    }
    PROF(0);
    // The S draws of a visit are independent of each other — a sample's candidates depend on the frequencies and on its own
    // multicluster state only, and every sample consumes exactly one uniform — so the copies of a narrow tile form `T` teams that
    // draw T samples at a time (team t takes sample s0 + t; inside a team the candidate blocks are shared as before).  The uniforms
    // are taken from the generator in sample order by every copy, and the picks are applied in sample order by every copy.
    const uint32_t T = c.t.copies > 1u && c.d().teams > 1u ? c.d().teams : 1u;
    const uint32_t tsz = c.t.copies / T, team = c.t.part / tsz, tpart = c.t.part - team * tsz;   // team >= T: no sample this round
    const uint32_t cw = 64u / c.t.copies, gl = (threadIdx.x & 63u) % cw, tlane0 = gl + team * tsz * cw;
    SPtrF<double, LANES> cum = c.cum() + (team < T ? team : 0u) * (c.d().D2m > 1 ? c.d().D2m : 1);
    for (uint32_t s0 = 0; s0 < P.S; s0 += T) {
      double u01 = 0;
      for (uint32_t j = 0; j < T && s0 + j < P.S; ++j) {   // LogDiscreteSampler::sample draws even for a single outcome (DiscreteSampler.cpp:120-125)
          const double u = rng_canonical(rng);
          if (j == team) u01 = u;
      }
      PROF(24);
      const uint32_t s = s0 + team;
      uint32_t mine = 0xFFFFFFFFu;
      if (team < T && s < P.S) {
        const uint16_t p1 = dip[2 * s], p2 = dip[2 * s + 1];
        const uint8_t ploidy = c.nest_ploidy()[s];
        uint32_t gen = 0;
        if (use_multi && nsub_m) {
            multi_refresh(c, P, s, p1, p2, nsub_m);
            gen = c.mgen()[s];
        }
        PROF(1);
        // Candidates in the reference's order.  The reference builds the cumulative log-sums with a chain of logAddition calls
        // (LogDiscreteSampler::addOutcome, DiscreteSampler.cpp:104-118) and picks upper_bound(cum, log(U) + cum.back()).
        // Small candidate sets do exactly that.  Larger sets take the same decision from linear-domain running sums of
        // exp(lp - max) against U * total (one exp per candidate instead of a dependent exp + log1p pair), and verify it: the
        // two formulations can only disagree when U * total lies within the accumulated rounding error of a boundary, so
        // whenever the threshold is closer than `margin` (>= 16x the worst-case bound) to either boundary of
        // the picked interval — about once in 10^5..10^6 draws — the reference's chain is evaluated after all.
        const uint32_t total = ploidy == 2 ? nnz * (nnz + 1) / 2 : (ploidy == 1 ? nnz : 0u);
        const bool chain_only = total <= BT_LINEAR_DRAW_MIN;
        const bool par = !chain_only && tsz > 1u;
        double lpmax = 0;
        // lp of every candidate -> cum[0..total), in order; returns through lpmax the maximum
        // Evaluated in blocks of 8: the cache words of a whole block are requested first (independent loads, one memory round trip
        // per block instead of one per candidate), then the block is finished in order.
        auto eval_candidates = [&]() {
            const bool dipl = ploidy == 2;
            const TileDesc BT_CAS &dd = c.d();
            const Vx::UCPtr uc = c.ucache();
            TPtr<double> mc = c.mcache();
            TPtr<uint32_t> uct = c.uctag(), mct = c.mctag(), mcg = c.mcgen();
            const bool multi = use_multi && nsub_m != 0;
            uint32_t a = 0, b = 0;   // enumeration state (diploid: b >= a)
            // skip `steps` candidates of the enumeration (row a holds nnz - a candidates when diploid, one when haploid)
            auto advance = [&](uint32_t steps) {
                if (!dipl) {
                    a += steps;
                    return;
                }
                while (steps > 0 && a < nnz) {
                    const uint32_t left = nnz - b;
                    if (steps < left) {
                        b += steps;
                        steps = 0;
                    } else {
                        steps -= left;
                        ++a;
                        b = a;
                    }
                }
            };
            // with copies of the group in the wavefront (Tile::part) every copy evaluates every copies-th block (dense tables: a slot
            // is private to a candidate; hashed tables: a slot is private to a copy, see hashed_slot)
            const uint32_t stride_blocks = par ? tsz : 1u;
            if (par) advance(EVB * tpart);
            lpmax = -__builtin_huge_val();
            for (uint32_t base = par ? EVB * tpart : 0u; base < total; base += EVB * stride_blocks) {
                const uint32_t nb = total - base < EVB ? total - base : EVB;
                uint16_t ha[EVB], hb[EVB];
                uint32_t uslot[EVB], ukey[EVB];
                double uval[EVB], mval[EVB], la[EVB], lb[EVB];
                uint32_t utag[EVB], mtag[EVB], mgn[EVB];
#pragma unroll
                for (uint32_t q = 0; q < EVB; ++q) {
                    if (q < nb) {
                        ha[q] = nzl[a];
                        hb[q] = dipl ? (uint16_t)nzl[b] : NOHAP;
                        if (dipl) {
                            if (++b == nnz) {
                                ++a;
                                b = a;
                            }
                        } else
                            ++a;
                        const uint32_t idx = dip_index(c, ha[q], hb[q]);
                        ukey[q] = s * dd.Dcm + idx + 1u;
                        uslot[q] = dd.cache_mode == 0 ? s * dd.Dcm + idx : hashed_slot(c, ukey[q]);
                        uval[q] = dd.cache_mode != 2 ? (double)uc[uslot[q]] : 0.0;
                        utag[q] = dd.cache_mode == 1 ? (uint32_t)uct[uslot[q]] : 0u;
                        if (multi) {
                            mtag[q] = mct[uslot[q]];
                            mgn[q] = mcg[uslot[q]];
                            mval[q] = mc[uslot[q]];
                        }
                        la[q] = logf[ha[q]];   // == log(freq[h]) (VariantClusterGenotyper.cpp:603-616 computes it per candidate)
                        lb[q] = dipl ? (double)logf[hb[q]] : 0.0;
                    }
                }
                if (multi) {
                    bool need[EVB], any = false;
#pragma unroll
                    for (uint32_t q = 0; q < EVB; ++q) {
                        need[q] = q < nb && !(mtag[q] == ukey[q] && mgn[q] == gen);
                        any = any || need[q];
                    }
                    if (any) {
                        double fresh[EVB];
                        multi_log_prob_block(c, P, s, ha, hb, need, nsub_m, fresh);
#pragma unroll
                        for (uint32_t q = 0; q < EVB; ++q)
                            if (need[q]) {
                                mval[q] = fresh[q];
                                mct[uslot[q]] = ukey[q];
                                mcg[uslot[q]] = gen;
                                mc[uslot[q]] = fresh[q];
                            }
                    }
                }
                {   // dense tables are complete (fill_unique_cache) or hold NaN where not computed yet; hashed tables fill on demand: the block's misses together
                    bool umiss[EVB], any = false;
#pragma unroll
                    for (uint32_t q = 0; q < EVB; ++q) {
                        umiss[q] = q < nb && !((dd.cache_mode == 0 && uval[q] == uval[q]) || (dd.cache_mode == 1 && utag[q] == ukey[q]));
                        any = any || umiss[q];
                    }
                    if (any) {
                        double fresh[EVB];
                        unique_log_prob_block(c, P, s, ha, hb, umiss, nsub_u, fresh);
#pragma unroll
                        for (uint32_t q = 0; q < EVB; ++q)
                            if (umiss[q]) uval[q] = fresh[q];
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < EVB; ++q) {
                    if (q < nb) {
                        double lp = 0;
                        if (!dipl) lp += la[q];
                        else if (ha[q] == hb[q]) lp += 2 * la[q];
                        else lp += BT_LN2 + la[q] + lb[q];
                        lp += uval[q];
                        if (multi) lp += mval[q];
                        lpmax = lp > lpmax ? lp : lpmax;
                        cum[base + q] = lp;
                    }
                }
                if (par) advance(EVB * (stride_blocks - 1u));
            }
            if (par) {   // maximum over the team's copies (lanes tlane0, tlane0 + cw, ...), then make their cum[] entries visible
                const double own = lpmax;
                for (uint32_t m = 0; m < tsz; ++m) {
                    const double o = __shfl(own, (int)(tlane0 + m * cw));
                    lpmax = o > lpmax ? o : lpmax;
                }
                copies_sync();
            }
        };
        // the reference's decision on cum[] holding the lp values: chain, then upper_bound (first index with cum > u)
        auto chain_pick = [&](double u01) -> uint32_t {
            if (total == 0) return 0u;
            double run = cum[0];
            for (uint32_t i = 1; i < total; ++i) {
                run = log_addition((double)cum[i], run);
                cum[i] = run;
            }
            const double u = bt_log(u01) + run;
            uint32_t lo = 0, hi = total;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (u < cum[mid]) hi = mid;
                else lo = mid + 1;
            }
            return lo < total ? lo : total - 1;
        };
        // LogDiscreteSampler::sample (DiscreteSampler.cpp:120-125): the draw happens even for a single outcome.  (The evaluation of
        // the candidates consumes no random numbers, so drawing first does not change the stream.)
        uint32_t pick = 0;
        // one evaluation site (the blocked evaluation is large: instantiating it twice would not fit the instruction cache)
        for (bool exact = chain_only;; exact = true) {
            eval_candidates();
            PROF(2);
            if (exact) {
                if (total == 0) (void)bt_log(u01);
                pick = chain_pick(u01);
                PROF(28);
                break;
            }
            double acc = 0, thr = 0, off = 0;     // off: cumulative weight before the searched range [s0, s1)
            uint32_t s0 = 0, s1 = total;
            {
                // The weights exp(lp - max) and their running sums are this kernel's own formulation of the draw (the decision is verified
                // against the reference's chain below), so their summation order is free: every copy takes one contiguous segment of the
                // candidates — exponentials and a local running sum in registers —, the segment totals are exchanged with lane shuffles,
                // and every copy searches the one segment that holds U * total.
                const uint32_t ncp = tsz, L = (total + ncp - 1u) / ncp;
                const uint32_t i0 = tpart * L < total ? tpart * L : total, i1 = i0 + L < total ? i0 + L : total;
                double run = 0;
                for (uint32_t b0 = i0; b0 < i1; b0 += 8) {
                    double e[8];
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q) e[q] = b0 + q < i1 ? (double)cum[b0 + q] : 0.0;
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q)
                        if (b0 + q < i1) e[q] = bt_exp(e[q] - lpmax);
#pragma unroll
                    for (uint32_t q = 0; q < 8; ++q)
                        if (b0 + q < i1) {
                            run += e[q];
                            cum[b0 + q] = run;
                        }
                }
                for (uint32_t m = 0; m < ncp; ++m) acc += __shfl(run, (int)(tlane0 + m * cw));
                thr = u01 * acc;
                uint32_t seg = ncp;
                for (uint32_t m = 0; m < ncp; ++m) {
                    const double tseg = __shfl(run, (int)(tlane0 + m * cw));
                    if (seg == ncp) {
                        if (thr < off + tseg) seg = m;
                        else off += tseg;
                    }
                }
                if (tsz > 1u) copies_sync();
                s0 = seg * L < total ? seg * L : total;     // seg == ncp (U * total not below the grand total): empty range -> not safe
                s1 = s0 + L < total ? s0 + L : total;
            }
            PROF(25);
            uint32_t lo = s0, hi = s1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (thr < off + (double)cum[mid]) hi = mid;
                else lo = mid + 1;
            }
            const double amax = fabs(lpmax) > 1.0 ? fabs(lpmax) : 1.0;
            double margin = 64.0 * (double)total * amax * BT_DBL_EPS;
            margin = (margin > 1e-6 ? margin : 1e-6) * acc;
            const bool safe = lo < s1 && off + (double)cum[lo] - thr > margin && (lo == 0 || thr - (lo == s0 ? off : off + (double)cum[lo - 1]) > margin);
            PROF(26);
            if (safe) {
                pick = lo;
                break;
            }
            PROF_CNT(14, 1000000);   // shows up as 1.0 per fallback in the "nested" column of scratch/prof_phases.py
        }
        uint16_t h1 = NOHAP, h2 = NOHAP;
        if (ploidy == 2) {
            uint32_t a = 0, rem = pick;   // invert the (a, b >= a) enumeration
            while (rem >= nnz - a) {
                rem -= nnz - a;
                ++a;
            }
            h1 = nzl[a];
            h2 = nzl[a + rem];
        } else if (ploidy == 1) {
            h1 = nzl[pick];
        }
        mine = (uint32_t)h1 | ((uint32_t)h2 << 16);
      }
      if (T > 1u) copies_sync();
      PROF(3);
      for (uint32_t j = 0; j < T && s0 + j < P.S; ++j) {   // apply the round's picks in sample order (every copy)
        const uint32_t ss = s0 + j;
        const uint32_t hh = T > 1u ? (uint32_t)__shfl((int)mine, (int)(gl + j * tsz * cw)) : mine;
        const uint16_t h1 = (uint16_t)(hh & 0xFFFFu), h2 = (uint16_t)(hh >> 16);
        const uint16_t p1 = dip[2 * ss], p2 = dip[2 * ss + 1];
        dip[2 * ss] = h1;
        dip[2 * ss + 1] = h2;
        if (!kSingle && c.d().nvm > 1u && (h1 != p1 || h2 != p2)) c.nver()[ss] += 1;
        hfd_increment(c, h1, is_sparse, hap_count);
        hfd_increment(c, h2, is_sparse, hap_count);
        PROF(4);
        update_multicluster_multiplicities(c, P, h1, h2, p1, p2, ss, nsub_m);
        PROF(5);
        if (tracing) trace_row[ss] = hh;
      }
    }
    mt_close(rng);
    sc[SC_HAP_COUNT] = hap_count;
#ifndef ABL_NOSTATS
    if (collect) update_allele_kmer_stats(env, vtx, nsub_u, nsub_m);
#endif
    PROF(6);
    sc[SC_USE_MULTI] = nsub_m != 0 ? 1u : 0u;
}

// Top the draw-ahead rings of a cluster's two generators up at the start of a visit: the whole wavefront refills together (one burst
// of independent state loads per generator), and the draws of the visit then read LDS.  Generating ahead does not change the stream.
__device__ BT_SWEEPFN void rng_topup(Env env, uint32_t vtx) {
    const Vx c = make_vx(make_tile(env), vtx);
    MtRing a = c.rng(0), b = c.rng(1);
    a.topup();
    b.topup();
    mt_close(a);
    mt_close(b);
}

// ---- SparseFrequencyDistribution::updateCachedSimplexProbVector (FrequencyDistribution.cpp:143-196) -> out[], returns length ----
// Every lgamma argument in that formula is a positive integer (<= H + 2S + 1): the values come from a host-computed table.
template <typename Q>
__device__ inline uint32_t simplex_prob_vector(const Vx &c, const GParams BT_CAS &P, Q out, uint32_t total_obs, uint32_t plus_size) {
    const uint32_t Hn = c.H;
    const double BT_GAS *lg = P.lgamma_int;
    const double sparsity = c.sparsity();
    const double lsp = bt_log(sparsity), l1sp = bt_log(1 - sparsity);
    double prob_z_log = plus_size * lsp + (Hn - plus_size) * l1sp;
    double prob_t_log = lg[plus_size] - lg[total_obs + plus_size];
    double prob_eq_z_log = 0.0 + prob_z_log + prob_t_log;
    double row_sum = prob_eq_z_log;
    uint32_t n = 0;
    out[n++] = row_sum;
    double prev = row_sum;
    for (uint32_t j = plus_size + 1; j < Hn + 1; ++j) {
        const double cardinal = lg[Hn - plus_size + 1] - (lg[j - plus_size + 1] + lg[Hn - j + 1]);
        prob_z_log = j * lsp + (Hn - j) * l1sp;
        prob_t_log = lg[j] - lg[total_obs + j];
        prob_eq_z_log = cardinal + prob_z_log + prob_t_log;
        row_sum += bt_log(1 + bt_exp(prob_eq_z_log - row_sum));
        out[n++] = row_sum;
        const double a = row_sum, b = prev;
        const double mn = a < b ? a : b;
        prev = row_sum;
        if (a == b || fabs(a - b) < fabs(mn) * BT_DBL_EPS * 100) break;   // Utils::doubleCompare
    }
    for (uint32_t i = 0; i < n; ++i) out[i] = bt_exp((double)out[i] - row_sum);
    return n;
}

// gamma_distribution(count + 1, 1): Marsaglia-Tsang with a2 = 1 / sqrt(9 (alpha - 1/3)) read from the host's table for the small integers alpha takes
// (the same two IEEE operations; GParams::gamma_a2) instead of a square root and a division per draw
template <class G>
__device__ inline double rng_gamma_count(G &rng, NormalState nd, const GParams BT_CAS &P, uint32_t count) {
    const uint32_t ai = count + 1u;
    const double a1 = (double)ai - 1.0 / 3.0;
    const double a2 = ai < P.gamma_n ? (double)P.gamma_a2[ai] : 1.0 / sqrt(9.0 * a1);
    return a1 * rng_gamma_v(rng, nd, a1, a2) * 1.0;
}

// ---- sampleHaplotypeFrequencies (VariantClusterGenotyper.cpp:781-785 -> HaplotypeFrequencyDistribution.cpp:127-138
//      -> FrequencyDistribution.cpp:75-93 / 209-303) ----
__device__ BT_SWEEPFN void sample_haplotype_frequencies(Env env, uint32_t vtx) {
    const Vx c = make_vx(make_tile(env), vtx);
    const GParams BT_CAS &P = env_params(env);
    const TileDesc BT_CAS &d = c.d();
    SPtrF<uint32_t, LANES> sc = c.sc();
    const uint32_t n_obs = sc[SC_HAP_COUNT];
    PROF_DECL;
    if (n_obs > 0) {
        MtRing rng = c.rng(1);
        const NormalState nd = c.fnd();
        SPtrF<uint32_t, LANES> obs = c.obs();
        SPtrF<double, LANES> freq = c.freq();
        SPtrF<uint8_t, LANES> nz = c.nz();
        if (!sc[SC_IS_SPARSE]) {
            double norm = 0;
            for (uint32_t h = 0; h < c.H; ++h) {
                const double f = rng_gamma_count(rng, nd, P, obs[h]);
                freq[h] = f;
                norm += f;
                obs[h] = 0;
            }
            SPtrF<double, LANES> logf = c.logf();
            for (uint32_t h = 0; h < c.H; ++h) {
                const double f = freq[h] / norm;
                freq[h] = f;
                logf[h] = bt_log(f);
            }
        } else {
            Vx::HSet plus = c.plus_set(), zero = c.zero_set();
            SPtrF<uint32_t, LANES> unext = c.unext();
            const uint32_t plus_size = uset_size(plus);
            // cached_simplex_prob_vectors[sum_observation_counts][|plus| - 1] (FrequencyDistribution.cpp:211-229): the vector is a pure
            // function of (n_obs, |plus|), so caching it or not is invisible; cached when the tile reserved room for it
            uint32_t len;
            TPtr<double> vec = c.simplex();
            const bool cached = d.scache_n && n_obs <= 2 * d.S && plus_size <= d.scache_p;
            const uint32_t ci = cached ? (n_obs - 1) * d.scache_p + (plus_size - 1) : 0u;
            if (cached) vec = c.scache() + (uint32_t)ci * d.scache_len;
            // the first entries of a cached vector are requested together with its length (one memory round trip instead of a
            // dependent chain of them: the search below usually ends within the first two entries)
            constexpr uint32_t PRE = 4;
            const uint32_t pre = d.scache_len < PRE ? d.scache_len : PRE;
            double head[PRE];
#pragma unroll
            for (uint32_t q = 0; q < PRE; ++q) head[q] = (cached && q < pre) ? (double)vec[q] : 0.0;
            len = cached ? (uint32_t)c.sclen()[ci] : 0u;
            if (len == 0) {
                len = simplex_prob_vector(c, P, vec, n_obs, plus_size);
                if (cached) c.sclen()[ci] = len;
#pragma unroll
                for (uint32_t q = 0; q < PRE; ++q) head[q] = q < len ? (double)vec[q] : 0.0;
            }
            PROF(21);
            const double u = rng_canonical(rng);
            uint32_t ub = 0;   // upper_bound over a non-decreasing vector
#pragma unroll
            for (uint32_t q = 0; q < PRE; ++q)
                if (ub == q && q < len && !(u < head[q])) ub = q + 1;
            if (ub == PRE)
                while (ub < len && !(u < vec[ub])) ++ub;
            const uint32_t simplex_size = ub + plus_size;
            double norm = 0;
            for (uint32_t e = uset_begin(plus); e != US_NONE; e = unext[e]) {
                const double f = rng_gamma_count(rng, nd, P, obs[e]);
                freq[e] = f;
                norm += f;
                nz[e] = 2;   // selected in this call (turned into 1 below)
            }
            PROF(22);
            while (uset_size(plus) < simplex_size) {
                const uint32_t pos = rng_uniform_int(rng, uset_size(zero));   // uniform_int(0, |zero| - 1)
                uint32_t e = uset_begin(zero);
                for (uint32_t i = 0; i < pos; ++i) e = unext[e];
                const double f = rng_gamma_count(rng, nd, P, 0u);
                freq[e] = f;
                norm += f;
                nz[e] = 2;
                // the two sets share the `next` words: leave the zero list before entering the plus list
                uset_erase(zero, e);
                uset_insert(plus, e);
            }
            PROF(23);
            // "for z in zero: freq = 0, nz = 0, obs = 0": the zero set is exactly the haplotypes not selected above, so the reset
            // runs over the arrays (independent accesses, eight in flight) instead of chasing the set's list links
            {
                uint32_t h = 0;
                for (; h + 8 <= c.H; h += 8) {
                    uint8_t m8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) m8[q] = nz[h + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (m8[q] == 2) nz[h + q] = 1;
                        else {
                            freq[h + q] = 0;
                            nz[h + q] = 0;
                            obs[h + q] = 0;
                        }
                    }
                }
                for (; h < c.H; ++h) {
                    if (nz[h] == 2) nz[h] = 1;
                    else {
                        freq[h] = 0;
                        nz[h] = 0;
                        obs[h] = 0;
                    }
                }
            }
            // "for p in plus: freq /= norm; zero.insert(p); obs = 0" then plus.clear(): record the plus iteration order
            // first (shared `next` words), clear plus, then insert into zero in that order — same final containers
            SPtrF<uint16_t, LANES> nzl = c.nzlist();
            uint32_t np = 0;
            for (uint32_t e = uset_begin(plus); e != US_NONE; e = unext[e]) nzl[np++] = (uint16_t)e;
            uset_clear(plus);
            for (uint32_t i = 0; i < np; ++i) {
                const uint32_t e = nzl[i];
                const double f = freq[e] / norm;
                freq[e] = f;
                c.logf()[e] = bt_log(f);
                uset_insert(zero, e);
                obs[e] = 0;
            }
        }
        mt_close(rng);
    }
    PROF(7);
    sc[SC_HAP_COUNT] = 0;
}

}  // namespace bt
