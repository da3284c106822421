// libstdc++-compatible random-number stack and unordered_set<uint> iteration-order emulation for the Gibbs
// kernels.  The reference draws every random decision from std::mt19937 through
// libstdc++'s distribution classes and iterates std::unordered_set<uint>; posteriors within 1e-4 require the
// same draw STREAM, so these are re-implemented here from the published algorithms:
//
//   mt19937                          Matsumoto & Nishimura 1998 (state 624 x u32, tempering constants of [rand.predef])
//   generate_canonical<double,53>    two 32-bit draws: (x0 + x1 * 2^32) / 2^64, clamped below 1   (bits/random.tcc:3362-3373)
//   uniform_int_distribution         Lemire's nearly-divisionless method on a 32-bit engine         (bits/uniform_int_dist.h:246-268)
//   bernoulli_distribution           canonical() < p                                                (bits/random.h:3635-3643)
//   normal_distribution              Marsaglia polar with a saved second variate                    (bits/random.tcc:1804-1835)
//   gamma_distribution               Marsaglia-Tsang over that normal; the normal's saved variate survives param() (bits/random.tcc:2337-2396)
//   std::shuffle                     two swap positions per draw                                    (bits/stl_algo.h:3706-3790)
//   std::unordered_set<unsigned>     _Hashtable: identity hash, prime bucket counts 13,29,59,..., insert-at-bucket-begin,
//                                    rehash re-threading in list order, clear() keeps the bucket array (bits/hashtable.h)
//
// All state lives in caller-provided memory (HBM on the device); every function is __host__ __device__ so the same
// code is exercised on the CPU through the bt_diag_* entry points.  Build with -ffp-contract=off: the reference's
// x86-64 build has no fused multiply-add, and the draw stream depends on individually rounded operations.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace bt {

#define BT_HD __host__ __device__ inline
// out-of-line device functions of the samplers; the translation unit of gibbs_simple_kernel inlines them under its own register budget
// BT_SWEEPFN: the functions a sweep calls per vertex visit.  Out of line they each save and restore ~120 callee-saved registers per call
// (30 KB of scratch traffic per wavefront and call); BT_SWEEP_INLINE makes them part of the kernel body instead.
#if defined(BT_SIMPLE_TU) || defined(BT_SWEEP_INLINE)
#define BT_SWEEPFN inline
#else
#define BT_SWEEPFN static __noinline__
#endif
#define BT_NOINLINE static __noinline__   // (internal linkage: the code generator then drops the callee-saved register convention for them)

// fp64 transcendental functions.  On the device they are out-of-line: ocml's double-precision log/exp/log1p/pow need many
// registers; keeping them as separate functions keeps the samplers' own allocation small enough for 2-4 waves per SIMD.
#if defined(__HIP_DEVICE_COMPILE__)
#define BT_MATH __host__ __device__ __noinline__
#ifdef BT_INLINE_MATH
#define BT_MATHI __host__ __device__ inline
#else
#define BT_MATHI BT_MATH
#endif
#else
#define BT_MATH __host__ __device__ inline
#define BT_MATHI BT_MATH
#endif
BT_MATHI double bt_log(double x) { return log(x); }
BT_MATHI double bt_exp(double x) { return exp(x); }
BT_MATH double bt_log1p(double x) { return log1p(x); }
BT_MATH double bt_pow(double x, double y) { return pow(x, y); }

// Pointer with an element stride: element i lives at p[i * STRIDE].  STRIDE = 1 is an ordinary array; STRIDE = 64 is the
// lane-interleaved layout of the Gibbs tiles (element i of all 64 lanes of a wavefront adjacent in memory, so a wave-uniform
// index is one coalesced transaction).
// Represented as a (wave-uniform) base pointer plus a 32-bit element offset, so that on the device an access is
// "scalar base + one vector offset register" instead of a 64-bit per-lane pointer pair; on the device the base is
// typed as a GLOBAL-address-space pointer so that the compiler emits global_* rather than flat_* instructions.
#if defined(__HIP_DEVICE_COMPILE__)
#define BT_GAS __attribute__((address_space(1)))
#define BT_CAS __attribute__((address_space(4)))   // read-only ("constant") global memory: eligible for scalar loads
#define BT_LAS __attribute__((address_space(3)))   // LDS
#else
#define BT_GAS
#define BT_CAS
#define BT_LAS
#endif
template <typename T, unsigned STRIDE>
struct SPtr {
    T BT_GAS *base;
    uint32_t off;
    BT_HD T BT_GAS &operator[](uint32_t i) const { return base[off + i * STRIDE]; }
    BT_HD SPtr<T, STRIDE> operator+(uint32_t i) const { return SPtr<T, STRIDE>{base, off + i * STRIDE}; }
};
template <typename T>
BT_HD SPtr<T, 1> sptr1(T *p) { return SPtr<T, 1>{(T BT_GAS *)p, 0u}; }
// Array of a Gibbs tile in HBM: element i of lane l at base[off + (i << sh)], the rows interleaved over the tile's OWN width 2^sh
// (4 .. 64 lanes; a run-time, wave-uniform shift).  A narrow tile's rows are as wide as the tile, so a cache line holds consecutive
// elements of the tile's own groups instead of one element of 64 lanes of which the tile uses a few.
template <typename T>
struct TPtr {
    T BT_GAS *base;
    uint32_t off;
    uint32_t sh;
    BT_HD T BT_GAS &operator[](uint32_t i) const { return base[off + (i << sh)]; }
    BT_HD TPtr<T> operator+(uint32_t i) const { return TPtr<T>{base, off + (i << sh), sh}; }
};
// same, with a generic ("flat") base pointer: may point into LDS as well as into HBM
template <typename T, unsigned STRIDE>
struct SPtrF {
    T *base;
    uint32_t off;
    uint32_t sh = STRIDE == 64 ? 6u : 0u;   // run-time: log2 of the lanes the rows are interleaved over (the tile's width)
    BT_HD T &operator[](uint32_t i) const { return base[off + (i << sh)]; }
    BT_HD SPtrF<T, STRIDE> operator+(uint32_t i) const { return SPtrF<T, STRIDE>{base, off + (i << sh), sh}; }
};
#if defined(__HIP_DEVICE_COMPILE__)
// tell the compiler that a pointer handed through a non-inlined call is wave-uniform (it then lives in scalar registers and
// loads through it can be scalar / use the scalar-base addressing mode)
template <typename T>
__device__ inline T *uniform_ptr(T *p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T *)(((uint64_t)hi << 32) | lo);
}
#else
template <typename T>
__host__ __device__ inline T *uniform_ptr(T *p) { return p; }
#endif

// ------------------------------------------------------------------------------------------------------------
// mt19937: st[0..623] state words, st[624] position
// ------------------------------------------------------------------------------------------------------------
constexpr unsigned MT_N = 624, MT_M = 397, MT_WORDS = 625;

BT_HD void mt_seed(uint32_t *st, uint32_t seed) {
    uint32_t x = seed;
    st[0] = x;
    for (unsigned i = 1; i < MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + i;
        st[i] = x;
    }
    st[MT_N] = 0;   // position of the next word to generate
}

// One output word.  The textbook generator regenerates all 624 words at once and then reads them out; here word p is
// regenerated in place right before it is read.  The recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) touches, at
// position p, only words that the block form has in the same old/new state (p+1 is still old, p+397 mod 624 is old for
// p < 227 and already new afterwards), so the output stream is identical — and no lane ever runs a 624-step refill loop
// while its wavefront neighbours wait.
BT_HD uint32_t mt_temper(uint32_t z) {
    z ^= (z >> 11);
    z ^= (z << 7) & 0x9d2c5680u;
    z ^= (z << 15) & 0xefc60000u;
    z ^= (z >> 18);
    return z;
}
BT_HD uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {   // new word from (st[p], st[p+1], st[p+397])
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
BT_HD uint32_t mt_wrap(uint32_t i) { return i >= MT_N ? i - MT_N : i; }

// A generator in use, DIRECT form: state words in memory + the position held in a register for the duration of a call sequence
// (mt_open loads it, mt_close writes it back).  With the position in a register the three state loads of a draw have
// independent addresses, i.e. ONE memory round trip per draw instead of a dependent chain of four.
struct Mt {
    uint32_t BT_GAS *st;
    uint32_t pos;
    BT_HD uint32_t next() {
        const uint32_t p = pos, p1 = mt_wrap(p + 1), pm = mt_wrap(p + MT_M);
        const uint32_t z = mt_twist(st[p], st[p1], st[pm]);
        st[p] = z;
        pos = p1;
        return mt_temper(z);
    }
    // two / four consecutive words with all state loads issued together: none of the regenerated words is an input of another (the
    // recurrence reaches 1 and 397 positions ahead and 227 behind), so loading everything first and storing afterwards gives the
    // textbook stream
    template <unsigned N>
    BT_HD void next_n(uint32_t (&out)[N]) {
        const uint32_t p = pos;
        uint32_t a[N + 1], c[N];
#pragma unroll
        for (uint32_t k = 0; k < N + 1; ++k) a[k] = st[mt_wrap(p + k)];
#pragma unroll
        for (uint32_t k = 0; k < N; ++k) c[k] = st[mt_wrap(mt_wrap(p + k) + MT_M)];
#pragma unroll
        for (uint32_t k = 0; k < N; ++k) {
            const uint32_t z = mt_twist(a[k], a[k + 1], c[k]);
            st[mt_wrap(p + k)] = z;
            out[k] = mt_temper(z);
        }
        pos = mt_wrap(p + N);
    }
};
BT_HD Mt mt_open(uint32_t *st) { return Mt{(uint32_t BT_GAS *)st, ((uint32_t BT_GAS *)st)[MT_N]}; }
BT_HD void mt_close(const Mt &m) { m.st[MT_N] = m.pos; }

// DRAW-AHEAD form (the Gibbs kernels): the same stream, but the generator's output is produced in bursts ahead of its use into a small ring that
// lives in LDS, and the sampler's draws read the ring.  A draw then costs an LDS access instead of a dependent HBM round trip per call; bursts are
// issued for the whole wavefront at fixed points (topup at the start of a cluster visit), so the lanes of a wavefront refill together.
// Ring block of one generator: [cap] tempered words, then {position of the next state word to generate, ring head, words available}.
//
// Round 4: production in ALIGNED CHUNKS OF FOUR WORDS.  The lanes of a wavefront are at different positions of their own (per-lane contiguous) states,
// so every state access is 64 separate requests whatever it fetches — the memory pipeline takes them one lane per cycle, and at three single-word
// loads and one store per generated word the generators alone kept a CU's address unit busy for most of a sweep (profiles/r04: 283 memory
// instructions per wavefront-sweep of two-haplotype clusters).  A chunk [p, p + 4), p a multiple of four (624 = 4 * 156: a chunk never wraps), is
// one 16-byte load of the four words, one word x[p + 4], one unaligned 16-byte load of x[p + 397 .. p + 400] and one 16-byte store: four
// requests per lane for four words.  Two of those reach past the end of the state — x[p + 4] at p = 620 and x[p + 397 ...] at p = 224 .. 226 — and
// read a MIRROR of words 0 .. 3 kept at [624, 628), rewritten with them (chunk 0).  None of a chunk's new words is an input of another (the recurrence
// reaches 1 and 397 positions ahead), so loading everything first gives the textbook stream.
constexpr unsigned MT_RING_HDR = 3;
constexpr unsigned MT_MIRROR = 4;   // words 0 .. 3 again at [MT_N, MT_N + 4)
BT_HD void bt_sched_fence() {   // the instruction scheduler moves nothing across this point
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
BT_HD bool bt_wave_any(bool p) {   // true in every lane of the wavefront when p holds in any of them
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p) != 0;
#else
    return p;
#endif
}
#ifdef BT_DIAG_FAKE_MT
// DIAGNOSTIC BUILD ONLY (tools/traffic_by_array.sh): the generators' words come from a counter hash instead of the mt19937 state, so a launch moves no
// generator state at all — its FETCH_SIZE / WRITE_SIZE against the product build's is what the states cost.  Results are NOT the reference's.
BT_HD uint32_t bt_fake_word(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
#endif
#if !defined(BT_MT_CHUNK) && !defined(BT_MT_BLOCK)
#define BT_MT_BLOCK 1   // the default since round 6; -DBT_MT_CHUNK builds the chunk form (rounds 4-5) for comparison
#endif
// ---- BLOCK form of a ring generator (round 6, BT_MT_BLOCK) ----------------------------------------------------------------------------------------
// The chunk form regenerates four words in place right before they are read: per lane and chunk a 16-byte load, a word, an unaligned 16-byte load and a
// 16-byte store — and since the lanes of a wavefront sit at different positions of their own (per-lane contiguous) states, every one of those instructions
// is 64 separate cache lines to the CU's L1 (5.5e9 such instructions per schedule of the bench batch: the two-haplotype class's bound).
// Block form: a generator has TWO state buffers (MT_BUF words apart).  The current one holds a block of 624 already twisted words of which [pos, 624) are
// unread; a refill is ONE aligned 16-byte load per lane and four words.  The NEXT block is twisted AHEAD, out of place, into the other buffer by the whole
// wavefront: new[i] = f(old[i], old[i + 1], i < 227 ? old[i + 397] : new[i - 227]) (i = 623: f(old[623], new[0], new[396])), so the lane that owns words
// 4q .. 4q + 3 of the three phases [0, 227), [227, 454), [454, 624) needs, beside old words, only its OWN results of the phase before (and new[0], three old
// words away): seven loads in flight, three stores, consecutive lanes on consecutive addresses — coalesced — and no exchange between lanes.  The twist-ahead
// runs at the convergent top-up of a visit once a lane has passed MT_AHEAD words of its block, so that reaching the end of a block — which happens wherever the
// lane is, usually inside a rejection loop with most of the wavefront masked off — is a flip of the buffer index.  A lane that does reach the end without a
// next block (chain starts draw hundreds of words between top-ups) gets it from the lanes that are active with it.  The output stream is the textbook's.
constexpr unsigned MT_BUF = 640;     // words between a ring generator's two state buffers
constexpr unsigned MT_AHEAD = 312;   // words of a block consumed before the next block is twisted ahead at a top-up
constexpr uint32_t MT_F_CUR = 1u << 16, MT_F_READY = 1u << 17;   // flags kept with the position word of the ring block: current buffer, next block ready
typedef uint32_t MtQuad __attribute__((ext_vector_type(4)));    // four consecutive state words at a 16-byte aligned address (a chunk)
typedef MtQuad MtQuadU __attribute__((aligned(4)));             // ... at a 4-byte aligned address (x[p + 397 ...])
#if defined(__HIP_DEVICE_COMPILE__)
// every active lane of the wavefront calls this with the SAME pointers: the block after `old` is written to `nw`
__device__ inline void mt_twist_block(const uint32_t BT_GAS *old, uint32_t BT_GAS *nw) {
    const unsigned long long ex = __ballot(1);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = (uint32_t)__popcll(ex & ((1ull << lane) - 1ull)), n = (uint32_t)__popcll(ex);
    for (uint32_t q = rank; q < 57u; q += n) {
        const uint32_t va = 227u - 4u * q < 4u ? 227u - 4u * q : 4u;                       // words of this quad in phases A and B (3 for the last quad)
        const uint32_t vc = q <= 42u ? (170u - 4u * q < 4u ? 170u - 4u * q : 4u) : 0u;     // ... in phase C (2 for quad 42, none beyond)
        const MtQuad a = *(const MtQuad BT_GAS *)(old + 4u * q);
        const uint32_t a4 = old[4u * q + 4u];
        const MtQuad b = *(const MtQuadU BT_GAS *)(old + 4u * q + 397u);
        const MtQuad c = *(const MtQuadU BT_GAS *)(old + 4u * q + 227u);
        const uint32_t c4 = old[4u * q + 231u];
        MtQuad d = {0u, 0u, 0u, 0u};
        uint32_t d4 = 0, o0 = 0, o1 = 0, o397 = 0;
        if (vc) {
            d = *(const MtQuadU BT_GAS *)(old + 4u * q + 454u);
            d4 = old[4u * q + 458u];
        }
        if (q == 42u) {   // word 623 reads new[0]
            o0 = old[0];
            o1 = old[1];
            o397 = old[397];
        }
        __builtin_amdgcn_sched_barrier(0);   // (all requests leave before the first word is used)
        MtQuad za, zb, zc;
        za.x = mt_twist(a.x, a.y, b.x);
        za.y = mt_twist(a.y, a.z, b.y);
        za.z = mt_twist(a.z, a.w, b.z);
        za.w = mt_twist(a.w, a4, b.w);
        zb.x = mt_twist(c.x, c.y, za.x);
        zb.y = mt_twist(c.y, c.z, za.y);
        zb.z = mt_twist(c.z, c.w, za.z);
        zb.w = mt_twist(c.w, c4, za.w);
        const uint32_t new0 = mt_twist(o0, o1, o397);
        zc.x = mt_twist(d.x, d.y, zb.x);
        zc.y = mt_twist(d.y, q == 42u ? new0 : d.z, zb.y);   // quad 42, word 1 is x[623]
        zc.z = mt_twist(d.z, d.w, zb.z);
        zc.w = mt_twist(d.w, d4, zb.w);
        if (va == 4u) {
            *(MtQuad BT_GAS *)(nw + 4u * q) = za;
            *(MtQuadU BT_GAS *)(nw + 4u * q + 227u) = zb;
        } else {   // the last quad: three words
            nw[4u * q] = za.x, nw[4u * q + 1u] = za.y, nw[4u * q + 2u] = za.z;
            nw[4u * q + 227u] = zb.x, nw[4u * q + 228u] = zb.y, nw[4u * q + 229u] = zb.z;
        }
        if (vc == 4u) *(MtQuadU BT_GAS *)(nw + 4u * q + 454u) = zc;
        else if (vc == 2u) nw[4u * q + 454u] = zc.x, nw[4u * q + 455u] = zc.y;
    }
    // the block's words were written by other lanes than the one that reads them: same CU, same L1 — visible once the stores are acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// the lanes with `want` set get the block after their current one twisted into their other buffer, one distinct state at a time (the lockstep copies of a
// narrow tile's group share a state: once for all of them), every active lane helping
// (a function of its own: it runs once per 624 words of a lane, and its body would otherwise sit at every refill site of the sweep)
__device__ static __noinline__ void mt_twist_for(bool want, uint32_t BT_GAS *st, uint32_t cur) {
    unsigned long long todo = __ballot(want);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        const uint64_t p = (uint64_t)(uintptr_t)st;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)p, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(p >> 32), src);
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cur, src);
        uint32_t BT_GAS *base = (uint32_t BT_GAS *)(uintptr_t)(((uint64_t)hi << 32) | lo);
        mt_twist_block(base + (c ? MT_BUF : 0u), base + (c ? 0u : MT_BUF));
        todo &= ~__ballot(want && (uint64_t)(uintptr_t)st == (((uint64_t)hi << 32) | lo));
    }
}
#endif
template <class RP>   // RP: pointer-like (operator[](uint32_t) -> uint32_t&) to the ring block
struct MtRingT {
    uint32_t BT_GAS *st;
    RP ring;                   // element i of the ring block at ring[i] (LDS when the cluster's hot arrays are resident, else HBM)
    uint32_t cap;              // power of two, 8 .. 64
    uint32_t pos, head, avail; // pos: multiple of four
#ifdef BT_MT_BLOCK
    uint32_t flags = 0;        // MT_F_CUR: the buffer that holds the current block; MT_F_READY: the other buffer holds the next block
#endif
#ifdef BT_DIAG_FAKE_MT
    uint32_t fake_ctr = 0;
#endif
#if defined(BT_MT_BLOCK) && defined(__HIP_DEVICE_COMPILE__) && !defined(BT_DIAG_FAKE_MT)
    __device__ inline uint32_t BT_GAS *cur_buf() const { return st + ((flags & MT_F_CUR) ? MT_BUF : 0u); }
    // lanes with `at_end` set have read their whole block: the next one is made if it was not made ahead, then the buffers change roles
    __device__ inline void next_block(bool at_end) {
        mt_twist_for(at_end && !(flags & MT_F_READY), st, (flags & MT_F_CUR) ? 1u : 0u);
        if (at_end) {
            flags = (flags ^ MT_F_CUR) & ~MT_F_READY;
            pos = 0;
        }
    }
    // at a convergent point: lanes past MT_AHEAD words of their block get the next block made now
    __device__ inline void twist_ahead() {
        const bool want = pos >= MT_AHEAD && !(flags & MT_F_READY);
        if (bt_wave_any(want)) {
            mt_twist_for(want, st, (flags & MT_F_CUR) ? 1u : 0u);
            if (want) flags |= MT_F_READY;
        }
    }
    __device__ inline void chunk(bool go) {
        if (bt_wave_any(go && pos == MT_N)) next_block(go && pos == MT_N);
        if (go) {
            const MtQuad z = *(const MtQuad BT_GAS *)(cur_buf() + pos);
            const uint32_t w = (head + avail) & (cap - 1u);   // a multiple of four: as many words were produced before
            ring[w] = mt_temper(z.x);
            ring[w + 1u] = mt_temper(z.y);
            ring[w + 2u] = mt_temper(z.z);
            ring[w + 3u] = mt_temper(z.w);
            pos += 4u;
            avail += 4u;
        }
    }
    template <unsigned NB>
    __device__ inline void batch(uint32_t want) {
        // up to NB quads of the current block with all loads in flight (a batch does not cross the end of a block: what is missing follows chunk by chunk)
        if (bt_wave_any(avail < want && pos == MT_N)) next_block(avail < want && pos == MT_N);
        uint32_t nc = avail < want ? (want - avail + 3u) >> 2 : 0u;
        nc = nc < NB ? nc : NB;
        const uint32_t left = (MT_N - pos) >> 2;
        nc = nc < left ? nc : left;
        MtQuad z[NB];
        const uint32_t BT_GAS *b = cur_buf() + pos;
#pragma unroll
        for (unsigned j = 0; j < NB; ++j)
            if (j < nc) z[j] = *(const MtQuad BT_GAS *)(b + 4u * j);
        bt_sched_fence();
#pragma unroll
        for (unsigned j = 0; j < NB; ++j)
            if (j < nc) {
                const uint32_t w = (head + avail) & (cap - 1u);
                ring[w] = mt_temper(z[j].x);
                ring[w + 1u] = mt_temper(z[j].y);
                ring[w + 2u] = mt_temper(z[j].z);
                ring[w + 3u] = mt_temper(z[j].w);
                avail += 4u;
                pos += 4u;
            }
    }
    __device__ inline void fill_to(uint32_t want) {
        while (bt_wave_any(avail < want)) chunk(avail < want);
    }
    template <unsigned NB>
    __device__ inline void fill_batched(uint32_t want) {
        if (bt_wave_any(avail < want)) batch<NB>(want);
        fill_to(want);
    }
    __device__ inline void topup() {
        twist_ahead();
        fill_batched<4>(cap - 3u);
    }
#else
    // one chunk for the lanes with `go` set
    BT_HD void chunk(bool go) {
#ifdef BT_DIAG_FAKE_MT
        if (go) {
            const uint32_t salt = (uint32_t)(uintptr_t)st * 2654435761u + fake_ctr;
            const uint32_t w = (head + avail) & (cap - 1u);
            for (uint32_t k = 0; k < 4u; ++k) ring[w + k] = bt_fake_word(salt + pos + k);
            fake_ctr += 0x9e3779b9u;
            pos = pos + 4u == MT_N ? 0u : pos + 4u;
            avail += 4u;
        }
        return;
#endif
        if (go) {   // (only the lanes that produce send their four requests)
            const uint32_t p = pos, q = p + MT_M < MT_N ? p + MT_M : p + MT_M - MT_N;
            const MtQuad a = *(const MtQuad BT_GAS *)(st + p);
            const uint32_t a4 = st[p + 4u];
            const MtQuad b = *(const MtQuadU BT_GAS *)(st + q);
            bt_sched_fence();   // (all three requests leave before the first word is used: one round trip, not two)
            MtQuad z;
            z.x = mt_twist(a.x, a.y, b.x);
            z.y = mt_twist(a.y, a.z, b.y);
            z.z = mt_twist(a.z, a.w, b.z);
            z.w = mt_twist(a.w, a4, b.w);
            *(MtQuad BT_GAS *)(st + p) = z;
            if (p == 0) *(MtQuad BT_GAS *)(st + MT_N) = z;
            const uint32_t w = (head + avail) & (cap - 1u);   // a multiple of four: as many words were produced before
            ring[w] = mt_temper(z.x);
            ring[w + 1u] = mt_temper(z.y);
            ring[w + 2u] = mt_temper(z.z);
            ring[w + 3u] = mt_temper(z.w);
            pos = p + 4u == MT_N ? 0u : p + 4u;
            avail += 4u;
        }
    }
    // up to NB chunks per lane with ALL their loads in flight before the first new word is computed: one memory round trip for the batch instead of
    // one per chunk (chunks at most a few apart never read each other's words: the recurrence reaches 397 words = 99 chunks ahead, and the mirror is
    // read by chunks 620 and 224 .. 226 only, each 56 or more chunks away from chunk 0 that writes it — or right before it, whose loads come first)
    template <unsigned NB>
    BT_HD void batch(uint32_t want) {
#ifdef BT_DIAG_FAKE_MT
        for (unsigned j = 0; j < NB; ++j) chunk(avail < want);
        return;
#endif
        uint32_t nc = avail < want ? (want - avail + 3u) >> 2 : 0u;
        nc = nc < NB ? nc : NB;
        MtQuad a[NB], b[NB];
        uint32_t a4[NB], pp[NB];
        uint32_t p = pos;
#pragma unroll
        for (unsigned j = 0; j < NB; ++j) {
            pp[j] = p;
            if (j < nc) {
                const uint32_t q = p + MT_M < MT_N ? p + MT_M : p + MT_M - MT_N;
                a[j] = *(const MtQuad BT_GAS *)(st + p);
                a4[j] = st[p + 4u];
                b[j] = *(const MtQuadU BT_GAS *)(st + q);
            }
            p = p + 4u == MT_N ? 0u : p + 4u;
        }
        bt_sched_fence();
#pragma unroll
        for (unsigned j = 0; j < NB; ++j)
            if (j < nc) {
                MtQuad z;
                z.x = mt_twist(a[j].x, a[j].y, b[j].x);
                z.y = mt_twist(a[j].y, a[j].z, b[j].y);
                z.z = mt_twist(a[j].z, a[j].w, b[j].z);
                z.w = mt_twist(a[j].w, a4[j], b[j].w);
                *(MtQuad BT_GAS *)(st + pp[j]) = z;
                if (pp[j] == 0) *(MtQuad BT_GAS *)(st + MT_N) = z;
                const uint32_t w = (head + avail) & (cap - 1u);
                ring[w] = mt_temper(z.x);
                ring[w + 1u] = mt_temper(z.y);
                ring[w + 2u] = mt_temper(z.z);
                ring[w + 3u] = mt_temper(z.w);
                avail += 4u;
                pos = pp[j] + 4u == MT_N ? 0u : pp[j] + 4u;
            }
    }
    // chunks until every lane holds at least `want` words (want <= cap - 3), the wavefront looping while any of its lanes is short
    BT_HD void fill_to(uint32_t want) {
        while (bt_wave_any(avail < want)) chunk(avail < want);
    }
    // the same with the first chunks of every lane as one batch (the top-ups at the start of a visit)
    template <unsigned NB>
    BT_HD void fill_batched(uint32_t want) {
        if (bt_wave_any(avail < want)) batch<NB>(want);
        fill_to(want);
    }
    BT_HD void topup() { fill_batched<4>(cap - 3u); }
#endif
    BT_HD void need(uint32_t n) {   // n <= 4
        if (avail < n) fill_to(cap - 3u < 16u ? cap - 3u : 16u);
    }
    BT_HD uint32_t next() {
        need(1);
        const uint32_t w = ring[head];
        head = (head + 1u) & (cap - 1u);
        avail -= 1u;
        return w;
    }
    template <unsigned N>
    BT_HD void next_n(uint32_t (&out)[N]) {
        need(N);
#pragma unroll
        for (uint32_t k = 0; k < N; ++k) out[k] = ring[(head + k) & (cap - 1u)];
        head = (head + N) & (cap - 1u);
        avail -= N;
    }
};
typedef MtRingT<SPtrF<uint32_t, 1>> MtRing;
template <class RP>
BT_HD MtRingT<RP> mt_ring_open_as(uint32_t *st, RP ring, uint32_t cap) {
    MtRingT<RP> m;
    m.st = (uint32_t BT_GAS *)st;
    m.ring = ring;
    m.cap = cap;
    m.pos = m.ring[cap];
    m.head = m.ring[cap + 1];
    m.avail = m.ring[cap + 2];
#ifdef BT_MT_BLOCK
    m.flags = m.pos & (MT_F_CUR | MT_F_READY);
    m.pos &= 0xFFFFu;
#endif
    return m;
}
template <class RP>
BT_HD MtRing mt_ring_open(uint32_t *st, RP ring, uint32_t cap) {
    MtRing m;
    m.st = (uint32_t BT_GAS *)st;
    m.ring = SPtrF<uint32_t, 1>{ring.base, ring.off, ring.sh};
    m.cap = cap;
    m.pos = m.ring[cap];
    m.head = m.ring[cap + 1];
    m.avail = m.ring[cap + 2];
#ifdef BT_MT_BLOCK
    m.flags = m.pos & (MT_F_CUR | MT_F_READY);
    m.pos &= 0xFFFFu;
#endif
    return m;
}
template <class RP>
BT_HD void mt_close(const MtRingT<RP> &m) {
#ifdef BT_MT_BLOCK
    m.ring[m.cap] = m.pos | m.flags;
#else
    m.ring[m.cap] = m.pos;
#endif
    m.ring[m.cap + 1] = m.head;
    m.ring[m.cap + 2] = m.avail;
}
template <class RP>
BT_HD void mt_ring_seed(uint32_t *st, RP ring, uint32_t cap, uint32_t seed) {
    mt_seed(st, seed);
    for (unsigned i = 0; i < MT_MIRROR; ++i) st[MT_N + i] = st[i];   // (a ring generator keeps its position in the ring block, not at st[MT_N])
#ifdef BT_MT_BLOCK
    ring[cap] = MT_N;   // the seeded words are the block BEFORE the first one: buffer 0 is "read to the end", the first refill twists (block form)
#else
    ring[cap] = 0;
#endif
    ring[cap + 1] = 0;
    ring[cap + 2] = 0;
}

template <class G>
BT_HD uint32_t mt_next(G &m) { return m.next(); }

// generate_canonical<double, 53>(mt19937): two draws
template <class G>
BT_HD double rng_canonical(G &m) {
    uint32_t w[2];
    m.template next_n<2>(w);
    double sum = (double)w[0];
    sum += (double)w[1] * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = 0.99999999999999988897769753748434595763683319091796875;   // nextafter(1, 0)
    return ret;
}

// two consecutive generate_canonical<double, 53> draws (four words): the polar method of normal_distribution draws its two uniforms
// back to back, and every memory round trip of a rejection loop is paid by the whole wavefront until its last lane accepts
template <class G>
BT_HD void rng_canonical2(G &m, double &first, double &second) {
    uint32_t w[4];
    m.template next_n<4>(w);
    const double top = 0.99999999999999988897769753748434595763683319091796875;   // nextafter(1, 0)
    double s0 = (double)w[0];
    s0 += (double)w[1] * 4294967296.0;
    first = s0 / 18446744073709551616.0;
    if (first >= 1.0) first = top;
    double s1 = (double)w[2];
    s1 += (double)w[3] * 4294967296.0;
    second = s1 / 18446744073709551616.0;
    if (second >= 1.0) second = top;
}

// uniform_int_distribution<>(0, b) for b < 2^32 - 1: range = b + 1
template <class G>
BT_HD uint32_t rng_uniform_int(G &st, uint32_t range) {
    uint64_t product = (uint64_t)mt_next(st) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)mt_next(st) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

template <class G>
BT_HD bool rng_bernoulli(G &st, double p) { return rng_canonical(st) < p; }

// std::shuffle over a uint32 array in caller memory
template <class G, typename Arr>
BT_HD void rng_shuffle_u32(G &st, Arr a, uint32_t n) {
    if (n == 0) return;
    const uint64_t urngrange = 0xFFFFFFFFull;
    if (urngrange / n >= n) {
        uint32_t i = 1;
        if ((n % 2u) == 0) {
            uint32_t j = rng_uniform_int(st, 2);
            uint32_t t = a[i]; a[i] = a[j]; a[j] = t;
            ++i;
        }
        while (i != n) {
            const uint32_t swap_range = i + 1;
            const uint32_t b1 = swap_range + 1;
            uint32_t x = rng_uniform_int(st, swap_range * b1);
            uint32_t p0 = x / b1, p1 = x % b1;
            uint32_t t = a[i]; a[i] = a[p0]; a[p0] = t;
            ++i;
            t = a[i]; a[i] = a[p1]; a[p1] = t;
            ++i;
        }
        return;
    }
    for (uint32_t i = 1; i < n; ++i) {
        uint32_t j = rng_uniform_int(st, i + 1);
        uint32_t t = a[i]; a[i] = a[j]; a[j] = t;
    }
}

// normal_distribution<double>(0,1) + gamma_distribution<double>; `nd` holds {saved value, saved_available flag}
struct NormalState {   // references to wherever the caller keeps the two fields (generic pointers: LDS or HBM)
    double *saved;
    uint32_t *available;
};

template <class G>
BT_HD double rng_normal(G &st, NormalState nd) {
    double ret;
    if (*nd.available) {
        *nd.available = 0;
        ret = *nd.saved;
    } else {
        double x, y, r2;
        do {
            double u1, u2;
            rng_canonical2(st, u1, u2);
            x = 2.0 * u1 - 1.0;
            y = 2.0 * u2 - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = sqrt(-2 * bt_log(r2) / r2);
        *nd.saved = x * mult;
        *nd.available = 1;
        ret = y * mult;
    }
    ret = ret * 1.0 + 0.0;
    return ret;
}

// single-precision natural logarithm for the screening test below (one hardware instruction on the device)
BT_HD float bt_fast_logf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __logf(x);
#else
    return logf(x);
#endif
}
// Marsaglia-Tsang's second acceptance test, log(u) > 0.5 n^2 + a1 (1 - v + log v), decides nothing but a branch.  It is screened in single
// precision with an error bound: only when the two sides are closer than the bound (about once in 10^4 evaluations) are the two double-precision
// logarithms evaluated.  The decision — and with it the draw stream — is the double-precision one in every case: a wavefront evaluates this
// test in nearly every iteration (some lane fails the cheap first test), so the two out-of-line logarithms were a fifth of a gamma draw.
// returns true when the candidate is REJECTED by the second test
BT_HD bool gamma_second_test_rejects(double u, double n, double v, double a1) {
    const float lu = bt_fast_logf((float)u), lv = bt_fast_logf((float)v);
    const double diff = (double)lu - (0.5 * n * n + a1 * (1.0 - v + (double)lv));
    // error of the screen: the rounding of u and v to single precision (relative 6e-8 -> 6e-8 absolute in the logarithm) plus the logarithm's own
    // (a few units in the last place; near 1 an absolute 1e-6 at most).  Bound taken 8x wider; comparisons are false on NaN / infinities -> exact path.
    const double err = 4e-5 * ((1.0 + fabs((double)lu)) + a1 * (1.0 + fabs((double)lv)));
    if (diff > err) return true;
    if (diff < -err) return false;
    return bt_log(u) > (0.5 * n * n + a1 * (1.0 - v + bt_log(v)));
}

// gamma_distribution(alpha, beta) for alpha >= 1 with a2 = 1 / sqrt(9 (alpha - 1/3)) handed in (callers whose alpha is a small integer — an
// observation count + 1 — read it from a table built with the same two IEEE operations: bt_gibbs.hip, GParams::gamma_a2)
// the accepted v of Marsaglia-Tsang's loop for a1 = alpha - 1/3 (alpha >= 1), a2 = 1 / sqrt(9 a1)
template <class G>
BT_HD double rng_gamma_v(G &st, NormalState nd, double a1, double a2) {
    double u, v, n;
    do {
        do {
            n = rng_normal(st, nd);
            v = 1.0 + a2 * n;
        } while (v <= 0.0);
        v = v * v * v;
        u = rng_canonical(st);
    } while (u > 1.0 - 0.0331 * n * n * n * n && gamma_second_test_rejects(u, n, v, a1));
    return v;
}
// gamma_distribution(alpha, beta) for alpha >= 1 with a2 = 1 / sqrt(9 (alpha - 1/3)) handed in (callers whose alpha is a small integer — an
// observation count + 1 — read it from a table built with the same two IEEE operations: bt_gibbs.hip, GParams::gamma_a2)
template <class G>
BT_HD double rng_gamma_ge1(G &st, NormalState nd, double alpha, double a2, double beta) {
    const double a1 = alpha - 1.0 / 3.0;
    const double v = rng_gamma_v(st, nd, a1, a2);
    return a1 * v * beta;
}

template <class G>
BT_HD double rng_gamma(G &st, NormalState nd, double alpha, double beta) {
    const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double a1 = malpha - 1.0 / 3.0;
    const double a2 = 1.0 / sqrt(9.0 * a1);
    const double v = rng_gamma_v(st, nd, a1, a2);
    if (alpha == malpha) return a1 * v * beta;
    double u;
    do u = rng_canonical(st);
    while (u == 0.0);
    return bt_pow(u, 1.0 / alpha) * a1 * v * beta;
}

// ------------------------------------------------------------------------------------------------------------
// std::unordered_set<unsigned> order emulation.  Elements are 0..universe-1; `next` (one word per element) may be
// shared by several sets as long as an element is in at most one of them at a time.
// hdr: [0] bucket count B, [1] head (before_begin.next), [2] size, [3] next_resize
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t US_NONE = 0xFFFFFFFFu, US_BEFORE = 0xFFFFFFFEu;

template <class PT>   // PT: any pointer-like with operator[](uint32_t) -> uint32_t&
struct USetP {
    PT hdr;    // 4 words
    PT bkt;    // `cap` words
    PT next;   // universe words
    // Bucket words stored.  An element e < universe lives in bucket e % B < min(B, universe), so min(uset_bucket_capacity(universe),
    // universe) words suffice whatever the bucket count B is; the loops over "all buckets" stop at min(B, cap).
    uint32_t cap;
};
typedef USetP<SPtr<uint32_t, 1>> USet;

BT_HD uint32_t uset_next_bucket_count(uint32_t min_needed) {   // next entry of libstdc++'s growth chain that is >= min_needed
    const uint32_t chain[15] = {13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753, 42043, 85229, 172933, 351061};
    for (int i = 0; i < 15; ++i)
        if (chain[i] >= min_needed) return chain[i];
    return 351061;
}
// bucket array capacity needed for a set that may hold up to `universe` elements
BT_HD uint32_t uset_bucket_capacity(uint32_t universe) {
    uint32_t b = 13;
    while (b < universe) b = uset_next_bucket_count(2 * b);
    return b < universe ? b : (universe ? universe : 1u);   // words to store: min(largest bucket count, universe)
}

// e % B; elements below the bucket count (every element of a set over fewer than B haplotypes) need no division
BT_HD uint32_t uset_mod(uint32_t e, uint32_t B) { return e < B ? e : e % B; }
template <class PT>
BT_HD void uset_init(USetP<PT> s) {
    s.hdr[0] = 1;
    s.hdr[1] = US_NONE;
    s.hdr[2] = 0;
    s.hdr[3] = 0;
    s.bkt[0] = US_NONE;
}
template <class PT>
BT_HD uint32_t uset_nxt(USetP<PT> s, uint32_t node) { return node == US_BEFORE ? s.hdr[1] : s.next[node]; }
template <class PT>
BT_HD void uset_set_nxt(USetP<PT> s, uint32_t node, uint32_t v) {
    if (node == US_BEFORE) s.hdr[1] = v;
    else s.next[node] = v;
}
template <class PT>
BT_HD void uset_rehash(USetP<PT> s, uint32_t newB) {
    for (uint32_t i = 0, n = newB < s.cap ? newB : s.cap; i < n; ++i) s.bkt[i] = US_NONE;
    uint32_t p = s.hdr[1];
    s.hdr[1] = US_NONE;
    uint32_t bbegin_bkt = 0;
    while (p != US_NONE) {
        uint32_t nx = s.next[p];
        uint32_t b = uset_mod(p, newB);
        if (s.bkt[b] == US_NONE) {
            s.next[p] = s.hdr[1];
            s.hdr[1] = p;
            s.bkt[b] = US_BEFORE;
            if (s.next[p] != US_NONE) s.bkt[bbegin_bkt] = p;
            bbegin_bkt = b;
        } else {
            uint32_t prev = s.bkt[b];
            s.next[p] = uset_nxt(s, prev);
            uset_set_nxt(s, prev, p);
        }
        p = nx;
    }
    s.hdr[0] = newB;
    s.hdr[3] = newB;   // floor(B * max_load_factor 1.0)
}
// insert an element that is not in the set
template <class PT>
BT_HD void uset_insert(USetP<PT> s, uint32_t e) {
    uint32_t B = s.hdr[0], size = s.hdr[2];
    if (size + 1 > s.hdr[3]) {
        uint32_t min_bkts = size + 1;
        if (s.hdr[3] == 0 && min_bkts < 11) min_bkts = 11;
        if (min_bkts >= B) {
            uint32_t want = min_bkts + 1 > 2 * B ? min_bkts + 1 : 2 * B;
            uset_rehash(s, uset_next_bucket_count(want));
            B = s.hdr[0];
        } else
            s.hdr[3] = B;
    }
    uint32_t b = uset_mod(e, B);
    if (s.bkt[b] != US_NONE) {
        uint32_t prev = s.bkt[b];
        s.next[e] = uset_nxt(s, prev);
        uset_set_nxt(s, prev, e);
    } else {
        s.next[e] = s.hdr[1];
        s.hdr[1] = e;
        if (s.next[e] != US_NONE) s.bkt[uset_mod(s.next[e], B)] = e;
        s.bkt[b] = US_BEFORE;
    }
    s.hdr[2] = size + 1;
}
// erase an element that is in the set
template <class PT>
BT_HD void uset_erase(USetP<PT> s, uint32_t e) {
    const uint32_t B = s.hdr[0];
    const uint32_t b = uset_mod(e, B);
    uint32_t prev = s.bkt[b];
    while (uset_nxt(s, prev) != e) prev = uset_nxt(s, prev);
    const uint32_t nn = s.next[e];
    if (prev == s.bkt[b]) {
        if (nn == US_NONE || uset_mod(nn, B) != b) {
            if (nn != US_NONE) s.bkt[uset_mod(nn, B)] = s.bkt[b];
            if (s.bkt[b] == US_BEFORE) s.hdr[1] = nn;
            s.bkt[b] = US_NONE;
        }
    } else if (nn != US_NONE) {
        uint32_t nb = uset_mod(nn, B);
        if (nb != b) s.bkt[nb] = prev;
    }
    uset_set_nxt(s, prev, nn);
    s.hdr[2] -= 1;
}
template <class PT>
BT_HD void uset_clear(USetP<PT> s) {
    const uint32_t B = s.hdr[0];
    for (uint32_t i = 0, n = B < s.cap ? B : s.cap; i < n; ++i) s.bkt[i] = US_NONE;
    s.hdr[1] = US_NONE;
    s.hdr[2] = 0;
}
template <class PT>
BT_HD uint32_t uset_begin(USetP<PT> s) { return s.hdr[1]; }
template <class PT>
BT_HD uint32_t uset_size(USetP<PT> s) { return s.hdr[2]; }

}  // namespace bt
