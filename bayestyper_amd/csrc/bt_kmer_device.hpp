// Device-side k-mer primitives for gfx950: 2-bit packing, canonical form, ntHash, Bloom probes.
//
// Reference behaviour restated here (not copied):
//   Nucleotide::ntToBit            include/bayesTyper/Nucleotide.hpp:40-70
//   KmerPair lexicographic lowest  include/bayesTyper/Kmer.tpp:225-255
//   NTP64(kmer,k) / (kmer,k,seed)  external/ntHash/nthash.hpp:262-282, seeds :18-28
//   BloomFilter::insertF/containsF external/ntHash/BloomFilter.hpp:56-66,149-161
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bt {

struct Kmer {            // nucleotide i at bits (2i, 2i+1) of the 128-bit value hi:lo
    uint64_t lo, hi;
};

// ntHash constants (external/ntHash/nthash.hpp:18-28)
constexpr uint64_t NT_SEED_A = 0x3c8bfbb395c60474ULL;
constexpr uint64_t NT_SEED_C = 0x3193c18562a02b4cULL;
constexpr uint64_t NT_SEED_G = 0x20323ed082572324ULL;
constexpr uint64_t NT_SEED_T = 0x295549f54be24456ULL;
constexpr uint64_t NT_MULTISEED = 0x90b45d39fb6da1faULL;
constexpr int NT_MULTISHIFT = 27;
constexpr uint32_t BT_ROUTE_SEED = 1029283129u;   // src/kmerBloom/KmerBloom.cpp:279
constexpr uint32_t BT_NUM_SUB_BLOOMS = 65536u;    // src/kmerBloom/KmerBloom.cpp:206

__host__ __device__ inline uint64_t rol64(uint64_t x, unsigned r) {
    r &= 63u;
    return r ? ((x << r) | (x >> (64u - r))) : x;
}

// reverse the order of the 32 two-bit groups of x
__host__ __device__ inline uint64_t rev2bit64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t y = __brevll(x);
#else
    uint64_t y = x;
    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
    y = ((y >> 2) & 0x3333333333333333ULL) | ((y & 0x3333333333333333ULL) << 2);
    y = ((y >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((y & 0x0F0F0F0F0F0F0F0FULL) << 4);
    y = __builtin_bswap64(y);
#endif
    // bit reversal also swapped the two bits inside every group: swap them back
    return ((y & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((y & 0x5555555555555555ULL) << 1);
}

// 128-bit helpers on Kmer (k <= 64 nucleotides)
__host__ __device__ inline Kmer kmer_mask(Kmer a, unsigned k) {
    unsigned bits = 2u * k;
    if (bits >= 128u) return a;
    if (bits >= 64u) {
        unsigned hb = bits - 64u;
        a.hi = hb ? (a.hi & ((1ULL << hb) - 1ULL)) : 0ULL;
    } else {
        a.hi = 0;
        a.lo &= ((1ULL << bits) - 1ULL);
    }
    return a;
}

// reverse complement of a k-mer: nucleotide i of the result = 3 - nucleotide (k-1-i)
__host__ __device__ inline Kmer kmer_revcomp(Kmer a, unsigned k) {
    // reverse all 64 two-bit groups of the 128-bit value, complement, shift down by 64-k groups
    uint64_t rlo = ~rev2bit64(a.hi);
    uint64_t rhi = ~rev2bit64(a.lo);
    unsigned sh = 2u * (64u - k);   // 0..126
    Kmer r;
    if (sh == 0) { r.lo = rlo; r.hi = rhi; }
    else if (sh < 64u) { r.lo = (rlo >> sh) | (rhi << (64u - sh)); r.hi = rhi >> sh; }
    else if (sh == 64u) { r.lo = rhi; r.hi = 0; }
    else { r.lo = rhi >> (sh - 64u); r.hi = 0; }
    return kmer_mask(r, k);
}

// "a < b" in the reference's lexicographic order: nucleotide 0 first, A<C<G<T
// (include/bayesTyper/Kmer.tpp:225-255).  Position 0 lives in the LOW bits, so compare the
// group-reversed words, low word first.
__host__ __device__ inline bool kmer_lex_less(Kmer a, Kmer b) {
    uint64_t ra = rev2bit64(a.lo), rb = rev2bit64(b.lo);
    if (ra != rb) return ra < rb;
    ra = rev2bit64(a.hi); rb = rev2bit64(b.hi);
    return ra < rb;
}

__host__ __device__ inline Kmer kmer_canonical(Kmer fw, unsigned k) {
    Kmer rc = kmer_revcomp(fw, k);
    // ties return the forward k-mer (Kmer.tpp:253)
    return kmer_lex_less(rc, fw) ? rc : fw;
}

__host__ __device__ inline uint64_t nt_seed(unsigned c) {
    uint64_t lo = (c & 1u) ? NT_SEED_C : NT_SEED_A;
    uint64_t hi = (c & 1u) ? NT_SEED_T : NT_SEED_G;
    return (c & 2u) ? hi : lo;
}

// NTP64(kmer, k): XOR_i rol(seed[c_i], (k-1-i) % 64), evaluated in Horner form
// h <- rol(h,1) ^ seed[c_i]  (identical value; nthash.hpp:262-267 uses the msTab lookup)
__host__ __device__ inline uint64_t nthash64(Kmer a, unsigned k) {
    uint64_t h = 0;
    unsigned n_lo = k < 32u ? k : 32u;
    uint64_t w = a.lo;
    for (unsigned i = 0; i < n_lo; ++i) {
        h = rol64(h, 1) ^ nt_seed((unsigned)(w & 3u));
        w >>= 2;
    }
    w = a.hi;
    for (unsigned i = 32u; i < k; ++i) {
        h = rol64(h, 1) ^ nt_seed((unsigned)(w & 3u));
        w >>= 2;
    }
    return h;
}

// NTP64(kmer, k, seed) given the unseeded value (nthash.hpp:275-282):
//   hVal *= seed ^ k * multiSeed;  hVal ^= hVal >> multiShift     ('*' binds tighter than '^')
__host__ __device__ inline uint64_t nthash64_seeded(uint64_t h, unsigned k, uint32_t seed) {
    h *= ((uint64_t)seed ^ ((uint64_t)k * NT_MULTISEED));
    h ^= h >> NT_MULTISHIFT;
    return h;
}

// x % m with a precomputed reciprocal M = floor((2^64 - 1) / m) (m >= 1)
struct FastMod {
    uint64_t m, M;
};
__host__ inline FastMod make_fastmod(uint64_t m) {
    FastMod f;
    f.m = m;
    f.M = m ? (~0ULL) / m : 0;
    return f;
}
__device__ inline uint64_t fastmod(uint64_t x, FastMod f) {
    uint64_t q = __umul64hi(x, f.M);
    uint64_t r = x - q * f.m;
    // q underestimates floor(x/m) by at most 2
    if (r >= f.m) r -= f.m;
    if (r >= f.m) r -= f.m;
    return r;
}

// Flat description of a KmerBloom / ThreadedKmerBloom living in HBM.
// Sub-filter s occupies bytes [s*stride, s*stride + (num_bits+7)/8); bit b of a sub-filter is
// byte b/8, mask 1 << (7 - b%8) (BloomFilter.hpp:59).  stride is a multiple of 4 so that
// 32-bit atomics stay inside one sub-filter.
struct BloomView {
    uint32_t *words;      // device
    uint64_t stride;      // bytes per sub-filter (multiple of 4)
    FastMod bits;         // num_bloom_bits of each (sub-)filter
    uint32_t num_hashes;
    uint32_t num_sub;     // 1 or 65536
    uint32_t k;
};

__device__ inline uint64_t bloom_probe_pos(uint64_t h, unsigned i, const BloomView &b) {
    if (i == 0) return fastmod(h, b.bits);
    // BloomFilter.hpp:60-63 / :153-156
    uint64_t mh = h * ((uint64_t)i ^ ((uint64_t)b.k * NT_MULTISEED));
    mh ^= mh >> NT_MULTISHIFT;
    return fastmod(mh, b.bits);
}

__device__ inline uint64_t bloom_sub_base(uint64_t h, const BloomView &b) {
    if (b.num_sub == 1u) return 0;
    uint64_t route = nthash64_seeded(h, b.k, BT_ROUTE_SEED) & (uint64_t)(BT_NUM_SUB_BLOOMS - 1u);
    return route * b.stride;
}

__device__ inline bool bloom_contains(uint64_t h, const BloomView &b) {
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(b.words) + bloom_sub_base(h, b);
    for (unsigned i = 0; i < b.num_hashes; ++i) {
        uint64_t pos = bloom_probe_pos(h, i, b);
        uint8_t byte = bytes[pos >> 3];
        if ((byte & (1u << (7u - (unsigned)(pos & 7u)))) == 0) return false;   // early exit as containsF
    }
    return true;
}

__device__ inline void bloom_insert(uint64_t h, const BloomView &b) {
    uint64_t base = bloom_sub_base(h, b);
    for (unsigned i = 0; i < b.num_hashes; ++i) {
        uint64_t pos = bloom_probe_pos(h, i, b);
        uint64_t byte = base + (pos >> 3);
        uint32_t bit = (uint32_t)((byte & 3u) * 8u + (7u - (unsigned)(pos & 7u)));   // little-endian word
        atomicOr(&b.words[byte >> 2], 1u << bit);
    }
}

// ASCII -> 2-bit code, -1 for anything outside ACGTacgt (Nucleotide.hpp:44-66)
__host__ __device__ inline int nt_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

}  // namespace bt
