#include "KmerHashOrder.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <random>

namespace bthost {

// libstdc++-v3/libsupc++/hash_bytes.cc, 64-bit size_t: _Hash_bytes(ptr, len, seed = 0xc70f6907)
static uint64_t hashBytes(const unsigned char *buf, size_t len) {
    const uint64_t mul = (0xc6a4a793ULL << 32) + 0x5bd1e995ULL;
    auto shift_mix = [](uint64_t v) { return v ^ (v >> 47); };
    uint64_t hash = 0xc70f6907ULL ^ (len * mul);
    const size_t len_aligned = len & ~(size_t)7;
    for (size_t at = 0; at < len_aligned; at += 8) {
        uint64_t w;
        std::memcpy(&w, buf + at, 8);
        const uint64_t data = shift_mix(w * mul) * mul;
        hash ^= data;
        hash *= mul;
    }
    if (len & 7) {
        uint64_t data = 0;
        for (int i = (int)(len & 7) - 1; i >= 0; i--) data = (data << 8) + buf[len_aligned + i];
        hash ^= data;
        hash *= mul;
    }
    hash = shift_mix(hash) * mul;
    return shift_mix(hash);
}

uint64_t bitsetHash(uint64_t lo, uint64_t hi, unsigned kmer_size) {
    unsigned char bytes[16];
    std::memcpy(bytes, &lo, 8);
    std::memcpy(bytes + 8, &hi, 8);
    return hashBytes(bytes, (2 * kmer_size + 7) / 8);
}

std::vector<uint32_t> hybridHashShuffledOrder(const uint64_t *kmers, uint64_t n, unsigned kmer_size, unsigned prng_seed, uint64_t root_hash_size) {
    std::vector<uint64_t> root(n);
    for (uint64_t i = 0; i < n; i++) root[i] = bitsetHash(kmers[2 * i], kmers[2 * i + 1], kmer_size) % root_hash_size;
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {   // root bucket, then BitsetLess (the 2k-bit value, ascending)
        if (root[a] != root[b]) return root[a] < root[b];
        if (kmers[2 * a + 1] != kmers[2 * b + 1]) return kmers[2 * a + 1] < kmers[2 * b + 1];
        return kmers[2 * a] < kmers[2 * b];
    });
    std::mt19937 prng(prng_seed);
    for (uint64_t a = 0; a < n;) {   // std::shuffle of every bucket in root order (empty buckets draw nothing)
        uint64_t b = a + 1;
        while (b < n && root[order[b]] == root[order[a]]) b++;
        std::shuffle(order.begin() + (std::ptrdiff_t)a, order.begin() + (std::ptrdiff_t)b, prng);
        a = b;
    }
    return order;
}

}  // namespace bthost
