// Iteration order of the reference's k-mer hash after KmerHash::shuffle — what decides WHICH parameter k-mers `bayesTyper cluster` writes
// to parameter_kmers.fa.gz (main.cpp:326-341: shuffle(seed), then the first <= 10^6 k-mers with value true in iteration order).
//
// HybridHash (include/bayesTyper/HybridHash.tpp:41-202): root bucket = std::hash<std::bitset<2k>>(kmer) % 4^12; a bucket is a
// PartialSortedLinearMap that KmerHash::addKmer keeps sorted (add_sorted = true, BitsetLess: most significant bit first); shuffle(seed)
// runs std::shuffle over every bucket in root order with ONE std::mt19937(seed) (LinearMap.tpp:203-208); iteration = root order, then
// bucket order.  std::hash<std::bitset<N>> is libstdc++'s _Hash_bytes over the first (N+7)/8 bytes of the bitset's words (restated
// below, checked against the library in tests/test_host_cli_cpu.py).
#pragma once
#include <cstdint>
#include <vector>

namespace bthost {

// libstdc++ std::hash<std::bitset<2k>> of a k-mer given as {lo, hi} (nucleotide i at bits 2i, 2i+1)
uint64_t bitsetHash(uint64_t lo, uint64_t hi, unsigned kmer_size);
// permutation: order[j] = index of the j-th k-mer of the iteration (kmers: n pairs {lo, hi})
std::vector<uint32_t> hybridHashShuffledOrder(const uint64_t *kmers, uint64_t n, unsigned kmer_size, unsigned prng_seed, uint64_t root_hash_size = 16777216);

}  // namespace bthost
