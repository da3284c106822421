// ORACLE TOOLING — builds oracle/_ref/libbtref.so: a thin extern "C" driver over the REFERENCE's own,
// unmodified translation units (compiled from /root/reference where they lie; see oracle/Makefile).
// Only the Boost-free part of the hot path can be built in this image (no Boost headers/libs, and no
// stand-ins are written for them): ntHash, BloomFilter, KmerBloom/ThreadedKmerBloom, the KMC reader,
// Kmer/Nucleotide templates, KmerCounts, KmerStats, NegativeBinomialDistribution, DiscreteSampler,
// SparsityEstimator, CountAllocation, Utils.  It is used by tests/ to pin the restatement in
// oracle_kmer.cpp / oracle_gibbs.cpp; it never ships and is never called by the product.
#include <bitset>
#include <unordered_map>
#include <unordered_set>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "KmerBloom.hpp"
#include "kmc_api/kmc_file.h"
#include "Utils.hpp"
#include "Kmer.hpp"
#include "Nucleotide.hpp"
#include "KmerCounts.hpp"
#include "KmerStats.hpp"
#include "NegativeBinomialDistribution.hpp"
#include "DiscreteSampler.hpp"
#include "SparsityEstimator.hpp"
#include "CountAllocation.hpp"
#include "HybridHash.hpp"

static const unsigned K = BT_KMER_SIZE;

extern "C" {

unsigned ref_kmer_size() { return K; }
uint64_t ref_ntp64(const char *kmer) { return NTP64(kmer, K); }
uint64_t ref_ntp64_seed(const char *kmer, unsigned seed) { return NTP64(kmer, K, seed); }

void ref_bloom_sizing(uint64_t n, float fpr, uint64_t *bits, unsigned *hashes) {
    *bits = KmerBloom<BT_KMER_SIZE>::calcOptNumBloomBits(fpr, n);
    *hashes = KmerBloom<BT_KMER_SIZE>::calcOptNumHashes(*bits, n);
}
void *ref_kmerbloom_new(uint64_t n, float fpr) { return new KmerBloom<BT_KMER_SIZE>(n, fpr); }
void *ref_kmerbloom_load(const char *prefix) { return new KmerBloom<BT_KMER_SIZE>(std::string(prefix)); }
void ref_kmerbloom_free(void *h) { delete (KmerBloom<BT_KMER_SIZE> *)h; }
void ref_kmerbloom_save(void *h, const char *prefix) { ((KmerBloom<BT_KMER_SIZE> *)h)->save(std::string(prefix)); }
void ref_kmerbloom_add(void *h, const char *kmers, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) ((KmerBloom<BT_KMER_SIZE> *)h)->addKmer(kmers + i * K);
}
void ref_kmerbloom_lookup(void *h, const char *kmers, uint64_t n, uint8_t *hits) {
    for (uint64_t i = 0; i < n; i++) hits[i] = ((KmerBloom<BT_KMER_SIZE> *)h)->lookup(kmers + i * K) ? 1 : 0;
}
// the bitset overloads (bitToNt path, KmerBloom.cpp:98-130,180-200)
void ref_kmerbloom_lookup_packed(void *h, const uint64_t *packed, uint64_t n, uint8_t *hits) {
    for (uint64_t i = 0; i < n; i++) {
        std::bitset<BT_KMER_SIZE * 2> b;
        for (unsigned j = 0; j < 2 * K; j++) b[j] = (packed[2 * i + j / 64] >> (j % 64)) & 1;
        hits[i] = ((KmerBloom<BT_KMER_SIZE> *)h)->lookup(b) ? 1 : 0;
    }
}
void *ref_tbloom_new(uint64_t n, float fpr) { return new ThreadedKmerBloom<BT_KMER_SIZE>(n, fpr); }
void ref_tbloom_free(void *h) { delete (ThreadedKmerBloom<BT_KMER_SIZE> *)h; }
void ref_tbloom_add(void *h, const char *kmers, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) ((ThreadedKmerBloom<BT_KMER_SIZE> *)h)->addKmer(kmers + i * K);
}
void ref_tbloom_lookup(void *h, const char *kmers, uint64_t n, uint8_t *hits) {
    for (uint64_t i = 0; i < n; i++) hits[i] = ((ThreadedKmerBloom<BT_KMER_SIZE> *)h)->lookup(kmers + i * K) ? 1 : 0;
}

// KmerPair sliding window (Kmer.tpp) -> packed lowest k-mer per position
void ref_kmers_from_sequence(const char *seq, uint64_t len, uint64_t *kmers, uint8_t *valid) {
    KmerPair<BT_KMER_SIZE> kp;
    for (uint64_t i = 0; i < len; i++) {
        kmers[2 * i] = kmers[2 * i + 1] = 0;
        valid[i] = 0;
        if (kp.move(Nucleotide::ntToBit<1>(seq[i]))) {
            auto low = kp.getLexicographicalLowestKmer();
            for (unsigned j = 0; j < 2 * K; j++)
                if (low[j]) kmers[2 * i + j / 64] |= (1ULL << (j % 64));
            valid[i] = 1;
        }
    }
}

// CKMCFile listing (kmc_file.cpp) exactly as KmerCounter.cpp:431-505 drives it
int64_t ref_kmc_total(const char *prefix, unsigned *k, unsigned *mode, unsigned *counter_size, unsigned *p) {
    CKMCFile f;
    if (!f.OpenForListing(prefix)) return -1;
    CKMCFileInfo info;
    f.Info(info);
    *k = info.kmer_length;
    *mode = info.mode;
    *counter_size = info.counter_size;
    *p = info.lut_prefix_length;
    return (int64_t)info.total_kmers;
}
int64_t ref_kmc_list(const char *prefix, char *kmers, uint32_t *counts, uint64_t max_n) {
    CKMCFile f;
    if (!f.OpenForListing(prefix)) return -1;
    CKmerAPI km(K);
    uint32 c;
    uint64_t n = 0;
    while (f.ReadNextKmer(km, c)) {
        if (n >= max_n) return -2;
        for (unsigned j = 0; j < K; j++) kmers[n * K + j] = km.get_asci_symbol(j);
        counts[n] = c;
        n++;
    }
    return (int64_t)n;
}

// ObservedKmerCounts<30> (KmerCounts.cpp)
void *ref_kc_new() { return new ObservedKmerCounts<30>(); }
void ref_kc_free(void *h) { delete (ObservedKmerCounts<30> *)h; }
void ref_kc_add_intercluster(void *h, int is_decoy, unsigned fp, unsigned mp) {
    std::vector<Utils::Ploidy> gp = {static_cast<Utils::Ploidy>(fp), static_cast<Utils::Ploidy>(mp)};
    ((KmerCounts *)(ObservedKmerCounts<30> *)h)->addInterclusterMultiplicity(is_decoy != 0, gp);
}
void ref_kc_add_cluster(void *h, unsigned mult, int is_mg) { ((KmerCounts *)(ObservedKmerCounts<30> *)h)->addClusterMultiplicity((uchar)mult, is_mg != 0); }
void ref_kc_add_sample_count(void *h, unsigned s, unsigned c) { ((ObservedKmerCounts<30> *)h)->addSampleCount(s, (uchar)c); }
void ref_kc_get(void *h, uint8_t *meta4, uint8_t *counts30, uint8_t *excluded) {
    auto *kc = (ObservedKmerCounts<30> *)h;
    meta4[0] = (uint8_t)((kc->hasClusterOccurrence() ? 1 : 0) | (kc->hasMulticlusterOccurrence() ? 2 : 0) | (kc->hasMultigroupOccurrence() ? 4 : 0) |
                         (kc->hasDecoyOccurrence() ? 8 : 0) | (kc->hasMaxMultiplicity() ? 16 : 0) | (kc->isParameter() ? 32 : 0));
    meta4[1] = 0;   // max_haploid_multiplicity is protected in the reference; observable only through hasMaxMultiplicity
    meta4[2] = kc->getInterclusterMultiplicity(Utils::Gender::Female);
    meta4[3] = kc->getInterclusterMultiplicity(Utils::Gender::Male);
    for (unsigned s = 0; s < 30; s++) counts30[s] = kc->getSampleCount(s);
    *excluded = kc->isExcluded() ? 1 : 0;
}

// NegativeBinomialDistribution (NegativeBinomialDistribution.cpp:68-147)
void ref_nb_moments(double mean, double var, double *p, double *size) {
    auto pr = NegativeBinomialDistribution::momentsToParameters(mean, var);
    *p = pr.first;
    *size = pr.second;
}
double ref_nb_logpmf(double p, double size, unsigned obs, unsigned scale) {
    NegativeBinomialDistribution nb(std::make_pair(p, size));
    return nb.logPmf(obs, scale);
}
double ref_log_addition(double a, double b) { return Utils::logAddition(a, b); }
int ref_double_compare(double a, double b) { return Utils::doubleCompare(a, b) ? 1 : 0; }

// LogDiscreteSampler / DiscreteSampler (DiscreteSampler.cpp:43-125)
void ref_logdiscrete_draws(const double *logw, unsigned n, unsigned seed, unsigned ndraws, unsigned *out) {
    std::mt19937 prng(seed);
    LogDiscreteSampler s(n);
    for (unsigned i = 0; i < n; i++) s.addOutcome(logw[i]);
    for (unsigned i = 0; i < ndraws; i++) out[i] = s.sample(&prng);
}
void ref_discrete_draws(const double *w, unsigned n, unsigned seed, unsigned ndraws, unsigned *out) {
    std::mt19937 prng(seed);
    DiscreteSampler s(n);
    for (unsigned i = 0; i < n; i++) s.addOutcome(w[i]);
    for (unsigned i = 0; i < ndraws; i++) out[i] = s.sample(&prng);
}

// KmerStats Welford (KmerStats.cpp:51-105)
void ref_kmerstats(const double *values, unsigned n, unsigned *count, double *fraction, double *mean, double *var) {
    KmerStats ks;
    for (unsigned i = 0; i < n; i++) ks.addValue(std::make_pair(values[i], true));
    *count = ks.getCount();
    *fraction = ks.getFraction().first;
    *mean = ks.getMean().first;
    *var = ks.getVariance().first;
}

// SparsityEstimator::estimateMinimumColumnCover (SparsityEstimator.cpp:41-87)
unsigned ref_sparsity_cover(const uint8_t *M, unsigned rows, unsigned cols, const uint8_t *row_mask, unsigned seed, unsigned *out) {
    Utils::MatrixXuchar m(rows, cols);
    Utils::RowVectorXbool mask(rows);
    for (unsigned r = 0; r < rows; r++) {
        mask(r) = row_mask[r] != 0;
        for (unsigned c = 0; c < cols; c++) m(r, c) = M[r * cols + c];
    }
    SparsityEstimator se(seed);
    auto cover = se.estimateMinimumColumnCover(m, mask, false);
    for (size_t i = 0; i < cover.size(); i++) out[i] = cover[i];
    return (unsigned)cover.size();
}


// ---- HybridHash / PartialSortedLinearMap (header-only: include/bayesTyper/HybridHash.tpp, LinearMap.tpp) ----
// root bucket of a k-mer: std::hash<std::bitset<2k>> % root_hash_size (HybridHash.tpp:78-82)
uint64_t ref_hybrid_hash_root(const char *kmer, uint64_t root_hash_size) {
    auto b = Nucleotide::ntToBit<BT_KMER_SIZE>(std::string(kmer, K));
    return std::hash<std::bitset<BT_KMER_SIZE * 2>>()(b.first) % root_hash_size;
}
// the reference's own container: n k-mers inserted as KmerHash::addKmer does (add_sorted = true, KmerHash.cpp:88-95), then
// shuffle(seed) (HybridHash.tpp:120-129) when do_shuffle, then iterated begin() .. end(): order_out[j] = index of the j-th k-mer
uint64_t ref_hybrid_hash_order(const char *kmers, uint64_t n, uint64_t root_hash_size, unsigned seed, int do_shuffle, uint32_t *order_out) {
    HybridHash<uint, BT_KMER_SIZE * 2> hash((uint)root_hash_size, n);
    for (uint64_t i = 0; i < n; i++) {
        auto b = Nucleotide::ntToBit<BT_KMER_SIZE>(std::string(kmers + i * K, K));
        hash.insert(b.first, (uint)i, true);
    }
    if (do_shuffle) hash.shuffle(seed);
    uint64_t j = 0;
    for (auto it = hash.begin(); it != hash.end(); it++) order_out[j++] = (*it).second;
    return j;
}
// the container KmerCounter::countPathMultigroupKmersCallback collects a group's path k-mers in (KmerCounter.cpp:111-119): a
// std::unordered_set<std::bitset<2k>> that is clear()ed between groups.  Groups are given back to back (group_off[g] .. group_off[g+1]);
// order_out receives, group by group, the indices (within the group) of its k-mers in iteration order; buckets_out[g] = bucket_count()
// after the group.  K-mers go through the reference's Nucleotide::ntToBit.
void ref_group_kmer_set_orders(const char *kmers, const uint64_t *group_off, uint64_t num_groups, uint32_t *order_out, uint64_t *buckets_out) {
    std::unordered_set<std::bitset<BT_KMER_SIZE * 2>> set;
    for (uint64_t g = 0; g < num_groups; g++) {
        set.clear();
        std::unordered_map<std::bitset<BT_KMER_SIZE * 2>, uint32_t> index;
        for (uint64_t i = group_off[g]; i < group_off[g + 1]; i++) {
            auto b = Nucleotide::ntToBit<BT_KMER_SIZE>(std::string(kmers + i * K, K));
            set.emplace(b.first);
            index.emplace(b.first, (uint32_t)(i - group_off[g]));
        }
        uint64_t j = group_off[g];
        for (auto &b : set) order_out[j++] = index.at(b);
        buckets_out[g] = set.bucket_count();
    }
}
// CountAllocation (src/bayesTyper/CountAllocation.cpp:34-57): addCount per (sample, count) then mergeInCountAllocations of a second one
void ref_count_allocation(unsigned short num_samples, const unsigned short *s1, const unsigned char *c1, uint64_t n1, const unsigned short *s2, const unsigned char *c2, uint64_t n2,
                          unsigned long *out /* [S*256] */) {
    CountAllocation a(num_samples), b(num_samples);
    for (uint64_t i = 0; i < n1; i++) a.addCount(s1[i], c1[i]);
    for (uint64_t i = 0; i < n2; i++) b.addCount(s2[i], c2[i]);
    a.mergeInCountAllocations(b);
    for (unsigned short s = 0; s < num_samples; s++)
        for (unsigned c = 0; c < 256; c++) out[(size_t)s * 256 + c] = a.getCounts().at(s).at(c);
}
}  // extern "C"
