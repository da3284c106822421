// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library, and only as the checker / CPU baseline.
//
// Scalar C++ restatement of the reference's per-cluster Gibbs genotyping path (BayesTyper v1.5):
//   CountDistribution LUTs, LogDiscreteSampler, (Sparse)FrequencyDistribution, SparsityEstimator,
//   VariantClusterHaplotypes, VariantClusterGenotyper, VariantClusterGroup and the default-mode
//   driver of InferenceEngine.  Every function cites the reference lines it follows.
//
// It uses libstdc++'s own <random>, std::shuffle and std::unordered_set — the very library code the
// reference is compiled against — so draw streams and container iteration orders
// are the reference's by construction.
//
// Parity status:
//   PINNED (bit-exact against the reference's own translation units, tests/test_oracle_gibbs.py):
//     LogDiscreteSampler/DiscreteSampler, NegativeBinomialDistribution::logPmf + momentsToParameters,
//     Utils::logAddition/doubleCompare, KmerStats, SparsityEstimator::estimateMinimumColumnCover,
//     and the known answers of SURVEY Appendix A.2/B.2 (gamma/shuffle/uniform_int/bernoulli streams).
//   PARITY UNPINNED (the reference TUs include Boost headers, Boost is absent from this image and no
//   stand-ins are written): FrequencyDistribution, CountDistribution, VariantClusterHaplotypes,
//   VariantClusterGenotyper, VariantClusterGroup, InferenceEngine are restated from the source text only.
//   boost::math::lgamma is restated with std::lgamma (ulp-level differences in the Poisson LUT and the
//   simplex CDF).
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../include/btgpu.h"   // only for the boundary data layout (bt_gibbs_params / bt_gibbs_batch)

namespace {

typedef unsigned int uint;
typedef unsigned short ushort;
typedef unsigned char uchar;
const ushort NOHAP = 0xFFFF;   // Utils::ushort_overflow
const double double_precision = std::numeric_limits<double>::epsilon();

// Utils.hpp:81-87
inline bool doubleCompare(const double a, const double b) { return ((a == b) or (std::abs(a - b) < std::abs(std::min(a, b)) * double_precision * 100)); }
// Utils.hpp:105-124
inline double logAddition(const double a, const double b) {
    if (a < b) return b + log1p(exp(a - b));
    return a + log1p(exp(b - a));
}

// ---- NegativeBinomialDistribution.cpp:68-79,122-147 ----
const double max_p = 0.99;
std::pair<double, double> nbMomentsToParameters(const double mean, double var) {
    if (max_p < (mean / var)) var = mean / max_p;
    double p = mean / var;
    double size = std::pow(mean, 2) / (var - mean);
    return std::make_pair(p, size);
}
double nbLogPmf(double p_, double size_, uint obs, uint size_scale) {
    double coef = std::lgamma(obs + size_ * size_scale) - std::lgamma(size_ * size_scale) - std::lgamma(obs + 1);
    return coef + std::log(p_) * size_ * size_scale + std::log(1 - p_) * obs;
}

// ---- CountDistribution.cpp:267-352 ----
double genomicCountLogPmf(double p, double size, uchar kmer_multiplicity, uchar kmer_count) {
    if (kmer_multiplicity == 0) return kmer_count == 0 ? 0 : -std::numeric_limits<double>::infinity();
    double v = nbLogPmf(p, size, kmer_count, kmer_multiplicity);
    if (kmer_count == 255) {
        uint limit = kmer_count;
        double prev = 0;
        do {
            limit++;
            prev = v;
            v = logAddition(v, nbLogPmf(p, size, limit, kmer_multiplicity));
            if (v > 0) {
                v = 0;
                break;
            }
        } while (!doubleCompare(prev, v));
    }
    return v;
}
double poissonLogProb(const uint value, const double rate) { return value * log(rate) - rate - std::lgamma(value + 1); }   // boost::math::lgamma in the reference
double noiseCountLogPmf(double rate, uchar kmer_count) {
    double v = poissonLogProb(kmer_count, rate);
    if (kmer_count == 255) {
        uint limit = kmer_count;
        double prev = 0;
        do {
            limit++;
            prev = v;
            v = logAddition(v, poissonLogProb(limit, rate));
            if (v > 0) {
                v = 0;
                break;
            }
        } while (!doubleCompare(prev, v));
    }
    return v;
}

struct CountDist {   // CountDistribution::calcCountLogProb, CountDistribution.cpp:255-265 (bias bins = 1)
    uint S = 0;
    std::vector<double> genomic;   // [S][256][256]
    std::vector<double> noise;     // [S][256]
    double calcCountLogProb(ushort s, uchar mult, uchar count) const {
        if (mult == 0) return noise[s * 256 + count];
        return genomic[((size_t)s * 256 + mult) * 256 + count];
    }
};

// ---- KmerStats.cpp:33-105, AlleleKmerStats :107-121 ----
struct KmerStats {
    uint count = 0;
    double fraction = 0, mean = 0, M2 = 0;
    void reset() { count = 0; fraction = 0; mean = 0; M2 = 0; }
    void addValue(const std::pair<double, bool> &value) {
        if (value.second) {
            count++;
            fraction += (static_cast<double>(!doubleCompare(value.first, 0)) - fraction) / count;
            double delta = value.first - mean;
            mean += delta / count;
            M2 += delta * (value.first - mean);
        }
    }
    std::pair<double, bool> getFraction() const { return count == 0 ? std::make_pair(-1.0, false) : std::make_pair(fraction, true); }
    std::pair<double, bool> getMean() const { return count == 0 ? std::make_pair(-1.0, false) : std::make_pair(mean, true); }
};
struct AlleleKmerStats {
    std::vector<KmerStats> count_stats, fraction_stats, mean_stats;
    AlleleKmerStats() {}
    explicit AlleleKmerStats(ushort n) : count_stats(n), fraction_stats(n), mean_stats(n) {}
    void addKmerStats(const KmerStats &ks, const ushort allele_idx) {
        count_stats.at(allele_idx).addValue(std::make_pair((double)ks.count, true));
        fraction_stats.at(allele_idx).addValue(ks.getFraction());
        mean_stats.at(allele_idx).addValue(ks.getMean());
    }
};

// ---- DiscreteSampler.cpp:43-125 ----
struct DiscreteSampler {
    std::vector<double> cum_probs;
    void addOutcome(double prob) { cum_probs.push_back(cum_probs.empty() ? prob : prob + cum_probs.back()); }
    uint search(double x) const {
        if (cum_probs.size() > 1) return (uint)(std::upper_bound(cum_probs.begin(), cum_probs.end(), x) - cum_probs.begin());
        return 0;
    }
    uint sample(std::mt19937 *prng) const { return search(std::generate_canonical<double, std::numeric_limits<double>::digits>(*prng) * cum_probs.back()); }
};
struct LogDiscreteSampler : DiscreteSampler {
    void addOutcome(double log_prob) { cum_probs.push_back(cum_probs.empty() ? log_prob : logAddition(log_prob, cum_probs.back())); }
    uint sample(std::mt19937 *prng) const { return search(log(std::generate_canonical<double, std::numeric_limits<double>::digits>(*prng)) + cum_probs.back()); }
};

// ---- SparsityEstimator.cpp:41-87 (do_weighted_sampling = false path) ----
std::vector<uint> estimateMinimumColumnCover(const uchar *M, uint rows, uint cols, const std::vector<uchar> &uncovered_rows, uint prng_seed) {
    std::mt19937 prng(prng_seed);
    std::vector<uchar> cur = uncovered_rows;
    std::vector<uint> cover;
    auto remaining = [&]() { uint s = 0; for (auto v : cur) s += v; return s; };
    while (remaining() > 0) {
        std::vector<uint> column_row_cover(cols, 0);
        for (uint r = 0; r < rows; r++)
            if (cur[r]) for (uint c = 0; c < cols; c++) column_row_cover[c] += M[(size_t)r * cols + c];
        uint max_row_cover = *std::max_element(column_row_cover.begin(), column_row_cover.end());
        assert(max_row_cover > 0);
        DiscreteSampler column_sampler;
        std::vector<uint> max_cols;
        for (uint c = 0; c < cols; c++)
            if (column_row_cover[c] == max_row_cover) {
                column_sampler.addOutcome(1);
                max_cols.push_back(c);
            }
        const uint sampled = max_cols.at(column_sampler.sample(&prng));
        cover.emplace_back(sampled);
        for (uint r = 0; r < rows; r++)
            if (cur[r] && M[(size_t)r * cols + sampled] != 0) cur[r] = 0;
    }
    return cover;
}

// ---- FrequencyDistribution.cpp:43-93 (dense) ----
const double dirichlet_parameter = 1;
struct FrequencyDistribution {
    const uint num_elements;
    std::vector<uint> observation_counts;
    std::vector<double> frequencies;
    std::vector<bool> non_zero_frequencies;
    std::mt19937 prng;
    std::gamma_distribution<> gamma_dist;
    FrequencyDistribution(uint n, uint seed) : num_elements(n) {
        prng = std::mt19937(seed);
        FrequencyDistribution::reset();
    }
    virtual ~FrequencyDistribution() {}
    virtual void reset() {
        observation_counts = std::vector<uint>(num_elements, 0);
        frequencies = std::vector<double>(num_elements, 1 / static_cast<double>(num_elements));
        non_zero_frequencies = std::vector<bool>(num_elements, true);
    }
    std::pair<bool, double> getElementFrequency(uint e) { return std::pair<bool, double>(non_zero_frequencies.at(e), frequencies.at(e)); }
    virtual void incrementObservationCount(uint e) { observation_counts.at(e)++; }
    virtual void sampleFrequencies(const uint) {
        double norm_const = 0;
        for (uint i = 0; i < frequencies.size(); i++) {
            gamma_dist.param(std::gamma_distribution<>::param_type(observation_counts.at(i) + 1, 1));
            frequencies.at(i) = gamma_dist(prng);
            norm_const += frequencies.at(i);
            observation_counts.at(i) = 0;
        }
        for (auto &f : frequencies) f /= norm_const;
    }
};

// ---- FrequencyDistribution.cpp:96-303 (sparse) ----
struct SparseFrequencyDistribution : FrequencyDistribution {
    double sparsity;
    std::unordered_map<uint, std::unordered_map<uint, std::vector<double>>> cached_simplex_prob_vectors;
    std::unordered_set<uint> plus_count_indices;
    std::unordered_set<uint> zero_count_indices;
    std::uniform_int_distribution<> uniform_int_dist;

    SparseFrequencyDistribution(double sparsity_in, uint n, uint seed) : FrequencyDistribution(n, seed) {
        sparsity = std::min(sparsity_in, 1 - double_precision * 100);
        assert(sparsity > 0);
        reset();
    }
    void reset() override {
        FrequencyDistribution::reset();
        plus_count_indices.clear();
        zero_count_indices.clear();
        for (uint i = 0; i < num_elements; i++) zero_count_indices.insert(i);
    }
    void updateCachedSimplexProbVector(std::vector<double> *v, const uint total_num_observations, const uint count_plus_size) {
        v->reserve(frequencies.size() - count_plus_size + 1);
        double cardinal_eq_z_log = 0;
        double prob_z_log = count_plus_size * log(sparsity) + (frequencies.size() - count_plus_size) * log(1 - sparsity);
        double prob_t_log = std::lgamma(count_plus_size * dirichlet_parameter) - std::lgamma(total_num_observations + count_plus_size * dirichlet_parameter);
        double prob_eq_z_log = cardinal_eq_z_log + prob_z_log + prob_t_log;
        double row_sum = prob_eq_z_log;
        v->push_back(row_sum);
        for (uint j = count_plus_size + 1; j < frequencies.size() + 1; j++) {
            cardinal_eq_z_log = std::lgamma(frequencies.size() - count_plus_size + 1) - (std::lgamma(j - count_plus_size + 1) + std::lgamma(frequencies.size() - j + 1));
            prob_z_log = j * log(sparsity) + (frequencies.size() - j) * log(1 - sparsity);
            prob_t_log = std::lgamma(j * dirichlet_parameter) - std::lgamma(total_num_observations + j * dirichlet_parameter);
            prob_eq_z_log = cardinal_eq_z_log + prob_z_log + prob_t_log;
            row_sum += log(1 + exp(prob_eq_z_log - row_sum));
            v->push_back(row_sum);
            if (doubleCompare(v->back(), *(v->rbegin() + 1))) break;
        }
        for (auto &prob : *v) prob = exp(prob - row_sum);
    }
    void incrementObservationCount(const uint e) override {
        if (observation_counts.at(e) == 0) {
            bool ok = plus_count_indices.insert(e).second;   // the reference does this inside assert() (asserts are enabled)
            assert(ok);
            size_t er = zero_count_indices.erase(e);
            assert(er);
            (void)ok; (void)er;
        }
        observation_counts.at(e)++;
    }
    void sampleFrequencies(const uint sum_observation_counts) override {
        auto &by_sum = cached_simplex_prob_vectors[sum_observation_counts];
        auto it = by_sum.find((uint)plus_count_indices.size() - 1);
        if (it == by_sum.end()) {
            it = by_sum.emplace((uint)plus_count_indices.size() - 1, std::vector<double>()).first;
            updateCachedSimplexProbVector(&(it->second), sum_observation_counts, (uint)plus_count_indices.size());
        }
        auto prob_vector = &it->second;
        uint simplex_size = uint(std::upper_bound(prob_vector->begin(), prob_vector->end(), std::generate_canonical<double, std::numeric_limits<double>::digits>(prng)) - prob_vector->begin()) +
                            (uint)plus_count_indices.size();
        assert(simplex_size > 0);
        double norm_const = 0;
        for (auto &plus_count_idx : plus_count_indices) {
            gamma_dist.param(std::gamma_distribution<>::param_type(observation_counts.at(plus_count_idx) + dirichlet_parameter, 1));
            frequencies.at(plus_count_idx) = gamma_dist(prng);
            norm_const += frequencies.at(plus_count_idx);
            non_zero_frequencies.at(plus_count_idx) = true;
        }
        gamma_dist.param(std::gamma_distribution<>::param_type(dirichlet_parameter, 1));
        while (plus_count_indices.size() < simplex_size) {
            uniform_int_dist.param(std::uniform_int_distribution<>::param_type(0, (int)zero_count_indices.size() - 1));
            uint sampled_position = uniform_int_dist(prng);
            auto zit = zero_count_indices.begin();
            uint pos = 0;
            while (pos < sampled_position) {
                zit++;
                pos++;
            }
            frequencies.at(*zit) = gamma_dist(prng);
            norm_const += frequencies.at(*zit);
            non_zero_frequencies.at(*zit) = true;
            uint e = *zit;
            bool ok = plus_count_indices.insert(e).second;
            assert(ok);
            zero_count_indices.erase(e);
            (void)ok;
        }
        for (auto &z : zero_count_indices) {
            frequencies.at(z) = 0;
            non_zero_frequencies.at(z) = false;
            observation_counts.at(z) = 0;
        }
        for (auto &p : plus_count_indices) {
            frequencies.at(p) /= norm_const;
            bool ok = zero_count_indices.insert(p).second;
            assert(ok);
            (void)ok;
            observation_counts.at(p) = 0;
        }
        plus_count_indices.clear();
    }
};

// ---- HaplotypeFrequencyDistribution.cpp:79-138 (SparseHaplotypeFrequencyDistribution) ----
struct HaplotypeFrequencyDistribution {
    uint num_haplotype_count = 0, num_missing_count = 0;
    FrequencyDistribution *fd = nullptr;
    HaplotypeFrequencyDistribution(const std::vector<uint> &non_zero_haplotypes, uint num_haplotypes, uint seed) {
        if (non_zero_haplotypes.empty()) fd = new FrequencyDistribution(num_haplotypes, seed);
        else fd = new SparseFrequencyDistribution(non_zero_haplotypes.size() / static_cast<double>(num_haplotypes), num_haplotypes, seed);
    }
    ~HaplotypeFrequencyDistribution() { delete fd; }
    void reset() {
        assert(num_haplotype_count == 0 && num_missing_count == 0);
        fd->reset();
    }
    std::pair<bool, double> getFrequency(ushort h) { return fd->getElementFrequency(h); }
    void incrementCount(ushort h) {
        if (h == NOHAP) num_missing_count++;
        else {
            num_haplotype_count++;
            fd->incrementObservationCount(h);
        }
    }
    void sampleFrequencies() {
        if (num_haplotype_count > 0) fd->sampleFrequencies(num_haplotype_count);
        num_haplotype_count = 0;
        num_missing_count = 0;
    }
};

// shared KmerCounts state of a multicluster k-mer: per-sample running multiplicities
// (ObservedKmerCounts::multiplicities, KmerCounts.cpp:196-223)
struct SharedRec {
    std::vector<uchar> multiplicities;
};

struct KmerInfo {   // VariantClusterHaplotypes.hpp:66-74
    bool has_counts = false;
    const uchar *counts = nullptr;   // [S]
    uchar ic_mult[2] = {0, 0};
    SharedRec *shared = nullptr;
    std::vector<std::pair<ushort, std::vector<bool>>> variant_haplotype_indices;
    uchar getSampleCount(ushort s) const { return counts[s]; }
};

typedef std::pair<ushort, ushort> Dip;
struct DipHash {   // iteration order of these maps never influences results
    size_t operator()(const Dip &d) const { return ((size_t)d.first << 16) ^ d.second; }
};

struct NestedVariantClusterInfo {   // VariantClusterHaplotypes.hpp:99-105
    uchar nested_ploidy;
    std::vector<KmerStats> nested_kmer_stats;
    explicit NestedVariantClusterInfo(uchar p) : nested_ploidy(p) {}
};

struct VariantInfoLite {
    ushort num_alleles;
    bool has_dependency;
    bool isMissing(ushort a) const { return has_dependency and (a == num_alleles - 1); }   // VariantInfo.hpp:83-95
};

// ---- VariantClusterHaplotypes.cpp ----
struct Haplotypes {
    uint H = 0, V = 0, K = 0, S = 0;
    const uchar *M = nullptr;   // K x H row-major
    std::vector<std::vector<ushort>> hap_alleles;        // [H][V]
    std::vector<std::vector<uint>> hap_nested;           // [H] sorted nested cluster indices
    std::vector<KmerInfo> kmers;
    std::vector<uint> unique_kmer_indices, multicluster_kmer_indices;
    std::vector<uint> unique_kmer_subset_indices, multicluster_kmer_subset_indices;
    std::vector<uchar> sample_multicluster_kmer_multiplicities;   // [nsub_m x S]
    struct KmerStatsCache {
        bool update = true;
        std::vector<KmerStats> haplotype_1, haplotype_2;
    };
    std::vector<KmerStatsCache> kmer_stats_cache;
    std::unordered_map<uint, std::vector<ushort>> nested_variant_cluster_dependency;

    uchar mult(uint k, ushort h) const { return M[(size_t)k * H + h]; }
    uchar getDiplotypeKmerMultiplicity(uint k, const Dip &d) const {   // :45-60
        uchar m = 0;
        if (d.first != NOHAP) m += mult(k, d.first);
        if (d.second != NOHAP) m += mult(k, d.second);
        return m;
    }
    uchar getUniqueKmerMultiplicity(uint k, const Dip &d, uchar gender) const {   // :62-74
        uchar m = getDiplotypeKmerMultiplicity(k, d);
        if (kmers[k].has_counts) m += kmers[k].ic_mult[gender];
        return m;
    }
    uchar getMulticlusterKmerMultiplicity(uint k, const Dip &d, const Dip &prev, ushort s, uchar gender) const {   // :76-93
        const KmerInfo &ki = kmers[k];
        if (ki.getSampleCount(s) == 0) return (uchar)(getDiplotypeKmerMultiplicity(k, d) + ki.ic_mult[gender]);
        return (uchar)(ki.shared->multiplicities[s] - getDiplotypeKmerMultiplicity(k, prev) + getDiplotypeKmerMultiplicity(k, d) + ki.ic_mult[gender]);
    }
    uchar getPreviousMulticlusterKmerMultiplicity(uint sub, const Dip &d, const Dip &prev, ushort s, uchar gender) const {   // :95-108
        const uint k = multicluster_kmer_subset_indices.at(sub);
        return (uchar)(sample_multicluster_kmer_multiplicities[(size_t)sub * S + s] - getDiplotypeKmerMultiplicity(k, prev) + getDiplotypeKmerMultiplicity(k, d) + kmers[k].ic_mult[gender]);
    }
    bool isMaxHaplotypeVariantKmer(std::vector<std::vector<uint>> *n, const uint maxk, const std::vector<std::pair<ushort, std::vector<bool>>> &vhi) {   // :155-177
        bool is_max = true;
        for (auto &vh : vhi)
            for (ushort h = 0; h < vh.second.size(); h++)
                if (vh.second.at(h) and (n->at(h).at(vh.first) < maxk)) {
                    n->at(h).at(vh.first)++;
                    is_max = false;
                }
        return is_max;
    }
    void sampleKmerSubset(std::mt19937 *prng, const float rate, const uint maxk, const ushort num_samples) {   // :110-153
        unique_kmer_subset_indices.clear();
        multicluster_kmer_subset_indices.clear();
        std::bernoulli_distribution bernoulli_dist(rate);
        std::vector<std::vector<uint>> n(H, std::vector<uint>(V, 0));
        std::shuffle(unique_kmer_indices.begin(), unique_kmer_indices.end(), *prng);
        for (auto &k : unique_kmer_indices)
            if (bernoulli_dist(*prng))
                if (!isMaxHaplotypeVariantKmer(&n, maxk, kmers.at(k).variant_haplotype_indices)) unique_kmer_subset_indices.push_back(k);
        std::shuffle(multicluster_kmer_indices.begin(), multicluster_kmer_indices.end(), *prng);
        for (auto &k : multicluster_kmer_indices)
            if (bernoulli_dist(*prng))
                if (!isMaxHaplotypeVariantKmer(&n, maxk, kmers.at(k).variant_haplotype_indices)) multicluster_kmer_subset_indices.push_back(k);
        sample_multicluster_kmer_multiplicities.assign(multicluster_kmer_subset_indices.size() * (size_t)num_samples, 0);
        for (auto &c : kmer_stats_cache) c.update = true;
    }
    bool isMulticlusterKmerUpdated(uint sub, ushort s) const {   // :180-195
        const uint k = multicluster_kmer_subset_indices.at(sub);
        return (kmers[k].getSampleCount(s) > 0) and (kmers[k].shared->multiplicities[s] != sample_multicluster_kmer_multiplicities[(size_t)sub * S + s]);
    }
    void updateMulticlusterKmerMultiplicities(const Dip &d, const Dip &prev, ushort s) {   // :197-233
        if (d != prev) {
            kmer_stats_cache.at(s).update = true;
            for (auto &k : multicluster_kmer_indices) {
                uchar cur = getDiplotypeKmerMultiplicity(k, d), pre = getDiplotypeKmerMultiplicity(k, prev);
                if (cur != pre) {
                    auto &m = kmers[k].shared->multiplicities[s];
                    assert(pre <= m);
                    m -= pre;
                    m += cur;
                }
            }
        }
        for (uint sub = 0; sub < multicluster_kmer_subset_indices.size(); sub++) {
            const uint k = multicluster_kmer_subset_indices.at(sub);
            if ((getDiplotypeKmerMultiplicity(k, d) > 0) and (kmers[k].getSampleCount(s) > 0) and
                (kmers[k].shared->multiplicities[s] != sample_multicluster_kmer_multiplicities[(size_t)sub * S + s]))
                kmer_stats_cache.at(s).update = true;
            sample_multicluster_kmer_multiplicities[(size_t)sub * S + s] = kmers[k].shared->multiplicities[s];
        }
    }
    void updateKmerStatsCache(const KmerInfo &ki, const Dip &d, ushort s, uchar kmer_multiplicity) {   // :300-330
        double kmer_count = 0;
        if (ki.has_counts) kmer_count = ki.getSampleCount(s) / static_cast<double>(kmer_multiplicity);
        for (auto &vh : ki.variant_haplotype_indices) {
            if (vh.second.at(d.first)) kmer_stats_cache.at(s).haplotype_1.at(vh.first).addValue(std::make_pair(kmer_count, true));
            if (d.second != NOHAP)
                if (vh.second.at(d.second)) kmer_stats_cache.at(s).haplotype_2.at(vh.first).addValue(std::make_pair(kmer_count, true));
        }
    }
    void addHaplotypeKmerStats(std::vector<std::vector<AlleleKmerStats>> *aks, const std::vector<KmerStats> &cache, const std::vector<VariantInfoLite> &vinfo, ushort s,
                               const std::vector<ushort> &alleles) {   // :332-358
        ushort last_non_missing = NOHAP;
        for (ushort v = 0; v < cache.size(); v++) {
            auto a = alleles.at(v);
            if (vinfo.at(v).isMissing(a)) {
                assert(last_non_missing != NOHAP);
                aks->at(v).at(s).addKmerStats(cache.at(last_non_missing), a);
            } else {
                aks->at(v).at(s).addKmerStats(cache.at(v), a);
                last_non_missing = v;
            }
        }
    }
    void updateAlleleKmerStats(std::vector<std::vector<AlleleKmerStats>> *aks, const std::vector<uchar> &gender, const std::vector<VariantInfoLite> &vinfo,
                               const std::vector<NestedVariantClusterInfo> &nested, const std::vector<Dip> &diplotypes) {   // :235-298
        for (ushort s = 0; s < S; s++) {
            auto d = diplotypes.at(s);
            auto &cache = kmer_stats_cache.at(s);
            if (cache.update) {
                cache.update = false;
                for (ushort v = 0; v < V; v++) {
                    cache.haplotype_1.at(v).reset();
                    cache.haplotype_2.at(v).reset();
                }
                if (d.first != NOHAP) {
                    for (auto &k : unique_kmer_subset_indices)
                        if (getDiplotypeKmerMultiplicity(k, d) > 0) updateKmerStatsCache(kmers.at(k), d, s, getUniqueKmerMultiplicity(k, d, gender[s]));
                    for (auto &k : multicluster_kmer_subset_indices)
                        if (getDiplotypeKmerMultiplicity(k, d) > 0) updateKmerStatsCache(kmers.at(k), d, s, getMulticlusterKmerMultiplicity(k, d, d, s, gender[s]));
                }
            }
            if (d.first != NOHAP) addHaplotypeKmerStats(aks, cache.haplotype_1, vinfo, s, hap_alleles.at(d.first));
            if (d.second != NOHAP) addHaplotypeKmerStats(aks, cache.haplotype_2, vinfo, s, hap_alleles.at(d.second));
            for (auto &ks : nested.at(s).nested_kmer_stats)   // addNestedHaplotypeKmerStats :360-372
                for (ushort v = 0; v < aks->size(); v++) aks->at(v).at(s).addKmerStats(ks, vinfo.at(v).num_alleles - 1);
        }
    }
};

// ---- VariantClusterGenotyper.cpp ----
struct Genotyper {
    uint S;
    const std::vector<uchar> &gender;
    std::mt19937 prng;
    bool use_multicluster_kmers = false;
    Haplotypes hap;
    std::vector<VariantInfoLite> vinfo;
    std::vector<std::vector<AlleleKmerStats>> allele_kmer_stats;   // [V][S]
    std::vector<std::unordered_map<Dip, double, DipHash>> unique_lp, multi_lp;
    std::vector<Dip> diplotypes;
    std::map<Dip, std::vector<uint>> diplotype_sampling_frequencies;   // ordered map: only sums are read from it
    HaplotypeFrequencyDistribution *hfd = nullptr;

    Genotyper(uint S_, const std::vector<uchar> &gender_, uint prng_seed, Haplotypes &&h, std::vector<VariantInfoLite> &&vi) : S(S_), gender(gender_), hap(std::move(h)), vinfo(std::move(vi)) {   // :59-106
        prng = std::mt19937(prng_seed);
        unique_lp.resize(S);
        multi_lp.resize(S);
        diplotypes.assign(S, Dip(NOHAP, NOHAP));
        for (auto &v : vinfo) allele_kmer_stats.emplace_back(std::vector<AlleleKmerStats>(S, AlleleKmerStats(v.num_alleles)));
        hap.kmer_stats_cache.assign(S, Haplotypes::KmerStatsCache());
        for (auto &c : hap.kmer_stats_cache) {
            c.haplotype_1.assign(hap.V, KmerStats());
            c.haplotype_2.assign(hap.V, KmerStats());
        }
        std::vector<uchar> non_zero_kmer_counts(hap.K, 0);
        for (uint k = 0; k < hap.K; k++) non_zero_kmer_counts[k] = hap.kmers[k].has_counts ? 1 : 0;
        auto cover = estimateMinimumColumnCover(hap.M, hap.K, hap.H, non_zero_kmer_counts, prng_seed);
        hfd = new HaplotypeFrequencyDistribution(cover, hap.H, prng_seed);
    }
    ~Genotyper() { delete hfd; }
    void reset(float rate, uint maxk) {   // :113-129
        use_multicluster_kmers = false;
        hap.sampleKmerSubset(&prng, rate, maxk, (ushort)S);
        clearCache();
        hfd->reset();
    }
    void clearCache() {   // :131-138
        for (ushort s = 0; s < S; s++) {
            unique_lp[s].clear();
            multi_lp[s].clear();
        }
    }
    void updateMulticlusterDiplotypeLogProb(const CountDist &cd, ushort s) {   // :569-595
        if (multi_lp[s].empty()) return;
        for (uint sub = 0; sub < hap.multicluster_kmer_subset_indices.size(); sub++) {
            if (!hap.isMulticlusterKmerUpdated(sub, s)) continue;
            for (auto &e : multi_lp[s]) {
                const uint k = hap.multicluster_kmer_subset_indices.at(sub);
                auto prev_m = hap.getPreviousMulticlusterKmerMultiplicity(sub, e.first, diplotypes.at(s), s, gender[s]);
                e.second -= cd.calcCountLogProb(s, prev_m, hap.kmers[k].getSampleCount(s));
                auto m = hap.getMulticlusterKmerMultiplicity(k, e.first, diplotypes.at(s), s, gender[s]);
                e.second += cd.calcCountLogProb(s, m, hap.kmers[k].getSampleCount(s));
            }
        }
    }
    double calcDiplotypeLogProb(const CountDist &cd, ushort s, const Dip &d) {   // :597-666
        double lp = 0;
        if (d.second == NOHAP) lp += log(hfd->getFrequency(d.first).second);
        else if (d.first == d.second) lp += 2 * log(hfd->getFrequency(d.first).second);
        else lp += log(2) + log(hfd->getFrequency(d.first).second) + log(hfd->getFrequency(d.second).second);
        auto ue = unique_lp[s].emplace(d, 0);
        if (ue.second) {
            for (auto &k : hap.unique_kmer_subset_indices) {
                auto m = hap.getUniqueKmerMultiplicity(k, d, gender[s]);
                ue.first->second += cd.calcCountLogProb(s, m, hap.kmers[k].has_counts ? hap.kmers[k].getSampleCount(s) : 0);
            }
        }
        lp += ue.first->second;
        if (use_multicluster_kmers) {
            auto me = multi_lp[s].emplace(d, 0);
            if (me.second) {
                for (auto &k : hap.multicluster_kmer_subset_indices) {
                    auto m = hap.getMulticlusterKmerMultiplicity(k, d, diplotypes.at(s), s, gender[s]);
                    me.first->second += cd.calcCountLogProb(s, m, hap.kmers[k].getSampleCount(s));
                }
            }
            lp += me.first->second;
        }
        assert(std::isfinite(lp));
        return lp;
    }
    void sampleDiplotype(const std::vector<ushort> &nz, const CountDist &cd, ushort s, uchar ploidy) {   // :707-755
        const size_t num_expected_diplotypes = (nz.size() * (nz.size() - 1)) / 2 + nz.size();   // :709-713: the sampler and the candidate list are sized up front
        LogDiscreteSampler sampler;
        sampler.cum_probs.reserve(num_expected_diplotypes);   // (LogDiscreteSampler(size_guess), DiscreteSampler.cpp:39)
        std::vector<Dip> cand;
        cand.reserve(num_expected_diplotypes);
        if (ploidy == 2) {
            for (size_t a = 0; a < nz.size(); a++)
                for (size_t b = a; b < nz.size(); b++) {
                    sampler.addOutcome(calcDiplotypeLogProb(cd, s, Dip(nz[a], nz[b])));
                    cand.emplace_back(nz[a], nz[b]);
                }
        } else if (ploidy == 1) {
            for (auto &h : nz) {
                sampler.addOutcome(calcDiplotypeLogProb(cd, s, Dip(h, NOHAP)));
                cand.emplace_back(h, NOHAP);
            }
        } else {
            sampler.addOutcome(0);
            cand.emplace_back(NOHAP, NOHAP);
        }
        diplotypes.at(s) = cand.at(sampler.sample(&prng));
        hfd->incrementCount(diplotypes.at(s).first);
        hfd->incrementCount(diplotypes.at(s).second);
    }
    void sampleDiplotypes(const CountDist &cd, const std::vector<NestedVariantClusterInfo> &nested, bool collect, std::vector<uint32_t> *trace) {   // :668-705
        std::vector<ushort> nz;
        nz.reserve(hap.H);   // (:672-673)
        for (ushort h = 0; h < hap.H; h++)
            if (hfd->getFrequency(h).first) nz.emplace_back(h);
        for (ushort s = 0; s < S; s++) {
            auto prev = diplotypes.at(s);
            updateMulticlusterDiplotypeLogProb(cd, s);
            sampleDiplotype(nz, cd, s, nested.at(s).nested_ploidy);
            hap.updateMulticlusterKmerMultiplicities(diplotypes.at(s), prev, s);
            if (trace) trace->push_back((uint32_t)diplotypes.at(s).first | ((uint32_t)diplotypes.at(s).second << 16));
            if (collect) {
                auto e = diplotype_sampling_frequencies.emplace(diplotypes.at(s), std::vector<uint>(S, 0));
                e.first->second.at(s)++;
            }
        }
        if (collect) hap.updateAlleleKmerStats(&allele_kmer_stats, gender, vinfo, nested, diplotypes);
        use_multicluster_kmers = !hap.multicluster_kmer_subset_indices.empty();
    }
    void getNoiseCounts(uint64_t *hist) {   // :757-779
        for (ushort s = 0; s < S; s++)
            for (auto &k : hap.unique_kmer_subset_indices)
                if (hap.getUniqueKmerMultiplicity(k, diplotypes.at(s), gender[s]) == 0) hist[s * 256 + (hap.kmers[k].has_counts ? hap.kmers[k].getSampleCount(s) : 0)]++;
    }
    void sampleHaplotypeFrequencies() { hfd->sampleFrequencies(); }   // :781-785
    void updateNestedVariantClusterInfo(std::vector<NestedVariantClusterInfo> *nested, const uint child_cluster_idx) {   // :182-206 (+ :140-180)
        for (ushort s = 0; s < S; s++) {
            for (int which = 0; which < 2; which++) {
                ushort h = which == 0 ? diplotypes.at(s).first : diplotypes.at(s).second;
                if (h == NOHAP) continue;
                const auto &nest = hap.hap_nested.at(h);
                if (std::binary_search(nest.begin(), nest.end(), child_cluster_idx)) continue;
                // updateNestedPloidy
                auto &pl = nested->at(s).nested_ploidy;
                assert(pl != 0);
                pl = (pl == 2) ? 1 : 0;
                // addNestedKmerStats
                auto dep = hap.nested_variant_cluster_dependency.find(child_cluster_idx);
                assert(dep != hap.nested_variant_cluster_dependency.end());
                ushort variant_idx = NOHAP;
                for (auto &nv : dep->second) {
                    auto a = hap.hap_alleles.at(h).at(nv);
                    if (!vinfo.at(nv).isMissing(a)) {
                        variant_idx = nv;
                        break;
                    }
                }
                assert(variant_idx != NOHAP);
                const auto &cache = which == 0 ? hap.kmer_stats_cache.at(s).haplotype_1 : hap.kmer_stats_cache.at(s).haplotype_2;
                nested->at(s).nested_kmer_stats.push_back(cache.at(variant_idx));
            }
        }
    }
};

// ---- VariantClusterGroup.cpp ----
struct Vertex {
    uint variant_cluster_idx;
    uint cluster;   // global cluster index in the batch
    Genotyper *genotyper = nullptr;
};

struct Group {
    uint index;   // i
    std::vector<Vertex> vertices;
    std::vector<std::vector<uint>> out_edges;
    std::vector<uint> source_vertices;
    std::vector<uchar> ploidy;   // [S]
    std::vector<SharedRec> shared;
    ~Group() {
        for (auto &v : vertices) delete v.genotyper;
    }
};

struct OracleGibbs {
    bt_gibbs_params P;
    std::vector<uchar> gender;
    const bt_gibbs_batch *B;   // caller keeps the arrays alive
    CountDist cd;
    std::vector<Group> groups;
    // prefix sums over clusters
    std::vector<uint64_t> mult_off, kvbits_off, hapvar_off;
    std::vector<uint32_t> hap_base, var_base;
    uint32_t trace_sweeps = 0;
    std::vector<std::vector<uint32_t>> traces;   // per group: [sweep][cluster in vertex order][S]
};

void buildGenotyper(OracleGibbs &O, Group &G, Vertex &vx, uint prng_seed) {   // VariantClusterGroup::initGenotyper :179-182 + getHaplotypeCandidates output
    const bt_gibbs_batch &B = *O.B;
    const uint c = vx.cluster, S = O.P.num_samples;
    Haplotypes h;
    h.H = B.num_haplotypes[c];
    h.V = B.num_variants[c];
    h.K = B.kmer_off[c + 1] - B.kmer_off[c];
    h.S = S;
    h.M = B.hap_kmer_mult + O.mult_off[c];
    const uint HW = (h.H + 31) / 32;
    h.hap_alleles.resize(h.H);
    h.hap_nested.resize(h.H);
    for (uint a = 0; a < h.H; a++) {
        h.hap_alleles[a].assign(B.hap_allele + O.hapvar_off[c] + (size_t)a * h.V, B.hap_allele + O.hapvar_off[c] + (size_t)(a + 1) * h.V);
        uint hb = O.hap_base[c] + a;
        h.hap_nested[a].assign(B.hapnest_idx + B.hapnest_off[hb], B.hapnest_idx + B.hapnest_off[hb + 1]);
    }
    h.kmers.resize(h.K);
    const uint r0 = B.kmer_off[c];
    uint64_t e_base = O.kvbits_off[c];
    for (uint k = 0; k < h.K; k++) {
        KmerInfo &ki = h.kmers[k];
        const uint r = r0 + k;
        ki.has_counts = B.kmer_has_counts[r] != 0;
        ki.counts = B.kmer_counts + (size_t)r * S;
        ki.ic_mult[0] = B.kmer_ic_mult[2 * r];
        ki.ic_mult[1] = B.kmer_ic_mult[2 * r + 1];
        ki.shared = B.kmer_shared[r] >= 0 ? &G.shared.at(B.kmer_shared[r]) : nullptr;
        for (uint e = B.kv_off[r]; e < B.kv_off[r + 1]; e++) {
            std::vector<bool> bits(h.H, false);
            const uint32_t *w = B.kv_bits + e_base + (uint64_t)(e - B.kv_off[r0]) * HW;
            for (uint a = 0; a < h.H; a++) bits[a] = (w[a / 32] >> (a % 32)) & 1;
            ki.variant_haplotype_indices.emplace_back(B.kv_var[e], bits);
        }
    }
    h.unique_kmer_indices.assign(B.unique_idx + B.unique_off[c], B.unique_idx + B.unique_off[c + 1]);
    h.multicluster_kmer_indices.assign(B.multi_idx + B.multi_off[c], B.multi_idx + B.multi_off[c + 1]);
    for (uint d = B.nestdep_off[c]; d < B.nestdep_off[c + 1]; d++)
        h.nested_variant_cluster_dependency.emplace(B.nestdep_cluster[d], std::vector<ushort>(B.nestdep_var + B.nestdep_var_off[d], B.nestdep_var + B.nestdep_var_off[d + 1]));
    std::vector<VariantInfoLite> vi(h.V);
    for (uint v = 0; v < h.V; v++) {
        vi[v].num_alleles = B.var_num_alleles[O.var_base[c] + v];
        vi[v].has_dependency = B.var_has_dependency[O.var_base[c] + v] != 0;
    }
    vx.genotyper = new Genotyper(S, O.gender, prng_seed, std::move(h), std::move(vi));
}

void initGenotyper(OracleGibbs &O, Group &G, uint prng_seed) {   // VariantClusterGroup.cpp:171-186
    for (auto &vx : G.vertices) {
        if (!vx.genotyper) buildGenotyper(O, G, vx, prng_seed + vx.variant_cluster_idx);
        vx.genotyper->reset(O.P.kmer_subsampling_rate, O.P.max_haplotype_variant_kmers);
    }
}
void shuffleBranchOrdering(Group &G, uint prng_seed) {   // :208-218
    std::mt19937 prng = std::mt19937(prng_seed);
    std::shuffle(G.source_vertices.begin(), G.source_vertices.end(), prng);
    for (auto &oe : G.out_edges) std::shuffle(oe.begin(), oe.end(), prng);
}
void runGibbsSample(OracleGibbs &O, Group &G, uint vertex_idx, const std::vector<NestedVariantClusterInfo> &nested, bool collect, std::vector<uint32_t> *trace_row) {   // :236-250
    Genotyper *g = G.vertices.at(vertex_idx).genotyper;
    std::vector<uint32_t> tr;
    g->sampleDiplotypes(O.cd, nested, collect, trace_row ? &tr : nullptr);
    if (trace_row) std::copy(tr.begin(), tr.end(), trace_row->begin() + (size_t)vertex_idx * O.P.num_samples);
    g->sampleHaplotypeFrequencies();
    for (auto &target : G.out_edges.at(vertex_idx)) {
        auto target_nested = nested;
        g->updateNestedVariantClusterInfo(&target_nested, G.vertices.at(target).variant_cluster_idx);
        runGibbsSample(O, G, target, target_nested, collect, trace_row);
    }
}
void estimateGenotypes(OracleGibbs &O, Group &G, bool collect, std::vector<uint32_t> *trace, uint32_t &sweep_no) {   // :220-234
    std::vector<NestedVariantClusterInfo> nested;
    for (auto &p : G.ploidy) nested.emplace_back(p);
    std::vector<uint32_t> row;
    bool tr = trace && sweep_no < O.trace_sweeps;
    if (tr) row.assign(G.vertices.size() * O.P.num_samples, 0xFFFFFFFFu);
    for (auto &sv : G.source_vertices) runGibbsSample(O, G, sv, nested, collect, tr ? &row : nullptr);
    if (tr) trace->insert(trace->end(), row.begin(), row.end());
    sweep_no++;
}

// InferenceEngine::estimateGenotypesCallback, per group (InferenceEngine.cpp:290-306)
void runGroupDefault(OracleGibbs &O, uint gi) {
    Group &G = O.groups[gi];
    uint32_t sweep_no = 0;
    std::vector<uint32_t> *trace = O.trace_sweeps ? &O.traces[gi] : nullptr;
    for (ushort chain = 0; chain < O.P.num_chains; chain++) {
        uint gseed = O.P.noise_seeding ? O.P.seed + (G.index + 1) * (chain + 1) : O.P.seed + (G.index + 1);
        initGenotyper(O, G, gseed);
        shuffleBranchOrdering(G, O.P.seed + (G.index + 1) * (chain + 1));
        for (ushort i = 0; i < O.P.burn_in; i++) estimateGenotypes(O, G, false, trace, sweep_no);
        for (ushort i = 0; i < O.P.num_iterations; i++) estimateGenotypes(O, G, true, trace, sweep_no);
    }
}

}  // namespace

extern "C" {

// CountDistribution LUTs from per-sample NB parameters (after size /= multiplicity) and noise rates
void orc_build_luts(unsigned S, const double *p, const double *size, const double *noise_rate, double *genomic, double *noise) {
    for (unsigned s = 0; s < S; s++) {
        for (unsigned m = 0; m < 256; m++)
            for (unsigned c = 0; c < 256; c++) genomic[((size_t)s * 256 + m) * 256 + c] = genomicCountLogPmf(p[s], size[s], (uchar)m, (uchar)c);
        for (unsigned c = 0; c < 256; c++) noise[s * 256 + c] = noiseCountLogPmf(noise_rate[s], (uchar)c);
    }
}
void orc_build_noise_lut(unsigned S, const double *noise_rate, double *noise) {
    for (unsigned s = 0; s < S; s++)
        for (unsigned c = 0; c < 256; c++) noise[s * 256 + c] = noiseCountLogPmf(noise_rate[s], (uchar)c);
}
void orc_nb_moments(double mean, double var, double *p, double *size) {
    auto r = nbMomentsToParameters(mean, var);
    *p = r.first;
    *size = r.second;
}
double orc_nb_logpmf(double p, double size, unsigned obs, unsigned scale) { return nbLogPmf(p, size, obs, scale); }
double orc_log_addition(double a, double b) { return logAddition(a, b); }
int orc_double_compare(double a, double b) { return doubleCompare(a, b) ? 1 : 0; }
void orc_logdiscrete_draws(const double *logw, unsigned n, unsigned seed, unsigned ndraws, unsigned *out) {
    std::mt19937 prng(seed);
    LogDiscreteSampler s;
    for (unsigned i = 0; i < n; i++) s.addOutcome(logw[i]);
    for (unsigned i = 0; i < ndraws; i++) out[i] = s.sample(&prng);
}
void orc_discrete_draws(const double *w, unsigned n, unsigned seed, unsigned ndraws, unsigned *out) {
    std::mt19937 prng(seed);
    DiscreteSampler s;
    for (unsigned i = 0; i < n; i++) s.addOutcome(w[i]);
    for (unsigned i = 0; i < ndraws; i++) out[i] = s.sample(&prng);
}
void orc_kmerstats(const double *values, unsigned n, unsigned *count, double *fraction, double *mean, double *var) {
    KmerStats ks;
    for (unsigned i = 0; i < n; i++) ks.addValue(std::make_pair(values[i], true));
    *count = ks.count;
    *fraction = ks.getFraction().first;
    *mean = ks.getMean().first;
    *var = ks.count < 2 ? -1 : ks.M2 / (ks.count - 1);
}
unsigned orc_sparsity_cover(const uint8_t *M, unsigned rows, unsigned cols, const uint8_t *row_mask, unsigned seed, unsigned *out) {
    std::vector<uchar> mask(row_mask, row_mask + rows);
    auto cover = estimateMinimumColumnCover(M, rows, cols, mask, seed);
    for (size_t i = 0; i < cover.size(); i++) out[i] = cover[i];
    return (unsigned)cover.size();
}
// libstdc++ facts the device code must reproduce
void orc_rng(unsigned seed, int kind, const double *a, const double *b, uint64_t n, double *out) {
    std::mt19937 prng(seed);
    if (kind == 0) for (uint64_t i = 0; i < n; i++) out[i] = (double)prng();
    else if (kind == 1) for (uint64_t i = 0; i < n; i++) out[i] = std::generate_canonical<double, std::numeric_limits<double>::digits>(prng);
    else if (kind == 2) {
        std::gamma_distribution<> g;
        for (uint64_t i = 0; i < n; i++) {
            g.param(std::gamma_distribution<>::param_type(a[i], b[i]));
            out[i] = g(prng);
        }
    } else if (kind == 3) {
        std::uniform_int_distribution<> u;
        for (uint64_t i = 0; i < n; i++) {
            u.param(std::uniform_int_distribution<>::param_type(0, (int)a[i]));
            out[i] = u(prng);
        }
    } else if (kind == 4) {
        std::bernoulli_distribution be((float)a[0]);
        for (uint64_t i = 0; i < n; i++) out[i] = be(prng) ? 1 : 0;
    } else if (kind == 5) {
        std::vector<uint> v((size_t)a[0]);
        for (size_t i = 0; i < v.size(); i++) v[i] = (uint)i;
        std::shuffle(v.begin(), v.end(), prng);
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    }
}
void orc_uset_replay(unsigned universe, const uint8_t *ops, const uint32_t *values, uint64_t num_ops, uint32_t *order, uint32_t *n) {
    std::unordered_set<uint> s;
    (void)universe;
    for (uint64_t i = 0; i < num_ops; i++) {
        if (ops[i] == 0) s.insert(values[i]);
        else if (ops[i] == 1) s.erase(values[i]);
        else s.clear();
    }
    uint32_t j = 0;
    for (auto v : s) order[j++] = v;
    *n = j;
}

void *orc_gibbs_create(const bt_gibbs_params *params, const bt_gibbs_batch *batch, const double *genomic, const double *noise) {
    OracleGibbs *O = new OracleGibbs();
    O->P = *params;
    O->gender.assign(params->gender, params->gender + params->num_samples);
    O->P.gender = nullptr;
    O->B = batch;
    const uint S = params->num_samples, C = batch->num_clusters;
    O->cd.S = S;
    O->cd.genomic.assign(genomic, genomic + (size_t)S * 65536);
    O->cd.noise.assign(noise, noise + (size_t)S * 256);
    O->mult_off.assign(C + 1, 0);
    O->kvbits_off.assign(C + 1, 0);
    O->hapvar_off.assign(C + 1, 0);
    O->hap_base.assign(C + 1, 0);
    O->var_base.assign(C + 1, 0);
    for (uint c = 0; c < C; c++) {
        const uint H = batch->num_haplotypes[c], V = batch->num_variants[c], K = batch->kmer_off[c + 1] - batch->kmer_off[c];
        O->mult_off[c + 1] = O->mult_off[c] + (uint64_t)K * H;
        const uint nnz = batch->kv_off[batch->kmer_off[c + 1]] - batch->kv_off[batch->kmer_off[c]];
        O->kvbits_off[c + 1] = O->kvbits_off[c] + (uint64_t)nnz * ((H + 31) / 32);
        O->hapvar_off[c + 1] = O->hapvar_off[c] + (uint64_t)H * V;
        O->hap_base[c + 1] = O->hap_base[c] + H;
        O->var_base[c + 1] = O->var_base[c] + V;
    }
    O->groups.resize(batch->num_groups);
    for (uint g = 0; g < batch->num_groups; g++) {
        Group &G = O->groups[g];
        G.index = batch->group_index[g];
        const uint c0 = batch->group_cluster_off[g], c1 = batch->group_cluster_off[g + 1];
        G.vertices.resize(c1 - c0);
        G.out_edges.resize(c1 - c0);
        for (uint c = c0; c < c1; c++) {
            G.vertices[c - c0].variant_cluster_idx = batch->cluster_idx[c];
            G.vertices[c - c0].cluster = c;
            G.out_edges[c - c0].assign(batch->edges + batch->edge_off[c], batch->edges + batch->edge_off[c + 1]);
        }
        G.source_vertices.assign(batch->group_sources + batch->group_source_off[g], batch->group_sources + batch->group_source_off[g + 1]);
        G.ploidy.assign(batch->group_ploidy + (size_t)g * S, batch->group_ploidy + (size_t)(g + 1) * S);
        G.shared.resize(batch->group_num_shared[g]);
        for (auto &sr : G.shared) sr.multiplicities.assign(S, 0);
    }
    return O;
}
void orc_gibbs_free(void *h) { delete (OracleGibbs *)h; }
void orc_gibbs_set_noise_lut(void *h, const double *noise) {
    OracleGibbs *O = (OracleGibbs *)h;
    O->cd.noise.assign(noise, noise + (size_t)O->P.num_samples * 256);
}
void orc_gibbs_trace_enable(void *h, uint32_t max_sweeps) {
    OracleGibbs *O = (OracleGibbs *)h;
    O->trace_sweeps = max_sweeps;
    O->traces.assign(O->groups.size(), std::vector<uint32_t>());
}
// default-mode schedule over all groups with `threads` workers (groups are independent)
void orc_gibbs_run(void *h, unsigned threads) {
    OracleGibbs *O = (OracleGibbs *)h;
    if (threads <= 1) {
        for (uint g = 0; g < O->groups.size(); g++) runGroupDefault(*O, g);
        return;
    }
    // groups are handed out in batch order from a shared counter, as the reference's worker threads pull group batches from its
    // ProducerConsumerQueue (InferenceEngine.cpp:335-382); batches arrive sorted by size, largest first (main.cpp:247)
    std::atomic<uint> next(0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++)
        pool.emplace_back([O, &next]() {
            for (uint g = next.fetch_add(1); g < O->groups.size(); g = next.fetch_add(1)) runGroupDefault(*O, g);
        });
    for (auto &th : pool) th.join();
}
// step-wise driving (noise drivers): InferenceEngine.cpp:60-98
void orc_gibbs_init_chain(void *h, uint32_t chain) {
    OracleGibbs *O = (OracleGibbs *)h;
    for (auto &G : O->groups) {
        uint gseed = O->P.noise_seeding ? O->P.seed + (G.index + 1) * (chain + 1) : O->P.seed + (G.index + 1);
        initGenotyper(*O, G, gseed);
        shuffleBranchOrdering(G, O->P.seed + (G.index + 1) * (chain + 1));
    }
}
void orc_gibbs_sweep(void *h, uint32_t n, int collect) {
    OracleGibbs *O = (OracleGibbs *)h;
    for (size_t gi = 0; gi < O->groups.size(); gi++) {   // (traces, when enabled, continue where the previous call stopped)
        Group &G = O->groups[gi];
        std::vector<uint32_t> *trace = O->trace_sweeps ? &O->traces[gi] : nullptr;
        uint32_t sweep_no = trace ? (uint32_t)(trace->size() / (G.vertices.size() * O->P.num_samples)) : 0xFFFFFFFFu;
        for (uint32_t i = 0; i < n; i++) estimateGenotypes(*O, G, collect != 0, trace, sweep_no);
    }
}
void orc_gibbs_noise_counts(void *h, uint64_t *hist, int zero_first) {   // getNoiseCounts + clearGenotyperCache
    OracleGibbs *O = (OracleGibbs *)h;
    if (zero_first) memset(hist, 0, (size_t)O->P.num_samples * 256 * 8);
    for (auto &G : O->groups)
        for (auto &vx : G.vertices) {
            vx.genotyper->getNoiseCounts(hist);
            vx.genotyper->clearCache();
        }
}
void orc_gibbs_reset_groups(void *h) {
    OracleGibbs *O = (OracleGibbs *)h;
    for (auto &G : O->groups)
        for (auto &vx : G.vertices) {
            delete vx.genotyper;
            vx.genotyper = nullptr;
        }
}
void orc_gibbs_result_sizes(void *h, uint64_t *n_dip, uint64_t *n_cells) {
    OracleGibbs *O = (OracleGibbs *)h;
    uint64_t nd = 0, nc = 0;
    for (auto &G : O->groups)
        for (auto &vx : G.vertices) {
            nd += vx.genotyper->diplotype_sampling_frequencies.size();
            for (auto &v : vx.genotyper->vinfo) nc += (uint64_t)v.num_alleles * O->P.num_samples;
        }
    *n_dip = nd;
    *n_cells = nc;
}
// same layout as bt_gibbs_result_fetch; diplotype entries sorted by (h1, h2) within a cluster
void orc_gibbs_result_fetch(void *h, uint64_t *dip_off, uint16_t *dip_h1, uint16_t *dip_h2, uint32_t *dip_freq, uint64_t *cell_off, double *stats) {
    OracleGibbs *O = (OracleGibbs *)h;
    const uint S = O->P.num_samples;
    uint64_t e = 0, cell = 0;
    for (auto &G : O->groups)
        for (auto &vx : G.vertices) {
            Genotyper *g = vx.genotyper;
            dip_off[vx.cluster] = e;
            cell_off[vx.cluster] = cell;
            for (auto &kv : g->diplotype_sampling_frequencies) {
                dip_h1[e] = kv.first.first;
                dip_h2[e] = kv.first.second;
                for (uint s = 0; s < S; s++) dip_freq[e * S + s] = kv.second[s];
                e++;
            }
            for (uint s = 0; s < S; s++)
                for (uint v = 0; v < g->vinfo.size(); v++)
                    for (uint a = 0; a < g->vinfo[v].num_alleles; a++) {
                        const AlleleKmerStats &ak = g->allele_kmer_stats[v][s];
                        const KmerStats *three[3] = {&ak.count_stats[a], &ak.fraction_stats[a], &ak.mean_stats[a]};
                        for (int t = 0; t < 3; t++) {
                            stats[cell * 12 + t * 4 + 0] = three[t]->count;
                            stats[cell * 12 + t * 4 + 1] = three[t]->fraction;
                            stats[cell * 12 + t * 4 + 2] = three[t]->mean;
                            stats[cell * 12 + t * 4 + 3] = three[t]->M2;
                        }
                        cell++;
                    }
        }
    dip_off[O->B->num_clusters] = e;
    cell_off[O->B->num_clusters] = cell;
}
// trace of group g: [sweep][vertex][S] words (0xFFFFFFFF for vertices not visited)
uint64_t orc_gibbs_trace_fetch(void *h, uint32_t group, uint32_t *out, uint64_t max_words) {
    OracleGibbs *O = (OracleGibbs *)h;
    auto &t = O->traces.at(group);
    uint64_t n = std::min<uint64_t>(t.size(), max_words);
    memcpy(out, t.data(), n * 4);
    return n;
}


// ---- VariantClusterGenotyper::getGenotypes for one cluster (VariantClusterGenotyper.cpp:208-567), from the arrays
// orc_gibbs_result_fetch returns.  Written along the reference's own structure: per-variant Genotypes objects with
// SampleStats built while iterating the diplotype_sampling_frequencies map. ----
namespace {
struct PairHash {
    size_t operator()(const std::pair<ushort, ushort> &p) const { return ((size_t)p.first << 16) ^ p.second; }
};
inline bool floatCompareO(const float a, const float b) { return ((a == b) or (std::abs(a - b) < std::abs(std::min(a, b)) * std::numeric_limits<float>::epsilon() * 100)); }
inline bool floatLessO(const float a, const float b) { return ((a < b) and !(floatCompareO(a, b))); }
}  // namespace
int orc_cluster_genotypes(unsigned S, unsigned H, unsigned V, const uint16_t *hap_allele, const uint16_t *var_num_alleles, const uint8_t *var_has_dependency,
                          unsigned long long num_diplotypes, const uint16_t *h1, const uint16_t *h2, const uint32_t *freq, const double *stats, const uint8_t *ploidy,
                          float min_gpp, float min_kmers, const float *min_fraction, unsigned Amax, float *gpp, float *app, uint16_t *filters, uint16_t *estimate,
                          uint32_t *gq, uint32_t *total_count, uint32_t *alt_counts, float *alt_freq, float *acp, float *max_alt_acp, uint8_t *non_covered) {
    const unsigned Gmax = Amax * (Amax + 1) / 2;
    std::unordered_map<std::pair<ushort, ushort>, std::vector<uint>, PairHash> diplotype_sampling_frequencies;
    for (unsigned long long e = 0; e < num_diplotypes; e++) diplotype_sampling_frequencies.emplace(std::make_pair(h1[e], h2[e]), std::vector<uint>(freq + e * S, freq + (e + 1) * S));
    uint allele_total = 0;
    std::vector<uint> allele_first(V);
    for (unsigned v = 0; v < V; v++) {
        allele_first[v] = allele_total;
        allele_total += var_num_alleles[v];
    }
    for (unsigned variant_idx = 0; variant_idx < V; variant_idx++) {
        const ushort num_alleles = var_num_alleles[variant_idx];
        auto haplotypeToAlleleIndex = [&](const ushort haplotype_idx) -> ushort {
            if (haplotype_idx != 0xFFFF) return hap_allele[(size_t)haplotype_idx * V + variant_idx];
            return num_alleles - 1;
        };
        {   // getNonCoveredAlleles
            std::vector<bool> is_allele_covered(num_alleles, false);
            for (unsigned h = 0; h < H; h++) is_allele_covered.at(hap_allele[(size_t)h * V + variant_idx]) = true;
            if (var_has_dependency[variant_idx]) is_allele_covered.back() = true;
            for (ushort a = 0; a < num_alleles; a++) non_covered[(size_t)variant_idx * Amax + a] = is_allele_covered[a] ? 0 : 1;
        }
        std::vector<std::vector<ushort>> estimates(S), sample_filters(S);
        std::vector<std::vector<float>> sample_allele_post(S);
        for (ushort sample_idx = 0; sample_idx < S; sample_idx++) {
            const size_t vs = (size_t)variant_idx * S + sample_idx;
            std::vector<float> genotype_posteriors, allele_posteriors;
            if (ploidy[sample_idx] == 2) {
                genotype_posteriors.assign((num_alleles * (num_alleles - 1)) / 2 + num_alleles, 0);
                allele_posteriors.assign(num_alleles, 0);
            } else if (ploidy[sample_idx] == 1) {
                genotype_posteriors.assign(num_alleles, 0);
                allele_posteriors.assign(num_alleles, 0);
            }
            std::vector<ushort> allele_filters(allele_posteriors.size(), 0);
            uint num_iterations = 0;
            std::pair<std::vector<std::pair<ushort, ushort>>, float> max_posterior_genotypes;
            max_posterior_genotypes.second = 0;
            for (auto &dsf : diplotype_sampling_frequencies) {
                if (dsf.second.at(sample_idx) > 0) {
                    std::pair<ushort, ushort> genotype_estimate(0xFFFF, 0xFFFF);
                    auto genotype_idx = genotype_estimate.first;
                    if (ploidy[sample_idx] == 2) {
                        genotype_estimate.first = haplotypeToAlleleIndex(dsf.first.first);
                        genotype_estimate.second = haplotypeToAlleleIndex(dsf.first.second);
                        if (genotype_estimate.first > genotype_estimate.second) std::swap(genotype_estimate.first, genotype_estimate.second);
                        genotype_idx = (genotype_estimate.second * (genotype_estimate.second + 1)) / 2 + genotype_estimate.first;
                        genotype_posteriors.at(genotype_idx) += dsf.second.at(sample_idx);
                        allele_posteriors.at(genotype_estimate.first) += dsf.second.at(sample_idx);
                        if (genotype_estimate.first != genotype_estimate.second) allele_posteriors.at(genotype_estimate.second) += dsf.second.at(sample_idx);
                    } else if (ploidy[sample_idx] == 1) {
                        genotype_estimate.first = haplotypeToAlleleIndex(dsf.first.first);
                        genotype_idx = genotype_estimate.first;
                        genotype_posteriors.at(genotype_estimate.first) += dsf.second.at(sample_idx);
                        allele_posteriors.at(genotype_estimate.first) += dsf.second.at(sample_idx);
                    }
                    num_iterations += dsf.second.at(sample_idx);
                    if (ploidy[sample_idx] != 0) {
                        if (floatCompareO(max_posterior_genotypes.second, genotype_posteriors.at(genotype_idx))) {
                            max_posterior_genotypes.first.push_back(genotype_estimate);
                        } else if (max_posterior_genotypes.second < genotype_posteriors.at(genotype_idx)) {
                            max_posterior_genotypes.first.clear();
                            max_posterior_genotypes.first.push_back(genotype_estimate);
                            max_posterior_genotypes.second = genotype_posteriors.at(genotype_idx);
                        }
                    }
                }
            }
            max_posterior_genotypes.second /= num_iterations;
            for (auto &posterior : genotype_posteriors) posterior /= num_iterations;
            for (auto &posterior : allele_posteriors) posterior /= num_iterations;
            for (ushort allele_idx = 0; allele_idx < allele_posteriors.size(); allele_idx++) {
                if (!floatCompareO(allele_posteriors.at(allele_idx), 0)) {
                    const double *cell = stats + ((size_t)sample_idx * allele_total + allele_first[variant_idx] + allele_idx) * 12;
                    const double count_mean = cell[2];      // count_stats.getMean().first
                    if (floatLessO(count_mean, min_kmers)) allele_filters.at(allele_idx) += 1;
                    if (!floatCompareO(count_mean, 0)) {
                        const double fraction_mean = cell[6];   // fraction_stats.getMean().first
                        if (floatLessO(fraction_mean, min_fraction[sample_idx])) allele_filters.at(allele_idx) += 2;
                    }
                }
            }
            uint genotype_quality;
            if (floatCompareO(max_posterior_genotypes.second, 1)) genotype_quality = 99;
            else if (floatCompareO(max_posterior_genotypes.second, 0)) genotype_quality = 0;
            else genotype_quality = -10 * std::log10(1 - max_posterior_genotypes.second);
            std::vector<ushort> genotype_estimate;
            if (ploidy[sample_idx] == 2) {
                genotype_estimate = std::vector<ushort>(2, 0xFFFF);
                if (max_posterior_genotypes.first.size() == 1) {
                    if (!floatLessO(max_posterior_genotypes.second, min_gpp)) {
                        if ((allele_filters.at(max_posterior_genotypes.first.front().first) == 0) and (allele_filters.at(max_posterior_genotypes.first.front().second) == 0)) {
                            genotype_estimate.front() = max_posterior_genotypes.first.front().first;
                            genotype_estimate.back() = max_posterior_genotypes.first.front().second;
                        }
                    }
                }
            } else if (ploidy[sample_idx] == 1) {
                genotype_estimate = std::vector<ushort>(1, 0xFFFF);
                if ((max_posterior_genotypes.first.size() == 1) and !floatLessO(max_posterior_genotypes.second, min_gpp)) {
                    if (allele_filters.at(max_posterior_genotypes.first.front().first) == 0) genotype_estimate.front() = max_posterior_genotypes.first.front().first;
                }
            }
            for (size_t i = 0; i < genotype_posteriors.size(); i++) gpp[vs * Gmax + i] = genotype_posteriors[i];
            for (size_t i = 0; i < allele_posteriors.size(); i++) {
                app[vs * Amax + i] = allele_posteriors[i];
                filters[vs * Amax + i] = allele_filters[i];
            }
            estimate[vs * 2] = estimate[vs * 2 + 1] = 0xFFFF;
            for (size_t i = 0; i < genotype_estimate.size(); i++) estimate[vs * 2 + i] = genotype_estimate[i];
            gq[vs] = genotype_quality;
            estimates[sample_idx] = genotype_estimate;
            sample_filters[sample_idx] = allele_filters;
            sample_allele_post[sample_idx] = allele_posteriors;
        }
        // getGenotypeVariantStats
        uint total = 0;
        std::vector<uint> alt_allele_counts(num_alleles - 1, 0);
        std::vector<float> allele_call_probabilities(num_alleles, 0);
        for (ushort sample_idx = 0; sample_idx < S; sample_idx++) {
            for (auto &allele_idx : estimates[sample_idx]) {
                if (allele_idx != 0xFFFF) {
                    total++;
                    if (allele_idx > 0) alt_allele_counts.at(allele_idx - 1)++;
                }
            }
            for (ushort allele_idx = 0; allele_idx < sample_allele_post[sample_idx].size(); allele_idx++)
                if (sample_filters[sample_idx].at(allele_idx) == 0)
                    allele_call_probabilities.at(allele_idx) = std::max(allele_call_probabilities.at(allele_idx), sample_allele_post[sample_idx].at(allele_idx));
        }
        float max_alt = 0;
        const ushort num_alt = num_alleles - 1 - (var_has_dependency[variant_idx] ? 1 : 0);
        for (ushort alt_allele_idx = 0; alt_allele_idx < num_alt; alt_allele_idx++) max_alt = std::max(max_alt, allele_call_probabilities.at(alt_allele_idx + 1));
        total_count[variant_idx] = total;
        max_alt_acp[variant_idx] = max_alt;
        for (ushort a = 0; a + 1 < num_alleles; a++) {
            alt_counts[(size_t)variant_idx * Amax + a] = alt_allele_counts[a];
            alt_freq[(size_t)variant_idx * Amax + a] = total > 0 ? alt_allele_counts[a] / static_cast<float>(total) : 0.f;
        }
        for (ushort a = 0; a < num_alleles; a++) acp[(size_t)variant_idx * Amax + a] = allele_call_probabilities[a];
    }
    return 0;
}


// The genotype-derived columns of GenotypeWriter's output line for every variant of a cluster (GenotypeWriter.cpp:174-230,261-345),
// from the arrays orc_cluster_genotypes fills.  One line per variant.
long long orc_cluster_output_columns(unsigned S, unsigned H, unsigned V, const uint16_t *hap_allele, const uint16_t *var_num_alleles, const uint8_t *var_has_dependency,
                                     unsigned long long num_diplotypes, const uint16_t *h1, const uint16_t *h2, const uint32_t *freq, const double *stats,
                                     const uint8_t *ploidy, float min_gpp, float min_kmers, const float *min_fraction, char *out, unsigned long long out_len) {
    unsigned Amax = 0, allele_total = 0;
    for (unsigned v = 0; v < V; v++) {
        Amax = std::max<unsigned>(Amax, var_num_alleles[v]);
        allele_total += var_num_alleles[v];
    }
    const unsigned Gmax = Amax * (Amax + 1) / 2;
    std::vector<float> gpp((size_t)V * S * Gmax, 0), app((size_t)V * S * Amax, 0), alt_freq((size_t)V * Amax, 0), acp((size_t)V * Amax, 0), max_alt(V, 0);
    std::vector<uint16_t> filters((size_t)V * S * Amax, 0), estimate((size_t)V * S * 2, 0);
    std::vector<uint32_t> gq((size_t)V * S, 0), total_count(V, 0), alt_counts((size_t)V * Amax, 0);
    std::vector<uint8_t> non_covered((size_t)V * Amax, 0);
    orc_cluster_genotypes(S, H, V, hap_allele, var_num_alleles, var_has_dependency, num_diplotypes, h1, h2, freq, stats, ploidy, min_gpp, min_kmers, min_fraction, Amax,
                          gpp.data(), app.data(), filters.data(), estimate.data(), gq.data(), total_count.data(), alt_counts.data(), alt_freq.data(), acp.data(),
                          max_alt.data(), non_covered.data());
    std::stringstream line;
    unsigned allele_first = 0;
    for (unsigned v = 0; v < V; v++) {
        const unsigned num_alleles = var_num_alleles[v];
        // writeQualityAndFilter
        if (floatCompareO(max_alt[v], 1)) line << "99";
        else if (floatCompareO(max_alt[v], 0)) line << "0";
        else line << -10 * std::log10(1 - max_alt[v]);
        line << (total_count[v] == 0 ? "\tAN0" : "\tPASS");
        // writeVariantStats
        line << "\tAC=";
        for (unsigned a = 0; a + 1 < num_alleles; a++) line << (a ? "," : "") << alt_counts[(size_t)v * Amax + a];
        line << ";AF=";
        for (unsigned a = 0; a + 1 < num_alleles; a++) line << (a ? "," : "") << alt_freq[(size_t)v * Amax + a];
        line << ";AN=" << total_count[v] << ";ACP=";
        for (unsigned a = 0; a < num_alleles; a++) line << (a ? "," : "") << acp[(size_t)v * Amax + a];
        // writeAlleleCover
        bool first = true;
        for (unsigned a = 0; a < num_alleles; a++)
            if (non_covered[(size_t)v * Amax + a]) {
                line << (first ? ";ANC=" : ",") << a;
                first = false;
            }
        // writeSamples
        for (unsigned s = 0; s < S; s++) {
            const size_t vs = (size_t)v * S + s;
            line << "\t";
            if (ploidy[s] == 0) {
                line << ":" << ".:.:.:.:.:.";
                continue;
            }
            for (unsigned i = 0; i < ploidy[s]; i++) {
                if (i > 0) line << "/";
                if (estimate[vs * 2 + i] != 0xFFFF) line << estimate[vs * 2 + i];
                else line << ".";
            }
            line << ":" << gq[vs] << ":";
            const unsigned ng = ploidy[s] == 2 ? (num_alleles * (num_alleles - 1)) / 2 + num_alleles : num_alleles;
            for (unsigned g = 0; g < ng; g++) line << (g ? "," : "") << gpp[vs * Gmax + g];
            line << ":";
            for (unsigned a = 0; a < num_alleles; a++) line << (a ? "," : "") << app[vs * Amax + a];
            line << ":";
            std::stringstream allele_kmer_counts, allele_kmer_fractions, allele_kmer_means;
            for (unsigned a = 0; a < num_alleles; a++) {
                const double *cell = stats + ((size_t)s * allele_total + allele_first + a) * 12;
                if (a) {
                    allele_kmer_counts << ",";
                    allele_kmer_fractions << ",";
                    allele_kmer_means << ",";
                }
                allele_kmer_counts << (cell[0] == 0 ? -1.0 : cell[2]);        // KmerStats::getMean(): (-1, false) when empty
                allele_kmer_fractions << (cell[4] == 0 ? -1.0 : cell[6]);
                allele_kmer_means << (cell[8] == 0 ? -1.0 : cell[10]);
            }
            line << allele_kmer_counts.str() << ":" << allele_kmer_fractions.str() << ":" << allele_kmer_means.str() << ":";
            for (unsigned a = 0; a < num_alleles; a++) line << (a ? "," : "") << filters[vs * Amax + a];
        }
        line << "\n";
        allele_first += num_alleles;
    }
    const std::string all = line.str();
    if (out && out_len >= all.size()) memcpy(out, all.data(), all.size());
    return (long long)all.size();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Noise drivers: InferenceEngine::estimateNoise (InferenceEngine.cpp:135-276) and ::estimateNoiseAndGenotypes (:384-472)
// with the noise half of CountDistribution (CountDistribution.cpp:51-67 ctor, :163-171 resetNoiseRates, :173-186
// sampleNoiseParameters, :188-199 calcCountSuffStats, :201-213 sampleGamma).  Single-threaded: the reference's worker
// threads only add per-group histograms (integers) and touch disjoint groups, so thread count does not change results.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct NoiseModel {
    std::vector<std::pair<float, float>> noise_rate_priors;
    std::mt19937 prng;
    std::gamma_distribution<> gamma_dist;
    std::vector<double> noise_rates;
    NoiseModel(uint S, float shape, float scale, uint seed) : noise_rate_priors(S, std::make_pair(shape, scale)), noise_rates(S, 0) {
        prng = std::mt19937(seed);
        resetNoiseRates();
    }
    double sampleGamma(const double shape, const double scale) {
        gamma_dist.param(std::gamma_distribution<>::param_type(shape, scale));
        return gamma_dist(prng);
    }
    void resetNoiseRates() {
        for (uint s = 0; s < noise_rates.size(); s++) noise_rates[s] = sampleGamma(noise_rate_priors[s].first, noise_rate_priors[s].second);
    }
    void sampleNoiseParameters(const uint64_t *hist) {
        for (uint s = 0; s < noise_rates.size(); s++) {
            ulong num_observations = 0, count_sum = 0;
            for (uint i = 0; i < 256; i++) {
                num_observations += hist[s * 256 + i];
                count_sum += i * hist[s * 256 + i];
            }
            noise_rates[s] = sampleGamma(noise_rate_priors[s].first + count_sum, noise_rate_priors[s].second / (num_observations * noise_rate_priors[s].second + 1));
        }
    }
};
void updateNoiseCache(OracleGibbs &O, const NoiseModel &nm) {
    for (uint s = 0; s < nm.noise_rates.size(); s++)
        for (uint c = 0; c < 256; c++) O.cd.noise[s * 256 + c] = noiseCountLogPmf(nm.noise_rates[s], (uchar)c);
}
void traceRow(std::vector<double> *trace, uint chain, uint iteration, const std::vector<double> &rates) {
    if (!trace) return;
    trace->push_back(chain);
    trace->push_back(iteration);
    trace->insert(trace->end(), rates.begin(), rates.end());
}
uint groupVariants(const OracleGibbs &O, const Group &G) {
    uint n = 0;
    for (auto &vx : G.vertices) n += O.B->num_variants[vx.cluster];
    return n;
}
}  // namespace

extern "C" {

// h: an orc_gibbs_create()d unit with params.noise_seeding = 1.  rows of (chain, iteration, rate_0..rate_{S-1}) are written to
// trace (capacity trace_cap doubles) exactly as the reference writes its noise parameter file; selected (optional, capacity
// selected_cap) receives, per chain, the number of groups followed by their indices.  Returns the number of trace doubles.
uint64_t orc_estimate_noise(void *h, float prior_shape, float prior_scale, uint32_t noise_variants_batch_size, double *trace, uint64_t trace_cap,
                            uint32_t *selected, uint64_t selected_cap, double *final_rates) {
    OracleGibbs &O = *(OracleGibbs *)h;
    const uint S = O.P.num_samples;
    NoiseModel cd(S, prior_shape, prior_scale, O.P.seed);
    updateNoiseCache(O, cd);
    std::vector<double> tr;
    std::vector<uint32_t> sel;
    std::vector<uint> noise_group_indices;
    for (uint g = 0; g < O.groups.size(); g++)
        if (O.groups[g].vertices.size() == 1) noise_group_indices.push_back(g);
    std::vector<double> mean_noise_rates(S, 0);
    std::mt19937 prng = std::mt19937(O.P.seed);
    for (ushort chain = 0; chain < O.P.num_chains; chain++) {
        std::shuffle(noise_group_indices.begin(), noise_group_indices.end(), prng);
        uint end = 0, num_noise_variants = 0;
        while ((num_noise_variants < noise_variants_batch_size) && (end < noise_group_indices.size())) {
            num_noise_variants += groupVariants(O, O.groups[noise_group_indices[end]]);
            end++;
        }
        std::sort(noise_group_indices.begin(), noise_group_indices.begin() + end);
        sel.push_back(end);
        for (uint j = 0; j < end; j++) sel.push_back(O.groups[noise_group_indices[j]].index);
        for (uint j = 0; j < end; j++) {   // initGenotypersCallback :60-75
            Group &G = O.groups[noise_group_indices[j]];
            initGenotyper(O, G, O.P.seed + (G.index + 1) * (chain + 1));
            shuffleBranchOrdering(G, O.P.seed + (G.index + 1) * (chain + 1));
        }
        traceRow(&tr, chain + 1, 0, cd.noise_rates);
        for (uint iteration = 1; iteration <= (uint)O.P.burn_in + O.P.num_iterations; iteration++) {
            std::vector<uint64_t> hist((size_t)S * 256, 0);
            for (uint j = 0; j < end; j++) {   // sampleGenotypesCallback :77-98
                Group &G = O.groups[noise_group_indices[j]];
                uint32_t sweep_no = 0xFFFFFFFFu;
                estimateGenotypes(O, G, false, nullptr, sweep_no);
                for (auto &vx : G.vertices) {
                    vx.genotyper->getNoiseCounts(hist.data());
                    vx.genotyper->clearCache();
                }
            }
            cd.sampleNoiseParameters(hist.data());
            updateNoiseCache(O, cd);
            traceRow(&tr, chain + 1, iteration, cd.noise_rates);
            if (O.P.burn_in < iteration)
                for (uint s = 0; s < S; s++) mean_noise_rates[s] += cd.noise_rates[s];
        }
        for (uint j = 0; j < end; j++)   // resetGroupsCallback :100-113
            for (auto &vx : O.groups[noise_group_indices[j]].vertices) {
                delete vx.genotyper;
                vx.genotyper = nullptr;
            }
        cd.resetNoiseRates();
        updateNoiseCache(O, cd);
    }
    for (uint s = 0; s < S; s++) mean_noise_rates[s] /= O.P.num_iterations * O.P.num_chains;
    cd.noise_rates = mean_noise_rates;   // setNoiseRates
    updateNoiseCache(O, cd);
    traceRow(&tr, 0, 0, cd.noise_rates);
    if (final_rates) std::copy(mean_noise_rates.begin(), mean_noise_rates.end(), final_rates);
    if (trace) std::copy(tr.begin(), tr.begin() + std::min<uint64_t>(tr.size(), trace_cap), trace);
    if (selected) std::copy(sel.begin(), sel.begin() + std::min<uint64_t>(sel.size(), selected_cap), selected);
    return tr.size();
}

// estimateNoiseAndGenotypes over every group of the unit; afterwards orc_gibbs_result_* hold the collected samples
// threads > 1: the groups of an iteration are dealt to worker threads as the reference's estimateNoiseAndGenotypes hands group batches
// to its threads (InferenceEngine.cpp:424-447); every thread tallies its own noise counts, summed after the join (integer sums: any order)
uint64_t orc_estimate_noise_and_genotypes_mt(void *h, float prior_shape, float prior_scale, double *trace, uint64_t trace_cap, unsigned threads);
uint64_t orc_estimate_noise_and_genotypes(void *h, float prior_shape, float prior_scale, double *trace, uint64_t trace_cap) {
    return orc_estimate_noise_and_genotypes_mt(h, prior_shape, prior_scale, trace, trace_cap, 1);
}
uint64_t orc_estimate_noise_and_genotypes_mt(void *h, float prior_shape, float prior_scale, double *trace, uint64_t trace_cap, unsigned threads) {
    OracleGibbs &O = *(OracleGibbs *)h;
    const uint S = O.P.num_samples;
    NoiseModel cd(S, prior_shape, prior_scale, O.P.seed);
    updateNoiseCache(O, cd);
    std::vector<double> tr;
    for (ushort chain = 0; chain < O.P.num_chains; chain++) {
        for (auto &G : O.groups) {
            initGenotyper(O, G, O.P.seed + (G.index + 1) * (chain + 1));
            shuffleBranchOrdering(G, O.P.seed + (G.index + 1) * (chain + 1));
        }
        traceRow(&tr, chain + 1, 0, cd.noise_rates);
        for (uint iteration = 1; iteration <= (uint)O.P.burn_in + O.P.num_iterations; iteration++) {
            std::vector<uint64_t> hist((size_t)S * 256, 0);
            auto one_group = [&](Group &G, uint64_t *into) {
                uint32_t sweep_no = 0xFFFFFFFFu;
                estimateGenotypes(O, G, iteration > O.P.burn_in, nullptr, sweep_no);
                for (auto &vx : G.vertices) {
                    vx.genotyper->getNoiseCounts(into);
                    vx.genotyper->clearCache();
                }
            };
            if (threads <= 1) {
                for (auto &G : O.groups) one_group(G, hist.data());
            } else {
                std::atomic<size_t> next(0);
                std::vector<std::vector<uint64_t>> part(threads, std::vector<uint64_t>((size_t)S * 256, 0));
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < threads; t++)
                    pool.emplace_back([&, t]() {
                        for (size_t g = next.fetch_add(1); g < O.groups.size(); g = next.fetch_add(1)) one_group(O.groups[g], part[t].data());
                    });
                for (auto &th : pool) th.join();
                for (auto &p : part)
                    for (size_t i = 0; i < hist.size(); i++) hist[i] += p[i];
            }
            cd.sampleNoiseParameters(hist.data());
            updateNoiseCache(O, cd);
            traceRow(&tr, chain + 1, iteration, cd.noise_rates);
        }
        cd.resetNoiseRates();
        updateNoiseCache(O, cd);
    }
    if (trace) std::copy(tr.begin(), tr.begin() + std::min<uint64_t>(tr.size(), trace_cap), trace);
    return tr.size();
}

}  // extern "C"
