// Host-side count model of the genotyping path: mirrors the reference's CountDistribution /
// NegativeBinomialDistribution / CountAllocation interface (include/bayesTyper/CountDistribution.hpp:53-63,
// NegativeBinomialDistribution.hpp, CountAllocation.hpp) for the pieces the GPU path needs.
//
// The sampler on the device only GATHERS from two log-pmf tables; they are computed here in fp64 exactly as the
// reference computes its caches (src/bayesTyper/CountDistribution.cpp:215-352) and uploaded with bt_gibbs_set_lut.
#pragma once
#include <cstdint>
#include <random>
#include <string>
#include <utility>
#include <vector>

namespace bthost {

class NegativeBinomialDistribution {
  public:
    NegativeBinomialDistribution();                                        // p = 0.99, size = p/(1-p)
    explicit NegativeBinomialDistribution(const std::pair<double, double> &parameters);
    static std::pair<double, double> momentsToParameters(double mean, double var);   // NegativeBinomialDistribution.cpp:68-79
    void setParameters(const std::pair<double, double> &parameters);
    double p() const { return p_; }
    double size() const { return size_; }
    double mean() const;
    double var() const;
    double logPmf(unsigned obs, unsigned size_scale) const;               // NegativeBinomialDistribution.cpp:122-147
  private:
    double p_, size_;
};

// 256-bin per-sample histogram of noise k-mer counts (CountAllocation.cpp:34-57)
class CountAllocation {
  public:
    explicit CountAllocation(unsigned short num_samples);
    void addCount(unsigned short sample_idx, unsigned char count);
    void mergeInCountAllocations(const CountAllocation &other);
    const std::vector<std::vector<unsigned long>> &getCounts() const { return sample_counts; }
    std::vector<std::vector<unsigned long>> &counts() { return sample_counts; }
  private:
    std::vector<std::vector<unsigned long>> sample_counts;
};

class CountDistribution {
  public:
    // noise_rate_prior = (shape, scale) of --noise-rate-prior (default 1,0.01); prng seeded with --random-seed
    CountDistribution(unsigned short num_samples, std::pair<float, float> noise_rate_prior, unsigned random_seed);

    // calcCountLogProb(sample, bias_idx = 0, multiplicity, count)  (CountDistribution.cpp:255-265)
    double calcCountLogProb(unsigned short sample_idx, unsigned char bias_idx, unsigned char multiplicity, unsigned char count) const;
    void sampleNoiseParameters(const CountAllocation &noise_counts);      // CountDistribution.cpp:173-186
    void setGenomicParameters(unsigned short sample_idx, const std::pair<double, double> &nb_parameters);   // after size /= multiplicity (:118-121)
    // fit from the mean/variance of the parameter k-mers of the best-populated multiplicity (CountDistribution.cpp:95-126)
    void setGenomicFromMoments(unsigned short sample_idx, double mean, double var, unsigned multiplicity);
    const std::vector<double> &getNoiseRates() const { return noise_rates; }
    void setNoiseRates(const std::vector<double> &rates);
    void resetNoiseRates();                                               // CountDistribution.cpp:163-171
    const std::vector<NegativeBinomialDistribution> &getGenomicCountDistributions() const { return genomic; }

    // flat tables in the layout of include/btgpu.h: genomic[(s*256+m)*256+c], noise[s*256+c]
    const std::vector<double> &genomicTable() const { return genomic_cache; }
    const std::vector<double> &noiseTable() const { return noise_cache; }

    // The generator the noise rates are drawn from, in libstdc++'s own terms (std::mt19937: 624 words + next index; the gamma distribution's
    // normal distribution: saved variate) — handed to the device for a whole chain of draws (bt_gibbs_noise_chain) and taken back afterwards.
    void exportGenerator(uint32_t *mt624, uint32_t *mt_pos, uint32_t *saved_available, double *saved) const;
    void importGenerator(const uint32_t *mt624, uint32_t mt_pos, uint32_t saved_available, double saved);
    const std::vector<std::pair<float, float>> &noiseRatePriors() const { return noise_rate_priors; }

  private:
    double sampleGamma(double shape, double scale);
    void updateGenomicCache();
    void updateNoiseCache();
    double genomicCountLogPmf(unsigned short s, unsigned char multiplicity, unsigned char count) const;
    double noiseCountLogPmf(unsigned short s, unsigned char count) const;

    unsigned short S;
    std::vector<std::pair<float, float>> noise_rate_priors;
    std::mt19937 prng;
    std::gamma_distribution<> gamma_dist;
    std::vector<NegativeBinomialDistribution> genomic;
    std::vector<double> noise_rates;
    std::vector<double> genomic_cache, noise_cache;
};

// Utils::logAddition / doubleCompare (include/bayesTyper/Utils.hpp:81-124)
double logAddition(double a, double b);
bool doubleCompare(double a, double b);
bool floatCompare(float a, float b);
bool floatLess(float a, float b);

}  // namespace bthost
