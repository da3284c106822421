#include "CountDistribution.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace bthost {

double logAddition(double a, double b) {
    if (a < b) return b + std::log1p(std::exp(a - b));
    return a + std::log1p(std::exp(b - a));
}
bool doubleCompare(double a, double b) {
    return (a == b) || (std::abs(a - b) < std::abs(std::min(a, b)) * std::numeric_limits<double>::epsilon() * 100);
}
bool floatCompare(float a, float b) { return (a == b) || (std::abs(a - b) < std::abs(std::min(a, b)) * std::numeric_limits<float>::epsilon() * 100); }
bool floatLess(float a, float b) { return (a < b) && !floatCompare(a, b); }

static const double kMaxP = 0.99;

NegativeBinomialDistribution::NegativeBinomialDistribution() : p_(kMaxP), size_(kMaxP / (1 - kMaxP)) {}
NegativeBinomialDistribution::NegativeBinomialDistribution(const std::pair<double, double> &parameters) { setParameters(parameters); }

std::pair<double, double> NegativeBinomialDistribution::momentsToParameters(double mean, double var) {
    if (kMaxP < (mean / var)) var = mean / kMaxP;
    const double p = mean / var;
    const double size = std::pow(mean, 2) / (var - mean);
    return std::make_pair(p, size);
}
void NegativeBinomialDistribution::setParameters(const std::pair<double, double> &parameters) {
    if (!(parameters.first > 0 && parameters.first < 1 && parameters.second > 0)) throw std::invalid_argument("negative binomial parameters out of range");
    p_ = parameters.first;
    size_ = parameters.second;
}
double NegativeBinomialDistribution::mean() const { return size_ * (1 - p_) / p_; }
double NegativeBinomialDistribution::var() const { return size_ * (1 - p_) / std::pow(p_, 2); }
double NegativeBinomialDistribution::logPmf(unsigned obs, unsigned size_scale) const {
    const double coef = std::lgamma(obs + size_ * size_scale) - std::lgamma(size_ * size_scale) - std::lgamma(obs + 1);
    return coef + std::log(p_) * size_ * size_scale + std::log(1 - p_) * obs;
}

CountAllocation::CountAllocation(unsigned short num_samples) : sample_counts(num_samples, std::vector<unsigned long>(256, 0)) {}
void CountAllocation::addCount(unsigned short sample_idx, unsigned char count) { sample_counts.at(sample_idx).at(count)++; }
void CountAllocation::mergeInCountAllocations(const CountAllocation &other) {
    assert(sample_counts.size() == other.sample_counts.size());
    for (size_t s = 0; s < sample_counts.size(); s++)
        for (size_t i = 0; i < 256; i++) sample_counts[s][i] += other.sample_counts[s][i];
}

CountDistribution::CountDistribution(unsigned short num_samples, std::pair<float, float> noise_rate_prior, unsigned random_seed)
    : S(num_samples), noise_rate_priors(num_samples, noise_rate_prior), genomic(num_samples), noise_rates(num_samples, 0),
      genomic_cache((size_t)num_samples * 65536), noise_cache((size_t)num_samples * 256) {
    prng = std::mt19937(random_seed);
    resetNoiseRates();
    updateGenomicCache();
}

double CountDistribution::calcCountLogProb(unsigned short s, unsigned char, unsigned char multiplicity, unsigned char count) const {
    if (multiplicity == 0) return noise_cache[(size_t)s * 256 + count];
    return genomic_cache[((size_t)s * 256 + multiplicity) * 256 + count];
}

void CountDistribution::setGenomicParameters(unsigned short s, const std::pair<double, double> &nb) {
    genomic.at(s).setParameters(nb);
    updateGenomicCache();
}
void CountDistribution::setGenomicFromMoments(unsigned short s, double mean, double var, unsigned multiplicity) {
    auto nb = NegativeBinomialDistribution::momentsToParameters(mean, var);
    nb.second /= multiplicity;
    setGenomicParameters(s, nb);
}
void CountDistribution::setNoiseRates(const std::vector<double> &rates) {
    if (rates.size() != S) throw std::invalid_argument("noise rate vector has the wrong size");
    noise_rates = rates;
    updateNoiseCache();
}
void CountDistribution::resetNoiseRates() {
    for (unsigned s = 0; s < S; s++) noise_rates[s] = sampleGamma(noise_rate_priors[s].first, noise_rate_priors[s].second);
    updateNoiseCache();
}
void CountDistribution::sampleNoiseParameters(const CountAllocation &noise_counts) {
    for (unsigned s = 0; s < S; s++) {
        const auto &counts = noise_counts.getCounts().at(s);
        unsigned long num_observations = 0, count_sum = 0;   // calcCountSuffStats (:188-200)
        for (unsigned i = 0; i < counts.size(); i++) {
            num_observations += counts[i];
            count_sum += i * counts[i];
        }
        noise_rates[s] = sampleGamma(noise_rate_priors[s].first + count_sum, noise_rate_priors[s].second / (num_observations * noise_rate_priors[s].second + 1));
    }
    updateNoiseCache();
}
// (the stream operators are the standard's way to reach the state; doubles are written with max_digits10, so the round trip is exact)
void CountDistribution::exportGenerator(uint32_t *mt624, uint32_t *mt_pos, uint32_t *saved_available, double *saved) const {
    std::stringstream ss;
    ss << prng;
    for (int i = 0; i < 624; i++) {
        unsigned long w;
        ss >> w;
        mt624[i] = (uint32_t)w;
    }
    unsigned long p;
    ss >> p;
    *mt_pos = (uint32_t)p;
    std::stringstream gs;
    gs << gamma_dist;   // alpha beta | mean stddev saved_available [saved]
    double alpha, beta, mean, stddev;
    int avail = 0;
    gs >> alpha >> beta >> mean >> stddev >> avail;
    *saved_available = avail ? 1u : 0u;
    *saved = 0;
    if (avail) gs >> *saved;
    if (!ss || !gs) throw std::runtime_error("CountDistribution: cannot read the generator state");
}
void CountDistribution::importGenerator(const uint32_t *mt624, uint32_t mt_pos, uint32_t saved_available, double saved) {
    std::stringstream ss;
    for (int i = 0; i < 624; i++) ss << mt624[i] << ' ';
    ss << mt_pos;
    ss >> prng;
    std::stringstream gs;
    gs.precision(std::numeric_limits<double>::max_digits10);
    gs << std::scientific << 1.0 << ' ' << 1.0 << ' ' << 0.0 << ' ' << 1.0 << ' ' << (saved_available ? 1 : 0);
    if (saved_available) gs << ' ' << saved;
    gs >> gamma_dist;
    if (!ss || !gs) throw std::runtime_error("CountDistribution: cannot set the generator state");
}
double CountDistribution::sampleGamma(double shape, double scale) {
    gamma_dist.param(std::gamma_distribution<>::param_type(shape, scale));
    return gamma_dist(prng);
}
void CountDistribution::updateGenomicCache() {
    for (unsigned s = 0; s < S; s++)
        for (unsigned m = 0; m < 256; m++)
            for (unsigned c = 0; c < 256; c++) genomic_cache[((size_t)s * 256 + m) * 256 + c] = genomicCountLogPmf(s, (unsigned char)m, (unsigned char)c);
}
void CountDistribution::updateNoiseCache() {
    for (unsigned s = 0; s < S; s++)
        for (unsigned c = 0; c < 256; c++) noise_cache[(size_t)s * 256 + c] = noiseCountLogPmf(s, (unsigned char)c);
}
// tail mass of counts >= 255 is folded into count 255 by iterating logAddition to convergence (CountDistribution.cpp:267-312)
double CountDistribution::genomicCountLogPmf(unsigned short s, unsigned char multiplicity, unsigned char count) const {
    if (multiplicity == 0) return count == 0 ? 0 : -std::numeric_limits<double>::infinity();
    double v = genomic[s].logPmf(count, multiplicity);
    if (count == 255) {
        unsigned limit = count;
        double prev = 0;
        do {
            limit++;
            prev = v;
            v = logAddition(v, genomic[s].logPmf(limit, multiplicity));
            if (v > 0) {
                v = 0;
                break;
            }
        } while (!doubleCompare(prev, v));
    }
    return v;
}
static double poissonLogProb(unsigned value, double rate) { return value * std::log(rate) - rate - std::lgamma(value + 1); }   // :349-352
double CountDistribution::noiseCountLogPmf(unsigned short s, unsigned char count) const {   // :314-347
    double v = poissonLogProb(count, noise_rates[s]);
    if (count == 255) {
        unsigned limit = count;
        double prev = 0;
        do {
            limit++;
            prev = v;
            v = logAddition(v, poissonLogProb(limit, noise_rates[s]));
            if (v > 0) {
                v = 0;
                break;
            }
        } while (!doubleCompare(prev, v));
    }
    return v;
}

}  // namespace bthost
