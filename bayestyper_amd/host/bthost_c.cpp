// C entry points of libbthost.so used by the Python tests and bench.py (ctypes).  The C++ classes in this directory
// are the host layer proper; this file only exposes them.
#include <cstring>

#include "CountDistribution.hpp"

using namespace bthost;

extern "C" {

// LUTs for S samples from per-sample (mean, var, multiplicity) of the parameter k-mers and explicit noise rates
int bth_build_luts(unsigned S, const double *mean, const double *var, const unsigned *multiplicity, const double *noise_rates, double *genomic, double *noise) {
    try {
        CountDistribution cd((unsigned short)S, std::make_pair(1.0f, 0.01f), 0);
        for (unsigned s = 0; s < S; s++) cd.setGenomicFromMoments((unsigned short)s, mean[s], var[s], multiplicity ? multiplicity[s] : 1);
        cd.setNoiseRates(std::vector<double>(noise_rates, noise_rates + S));
        std::memcpy(genomic, cd.genomicTable().data(), (size_t)S * 65536 * 8);
        std::memcpy(noise, cd.noiseTable().data(), (size_t)S * 256 * 8);
        return 0;
    } catch (...) {
        return 1;
    }
}

void *bth_count_distribution_new(unsigned S, float prior_shape, float prior_scale, unsigned seed) {
    return new CountDistribution((unsigned short)S, std::make_pair(prior_shape, prior_scale), seed);
}
void bth_count_distribution_free(void *h) { delete (CountDistribution *)h; }
void bth_count_distribution_set_genomic(void *h, unsigned s, double mean, double var, unsigned multiplicity) {
    ((CountDistribution *)h)->setGenomicFromMoments((unsigned short)s, mean, var, multiplicity);
}
void bth_count_distribution_noise_rates(void *h, double *out) {
    auto &r = ((CountDistribution *)h)->getNoiseRates();
    std::memcpy(out, r.data(), r.size() * 8);
}
void bth_count_distribution_set_noise_rates(void *h, const double *rates, unsigned S) { ((CountDistribution *)h)->setNoiseRates(std::vector<double>(rates, rates + S)); }
void bth_count_distribution_reset_noise_rates(void *h) { ((CountDistribution *)h)->resetNoiseRates(); }
// sampleNoiseParameters from a [S*256] u64 histogram
void bth_count_distribution_sample_noise(void *h, const unsigned long long *hist, unsigned S) {
    CountAllocation ca((unsigned short)S);
    for (unsigned s = 0; s < S; s++)
        for (unsigned i = 0; i < 256; i++) ca.counts()[s][i] = hist[s * 256 + i];
    ((CountDistribution *)h)->sampleNoiseParameters(ca);
}
void bth_count_distribution_tables(void *h, double *genomic, double *noise, unsigned S) {
    auto *cd = (CountDistribution *)h;
    if (genomic) std::memcpy(genomic, cd->genomicTable().data(), (size_t)S * 65536 * 8);
    if (noise) std::memcpy(noise, cd->noiseTable().data(), (size_t)S * 256 * 8);
}

}  // extern "C"
