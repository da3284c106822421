// Deterministic host pieces of the reference's noise drivers (src/bayesTyper/InferenceEngine.cpp:135-276, 384-472):
// which groups a chain of estimateNoise samples, and the rows of <prefix>_noise_parameters.txt.  The iteration loop itself
// (launch a sweep on the GPU, add up the noise-count histograms — across ranks too —, draw the noise rates, upload the noise
// table) lives in bayestyper_amd/host/inference_engine.py, above the C ABI of libbtgpu.so.
#pragma once
#include <cstdint>
#include <random>
#include <string>
#include <vector>

namespace bthost {

// InferenceEngine.cpp:141-151 (candidates: groups with exactly one variant cluster, in group order) and :172-189 (per chain:
// std::shuffle of the WHOLE candidate vector with one mt19937(seed) that lives across chains, then the shortest prefix with
// >= noise_variants_batch_size variants, sorted in place — the partly sorted vector is what the next chain shuffles).
class NoiseGroupSelector {
  public:
    static constexpr uint32_t noise_variants_batch_size = 100000;   // InferenceEngine.cpp:50
    NoiseGroupSelector(const uint32_t *clusters_per_group, const uint32_t *variants_per_group, uint32_t num_groups, unsigned prng_seed,
                       uint32_t variants_batch_size = noise_variants_batch_size);
    // group indices (ascending) the next chain runs on
    std::vector<uint32_t> nextChain();
    uint32_t numCandidates() const { return (uint32_t)noise_group_indices.size(); }
    uint32_t lastNumVariants() const { return num_noise_variants; }   // < batch size => the reference prints its "low number of variants" warning (:270-274)

  private:
    std::vector<uint32_t> noise_group_indices;
    std::vector<uint32_t> variants;   // per group of the unit
    std::mt19937 prng;
    uint32_t batch_size, num_noise_variants = 0;
};

// one row of the noise parameter file: "<chain>\t<iteration>\t<rate_0>\t...\n" with the default ostream formatting (6 significant
// digits) the reference writes with (InferenceEngine.cpp:205,236,267; Utils.hpp:209-224)
std::string noiseParameterHeader(const std::vector<std::string> &sample_names);
std::string noiseParameterRow(unsigned chain, unsigned iteration, const std::vector<double> &rates);

}  // namespace bthost
