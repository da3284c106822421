#include "InferenceEngine.hpp"

#include <algorithm>
#include <sstream>

namespace bthost {

NoiseGroupSelector::NoiseGroupSelector(const uint32_t *clusters_per_group, const uint32_t *variants_per_group, uint32_t num_groups, unsigned prng_seed,
                                       uint32_t variants_batch_size)
    : variants(variants_per_group, variants_per_group + num_groups), prng(prng_seed), batch_size(variants_batch_size) {
    for (uint32_t g = 0; g < num_groups; g++)
        if (clusters_per_group[g] == 1) noise_group_indices.push_back(g);
}

std::vector<uint32_t> NoiseGroupSelector::nextChain() {
    std::shuffle(noise_group_indices.begin(), noise_group_indices.end(), prng);
    size_t end = 0;
    num_noise_variants = 0;
    while (num_noise_variants < batch_size && end < noise_group_indices.size()) num_noise_variants += variants[noise_group_indices[end++]];
    std::sort(noise_group_indices.begin(), noise_group_indices.begin() + end);
    return std::vector<uint32_t>(noise_group_indices.begin(), noise_group_indices.begin() + end);
}

std::string noiseParameterHeader(const std::vector<std::string> &sample_names) {
    std::string s = "Chain\tIteration";
    for (auto &n : sample_names) s += "\t" + n;
    return s + "\n";
}

std::string noiseParameterRow(unsigned chain, unsigned iteration, const std::vector<double> &rates) {
    std::ostringstream os;
    os << chain << "\t" << iteration;
    for (double r : rates) os << "\t" << r;
    os << "\n";
    return os.str();
}

}  // namespace bthost
