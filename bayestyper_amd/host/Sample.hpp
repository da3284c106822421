// Samples file and chromosome ploidy of the two command lines: Sample (src/bayesTyper/Sample.cpp:38-67: "<id>\t<F|Female|M|Male>\t<KMC
// prefix>") and ChromosomePloidy (src/bayesTyper/ChromosomePloidy.cpp:40-200: human defaults — X diploid / haploid, Y absent / haploid —
// or a "<chromosome>\t<female ploidy>\t<male ploidy>" file).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "VariantFileParser.hpp"

namespace bthost {

struct Sample {
    std::string name, file;
    uint8_t gender = 0;   // 0 female, 1 male (Utils::Gender)
    explicit Sample(const std::string &sample_line);   // throws std::runtime_error with the reference's message
};
// the samples of a samples file (main.cpp:163-192: at least one, at most 30)
std::vector<Sample> readSamples(const std::string &samples_filename);

class ChromosomePloidy {
  public:
    ChromosomePloidy(const std::string &chrom_ploidy_filename, const Chromosomes &chromosomes, const std::vector<Sample> &samples);
    const std::vector<uint8_t> &getGenderPloidy(const std::string &chrom_name) const;   // [female, male]: 0 Null, 1 Haploid, 2 Diploid
    const std::vector<uint8_t> &getSamplePloidy(const std::string &chrom_name) const;   // per sample

  private:
    std::unordered_map<std::string, std::vector<uint8_t>> gender_ploidy, sample_ploidy;
};

}  // namespace bthost
