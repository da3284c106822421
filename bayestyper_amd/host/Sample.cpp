#include "Sample.hpp"

#include <algorithm>
#include <fstream>
#include <stdexcept>

namespace bthost {

static std::vector<std::string> splitTabs(const std::string &line) {
    std::vector<std::string> out(1);
    for (char c : line) {
        if (c == '\t') out.emplace_back();
        else out.back().push_back(c);
    }
    return out;
}

Sample::Sample(const std::string &sample_line) {
    const auto cols = splitTabs(sample_line);
    if (cols.size() != 3)
        throw std::runtime_error("Line \"" + sample_line + "\" in the samples file should contain three tab-seperated columns (<Sample ID>, <Gender> & <KMC Output Prefix>)");
    name = cols[0];
    if (cols[1] == "F" || cols[1] == "Female") gender = 0;
    else if (cols[1] == "M" || cols[1] == "Male") gender = 1;
    else throw std::runtime_error("Gender (column two) in line \"" + sample_line + "\" in the samples file should be either \"F\" (Female) or \"M\" (Male)");
    file = cols[2];
}

std::vector<Sample> readSamples(const std::string &samples_filename) {
    std::ifstream in(samples_filename);
    if (!in.is_open()) throw std::runtime_error("Unable to open file " + samples_filename);
    std::vector<Sample> samples;
    for (std::string line; std::getline(in, line);) samples.emplace_back(line);
    if (samples.empty()) throw std::runtime_error("Samples file empty");
    if (samples.size() > 30) throw std::runtime_error("The maximum number of samples supported by BayesTyper is currently 30");
    return samples;
}

ChromosomePloidy::ChromosomePloidy(const std::string &chrom_ploidy_filename, const Chromosomes &chromosomes, const std::vector<Sample> &samples) {
    const size_t S = samples.size();
    if (chrom_ploidy_filename.empty()) {
        for (size_t i = 0; i < chromosomes.size(); i++) {
            const std::string &chrom = chromosomes.name(i);
            if (chromosomes.isDecoy(chrom)) continue;
            std::vector<uint8_t> gender(2, 2), per_sample(S, 2);
            std::string lower = chrom;
            std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
            if (lower == "x" || lower == "chrx") {
                gender = {2, 1};
                for (size_t s = 0; s < S; s++)
                    if (samples[s].gender == 1) per_sample[s] = 1;
            } else if (lower == "y" || lower == "chry") {
                gender = {0, 1};
                for (size_t s = 0; s < S; s++) per_sample[s] = samples[s].gender == 0 ? 0 : 1;
            }
            gender_ploidy.emplace(chrom, gender);
            sample_ploidy.emplace(chrom, per_sample);
        }
        return;
    }
    std::ifstream in(chrom_ploidy_filename);
    if (!in.is_open()) throw std::runtime_error("Unable to open file " + chrom_ploidy_filename);
    std::unordered_map<std::string, std::pair<int, int>> ploidies;
    for (std::string line; std::getline(in, line);) {
        const auto cols = splitTabs(line);
        if (cols.size() != 3)
            throw std::runtime_error("Line \"" + line + "\" in the chromosome ploidy file should contain three tab-seperated columns (<Chromosome name>, <Female Ploidy> & <Male Ploidy>)");
        const int f = std::stoi(cols[1]), m = std::stoi(cols[2]);
        if (!ploidies.emplace(cols[0], std::make_pair(f, m)).second)
            throw std::runtime_error("Chromosome (column one) in line \"" + line + "\" appear multiple times in the chromosome ploidy file");
        if (f < 0 || f > 2)
            throw std::runtime_error("Female ploidy (column two) in line \"" + line +
                                     "\" in the chromosome ploidy file should be between zero and two; only ploidy levels up to diploid are currently supported");
        if (m < 0 || m > 2)
            throw std::runtime_error("Male ploidy (column three) in line \"" + line +
                                     "\" in the chromosome ploidy file should be between zero and two; only ploidy levels up to diploid are currently supported");
    }
    for (size_t i = 0; i < chromosomes.size(); i++) {
        const std::string &chrom = chromosomes.name(i);
        if (chromosomes.isDecoy(chrom)) continue;
        auto it = ploidies.find(chrom);
        if (it == ploidies.end()) throw std::runtime_error("Chromosome \"" + chrom + "\" in reference genome does not appear in the chromosome ploidy file");
        gender_ploidy.emplace(chrom, std::vector<uint8_t>{(uint8_t)it->second.first, (uint8_t)it->second.second});
        std::vector<uint8_t> per_sample(S);
        for (size_t s = 0; s < S; s++) per_sample[s] = (uint8_t)(samples[s].gender == 0 ? it->second.first : it->second.second);
        sample_ploidy.emplace(chrom, per_sample);
    }
}

const std::vector<uint8_t> &ChromosomePloidy::getGenderPloidy(const std::string &chrom_name) const {
    auto it = gender_ploidy.find(chrom_name);
    if (it == gender_ploidy.end()) throw std::runtime_error("no ploidy for chromosome " + chrom_name);
    return it->second;
}
const std::vector<uint8_t> &ChromosomePloidy::getSamplePloidy(const std::string &chrom_name) const {
    auto it = sample_ploidy.find(chrom_name);
    if (it == sample_ploidy.end()) throw std::runtime_error("no ploidy for chromosome " + chrom_name);
    return it->second;
}

}  // namespace bthost
