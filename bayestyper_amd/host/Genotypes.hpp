// Genotype summaries of one variant cluster from the sampler's results — the host half of
// VariantClusterGenotyper::getGenotypes (src/bayesTyper/VariantClusterGenotyper.cpp:208-567): genotype / allele posteriors
// (GPP, APP), genotype quality (GQ), allele filters (NAK / FAK), the genotype call, and the per-variant call statistics
// (AC, AF, ACP, AN).  Input is exactly what bt_gibbs_result_fetch returns for a cluster.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bthost {

struct Filters {                       // src/bayesTyper/Filters.cpp:33-54
    float min_genotype_posterior = 0.99f;
    float min_number_of_kmers = 1.0f;
    std::vector<float> min_fraction_observed_kmers;   // per sample; 1 - exp(-0.275 * mean of the sample's genomic NB), 0 when disabled
    static float minFractionObservedKmers(double genomic_mean) ;
};

struct KmerStatsView {                 // KmerStats (KmerStats.cpp:35-105): {count, fraction, mean, M2}
    double count, fraction, mean, m2;
};

struct SampleStats {                   // Genotypes::SampleStats (include/bayesTyper/Genotypes.hpp:80-97)
    std::vector<uint16_t> genotype_estimate;      // empty (ploidy 0), 1 or 2 allele indices; 0xFFFF = no call
    uint32_t genotype_quality = 0;
    std::vector<float> genotype_posteriors;       // diploid: index b(b+1)/2 + a for a <= b; haploid: allele index
    std::vector<float> allele_posteriors;
    std::vector<uint16_t> allele_filters;         // bit 0: too few k-mers (NAK), bit 1: too low observed fraction (FAK)
};

struct VariantStats {                  // Genotypes::VariantStats (Genotypes.hpp:62-76)
    uint32_t total_count = 0;
    std::vector<uint32_t> alt_allele_counts;
    std::vector<float> alt_allele_frequency;
    float max_alt_allele_call_probability = 0;
    std::vector<float> allele_call_probabilities;
};

struct VariantGenotypes {
    std::vector<uint16_t> non_covered_alleles;
    std::vector<SampleStats> sample_stats;
    VariantStats variant_stats;
};

struct ClusterResults {
    uint32_t S = 0, H = 0, V = 0;
    const uint16_t *hap_allele = nullptr;          // [H*V] variant_allele_indices
    const uint16_t *var_num_alleles = nullptr;     // [V] numberOfAlleles() incl. the missing allele
    const uint8_t *var_has_dependency = nullptr;   // [V]
    uint64_t num_diplotypes = 0;                   // distinct sampled diplotypes
    const uint16_t *h1 = nullptr, *h2 = nullptr;   // 0xFFFF = none
    const uint32_t *freq = nullptr;                // [num_diplotypes*S] sampling frequencies
    const double *stats = nullptr;                 // [(s*A_total + allele_base(v) + a)*12]: count_stats, fraction_stats, mean_stats
    const uint8_t *ploidy = nullptr;               // [S] 0 Null, 1 Haploid, 2 Diploid
};

std::vector<VariantGenotypes> getGenotypes(const ClusterResults &r, const Filters &filters);

// The genotype-derived columns of one variant's output line as GenotypeWriter writes them (src/bayesTyper/GenotypeWriter.cpp):
// "<QUAL>\t<FILTER>\tAC=..;AF=..;AN=..;ACP=..[;ANC=..]" (writeQualityAndFilter :174-200, writeVariantStats :202-218, writeAlleleCover
// :220-230) and, per sample, "\tGT:GQ:GPP:APP:NAK:FAK:MAC:SAF" (writeSamples :261-322, writeAlleleKmerStats :324-345) with the
// default ostream formatting the reference uses.  `variant` = index of the variant inside the cluster (for the k-mer statistics).
std::string formatVariantStatsColumns(const VariantGenotypes &g);
// the two halves of the above: GenotypeWriter puts the cluster annotations (VCS .. HC) between them
std::string formatQualityFilterAndStats(const VariantGenotypes &g);   // "<QUAL>\t<FILTER>\tAC=..;AF=..;AN=..;ACP=.."
std::string formatAlleleCover(const VariantGenotypes &g);             // ";ANC=.." or ""
std::string formatSampleColumns(const ClusterResults &r, uint32_t variant, const VariantGenotypes &g);

}  // namespace bthost
