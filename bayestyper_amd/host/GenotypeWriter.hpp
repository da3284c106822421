// Output VCF of the genotyping stage: GenotypeWriter (include/bayesTyper/GenotypeWriter.hpp, src/bayesTyper/GenotypeWriter.cpp:57-556).
// One line per genotyped variant: REF/ALT rebuilt from the cluster's variant records and the reference sequence, QUAL/FILTER and
// the call statistics (Genotypes.hpp), the cluster annotations (VCS, VCR, VCGS, VCGR, HC), ANC, ACO and the per-sample columns;
// lines sorted by contig (genome order) and position; the header names every non-decoy contig.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "Genotypes.hpp"
#include "VariantFileParser.hpp"

namespace bthost {

struct VariantInfo {   // include/bayesTyper/VariantInfo.hpp:66-121
    uint32_t position = 0;   // 1-based
    std::string id;
    bool has_dependency = false;
    std::vector<AlleleInfo> alt_alleles;
    uint16_t numberOfAlleles() const { return (uint16_t)(1 + (has_dependency ? 1 : 0) + alt_alleles.size()); }
    uint32_t maxReferenceLength() const;
};
// variant_cluster_info of a cluster's graph (VariantClusterGraph.cpp:69-93) and the region getGenotypes reports (VariantClusterGenotyper.cpp:535-555)
std::vector<VariantInfo> variantClusterInfo(const VariantCluster &cluster);
std::string variantClusterRegion(const std::string &chrom_name, const std::vector<VariantInfo> &variant_cluster_info);

struct ClusterAnnotation {   // the Genotypes fields that describe where a variant was genotyped (Genotypes.hpp:47-58)
    std::string chrom_name;
    uint32_t variant_cluster_size = 0;
    std::string variant_cluster_region;
    uint32_t variant_cluster_group_size = 0;
    std::string variant_cluster_group_region;
    uint32_t num_candidates = 0;   // haplotype candidates of the cluster (HC)
};

class GenotypeWriter {
  public:
    GenotypeWriter(std::vector<std::string> sample_names, const Chromosomes &chromosomes);
    struct GenotypedVariant {
        uint32_t position, max_ref_length;
        std::string variant_id, genotypes;   // genotypes = everything after the REF column
    };
    // one genotyped variant (GenotypeWriter::writeGenotypes :84-128); sample_columns = formatSampleColumns(...) of that variant
    void addGenotypes(const ClusterAnnotation &where, const VariantInfo &variant_info, const VariantGenotypes &genotypes, const std::string &sample_columns);
    // the same in two steps, so that worker threads can format (const, thread-safe) and one thread appends in the order of a one-thread run
    GenotypedVariant formatGenotypes(const ClusterAnnotation &where, const VariantInfo &variant_info, const VariantGenotypes &genotypes, const std::string &sample_columns) const;
    void append(const std::string &chrom_name, GenotypedVariant &&variant) { genotyped_variants[chrom_name].push_back(std::move(variant)); }
    std::string generateHeader(const std::string &genome_filename, const std::string &graph_options_header, const std::string &genotype_options_header) const;   // :494-551
    // header + sorted lines (finalise :352-492); the file variant writes <output_prefix>.vcf or .vcf.gz and returns the number of variants
    std::string vcfText(const std::string &genome_filename, const std::string &graph_options_header, const std::string &genotype_options_header);
    uint32_t finalise(const std::string &output_prefix, bool gzip_output, const std::string &genome_filename, const std::string &graph_options_header,
                      const std::string &genotype_options_header);

  private:
    std::vector<std::string> samples;
    const Chromosomes &chromosomes;
    std::unordered_map<std::string, std::vector<GenotypedVariant>> genotyped_variants;
};

}  // namespace bthost
