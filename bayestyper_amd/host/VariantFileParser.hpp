// The cluster stage's front end: candidate-variant VCF -> variant clusters -> variant-cluster groups, plus the intercluster
// regions (everything of the genome that is not inside a cluster).  Mirrors the reference's VariantFileParser
// (include/bayesTyper/VariantFileParser.hpp:57-141, src/bayesTyper/VariantFileParser.cpp:59-1235), Chromosomes
// (include/bayesTyper/Chromosomes.hpp:44-82) and the constructor of VariantClusterGroup (src/bayesTyper/VariantClusterGroup.cpp:47-107).
//
// Host C++: this is sequential text parsing with small ordered containers, run once per unit; its output (clusters with their
// variants, contained clusters and group structure) is what VariantClusterGraph (VariantClusterGraph.hpp) turns into the graphs the
// GPU stages consume.  Cluster indices, the order of a group's vertices and of its edges follow the iteration order of libstdc++'s
// hash containers in the reference, so the same containers are used here.
#pragma once
#include <cstdint>
#include <istream>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "VariantClusterGraph.hpp"

namespace bthost {

// reference genome (+ decoys) in FASTA order
class Chromosomes {
  public:
    void addFasta(const std::string &fasta_filename, bool is_decoy);                      // Chromosomes.cpp:72-113
    void addSequence(const std::string &name, const std::string &sequence, bool is_decoy);   // :52-70
    void convertToUpper();                                                                // Chromosomes::convertToUpper (main.cpp:209,462)
    size_t decoyCount() const { return decoys.size(); }
    int find(const std::string &name) const;                                              // index or -1 (:125-137)
    bool isDecoy(const std::string &name) const { return decoys.count(name) > 0; }
    size_t size() const { return seqs.size(); }
    const std::string &name(size_t i) const { return seqs[i].first; }
    const std::string &sequence(size_t i) const { return seqs[i].second; }
    uint64_t getTotalLength() const { return total_length; }
    uint64_t getDecoyLength() const { return decoy_length; }

  private:
    std::vector<std::pair<std::string, std::string>> seqs;
    std::unordered_map<std::string, uint32_t> order;
    std::unordered_set<std::string> decoys;
    uint64_t total_length = 0, decoy_length = 0;
};

struct InterClusterRegion {   // VariantFileParser.hpp:64-72
    std::string chrom_name;
    bool is_decoy;
    uint32_t start_position, end_position;   // 0-based, inclusive
};

// what VariantClusterGroup's constructor keeps of a group (VariantClusterGroup.cpp:47-107)
struct ClusterGroup {
    std::string chrom_name;
    uint32_t start_position = 0, end_position = 0;   // 1-based (region() = chrom:start-end)
    uint32_t num_variants = 0;
    std::vector<VariantCluster> clusters;            // in vertex order; contained_clusters filled
    std::vector<uint32_t> source_vertices;
    std::vector<std::vector<uint32_t>> out_edges;    // per vertex: nested clusters (vertex ids), in the reference's edge order
    std::string region() const { return chrom_name + ":" + std::to_string(start_position) + "-" + std::to_string(end_position); }
};
bool ClusterGroupCompare(const ClusterGroup &first, const ClusterGroup &second);   // VariantClusterGroup.cpp:277-292 (main.cpp:247 sorts a unit with it)
// canonical text form of a unit's groups (tests compare it with the oracle's; one line per group / vertex / variant)
std::string dumpClusterGroups(const std::vector<ClusterGroup> &groups);

class VariantFileParser {
  public:
    enum AlleleCount { Total = 0, Excluded_decoy, Excluded_genome, Excluded_match, Excluded_end, Excluded_length, ALLELE_COUNT_SIZE };

    // vcf: the whole (decompressed) text of the candidate VCF; options as in main.cpp:135-136
    VariantFileParser(std::string vcf_text, unsigned kmer_size, uint32_t max_allele_length = 500000, float copy_number_variant_threshold = 0.5f);
    static std::string readVariantFile(const std::string &variant_filename);   // ".vcf" or ".vcf.gz" (VariantFileParser.cpp:122-146)

    // parses the next unit (at least min_unit_variants variants, cut where two variants are >= k apart); appends its groups in
    // creation order and returns true when the file is exhausted (VariantFileParser.cpp:185-235)
    bool constructVariantClusterGroups(std::vector<ClusterGroup> *groups, uint32_t min_unit_variants, const Chromosomes &chromosomes);

    void sortInterclusterRegions();                                   // by length, descending (:59-65,1186-1189)
    const std::vector<InterClusterRegion> &getInterclusterRegions() const { return intercluster_regions; }
    std::string interclusterRegionsText() const;                      // the rows of intercluster_regions.txt(.gz) (:1191-1212)
    uint64_t getInterclusterRegionLength() const { return intercluster_regions_length; }
    uint64_t getNumberOfInterclusterRegionKmers() const;              // :1224-1228
    uint32_t getNumberOfVariants() const { return total_num_variants; }
    uint32_t numParsedVariants() const { return num_variants; }
    uint32_t numVariantClusters() const { return num_variant_clusters; }
    uint32_t numVariantClusterGroups() const { return num_variant_cluster_groups; }
    const std::vector<uint32_t> &alleleTypeCounter() const { return allele_type_counter; }
    const std::vector<uint32_t> &variantTypeCounter() const { return variant_type_counter; }

    // helpers of the allele handling, public for the tests
    static void rightTrimAllele(std::string *ref_allele, std::string *alt_allele);                     // :563-580
    static VariantType classifyAllele(int reference_size, int allele_size);                           // :624-647
    uint32_t copyNumberVariantLength(const std::string &allele_sequence, const std::string &chrom_sequence, uint32_t chrom_start_position) const;   // :649-733

  private:
    typedef std::unordered_map<uint32_t, std::unique_ptr<VariantCluster>> Group;   // cluster_idx -> cluster, while a group is open
    struct Open {   // the group under construction
        Group clusters;
        std::map<uint32_t, VariantCluster *> flanks;
        std::list<std::unordered_set<uint32_t>> merge_sets;
    };
    struct UnitScan;
    bool arrive(UnitScan *u, int position, uint32_t min_unit_variants);
    bool screen(UnitScan *u, int position, const std::vector<std::string> &ref, const std::vector<std::string> &alt, const std::string &ref_field, std::vector<bool> *keep);
    bool updateVariantLine();
    void addSequenceToInterclusterRegions(const std::string &chrom_name, bool is_decoy, uint32_t start_position, uint32_t end_position);
    void addAlternativeAllele(Variant *cur_variant, const std::string &ref_allele, const std::string &alt_allele, const std::string &origin_att) const;
    void clusterVariants(const Variant &cur_variant, uint32_t cur_position, const std::set<uint32_t> &cur_end_positions, const std::string &cur_chrom_name, Open *open);
    void closeGroup(Open *open, std::vector<ClusterGroup> *groups);
    static void mergeVariantClusters(Group *group, const std::list<std::unordered_set<uint32_t>> &merge_sets);
    static std::unordered_map<uint32_t, uint32_t> getVariantClusterGroupDependencies(Group *group);

    const unsigned kmer_size;
    const uint32_t max_allele_length;
    const float copy_number_variant_threshold;
    std::vector<uint32_t> allele_type_counter, variant_type_counter;
    uint32_t num_variants = 0, num_variant_clusters = 0, num_variant_cluster_groups = 0, total_num_variants = 0;
    std::vector<InterClusterRegion> intercluster_regions;
    uint64_t intercluster_regions_length = 0;
    std::unordered_set<std::string> intercluster_chromosomes;
    std::string prev_chrom_name;
    int prev_position = -1, prev_var_end_position = -1;
    // the VCF text and the read cursor
    std::string text;
    size_t cursor = 0;
    bool has_format = false, line_good = false;
    std::vector<std::string> variant_line;   // CHROM, POS, ID, REF, ALT, INFO
};

}  // namespace bthost
