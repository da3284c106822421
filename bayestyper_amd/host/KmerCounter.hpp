// KmerCounter (include/bayesTyper/KmerCounter.hpp:59-67, src/bayesTyper/KmerCounter.cpp): the k-mer passes of the two stages, driven
// from the host over the C ABI of libbtgpu.so.  Method names, argument meaning and the order of the passes are the reference's; the
// work itself — best-path search, path k-mer enumeration, Bloom / table updates, the KMC scans — runs on the GPU.
//
//   cluster:   findVariantClusterPaths -> countPathMultigroupKmers (per unit) ... countInterclusterParameterKmers
//   genotype:  countPathKmers -> countInterclusterKmers -> parseSampleKmers -> classifyPathKmers (-> haplotype candidates)
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/btgpu.h"
#include "InferenceUnit.hpp"
#include "Sample.hpp"

namespace bthost {

// RAII holders of libbtgpu handles
struct BloomHandle {
    bt_bloom *h = nullptr;
    ~BloomHandle() { bt_bloom_destroy(h); }
};
struct TableHandle {
    bt_table *h = nullptr;
    ~TableHandle() { bt_table_destroy(h); }
};
struct PathsHandle {
    bt_paths *h = nullptr;
    ~PathsHandle() { bt_paths_destroy(h); }
};

// the graphs of a unit in unit order (groups, then vertices) and what locates a cluster in it
struct UnitGraphs {
    std::vector<VariantClusterGraph> graphs;
    std::vector<uint32_t> cluster_group, cluster_vertex, group_first;   // group_first[g] = index of group g's first cluster (+ end sentinel)
    // `threads` host threads construct the graphs (the reference threads graph construction too: VariantFileParser.cpp:1044-1106); the result does not
    // depend on the thread count (every graph is built from its own cluster)
    UnitGraphs(const InferenceUnit &unit, const Chromosomes &chromosomes, unsigned kmer_size, unsigned threads = 1);
};

// the flattened VariantClusterHaplotypes bundles of a unit + its group structure: the host copy of bt_gibbs_batch
struct GibbsBatchData {
    uint32_t S = 0;
    std::vector<uint32_t> group_index, group_cluster_off, group_source_off, group_sources, group_num_shared, cluster_idx, edge_off, edges, num_haplotypes, num_variants, kmer_off,
        kv_off, kv_bits, unique_off, unique_idx, multi_off, multi_idx, hapnest_off, hapnest_idx, nestdep_off, nestdep_cluster, nestdep_var_off;
    std::vector<uint8_t> group_ploidy, hap_kmer_mult, kmer_has_counts, kmer_counts, kmer_ic_mult, var_has_dependency;
    std::vector<int32_t> kmer_shared;
    std::vector<uint16_t> kv_var, hap_allele, var_num_alleles, nestdep_var;
    bt_gibbs_batch view() const;
    static GibbsBatchData fromView(const bt_gibbs_batch &b, uint32_t S);   // a deep copy of a batch handed in as plain arrays (include/btgpu.h: bt_gibbs_batch)
    // the groups `ids` (ascending) as a batch of their own; group_index keeps the unit-wide index
    GibbsBatchData take(const std::vector<uint32_t> &ids) const;
    uint32_t numGroups() const { return (uint32_t)group_index.size(); }
    uint32_t numClusters() const { return (uint32_t)cluster_idx.size(); }
};

class Comm;

class KmerCounter {
  public:
    KmerCounter(bt_ctx *ctx, const std::vector<Sample> &samples, unsigned kmer_size, unsigned prng_seed);

    // ---- cluster stage (KmerCounter.cpp:59-250) ----
    void findVariantClusterPaths(InferenceUnit *unit, const UnitGraphs &graphs, uint16_t max_sample_haplotypes);
    void countPathMultigroupKmers(bt_table *multigroup_table, bt_bloom *path_bloom, InferenceUnit *unit, const UnitGraphs &graphs);
    void countInterclusterParameterKmers(bt_table *parameter_table, const std::vector<InterClusterRegion> &regions, const Chromosomes &chromosomes, bt_bloom *path_bloom,
                                         float parameter_kmer_fraction);
    // ---- genotype stage (KmerCounter.cpp:252-555) ----
    void countPathKmers(bt_bloom *path_bloom, const InferenceUnit &unit, const UnitGraphs &graphs);   // keeps the enumerated paths for classifyPathKmers
    void countInterclusterKmers(bt_table *table, bt_bloom *path_bloom, const std::string &intercluster_regions_prefix, const Chromosomes &chromosomes, const ChromosomePloidy &chrom_ploidy);
    // comm (several ranks): every rank scans its byte range of every sample's database into its replica of the table; the records with
    // counts are then all-gathered and merged, so that every rank holds the table a one-rank scan fills
    void parseSampleKmers(bt_table *table, bt_bloom *path_bloom, Comm *comm = nullptr);
    // classifyPathKmers, then getHaplotypeCandidates of every cluster against the classified table -> the unit's Gibbs batch
    GibbsBatchData classifyPathKmers(bt_table *table, const InferenceUnit &unit, const UnitGraphs &graphs, const std::string &multigroup_kmers_bloom_prefix,
                                     const ChromosomePloidy &chrom_ploidy);

    static void checkTable(bt_table *table, const char *stage);   // throws when the table dropped records (overflow)

  private:
    bt_ctx *ctx;
    const std::vector<Sample> &samples;
    unsigned kmer_size, prng_seed;
    std::unique_ptr<PathsHandle> unit_paths;   // bt_paths of the unit being genotyped (countPathKmers .. classifyPathKmers)
};

}  // namespace bthost
