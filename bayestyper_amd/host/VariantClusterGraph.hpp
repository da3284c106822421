// Construction of a variant cluster's graph — VariantClusterGraph::VariantClusterGraph / addVertices / initVertex
// (src/bayesTyper/VariantClusterGraph.cpp:62-377) — and its flattening into bt_paths_batch (include/btgpu.h), the input of
// bt_find_paths_* and bt_paths_*.  Host code: one pass over a cluster's variants, run once per cluster in the cluster stage.
#pragma once
#include <cstdint>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "../../include/btgpu.h"

namespace bthost {

enum class VariantType : uint8_t { SNV = 0, Insertion, Deletion, Complex, Mixture, Unsupported, VARIANT_TYPE_SIZE };   // VariantCluster.hpp:52

struct AlleleInfo {                    // VariantInfo.hpp:41-63
    uint32_t ref_length = 0;
    std::string sequence;
    std::string aco_att;               // the allele's ACO (allele call-set origin) attribute, carried to the output VCF
};
struct Variant {                       // VariantCluster::Variant (VariantCluster.hpp:56-71)
    std::string id;
    bool has_dependency = false;
    VariantType type = VariantType::Unsupported;
    uint32_t num_redundant_nucleotides = 0xFFFFFFFFu;
    std::vector<AlleleInfo> alt_alleles;
};
struct ContainedCluster {              // VariantCluster::ContainedCluster (:73-80)
    uint32_t cluster_idx, left_flank, right_flank;
};
struct VariantCluster {                // VariantCluster.hpp:99-108
    uint32_t cluster_idx = 0, left_flank = 0, right_flank = 0;   // flanks: 0-based positions of the first / last reference nucleotide covered
    std::string chrom_name;
    std::map<uint32_t, Variant> variants;               // 0-based position of the first reference nucleotide -> variant
    std::list<ContainedCluster> contained_clusters;     // sorted by left flank, disjoint
};

struct GraphVertex {                   // VariantClusterGraphVertex (VariantClusterGraphVertex.hpp:43-73)
    uint16_t variant = 0xFFFF, allele = 0xFFFF;
    std::vector<uint16_t> reference_variant_indices;
    uint32_t nested_variant_cluster_index = 0xFFFFFFFFu;
    bool is_disconnected = false, is_first_nucleotides_redundant = false;
    std::vector<uint8_t> sequence;     // 2-bit codes
};

class VariantClusterGraph {
  public:
    VariantClusterGraph(VariantCluster variant_cluster, const std::string &chrom_sequence, unsigned kmer_size);

    std::vector<GraphVertex> vertices;
    std::vector<std::pair<uint32_t, uint32_t>> edges;   // (source, target) in insertion order
    std::vector<uint16_t> var_num_alleles;              // variant_cluster_info[v].numberOfAlleles()
    std::vector<uint8_t> var_has_dependency;

};

// Collects graphs (and, when known, their best paths) into the flat arrays of bt_paths_batch.
class PathsBatchBuilder {
  public:
    void add(const VariantClusterGraph &graph, const std::vector<std::vector<uint8_t>> &best_paths = {});
    const bt_paths_batch &batch();

  private:
    bt_paths_batch b{};
    std::vector<uint32_t> vertex_off{0}, num_paths, vertex_nested, refvar_off{0}, var_off{0}, in_off{0}, in_src;
    std::vector<uint64_t> seq_off{0}, path_off{0};
    std::vector<uint8_t> seq, vertex_flags, path_vertices, var_has_dependency;
    std::vector<uint16_t> vertex_variant, vertex_allele, refvar, var_num_alleles;
};

}  // namespace bthost
