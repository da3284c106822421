// ORACLE — TEST INFRASTRUCTURE ONLY (never linked, imported or executed by the product; see tests/test_abi.py).
//
// CPU restatement of the reference's variant-cluster graph construction, VariantClusterGraph::VariantClusterGraph / addVertices /
// initVertex (src/bayesTyper/VariantClusterGraph.cpp:62-377), on std:: containers (std::map, std::unordered_set<ushort>: the
// iteration order of the latter decides the order of a vertex's reference_variant_indices, and it is libstdc++'s here as in the
// reference).  It is the checker of the product's graph builder (bayestyper_amd/host/VariantClusterGraph.cpp) and the graph source of
// the oracle side of the end-to-end tests — independent of bayestyper_amd/synth_graphs.py.
// PARITY UNPINNED: the reference's TU includes Boost (BGL) headers that are absent from this image, so it cannot be compiled here;
// this file restates its source text statement by statement.
#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

typedef unsigned int uint;
typedef unsigned short ushort;
const uint uint_overflow = 0xFFFFFFFFu;
const ushort ushort_overflow = 0xFFFF;

struct Alt {
    uint ref_length;
    std::string sequence;
};
struct Var {
    bool has_dependency;
    uint num_redundant_nucleotides;
    std::vector<Alt> alt_alleles;
};
struct Contained {
    uint cluster_idx, left_flank, right_flank;
};
struct Vertex {   // VariantClusterGraphVertex.hpp:43-73
    std::pair<ushort, ushort> variant_allele_idx;
    std::vector<ushort> reference_variant_indices;
    uint nested_variant_cluster_index;
    bool is_disconnected, is_first_nucleotides_redundant;
    std::vector<uint8_t> sequence;   // one 2-bit code per nucleotide
};
typedef std::pair<std::string::const_iterator, std::string::const_iterator> StringItPair;

struct Graph {
    uint kmer_size;
    std::vector<Vertex> graph;
    std::vector<std::pair<uint, uint>> edges;
    std::vector<ushort> num_alleles;
    std::vector<uint8_t> has_dependency;

    uint add_vertex() {
        graph.emplace_back();
        return (uint)graph.size() - 1;
    }
    void add_edge(uint a, uint b) { edges.emplace_back(a, b); }

    // VariantClusterGraph.cpp:323-377
    void initVertex(uint *cur_vertex, StringItPair vertex_sequence, const std::pair<ushort, ushort> &variant_allele_idx, const std::vector<ushort> &vertex_reference_variant_indices,
                    const uint nested_variant_cluster_index, const bool is_first_nucleotides_redundant) {
        graph[*cur_vertex].variant_allele_idx = variant_allele_idx;
        graph[*cur_vertex].reference_variant_indices = vertex_reference_variant_indices;
        graph[*cur_vertex].nested_variant_cluster_index = nested_variant_cluster_index;
        graph[*cur_vertex].is_first_nucleotides_redundant = is_first_nucleotides_redundant;
        graph[*cur_vertex].is_disconnected = nested_variant_cluster_index != uint_overflow;
        bool prev_is_disconnected = false;
        while (vertex_sequence.first != vertex_sequence.second) {
            int code;
            switch (*vertex_sequence.first) {   // Nucleotide::ntToBit<1> (Nucleotide.hpp:40-70)
                case 'A': case 'a': code = 0; break;
                case 'C': case 'c': code = 1; break;
                case 'G': case 'g': code = 2; break;
                case 'T': case 't': code = 3; break;
                default: code = -1;
            }
            if (code < 0) {
                if (!prev_is_disconnected) {
                    uint prev_vertex = *cur_vertex;
                    *cur_vertex = add_vertex();
                    add_edge(prev_vertex, *cur_vertex);
                    graph[*cur_vertex].variant_allele_idx = variant_allele_idx;
                    graph[*cur_vertex].reference_variant_indices = vertex_reference_variant_indices;
                    graph[*cur_vertex].nested_variant_cluster_index = uint_overflow;
                    graph[*cur_vertex].is_first_nucleotides_redundant = false;
                    graph[*cur_vertex].is_disconnected = true;
                }
                prev_is_disconnected = true;
            } else {
                graph[*cur_vertex].sequence.push_back((uint8_t)code);
                prev_is_disconnected = false;
            }
            vertex_sequence.first++;
        }
    }

    // VariantClusterGraph.cpp:285-321
    void addVertices(uint *cur_vertex, const std::vector<StringItPair> &vertex_sequences, const std::pair<ushort, ushort> &variant_allele_idx,
                     const std::unordered_set<ushort> &reference_variant_indices, const std::vector<uint> &nested_variant_cluster_indices, const bool is_first_nucleotides_redundant) {
        std::vector<ushort> vertex_reference_variant_indices;
        vertex_reference_variant_indices.reserve(reference_variant_indices.size());
        for (auto &reference_variant_idx : reference_variant_indices)
            if (reference_variant_idx != variant_allele_idx.first) vertex_reference_variant_indices.push_back(reference_variant_idx);
        initVertex(cur_vertex, vertex_sequences.front(), variant_allele_idx, vertex_reference_variant_indices, uint_overflow, is_first_nucleotides_redundant);
        for (uint vertex_idx = 1; vertex_idx < vertex_sequences.size(); vertex_idx++) {
            uint prev_vertex = *cur_vertex;
            *cur_vertex = add_vertex();
            add_edge(prev_vertex, *cur_vertex);
            initVertex(cur_vertex, vertex_sequences.at(vertex_idx), variant_allele_idx, vertex_reference_variant_indices, nested_variant_cluster_indices.at(vertex_idx - 1), false);
        }
    }

    // VariantClusterGraph.cpp:62-283
    Graph(std::map<uint, Var> &variants, std::list<Contained> &contained_clusters, const std::string &chrom_sequence, uint k) : kmer_size(k) {
        std::map<uint, std::pair<std::vector<uint>, std::vector<ushort>>> added_vertices;
        std::unordered_set<ushort> reference_variant_indices;
        auto variants_it = variants.begin();
        auto chrom_sequence_it = chrom_sequence.begin();
        uint cur_vertex = add_vertex();
        addVertices(&cur_vertex, std::vector<StringItPair>(1, StringItPair(chrom_sequence_it + variants_it->first - (kmer_size - 1), chrom_sequence_it + variants_it->first)),
                    std::make_pair(ushort_overflow, ushort_overflow), reference_variant_indices, std::vector<uint>(), false);
        uint prev_vertex = cur_vertex;
        added_vertices.insert({variants_it->first, std::make_pair(std::vector<uint>(1, cur_vertex), std::vector<ushort>())});
        uint cur_last_position = 0, next_position = 0;
        ushort variant_counter = 0;
        while (variants_it != variants.end()) {
            num_alleles.push_back((ushort)(1 + (variants_it->second.has_dependency ? 1 : 0) + variants_it->second.alt_alleles.size()));   // VariantInfo::numberOfAlleles
            has_dependency.push_back(variants_it->second.has_dependency ? 1 : 0);
            const bool is_first_nucleotides_redundant = variants_it->second.num_redundant_nucleotides > 0;
            uint max_reference_length = 0;
            for (ushort alt_allele_idx = 0; alt_allele_idx < variants_it->second.alt_alleles.size(); alt_allele_idx++) {
                max_reference_length = std::max(max_reference_length, variants_it->second.alt_alleles.at(alt_allele_idx).ref_length);
                uint next_vertex = add_vertex();
                add_edge(cur_vertex, next_vertex);
                const std::string &s = variants_it->second.alt_alleles.at(alt_allele_idx).sequence;
                addVertices(&next_vertex, std::vector<StringItPair>(1, StringItPair(s.begin(), s.end())), std::make_pair(variant_counter, (ushort)(alt_allele_idx + 1)),
                            reference_variant_indices, std::vector<uint>(), is_first_nucleotides_redundant);
                auto added_vertices_insert =
                    added_vertices.insert({variants_it->first + variants_it->second.alt_alleles.at(alt_allele_idx).ref_length, std::make_pair(std::vector<uint>(), std::vector<ushort>())});
                added_vertices_insert.first->second.first.push_back(next_vertex);
            }
            auto added_vertices_it = added_vertices.find(variants_it->first + max_reference_length);
            added_vertices_it->second.second.push_back(variant_counter);
            reference_variant_indices.insert(variant_counter);
            variants_it++;
            added_vertices_it = added_vertices.begin();
            bool more_edges = true, last_variant = false;
            if (variants_it != variants.end()) next_position = variants_it->first;
            else {
                next_position = uint_overflow;
                last_variant = true;
            }
            while (more_edges) {
                auto cur_position = added_vertices_it->first;
                auto next_vertices = added_vertices_it->second.first;
                for (auto &variant_idx : added_vertices_it->second.second) reference_variant_indices.erase(variant_idx);
                added_vertices.erase(added_vertices_it);
                if (added_vertices.empty()) {
                    more_edges = false;
                    if (last_variant) cur_last_position = cur_position + kmer_size - 1;
                    else cur_last_position = next_position;
                } else {
                    added_vertices_it = added_vertices.begin();
                    cur_last_position = added_vertices_it->first;
                    if (!last_variant && (cur_last_position > next_position)) {
                        more_edges = false;
                        cur_last_position = next_position;
                    }
                }
                std::vector<StringItPair> contained_vertices;
                std::vector<uint> nested_variant_cluster_indices;
                auto contained_cluster_it = contained_clusters.begin();
                uint prev_contained_edge = uint_overflow;
                while ((contained_cluster_it != contained_clusters.end()) && (contained_cluster_it->left_flank < cur_last_position)) {
                    if (prev_contained_edge < uint_overflow) nested_variant_cluster_indices.emplace_back(prev_contained_edge);
                    contained_vertices.emplace_back(chrom_sequence_it + cur_position, chrom_sequence_it + contained_cluster_it->left_flank);
                    prev_contained_edge = contained_cluster_it->cluster_idx;
                    cur_position = contained_cluster_it->right_flank + 1;
                    contained_clusters.erase(contained_cluster_it++);
                }
                if (prev_contained_edge < uint_overflow) nested_variant_cluster_indices.emplace_back(prev_contained_edge);
                contained_vertices.emplace_back(chrom_sequence_it + cur_position, chrom_sequence_it + cur_last_position);
                cur_vertex = add_vertex();
                bool is_reference_allele = false;
                for (auto &vit : next_vertices) {
                    if (vit == prev_vertex) is_reference_allele = true;
                    add_edge(vit, cur_vertex);
                }
                if (is_reference_allele)
                    addVertices(&cur_vertex, contained_vertices, std::pair<ushort, ushort>(variant_counter, 0), reference_variant_indices, nested_variant_cluster_indices, is_first_nucleotides_redundant);
                else
                    addVertices(&cur_vertex, contained_vertices, std::pair<ushort, ushort>(ushort_overflow, ushort_overflow), reference_variant_indices, nested_variant_cluster_indices, false);
                auto added_vertices_insert = added_vertices.insert({cur_last_position, std::make_pair(std::vector<uint>(), std::vector<ushort>())});
                added_vertices_insert.first->second.first.push_back(cur_vertex);
            }
            variant_counter++;
            prev_vertex = cur_vertex;
        }
    }
};

}  // namespace

extern "C" {

// one cluster's graph from flat variant arrays (same argument layout as the product's test entry point bth_graph_build)
void *orc_graph_build(unsigned k, const char *chrom, unsigned long long chrom_len, unsigned nvar, const uint32_t *var_pos, const uint32_t *var_nalt, const uint32_t *var_redundant,
                      const uint8_t *var_dep, const uint32_t *alt_ref_len, const uint32_t *alt_off, const char *alt_seq, unsigned ncont, const uint32_t *cont_lf, const uint32_t *cont_rf,
                      const uint32_t *cont_idx) {
    std::map<uint, Var> variants;
    unsigned a = 0;
    for (unsigned v = 0; v < nvar; v++) {
        Var var;
        var.has_dependency = var_dep[v] != 0;
        var.num_redundant_nucleotides = var_redundant[v];
        for (unsigned i = 0; i < var_nalt[v]; i++, a++) var.alt_alleles.push_back(Alt{alt_ref_len[a], std::string(alt_seq + alt_off[a], alt_seq + alt_off[a + 1])});
        variants.emplace(var_pos[v], var);
    }
    std::list<Contained> contained;
    for (unsigned i = 0; i < ncont; i++) contained.push_back(Contained{cont_idx[i], cont_lf[i], cont_rf[i]});
    return new Graph(variants, contained, std::string(chrom, chrom + chrom_len), k);
}
void orc_graph_free(void *h) { delete (Graph *)h; }
void orc_graph_sizes(void *h, uint64_t *sizes) {
    auto *g = (Graph *)h;
    sizes[0] = g->graph.size();
    sizes[1] = g->edges.size();
    sizes[2] = sizes[3] = 0;
    for (auto &v : g->graph) {
        sizes[2] += v.sequence.size();
        sizes[3] += v.reference_variant_indices.size();
    }
}
void orc_graph_fetch(void *h, uint64_t *seq_off, uint8_t *seq, uint16_t *vvar, uint16_t *vall, uint8_t *vflags, uint32_t *vnested, uint32_t *refvar_off, uint16_t *refvar, uint32_t *edges,
                     uint16_t *var_num_alleles, uint8_t *var_dep) {
    auto *g = (Graph *)h;
    uint64_t so = 0;
    uint32_t ro = 0;
    seq_off[0] = 0;
    refvar_off[0] = 0;
    for (size_t v = 0; v < g->graph.size(); v++) {
        auto &x = g->graph[v];
        for (uint8_t c : x.sequence) seq[so++] = c;
        seq_off[v + 1] = so;
        vvar[v] = x.variant_allele_idx.first;
        vall[v] = x.variant_allele_idx.second;
        vflags[v] = (uint8_t)((x.is_disconnected ? 1 : 0) | (x.is_first_nucleotides_redundant ? 2 : 0));
        vnested[v] = x.nested_variant_cluster_index;
        for (ushort r : x.reference_variant_indices) refvar[ro++] = r;
        refvar_off[v + 1] = ro;
    }
    for (size_t e = 0; e < g->edges.size(); e++) {
        edges[2 * e] = g->edges[e].first;
        edges[2 * e + 1] = g->edges[e].second;
    }
    for (size_t v = 0; v < g->num_alleles.size(); v++) {
        var_num_alleles[v] = g->num_alleles[v];
        var_dep[v] = g->has_dependency[v];
    }
}

}  // extern "C"
