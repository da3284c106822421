#include "VariantClusterGraph.hpp"

#include <algorithm>
#include <set>
#include <stdexcept>

namespace bthost {

namespace {
inline int ntCode(char c) {   // Nucleotide::ntToBit (Nucleotide.hpp:40-70): anything else splits the vertex
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}
constexpr uint16_t NONE16 = 0xFFFF;
constexpr uint32_t NONE32 = 0xFFFFFFFFu;
}  // namespace

uint32_t VariantClusterGraph::addVertex() {
    vertices.emplace_back();
    return (uint32_t)vertices.size() - 1;
}

// VariantClusterGraph.cpp:62-262
VariantClusterGraph::VariantClusterGraph(VariantCluster variant_cluster, const std::string &chrom_sequence, unsigned kmer_size) {
    if (variant_cluster.variants.empty()) throw std::invalid_argument("VariantClusterGraph: cluster without variants");
    if (variant_cluster.variants.size() >= NONE16) throw std::invalid_argument("VariantClusterGraph: too many variants");
    std::map<uint32_t, std::pair<std::vector<uint32_t>, std::vector<uint16_t>>> added_vertices;
    std::set<uint16_t> reference_variant_indices;   // the reference's unordered_set: only membership matters downstream
    auto refvec = [&]() { return std::vector<uint16_t>(reference_variant_indices.begin(), reference_variant_indices.end()); };
    auto variants_it = variant_cluster.variants.begin();
    const auto chrom_it = chrom_sequence.begin();
    if (variants_it->first < kmer_size - 1) throw std::invalid_argument("VariantClusterGraph: first variant closer than k-1 to the chromosome start");
    uint32_t cur_vertex = addVertex();
    addVertices(&cur_vertex, std::vector<StringItPair>(1, StringItPair(chrom_it + variants_it->first - (kmer_size - 1), chrom_it + variants_it->first)), {NONE16, NONE16},
                refvec(), {}, false);
    uint32_t prev_vertex = cur_vertex;
    added_vertices.insert({variants_it->first, {std::vector<uint32_t>(1, cur_vertex), {}}});
    uint32_t cur_last_position = 0, next_position = 0;
    uint16_t variant_counter = 0;
    while (variants_it != variant_cluster.variants.end()) {
        const Variant &var = variants_it->second;
        var_num_alleles.push_back((uint16_t)(1 + (var.has_dependency ? 1 : 0) + var.alt_alleles.size()));
        var_has_dependency.push_back(var.has_dependency ? 1 : 0);
        const bool is_first_nucleotides_redundant = var.num_redundant_nucleotides > 0;
        uint32_t max_reference_length = 0;
        for (uint16_t alt_allele_idx = 0; alt_allele_idx < var.alt_alleles.size(); alt_allele_idx++) {
            const AlleleInfo &alt = var.alt_alleles[alt_allele_idx];
            max_reference_length = std::max(max_reference_length, alt.ref_length);
            uint32_t next_vertex = addVertex();
            edges.emplace_back(cur_vertex, next_vertex);
            addVertices(&next_vertex, std::vector<StringItPair>(1, StringItPair(alt.sequence.begin(), alt.sequence.end())), {variant_counter, (uint16_t)(alt_allele_idx + 1)},
                        refvec(), {}, is_first_nucleotides_redundant);
            added_vertices[variants_it->first + alt.ref_length].first.push_back(next_vertex);
        }
        if (max_reference_length == 0) throw std::invalid_argument("VariantClusterGraph: variant without alternative alleles");
        added_vertices.at(variants_it->first + max_reference_length).second.push_back(variant_counter);
        reference_variant_indices.insert(variant_counter);
        variants_it++;
        bool more_edges = true, last_variant = false;
        if (variants_it != variant_cluster.variants.end()) next_position = variants_it->first;
        else {
            next_position = NONE32;
            last_variant = true;
        }
        while (more_edges) {
            auto added_it = added_vertices.begin();
            uint32_t cur_position = added_it->first;
            const std::vector<uint32_t> next_vertices = added_it->second.first;
            for (uint16_t variant_idx : added_it->second.second) reference_variant_indices.erase(variant_idx);
            added_vertices.erase(added_it);
            if (added_vertices.empty()) {
                more_edges = false;
                cur_last_position = last_variant ? cur_position + kmer_size - 1 : next_position;
            } else {
                cur_last_position = added_vertices.begin()->first;
                if (!last_variant && cur_last_position > next_position) {
                    more_edges = false;
                    cur_last_position = next_position;
                }
            }
            if (cur_last_position > chrom_sequence.size()) throw std::invalid_argument("VariantClusterGraph: cluster runs past the chromosome end");
            std::vector<StringItPair> contained_vertices;
            std::vector<uint32_t> nested_variant_cluster_indices;
            uint32_t prev_contained_edge = NONE32;
            auto contained_it = variant_cluster.contained_clusters.begin();
            while (contained_it != variant_cluster.contained_clusters.end() && contained_it->left_flank < cur_last_position) {
                if (!(cur_position <= contained_it->left_flank) || !(contained_it->right_flank <= cur_last_position - kmer_size))
                    throw std::invalid_argument("VariantClusterGraph: contained cluster does not fit its reference segment");
                if (prev_contained_edge < NONE32) nested_variant_cluster_indices.push_back(prev_contained_edge);
                contained_vertices.emplace_back(chrom_it + cur_position, chrom_it + contained_it->left_flank);
                prev_contained_edge = contained_it->cluster_idx;
                cur_position = contained_it->right_flank + 1;
                contained_it = variant_cluster.contained_clusters.erase(contained_it);
            }
            if (prev_contained_edge < NONE32) nested_variant_cluster_indices.push_back(prev_contained_edge);
            contained_vertices.emplace_back(chrom_it + cur_position, chrom_it + cur_last_position);
            cur_vertex = addVertex();
            bool is_reference_allele = false;
            for (uint32_t v : next_vertices) {
                if (v == prev_vertex) is_reference_allele = true;
                edges.emplace_back(v, cur_vertex);
            }
            if (is_reference_allele)
                addVertices(&cur_vertex, contained_vertices, {variant_counter, (uint16_t)0}, refvec(), nested_variant_cluster_indices, is_first_nucleotides_redundant);
            else
                addVertices(&cur_vertex, contained_vertices, {NONE16, NONE16}, refvec(), nested_variant_cluster_indices, false);
            added_vertices[cur_last_position].first.push_back(cur_vertex);
        }
        variant_counter++;
        prev_vertex = cur_vertex;
    }
}

// VariantClusterGraph.cpp:290-316
void VariantClusterGraph::addVertices(uint32_t *cur_vertex, const std::vector<StringItPair> &vertex_sequences, std::pair<uint16_t, uint16_t> variant_allele_idx,
                                      const std::vector<uint16_t> &reference_variant_indices, const std::vector<uint32_t> &nested_variant_cluster_indices,
                                      bool is_first_nucleotides_redundant) {
    std::vector<uint16_t> vertex_reference_variant_indices;
    for (uint16_t r : reference_variant_indices)
        if (r != variant_allele_idx.first) vertex_reference_variant_indices.push_back(r);
    initVertex(cur_vertex, vertex_sequences.front(), variant_allele_idx, vertex_reference_variant_indices, NONE32, is_first_nucleotides_redundant);
    for (size_t i = 1; i < vertex_sequences.size(); i++) {
        const uint32_t prev_vertex = *cur_vertex;
        *cur_vertex = addVertex();
        edges.emplace_back(prev_vertex, *cur_vertex);
        initVertex(cur_vertex, vertex_sequences[i], variant_allele_idx, vertex_reference_variant_indices, nested_variant_cluster_indices.at(i - 1), false);
    }
}

// VariantClusterGraph.cpp:318-377: a run of non-ACGT characters closes the vertex and opens a disconnected one
void VariantClusterGraph::initVertex(uint32_t *cur_vertex, StringItPair vertex_sequence, std::pair<uint16_t, uint16_t> variant_allele_idx,
                                     const std::vector<uint16_t> &vertex_reference_variant_indices, uint32_t nested_variant_cluster_index, bool is_first_nucleotides_redundant) {
    GraphVertex *v = &vertices[*cur_vertex];
    v->variant = variant_allele_idx.first;
    v->allele = variant_allele_idx.second;
    v->reference_variant_indices = vertex_reference_variant_indices;
    v->nested_variant_cluster_index = nested_variant_cluster_index;
    v->is_first_nucleotides_redundant = is_first_nucleotides_redundant;
    v->is_disconnected = nested_variant_cluster_index != NONE32;
    bool prev_is_disconnected = false;
    while (vertex_sequence.first != vertex_sequence.second) {
        const int code = ntCode(*vertex_sequence.first);
        if (code < 0) {
            if (!prev_is_disconnected) {
                const uint32_t prev_vertex = *cur_vertex;
                *cur_vertex = addVertex();
                edges.emplace_back(prev_vertex, *cur_vertex);
                v = &vertices[*cur_vertex];
                v->variant = variant_allele_idx.first;
                v->allele = variant_allele_idx.second;
                v->reference_variant_indices = vertex_reference_variant_indices;
                v->nested_variant_cluster_index = NONE32;
                v->is_first_nucleotides_redundant = false;
                v->is_disconnected = true;
            }
            prev_is_disconnected = true;
        } else {
            v->sequence.push_back((uint8_t)code);
            prev_is_disconnected = false;
        }
        vertex_sequence.first++;
    }
}

void PathsBatchBuilder::add(const VariantClusterGraph &g, const std::vector<std::vector<uint8_t>> &best_paths) {
    const uint32_t nv = (uint32_t)g.vertices.size();
    std::vector<std::vector<uint32_t>> ins(nv);
    for (auto &e : g.edges) ins[e.second].push_back(e.first);
    for (uint32_t v = 0; v < nv; v++) {
        const GraphVertex &x = g.vertices[v];
        seq.insert(seq.end(), x.sequence.begin(), x.sequence.end());
        seq_off.push_back(seq.size());
        vertex_variant.push_back(x.variant);
        vertex_allele.push_back(x.allele);
        vertex_flags.push_back((uint8_t)((x.is_disconnected ? 1 : 0) | (x.is_first_nucleotides_redundant ? 2 : 0)));
        vertex_nested.push_back(x.nested_variant_cluster_index);
        refvar.insert(refvar.end(), x.reference_variant_indices.begin(), x.reference_variant_indices.end());
        refvar_off.push_back((uint32_t)refvar.size());
        in_src.insert(in_src.end(), ins[v].begin(), ins[v].end());
        in_off.push_back((uint32_t)in_src.size());
    }
    vertex_off.push_back(vertex_off.back() + nv);
    num_paths.push_back((uint32_t)best_paths.size());
    for (auto &row : best_paths) {
        if (row.size() != nv) throw std::invalid_argument("PathsBatchBuilder: best path row of the wrong length");
        path_vertices.insert(path_vertices.end(), row.begin(), row.end());
    }
    path_off.push_back(path_vertices.size());
    var_num_alleles.insert(var_num_alleles.end(), g.var_num_alleles.begin(), g.var_num_alleles.end());
    var_has_dependency.insert(var_has_dependency.end(), g.var_has_dependency.begin(), g.var_has_dependency.end());
    var_off.push_back((uint32_t)var_num_alleles.size());
}

const bt_paths_batch &PathsBatchBuilder::batch() {
    b.num_clusters = (uint32_t)num_paths.size();
    b.vertex_off = vertex_off.data();
    b.num_paths = num_paths.data();
    b.seq_off = seq_off.data();
    b.seq = seq.data();
    b.vertex_variant = vertex_variant.data();
    b.vertex_allele = vertex_allele.data();
    b.vertex_flags = vertex_flags.data();
    b.vertex_nested = vertex_nested.data();
    b.refvar_off = refvar_off.data();
    b.refvar = refvar.data();
    b.path_off = path_off.data();
    b.path_vertices = path_vertices.data();
    b.var_off = var_off.data();
    b.var_num_alleles = var_num_alleles.data();
    b.var_has_dependency = var_has_dependency.data();
    b.in_off = in_off.data();
    b.in_src = in_src.data();
    return b;
}

}  // namespace bthost
