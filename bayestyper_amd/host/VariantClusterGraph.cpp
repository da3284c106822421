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

// ---- graph assembly ------------------------------------------------------------------------------------------------------------------
// The graph of a cluster is its stretch of the reference (k-1 flank before the first variant .. k-1 after the last allele end) cut into
// BACKBONE pieces at every position where a variant starts or an alternative allele re-joins, plus one ALLELE piece per alternative
// allele hanging between the backbone vertex that ends at its variant's position and the backbone piece that starts where the allele
// ends.  Pieces are emitted left to right: a variant's allele pieces first (in allele order), then the backbone up to the next
// variant.  A piece becomes one vertex, or several in a row when it is cut — at a contained (nested) cluster, whose own graph replaces
// that part of the reference, and at runs of non-ACGT characters; every cut starts a "disconnected" vertex (no k-mer spans it).
// Vertex and edge numbering follow the reference's constructor (VariantClusterGraph.cpp:62-377), which the path search and the
// k-mer enumeration rely on (checked against an independent restatement of that constructor in tests/test_host_graph_cpu.py).
namespace {

struct Tail {                 // vertices whose sequence ends right before `position` and that still wait for their successor
    uint32_t position;
    std::vector<uint32_t> vertices;
    std::vector<uint16_t> closing;   // variants whose longest allele ends here: they stop overlapping the backbone from here on
};

class Assembler {
  public:
    Assembler(std::vector<GraphVertex> *vertices_in, std::vector<std::pair<uint32_t, uint32_t>> *edges_in, const std::string &chrom_in, unsigned k_in,
              std::list<ContainedCluster> *contained_in)
        : vertices(*vertices_in), edges(*edges_in), chrom(chrom_in), k(k_in), contained(*contained_in) {}

    // a piece over characters [begin, end) of `text`: one vertex per maximal ACGT stretch; returns the piece's LAST vertex.  `first` is
    // the piece's first vertex (already created by the caller, so that the caller can wire its in-edges).
    uint32_t fill(uint32_t first, const std::string &text, size_t begin, size_t end, uint16_t variant, uint16_t allele, uint32_t nested, bool redundant) {
        uint32_t at = first;
        stamp(at, variant, allele, nested, redundant, nested != NONE32);
        bool in_gap = false;
        for (size_t i = begin; i < end; i++) {
            const int code = ntCode(text[i]);
            if (code >= 0) {
                vertices[at].sequence.push_back((uint8_t)code);
                in_gap = false;
            } else if (!in_gap) {   // first character of a run of N (or anything else): the rest continues in a disconnected vertex
                const uint32_t next = append();
                edges.emplace_back(at, next);
                at = next;
                stamp(at, variant, allele, NONE32, false, true);
                in_gap = true;
            }
        }
        return at;
    }

    uint32_t append() {
        vertices.emplace_back();
        return (uint32_t)vertices.size() - 1;
    }

    // the backbone piece [from, to) of the reference, entered from `sources`; contained clusters inside it split it (each one's
    // stretch of the reference is skipped, the vertex after it carries the nested cluster's index).  Returns its last vertex.
    uint32_t backbone(uint32_t from, uint32_t to, const std::vector<uint32_t> &sources, uint16_t variant, uint16_t allele, bool redundant) {
        if (to > chrom.size()) throw std::invalid_argument("VariantClusterGraph: cluster runs past the chromosome end");
        uint32_t at = append();
        for (uint32_t s : sources) edges.emplace_back(s, at);
        uint32_t nested = NONE32, cursor = from;
        bool first_part = true;
        while (!contained.empty() && contained.front().left_flank < to) {
            const ContainedCluster inner = contained.front();
            contained.pop_front();
            if (inner.left_flank < cursor || inner.right_flank + k > to) throw std::invalid_argument("VariantClusterGraph: contained cluster does not fit its reference segment");
            at = part(at, cursor, inner.left_flank, variant, allele, nested, redundant && first_part, first_part);
            first_part = false;
            nested = inner.cluster_idx;
            cursor = inner.right_flank + 1;
        }
        return part(at, cursor, to, variant, allele, nested, redundant && first_part, first_part);
    }

    void openVariant(uint16_t v) { open.push_back(v); }
    void closeVariants(const std::vector<uint16_t> &vs) {
        for (uint16_t v : vs) open.erase(std::find(open.begin(), open.end(), v));
    }
    // a tail list kept sorted by position (a handful of entries at most)
    Tail &tailAt(uint32_t position) {
        auto it = std::lower_bound(tails.begin(), tails.end(), position, [](const Tail &t, uint32_t p) { return t.position < p; });
        if (it == tails.end() || it->position != position) it = tails.insert(it, Tail{position, {}, {}});
        return *it;
    }
    std::vector<Tail> tails;

  private:
    uint32_t part(uint32_t at, uint32_t from, uint32_t to, uint16_t variant, uint16_t allele, uint32_t nested, bool redundant, bool is_first) {
        if (!is_first) {
            const uint32_t next = append();
            edges.emplace_back(at, next);
            at = next;
        }
        return fill(at, chrom, from, to, variant, allele, nested, redundant);
    }
    void stamp(uint32_t v, uint16_t variant, uint16_t allele, uint32_t nested, bool redundant, bool disconnected) {
        GraphVertex &x = vertices[v];
        x.variant = variant;
        x.allele = allele;
        x.nested_variant_cluster_index = nested;
        x.is_first_nucleotides_redundant = redundant;
        x.is_disconnected = disconnected;
        x.reference_variant_indices.clear();
        for (uint16_t o : open)   // the variants still "open" over this stretch, except the one the vertex is an allele of
            if (o != variant) x.reference_variant_indices.push_back(o);
        std::sort(x.reference_variant_indices.begin(), x.reference_variant_indices.end());
    }

    std::vector<GraphVertex> &vertices;
    std::vector<std::pair<uint32_t, uint32_t>> &edges;
    const std::string &chrom;
    const unsigned k;
    std::list<ContainedCluster> &contained;
    std::vector<uint16_t> open;
};

}  // namespace

VariantClusterGraph::VariantClusterGraph(VariantCluster variant_cluster, const std::string &chrom_sequence, unsigned kmer_size) {
    const auto &variants = variant_cluster.variants;
    if (variants.empty()) throw std::invalid_argument("VariantClusterGraph: cluster without variants");
    if (variants.size() >= NONE16) throw std::invalid_argument("VariantClusterGraph: too many variants");
    const uint32_t first_position = variants.begin()->first;
    if (first_position < kmer_size - 1) throw std::invalid_argument("VariantClusterGraph: first variant closer than k-1 to the chromosome start");
    Assembler as(&vertices, &edges, chrom_sequence, kmer_size, &variant_cluster.contained_clusters);

    // left flank: the k-1 reference nucleotides before the first variant
    uint32_t trunk = as.fill(as.append(), chrom_sequence, first_position - (kmer_size - 1), first_position, NONE16, NONE16, NONE32, false);
    as.tailAt(first_position).vertices.push_back(trunk);

    uint16_t index = 0;
    for (auto it = variants.begin(); it != variants.end(); ++it, ++index) {
        const uint32_t position = it->first;
        const Variant &variant = it->second;
        if (variant.alt_alleles.empty()) throw std::invalid_argument("VariantClusterGraph: variant without alternative alleles");
        var_num_alleles.push_back((uint16_t)(1 + (variant.has_dependency ? 1 : 0) + variant.alt_alleles.size()));
        var_has_dependency.push_back(variant.has_dependency ? 1 : 0);
        const bool redundant = variant.num_redundant_nucleotides > 0;

        // the alternative alleles branch off the trunk vertex that ends at the variant and re-join where their reference stretch ends
        uint32_t longest = 0;
        for (size_t a = 0; a < variant.alt_alleles.size(); a++) {
            const AlleleInfo &alt = variant.alt_alleles[a];
            const uint32_t head = as.append();
            edges.emplace_back(trunk, head);
            const uint32_t tail = as.fill(head, alt.sequence, 0, alt.sequence.size(), index, (uint16_t)(a + 1), NONE32, redundant);
            as.tailAt(position + alt.ref_length).vertices.push_back(tail);
            longest = std::max(longest, alt.ref_length);
        }
        if (longest == 0) throw std::invalid_argument("VariantClusterGraph: variant whose alleles cover no reference nucleotide");
        as.tailAt(position + longest).closing.push_back(index);
        as.openVariant(index);

        // the trunk from the variant's position up to the next variant (or, after the last one, k-1 past the last allele end): one
        // backbone piece per stretch between consecutive re-join positions.  The piece entered from the old trunk vertex is the
        // variant's reference allele.
        const auto next = std::next(it);
        const bool last = next == variants.end();
        const uint32_t limit = last ? NONE32 : next->first;
        const uint32_t branch_vertex = trunk;
        for (bool done = false; !done;) {
            const Tail here = as.tails.front();
            as.tails.erase(as.tails.begin());
            as.closeVariants(here.closing);
            uint32_t until;
            if (as.tails.empty()) {
                until = last ? here.position + kmer_size - 1 : limit;
                done = true;
            } else {
                until = as.tails.front().position;
                if (!last && until > limit) {
                    until = limit;
                    done = true;
                }
            }
            const bool reference_allele = std::find(here.vertices.begin(), here.vertices.end(), branch_vertex) != here.vertices.end();
            trunk = as.backbone(here.position, until, here.vertices, reference_allele ? index : NONE16, reference_allele ? (uint16_t)0 : NONE16, reference_allele && redundant);
            as.tailAt(until).vertices.push_back(trunk);
        }
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
