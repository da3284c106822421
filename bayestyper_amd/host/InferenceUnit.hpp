// InferenceUnit (include/bayesTyper/InferenceUnit.hpp:40-71): what `bayesTyper cluster` hands to `bayesTyper genotype` per unit — the
// unit's variant-cluster groups (sorted as main.cpp:247 sorts them) with their clusters, and per cluster the best paths found over all
// samples (best_paths_indices, VariantClusterGraph.hpp:98) — plus the file it travels in, <prefix>_unit_<i>/variant_clusters.bin.
//
// File format.  The reference writes gzip(boost::archive::binary_oarchive(InferenceUnit)); Boost is not part of this build, so the file
// here is gzip(own binary layout, magic "BTAMDUNIT1"): the two formats are NOT interchangeable.  The graphs
// themselves are not stored: `genotype` rebuilds them from the clusters and the genome (VariantClusterGraph is a pure function of both).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "VariantFileParser.hpp"

namespace bthost {

struct InferenceUnit {
    uint32_t index = 0;
    std::string cluster_options_header;
    uint32_t num_variants = 0, num_variant_clusters = 0;
    uint64_t num_path_kmers = 0;
    std::vector<ClusterGroup> variant_cluster_groups;
    // best_paths[g][v]: rows of |V(graph)| bytes (0/1), one per haplotype candidate of cluster v of group g
    std::vector<std::vector<std::vector<std::vector<uint8_t>>>> best_paths;

    void write(const std::string &filename) const;   // throws std::runtime_error
    static InferenceUnit read(const std::string &filename);
};

// gzip text/binary files (the reference's boost::iostreams gzip filters)
void writeGzFile(const std::string &filename, const std::string &content, unsigned threads = 1);   // one gzip member whatever the thread count (large contents: 4 MB pieces of one deflate stream, compressed on `threads` threads)
std::string readGzFile(const std::string &filename);   // also reads uncompressed files
// The k-mers of a text of one k-mer per line from offset `begin` on (parameter_kmers.fa.gz after its header line), [lo, hi] per k-mer in the packing of
// Nucleotide::ntToBit (symbol i in bits 2i, 2i+1); the lines are cut among `threads` threads when they all have k symbols (what the cluster stage writes).
// Throws std::runtime_error naming the first malformed line.
std::vector<uint64_t> parseKmerLines(const std::string &text, size_t begin, uint32_t k, unsigned threads);

}  // namespace bthost
