#include "GenotypeWriter.hpp"

#include <zlib.h>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace bthost {

uint32_t VariantInfo::maxReferenceLength() const {
    uint32_t max_ref_length = 0;
    for (auto &alt_allele : alt_alleles) max_ref_length = std::max(max_ref_length, alt_allele.ref_length);
    return max_ref_length;
}

std::vector<VariantInfo> variantClusterInfo(const VariantCluster &cluster) {
    std::vector<VariantInfo> info;
    info.reserve(cluster.variants.size());
    for (auto &v : cluster.variants) {
        VariantInfo vi;
        vi.position = v.first + 1;
        vi.id = v.second.id;
        vi.has_dependency = v.second.has_dependency;
        vi.alt_alleles = v.second.alt_alleles;
        info.push_back(std::move(vi));
    }
    return info;
}

std::string variantClusterRegion(const std::string &chrom_name, const std::vector<VariantInfo> &variant_cluster_info) {
    const uint32_t start_position = variant_cluster_info.front().position;
    uint32_t end_position = 0;
    for (auto &variant_info : variant_cluster_info) end_position = std::max(end_position, variant_info.position + variant_info.maxReferenceLength() - 1);
    return chrom_name + ":" + std::to_string(start_position) + "-" + std::to_string(end_position);
}

GenotypeWriter::GenotypeWriter(std::vector<std::string> sample_names, const Chromosomes &chromosomes_in) : samples(std::move(sample_names)), chromosomes(chromosomes_in) {}

void GenotypeWriter::addGenotypes(const ClusterAnnotation &where, const VariantInfo &variant_info, const VariantGenotypes &genotypes, const std::string &sample_columns) {
    append(where.chrom_name, formatGenotypes(where, variant_info, genotypes, sample_columns));
}

GenotypeWriter::GenotypedVariant GenotypeWriter::formatGenotypes(const ClusterAnnotation &where, const VariantInfo &variant_info, const VariantGenotypes &genotypes,
                                                                 const std::string &sample_columns) const {
    const int chrom = chromosomes.find(where.chrom_name);
    if (chrom < 0) throw std::runtime_error("GenotypeWriter: unknown chromosome " + where.chrom_name);
    const std::string &chrom_sequence = chromosomes.sequence((size_t)chrom);
    const uint32_t max_ref_length = variant_info.maxReferenceLength();
    if (variant_info.alt_alleles.empty() || max_ref_length == 0) throw std::runtime_error("GenotypeWriter: variant without alternative allele");
    std::ostringstream os;
    // ALT: every alternative allele padded with the reference nucleotides the longest reference allele covers beyond it (:145-172)
    for (size_t a = 0; a < variant_info.alt_alleles.size(); a++) {
        const AlleleInfo &alt = variant_info.alt_alleles[a];
        os << (a ? "," : "") << alt.sequence << chrom_sequence.substr(variant_info.position + alt.ref_length - 1, max_ref_length - alt.ref_length);
    }
    if (variant_info.has_dependency) os << ",*";
    os << "\t" << formatQualityFilterAndStats(genotypes);   // QUAL, FILTER, AC/AF/AN/ACP
    os << ";VCS=" << where.variant_cluster_size << ";VCR=" << where.variant_cluster_region << ";VCGS=" << where.variant_cluster_group_size << ";VCGR=" << where.variant_cluster_group_region
       << ";HC=" << where.num_candidates;
    os << formatAlleleCover(genotypes);   // ";ANC=..." or nothing
    os << ";ACO=";                        // :232-259
    for (size_t a = 0; a < variant_info.alt_alleles.size(); a++) os << (a ? "," : "") << (variant_info.alt_alleles[a].aco_att.empty() ? "." : variant_info.alt_alleles[a].aco_att);
    if (variant_info.has_dependency) os << ",.";
    os << "\tGT:GQ:GPP:APP:NAK:FAK:MAC:SAF" << sample_columns;
    return GenotypedVariant{variant_info.position, max_ref_length, variant_info.id, os.str()};
}

std::string GenotypeWriter::generateHeader(const std::string &genome_filename, const std::string &graph_options_header, const std::string &genotype_options_header) const {
    std::ostringstream h;
    h << "##fileformat=VCFv4.2\n";
    h << "##reference=file:" << genome_filename << "\n";
    for (size_t c = 0; c < chromosomes.size(); c++)
        if (!chromosomes.isDecoy(chromosomes.name(c))) h << "##contig=<ID=" << chromosomes.name(c) << ",length=" << chromosomes.sequence(c).size() << ">\n";
    h << graph_options_header << genotype_options_header;
    h << "##FILTER=<ID=AN0,Description=\"No called genotypes (AN = 0)\">\n";
    static const char *const info[][4] = {
        {"AC", "A", "Integer", "Alternative allele counts in called genotypes"},
        {"AF", "A", "Float", "Alternative allele frequencies in called genotypes"},
        {"AN", "1", "Integer", "Total number of alleles in called genotypes"},
        {"ACP", "R", "Float", "Allele call probabilites (maximum APP across samples)"},
        {"VCS", "1", "Integer", "Variant cluster size"},
        {"VCR", "1", "String", "Variant cluster region (<chromosome>:<start>-<end>)"},
        {"VCGS", "1", "Integer", "Variant cluster group size (number of variant clusters)"},
        {"VCGR", "1", "String", "Variant cluster group region (<chromosome>:<start>-<end>)"},
        {"HC", "1", "Integer", "Number of haplotype candidates used for inference in variant cluster"},
        {"ANC", ".", "String", "Allele(s) not covered by a haplotype candidate ('0': Reference allele)"},
        {"ACO", "A", "String", "Alternative allele call-set origin(s) (<call-set>:...)"}};
    for (auto &f : info) h << "##INFO=<ID=" << f[0] << ",Number=" << f[1] << ",Type=" << f[2] << ",Description=\"" << f[3] << "\">\n";
    static const char *const format[][4] = {
        {"GT", "1", "String", "Genotype"},
        {"GQ", "1", "Integer", "Genotype quality (phred-scaled 1 - max(GPP))"},
        {"GPP", "G", "Float", "Genotype posterior probabilities"},
        {"APP", "R", "Float", "Allele posterior probabilities"},
        {"NAK", "R", "Float", "Mean number of allele kmers across gibbs samples ('-1': Not sampled)"},
        {"FAK", "R", "Float", "Mean fraction of observed allele kmers across gibbs samples ('-1': Not sampled or NAK = 0)"},
        {"MAC", "R", "Float", "Mean allele kmer coverage (mean value) across gibbs samples ('-1': Not sampled or NAK = 0)"},
        {"SAF", "R", "Integer", "Sample specific allele filter ('0': PASS, '1': NAK, '2': FAK, '3': NAK and FAK)"}};
    for (auto &f : format) h << "##FORMAT=<ID=" << f[0] << ",Number=" << f[1] << ",Type=" << f[2] << ",Description=\"" << f[3] << "\">\n";
    h << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
    for (auto &s : samples) h << "\t" << s;
    h << "\n";
    return h.str();
}

std::string GenotypeWriter::vcfText(const std::string &genome_filename, const std::string &graph_options_header, const std::string &genotype_options_header) {
    std::ostringstream os;
    os << generateHeader(genome_filename, graph_options_header, genotype_options_header);
    for (size_t c = 0; c < chromosomes.size(); c++) {   // contigs in genome order, variants by position (:452-470)
        auto it = genotyped_variants.find(chromosomes.name(c));
        if (it == genotyped_variants.end()) continue;
        std::sort(it->second.begin(), it->second.end(), [](const GenotypedVariant &a, const GenotypedVariant &b) { return a.position < b.position; });
        for (auto &gv : it->second)
            os << it->first << "\t" << gv.position << "\t" << gv.variant_id << "\t" << chromosomes.sequence(c).substr(gv.position - 1, gv.max_ref_length) << "\t" << gv.genotypes << "\n";
    }
    return os.str();
}

uint32_t GenotypeWriter::finalise(const std::string &output_prefix, bool gzip_output, const std::string &genome_filename, const std::string &graph_options_header,
                                  const std::string &genotype_options_header) {
    const std::string text = vcfText(genome_filename, graph_options_header, genotype_options_header);
    const std::string filename = output_prefix + (gzip_output ? ".vcf.gz" : ".vcf");
    if (gzip_output) {
        gzFile f = gzopen(filename.c_str(), "wb");
        if (!f) throw std::runtime_error("Unable to write file " + filename);
        const bool ok = text.empty() || gzwrite(f, text.data(), (unsigned)text.size()) == (int)text.size();
        if (gzclose(f) != Z_OK || !ok) throw std::runtime_error("Error while writing " + filename);
    } else {
        std::ofstream f(filename);
        if (!f.is_open()) throw std::runtime_error("Unable to write file " + filename);
        f << text;
    }
    uint32_t n = 0;
    for (auto &c : genotyped_variants) n += (uint32_t)c.second.size();
    return n;
}

}  // namespace bthost
