#include "VariantFileParser.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "CountDistribution.hpp"   // doubleCompare

namespace bthost {

// ---------------------------------------------------------------------------------------------------------------
// Chromosomes
// ---------------------------------------------------------------------------------------------------------------
void Chromosomes::addSequence(const std::string &name, const std::string &sequence, bool is_decoy) {
    if (name.empty()) throw std::runtime_error("Chromosomes: empty sequence name");
    if (!order.emplace(name, (uint32_t)seqs.size()).second) throw std::runtime_error("Chromosome \"" + name + "\" appears multiple times in fasta file(s)");
    seqs.emplace_back(name, sequence);
    if (is_decoy) {
        decoys.insert(name);
        decoy_length += sequence.size();
    }
    total_length += sequence.size();
}

void Chromosomes::addFasta(const std::string &fasta_filename, bool is_decoy) {
    if (fasta_filename.empty()) return;
    std::ifstream in(fasta_filename);
    if (!in.is_open()) throw std::runtime_error("Unable to open file " + fasta_filename);
    bool any = false;
    for (std::string line; std::getline(in, line);) {
        if (!line.empty() && line[0] == '>') {
            const size_t cut = line.find_first_of("\t ");   // the name ends at the first blank
            addSequence(line.substr(1, cut == std::string::npos ? std::string::npos : cut - 1), "", is_decoy);
            any = true;
        } else {
            if (!any) throw std::runtime_error("Fasta file " + fasta_filename + " does not start with a header line");
            seqs.back().second.append(line);
            total_length += line.size();
            if (is_decoy) decoy_length += line.size();
        }
    }
}

void Chromosomes::convertToUpper() {
    for (auto &s : seqs)
        for (char &c : s.second) c = (char)std::toupper((unsigned char)c);
}

int Chromosomes::find(const std::string &name) const {
    auto it = order.find(name);
    return it == order.end() ? -1 : (int)it->second;
}

bool ClusterGroupCompare(const ClusterGroup &first, const ClusterGroup &second) {
    if (first.num_variants != second.num_variants) return first.num_variants > second.num_variants;
    return first.region() > second.region();
}

std::string dumpClusterGroups(const std::vector<ClusterGroup> &groups) {
    std::ostringstream os;
    for (size_t g = 0; g < groups.size(); g++) {
        const ClusterGroup &G = groups[g];
        os << "GROUP " << g << " region=" << G.region() << " nvar=" << G.num_variants << " sources=";
        for (size_t i = 0; i < G.source_vertices.size(); i++) os << (i ? "," : "") << G.source_vertices[i];
        os << "\n";
        for (size_t v = 0; v < G.clusters.size(); v++) {
            const VariantCluster &c = G.clusters[v];
            os << " VERTEX " << v << " cluster_idx=" << c.cluster_idx << " chrom=" << c.chrom_name << " left=" << c.left_flank << " right=" << c.right_flank << " edges=";
            for (size_t i = 0; i < G.out_edges[v].size(); i++) os << (i ? "," : "") << G.out_edges[v][i];
            os << " contained=";
            bool first = true;
            for (auto &cc : c.contained_clusters) {
                os << (first ? "" : ";") << cc.cluster_idx << ":" << cc.left_flank << ":" << cc.right_flank;
                first = false;
            }
            os << "\n";
            for (auto &pv : c.variants) {
                const Variant &x = pv.second;
                os << "  VAR pos=" << pv.first << " id=" << x.id << " dep=" << (x.has_dependency ? 1 : 0) << " type=" << (int)x.type << " red=" << x.num_redundant_nucleotides << " alts=";
                for (size_t a = 0; a < x.alt_alleles.size(); a++) os << (a ? "|" : "") << x.alt_alleles[a].ref_length << ":" << x.alt_alleles[a].sequence << ":" << x.alt_alleles[a].aco_att;
                os << "\n";
            }
        }
    }
    return os.str();
}

// ---------------------------------------------------------------------------------------------------------------
// reading
// ---------------------------------------------------------------------------------------------------------------
std::string VariantFileParser::readVariantFile(const std::string &variant_filename) {
    const auto ends_with = [&](const char *suffix) {
        const std::string s(suffix);
        return variant_filename.size() >= s.size() && variant_filename.compare(variant_filename.size() - s.size(), s.size(), s) == 0;
    };
    if (!ends_with(".vcf") && !ends_with(".vcf.gz")) throw std::runtime_error("Variant file " + variant_filename + " is neither .vcf nor .vcf.gz");
    gzFile f = gzopen(variant_filename.c_str(), "rb");   // reads plain files transparently
    if (!f) throw std::runtime_error("Unable to open file " + variant_filename);
    std::string out;
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
    gzclose(f);
    if (n < 0) throw std::runtime_error("Error while reading " + variant_filename);
    return out;
}

VariantFileParser::VariantFileParser(std::string vcf_text, unsigned kmer_size_in, uint32_t max_allele_length_in, float copy_number_variant_threshold_in)
    : kmer_size(kmer_size_in), max_allele_length(max_allele_length_in), copy_number_variant_threshold(copy_number_variant_threshold_in),
      allele_type_counter(ALLELE_COUNT_SIZE, 0), variant_type_counter((size_t)VariantType::VARIANT_TYPE_SIZE, 0), text(std::move(vcf_text)), variant_line(6) {
    // total number of variant lines, then position the cursor behind the "#CHROM" line (VariantFileParser.cpp:86-119)
    size_t header_end = std::string::npos;
    for (size_t p = 0; p < text.size();) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        if (text[p] != '#') {
            if (e > p) total_num_variants++;
        } else if (text.compare(p, 6, "#CHROM") == 0) {
            const size_t columns = (size_t)std::count(text.begin() + p, text.begin() + e, '\t') + 1;
            if (columns < 8) throw std::runtime_error("VCF header line has fewer than 8 columns");
            has_format = columns > 8;
            header_end = e + 1;
        }
        p = e + 1;
    }
    if (header_end == std::string::npos) throw std::runtime_error("VCF has no #CHROM header line");
    cursor = std::min(header_end, text.size());
    line_good = updateVariantLine();
}

// one data line into variant_line; false when none is left (VariantFileParser.cpp:148-171)
bool VariantFileParser::updateVariantLine() {
    if (cursor >= text.size()) {
        for (auto &f : variant_line) f.clear();
        return false;
    }
    size_t e = text.find('\n', cursor);
    const bool terminated = e != std::string::npos;
    if (!terminated) e = text.size();
    size_t p = cursor;
    auto field = [&]() {
        size_t t = text.find('\t', p);
        if (t == std::string::npos || t > e) t = e;
        std::string s = text.substr(p, t - p);
        p = std::min(t + 1, e);
        return s;
    };
    for (int i = 0; i < 5; i++) variant_line[i] = field();
    field();   // QUAL
    field();   // FILTER
    variant_line[5] = field();   // INFO (up to the FORMAT column or the end of the line)
    cursor = terminated ? e + 1 : text.size();
    return !variant_line[0].empty();
}

void VariantFileParser::addSequenceToInterclusterRegions(const std::string &chrom_name, bool is_decoy, uint32_t start_position, uint32_t end_position) {
    intercluster_regions_length += (uint64_t)end_position - start_position + 1;
    if ((uint64_t)end_position - start_position + 1 >= kmer_size) intercluster_regions.push_back(InterClusterRegion{chrom_name, is_decoy, start_position, end_position});
}

static std::string upper(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), ::toupper);
    return s;
}
static std::vector<std::string> split(const std::string &s, char sep) {   // boost::split on one separator: n separators -> n + 1 fields
    std::vector<std::string> out;
    size_t p = 0;
    while (true) {
        const size_t t = s.find(sep, p);
        out.push_back(s.substr(p, t == std::string::npos ? std::string::npos : t - p));
        if (t == std::string::npos) break;
        p = t + 1;
    }
    return out;
}
static std::pair<std::string, bool> getInfoAttributeString(const std::string &info_str, const std::string &att_name) {   // :547-561
    for (const std::string &att : split(info_str, ';'))
        if (att.compare(0, att_name.size(), att_name) == 0 && att.size() > att_name.size() && att[att_name.size()] == '=') return {att.substr(att_name.size() + 1), true};
    return {"", false};
}

// ---------------------------------------------------------------------------------------------------------------
// one unit
// ---------------------------------------------------------------------------------------------------------------
bool VariantFileParser::constructVariantClusterGroups(std::vector<ClusterGroup> *groups, uint32_t min_unit_variants, const Chromosomes &chromosomes) {
    const int k = (int)kmer_size;
    bool is_first_unit_variant = true;
    uint32_t unit_variant_counter = 0;
    std::string cur_chrom_name;
    int chrom = prev_chrom_name.empty() ? -1 : chromosomes.find(prev_chrom_name);
    int cur_position = 0;
    int cur_group_end_position = prev_var_end_position;
    Open open;
    std::set<uint32_t> variant_dependencies;   // last reference positions of the alleles that are still open

    while (is_first_unit_variant ? line_good : (line_good = updateVariantLine())) {   // the first line of a unit is the one the previous unit stopped at
        is_first_unit_variant = false;
        cur_chrom_name = variant_line[0];
        cur_position = std::stoi(variant_line[1]) - 1;
        if (cur_chrom_name != prev_chrom_name) {
            if (!prev_chrom_name.empty()) {
                const int prev_chrom = chromosomes.find(prev_chrom_name);
                if (prev_chrom >= 0) {
                    closeGroup(&open, groups);
                    addSequenceToInterclusterRegions(prev_chrom_name, chromosomes.isDecoy(prev_chrom_name), (uint32_t)(prev_var_end_position + 1), (uint32_t)chromosomes.sequence(prev_chrom).size() - 1);
                    if (!intercluster_chromosomes.insert(prev_chrom_name).second)
                        throw std::runtime_error("Variants need to be sorted by contig; variants on contig \"" + prev_chrom_name + "\" is unordered");
                }
            }
            chrom = chromosomes.find(cur_chrom_name);
            prev_var_end_position = -1;
            cur_group_end_position = -1;
            variant_dependencies.clear();
        } else if (prev_position > cur_position) {
            throw std::runtime_error("Variants need to be sorted by position; \"" + std::to_string(prev_position) + "\" is before \"" + std::to_string(cur_position) + "\" on contig \"" +
                                     cur_chrom_name + "\"");
        } else if (prev_position == cur_position) {
            // the reference prints this message and then fails on the duplicate map key
            throw std::runtime_error("Variants on the same position need to be multi-allelic; multiple variants observed on position \"" + std::to_string(prev_position) + "\" on contig \"" +
                                     cur_chrom_name + "\"");
        }
        prev_chrom_name = cur_chrom_name;
        while (!variant_dependencies.empty() && (int)*variant_dependencies.begin() < cur_position) variant_dependencies.erase(variant_dependencies.begin());
        if (unit_variant_counter >= min_unit_variants && (cur_position - cur_group_end_position) >= k) break;   // the line stays pending for the next unit
        prev_position = cur_position;

        const std::string var_ref_seq = upper(variant_line[3]);
        std::vector<std::string> alt_alleles = split(variant_line[4], ',');
        std::vector<std::string> origin_allele_att;
        const auto origin_att_str = getInfoAttributeString(variant_line[5], "ACO");
        if (origin_att_str.second) origin_allele_att = split(origin_att_str.first, ',');
        else origin_allele_att.assign(alt_alleles.size(), "");
        if (origin_allele_att.size() != alt_alleles.size()) throw std::runtime_error("ACO attribute does not have one entry per alternative allele (" + variant_line[2] + ")");

        Variant cur_variant;
        cur_variant.id = variant_line[2];
        cur_variant.has_dependency = !variant_dependencies.empty();
        num_variants += 1;
        unit_variant_counter += 1;
        if (alt_alleles.back() == "*") alt_alleles.pop_back();   // the missing allele is implied by has_dependency
        if (alt_alleles.empty()) throw std::runtime_error("Variant without alternative allele (" + variant_line[2] + ")");
        allele_type_counter[Total] += (uint32_t)alt_alleles.size();

        if (chrom >= 0 && chromosomes.isDecoy(cur_chrom_name)) {
            allele_type_counter[Excluded_decoy] += (uint32_t)alt_alleles.size();
            variant_type_counter[(size_t)VariantType::Unsupported]++;
            continue;
        }
        if (chrom < 0) {
            allele_type_counter[Excluded_genome] += (uint32_t)alt_alleles.size();
            variant_type_counter[(size_t)VariantType::Unsupported]++;
            continue;
        }
        const std::string &chrom_sequence = chromosomes.sequence(chrom);
        std::vector<std::string> ref_alleles(alt_alleles.size(), var_ref_seq);
        for (size_t i = 0; i < alt_alleles.size(); i++) {
            alt_alleles[i] = upper(alt_alleles[i]);
            rightTrimAllele(&ref_alleles[i], &alt_alleles[i]);
        }
        bool is_excluded = false;
        const std::string gen_ref_seq = upper((size_t)cur_position <= chrom_sequence.size() ? chrom_sequence.substr(cur_position, variant_line[3].size()) : std::string());
        if (var_ref_seq != gen_ref_seq) {
            allele_type_counter[Excluded_match] += (uint32_t)alt_alleles.size();
            is_excluded = true;
        }
        if (cur_position < k - 1) {
            allele_type_counter[Excluded_end] += (uint32_t)alt_alleles.size();
            is_excluded = true;
        }
        std::unordered_set<uint16_t> excluded_alleles;
        if (!is_excluded) {
            for (size_t i = 0; i < alt_alleles.size(); i++) {
                if ((uint64_t)cur_position + ref_alleles[i].size() - 1 + kmer_size > chrom_sequence.size()) {
                    allele_type_counter[Excluded_end]++;
                    excluded_alleles.insert((uint16_t)i);
                } else if (ref_alleles[i].size() > max_allele_length || alt_alleles[i].size() > max_allele_length) {
                    allele_type_counter[Excluded_length]++;
                    excluded_alleles.insert((uint16_t)i);
                } else {
                    variant_dependencies.insert((uint32_t)(cur_position + ref_alleles[i].size() - 1));
                }
            }
        }
        if (is_excluded || excluded_alleles.size() == alt_alleles.size()) {
            variant_type_counter[(size_t)VariantType::Unsupported]++;
            continue;
        }
        if ((cur_position - cur_group_end_position) >= k) closeGroup(&open, groups);
        if (cur_position > prev_var_end_position + 1) addSequenceToInterclusterRegions(cur_chrom_name, false, (uint32_t)(prev_var_end_position + 1), (uint32_t)(cur_position - 1));

        std::set<uint32_t> cur_end_positions;
        for (size_t a = 0; a < alt_alleles.size(); a++) {
            if (excluded_alleles.count((uint16_t)a)) continue;
            addAlternativeAllele(&cur_variant, ref_alleles[a], alt_alleles[a], origin_allele_att[a]);
            const uint32_t after_ref = (uint32_t)(cur_position + ref_alleles[a].size());
            const uint32_t ref_cnv = copyNumberVariantLength(ref_alleles[a], chrom_sequence, after_ref);
            const uint32_t alt_cnv = copyNumberVariantLength(alt_alleles[a], chrom_sequence, after_ref);
            cur_end_positions.insert(after_ref - 1);
            cur_group_end_position = std::max(cur_group_end_position, (int)(after_ref - 1 + std::max(ref_cnv, alt_cnv)));
        }
        prev_var_end_position = std::max(prev_var_end_position, (int)*cur_end_positions.rbegin());
        clusterVariants(cur_variant, (uint32_t)cur_position, cur_end_positions, cur_chrom_name, &open);
        variant_type_counter[(size_t)cur_variant.type]++;
    }
    closeGroup(&open, groups);

    if (total_num_variants == num_variants) {   // the whole file has been read: close the last contig, add the untouched ones (:519-544)
        if (chrom >= 0) {
            addSequenceToInterclusterRegions(cur_chrom_name, chromosomes.isDecoy(cur_chrom_name), (uint32_t)(prev_var_end_position + 1), (uint32_t)chromosomes.sequence(chrom).size() - 1);
            intercluster_chromosomes.insert(cur_chrom_name);
        }
        for (size_t c = 0; c < chromosomes.size(); c++)
            if (intercluster_chromosomes.insert(chromosomes.name(c)).second)
                addSequenceToInterclusterRegions(chromosomes.name(c), chromosomes.isDecoy(chromosomes.name(c)), 0, (uint32_t)chromosomes.sequence(c).size() - 1);
    }
    return num_variants == total_num_variants;
}

// ---------------------------------------------------------------------------------------------------------------
// alleles
// ---------------------------------------------------------------------------------------------------------------
void VariantFileParser::rightTrimAllele(std::string *ref_allele, std::string *alt_allele) {
    while (ref_allele->size() > 1 && alt_allele->size() > 1 && ref_allele->back() == alt_allele->back()) {
        ref_allele->pop_back();
        alt_allele->pop_back();
    }
}

VariantType VariantFileParser::classifyAllele(int reference_size, int allele_size) {
    if (reference_size == 1 && allele_size == 1) return VariantType::SNV;
    if (reference_size == 0 || allele_size == 0) return allele_size > reference_size ? VariantType::Insertion : VariantType::Deletion;
    return VariantType::Complex;
}

void VariantFileParser::addAlternativeAllele(Variant *cur_variant, const std::string &ref_allele, const std::string &alt_allele, const std::string &origin_att) const {   // :582-622
    uint32_t identical_left_nucleotides = 0;
    while (identical_left_nucleotides < ref_allele.size() && identical_left_nucleotides < alt_allele.size() && ref_allele[identical_left_nucleotides] == alt_allele[identical_left_nucleotides])
        identical_left_nucleotides++;
    cur_variant->num_redundant_nucleotides = std::min(cur_variant->num_redundant_nucleotides, identical_left_nucleotides);
    cur_variant->alt_alleles.push_back(AlleleInfo{(uint32_t)ref_allele.size(), alt_allele, origin_att});
    const VariantType variant_type = classifyAllele((int)(ref_allele.size() - identical_left_nucleotides), (int)(alt_allele.size() - identical_left_nucleotides));
    if (cur_variant->type == VariantType::Unsupported) cur_variant->type = variant_type;
    else if (cur_variant->type != variant_type) cur_variant->type = VariantType::Mixture;
}

namespace {
// canonical k-mers of a sequence window by window (Kmer.tpp:182-255): position 0 is the most significant pair of bits, A < C < G < T
struct CanonicalRoller {
    typedef unsigned __int128 u128;
    const unsigned k;
    const u128 mask;
    u128 fwd = 0, rc = 0;
    unsigned filled = 0;
    explicit CanonicalRoller(unsigned k_in) : k(k_in), mask(k_in == 64 ? ~(u128)0 : (((u128)1 << (2 * k_in)) - 1)) {}
    void reset() { filled = 0; }
    bool move(char nt) {   // true when a complete window ends at this nucleotide
        unsigned code;
        switch (nt) {
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': code = 3; break;
            default: reset(); return false;
        }
        fwd = ((fwd << 2) | code) & mask;
        rc = (rc >> 2) | ((u128)(3 - code) << (2 * (k - 1)));
        if (filled < k) filled++;
        return filled == k;
    }
    u128 lowest() const { return fwd < rc ? fwd : rc; }
};
}  // namespace

// How far the sequence following an allele repeats the allele's own k-mers (a copy-number-like context extends the group's end
// so that no other group can start inside it): VariantFileParser.cpp:649-733
uint32_t VariantFileParser::copyNumberVariantLength(const std::string &allele_sequence, const std::string &chrom_sequence, uint32_t chrom_start_position) const {
    uint32_t copy_number_variant_length = 0;
    if (allele_sequence.size() < kmer_size) return 0;
    CanonicalRoller roller(kmer_size);
    std::vector<CanonicalRoller::u128> allele_kmers;
    for (char nt : allele_sequence)
        if (roller.move(nt)) allele_kmers.push_back(roller.lowest());
    if (allele_kmers.empty()) return 0;
    std::sort(allele_kmers.begin(), allele_kmers.end());
    uint32_t chrom_window_end_position = (uint32_t)std::min<uint64_t>((uint64_t)chrom_start_position + copy_number_variant_length + allele_sequence.size(), chrom_sequence.size());
    while (true) {
        roller.reset();
        uint32_t num_bases = 0, num_identical_kmers = 0;
        std::pair<double, uint32_t> highest_scoring_window(0, 0);
        for (uint32_t chrom_position = chrom_start_position + copy_number_variant_length; chrom_position < chrom_window_end_position; chrom_position++) {
            if (roller.move(chrom_sequence[chrom_position]) && std::binary_search(allele_kmers.begin(), allele_kmers.end(), roller.lowest())) num_identical_kmers++;
            num_bases++;
            if (num_identical_kmers > 0) {
                const double identical_kmer_fraction = num_identical_kmers / static_cast<double>(num_bases - kmer_size + 1);
                if (doubleCompare(identical_kmer_fraction, highest_scoring_window.first) || identical_kmer_fraction > highest_scoring_window.first) {
                    highest_scoring_window.first = identical_kmer_fraction;
                    highest_scoring_window.second = num_bases;
                }
            }
        }
        if (highest_scoring_window.first < copy_number_variant_threshold) break;
        copy_number_variant_length += highest_scoring_window.second;
        if (chrom_window_end_position == chrom_sequence.size()) break;
        chrom_window_end_position = (uint32_t)std::min<uint64_t>((uint64_t)chrom_start_position + copy_number_variant_length + allele_sequence.size(), chrom_sequence.size());
    }
    return copy_number_variant_length;
}

// ---------------------------------------------------------------------------------------------------------------
// clustering inside the open group (VariantFileParser.cpp:735-978)
// ---------------------------------------------------------------------------------------------------------------
void VariantFileParser::clusterVariants(const Variant &cur_variant, uint32_t cur_position, const std::set<uint32_t> &cur_end_positions, const std::string &cur_chrom_name, Open *open) {
    const int k = (int)kmer_size;
    auto &flanks = open->flanks;
    // flank positions a full k-mer behind the variant can no longer be reached
    while (!flanks.empty() && (int)(cur_position - flanks.begin()->first) >= k) flanks.erase(flanks.begin());

    // The clusters this variant touches: its start or one of its allele ends lies within a k-mer of one of their flank positions,
    // or an allele spans a flank position.  The first one found (ascending flank position) takes the variant, the others are
    // merged with it when the group is closed.  The reference keeps those "second overlaps" in a set ordered by heap address;
    // clusters are allocated one after the other, so creation order (= cluster index) is used here.
    VariantCluster *first = nullptr;
    std::map<uint32_t, VariantCluster *> second_overlaps;   // cluster_idx -> cluster
    auto touch = [&](VariantCluster *c) {
        if (first == nullptr) first = c;
        else if (first != c) second_overlaps.emplace(c->cluster_idx, c);
    };
    for (auto vit = flanks.begin(); vit != flanks.end();) {
        if (std::abs((int)(cur_position - vit->first)) + 1 <= k) {
            const bool was_unset = first == nullptr;
            touch(vit->second);
            if (was_unset && cur_position >= vit->first) {   // a flank behind the variant that led to its cluster is used up
                vit = flanks.erase(vit);
                continue;
            }
        }
        for (uint32_t end : cur_end_positions) {
            if (std::abs((int)(end - vit->first)) + 1 <= k) touch(vit->second);
            else if (cur_position < vit->first && end > vit->first) touch(vit->second);
        }
        ++vit;
    }
    const uint32_t last_end = *cur_end_positions.rbegin();
    if (first == nullptr) {
        std::unique_ptr<VariantCluster> cluster(new VariantCluster());
        cluster->cluster_idx = (uint32_t)open->clusters.size();
        cluster->left_flank = cur_position;
        cluster->right_flank = last_end;
        cluster->chrom_name = cur_chrom_name;
        cluster->variants.emplace(cur_position, cur_variant);
        first = cluster.get();
        open->clusters.emplace(cluster->cluster_idx, std::move(cluster));
    } else {
        if (!first->variants.emplace(cur_position, cur_variant).second) throw std::runtime_error("two variants at position " + std::to_string(cur_position + 1) + " of " + cur_chrom_name);
        first->right_flank = std::max(last_end, first->right_flank);
    }
    // flank positions of the variant: every allele end, and its start when an allele is at least a k-mer long (an entry that
    // already belongs to another — overlapping — cluster stays)
    for (uint32_t end : cur_end_positions) flanks.emplace(end, first);
    if (last_end - cur_position >= kmer_size) flanks.emplace(cur_position, first);

    if (second_overlaps.empty()) return;
    // union of the merge sets that hold any of the clusters involved (:931-977)
    auto &sets = open->merge_sets;
    auto found_set = sets.end();
    for (auto sit = sets.begin(); sit != sets.end();) {
        if (sit->count(first->cluster_idx) > 0) {
            if (found_set == sets.end()) found_set = sit;
            else if (sit != found_set) {
                found_set->insert(sit->begin(), sit->end());
                sit = sets.erase(sit);
                continue;
            }
        }
        bool merged_cluster_merge_sets = false;
        for (auto &lit : second_overlaps) {
            if (sit->count(lit.first) > 0) {
                if (found_set == sets.end()) found_set = sit;
                else if (sit != found_set) {
                    found_set->insert(sit->begin(), sit->end());
                    sit = sets.erase(sit);
                    merged_cluster_merge_sets = true;
                    break;
                }
            }
        }
        if (!merged_cluster_merge_sets) ++sit;
    }
    if (found_set == sets.end()) {
        sets.emplace_back();
        found_set = std::prev(sets.end());
    }
    found_set->insert(first->cluster_idx);
    for (auto &lit : second_overlaps) found_set->insert(lit.first);
}

// every merge set collapses into the cluster its iteration starts with (VariantFileParser.cpp:1003-1042)
void VariantFileParser::mergeVariantClusters(Group *group, const std::list<std::unordered_set<uint32_t>> &merge_sets) {
    for (auto &merge_set : merge_sets) {
        auto cur_cluster = merge_set.begin();
        VariantCluster &into = *group->at(*cur_cluster);
        for (++cur_cluster; cur_cluster != merge_set.end(); ++cur_cluster) {
            VariantCluster &from = *group->at(*cur_cluster);
            into.left_flank = std::min(into.left_flank, from.left_flank);
            into.right_flank = std::max(into.right_flank, from.right_flank);
            for (auto &variant : from.variants) into.variants.insert(variant);
            group->erase(*cur_cluster);
        }
    }
}

// child cluster -> the innermost cluster whose flanks strictly contain it (VariantFileParser.cpp:1108-1160)
std::unordered_map<uint32_t, uint32_t> VariantFileParser::getVariantClusterGroupDependencies(Group *group) {
    std::unordered_map<uint32_t, uint32_t> dependencies;
    for (auto first = group->begin(); first != group->end(); ++first) {
        auto nested = group->end();
        for (auto second = group->begin(); second != group->end(); ++second) {
            if (first == second) continue;
            if (first->second->left_flank > second->second->left_flank && first->second->right_flank < second->second->right_flank) {
                if (nested == group->end() || (second->second->left_flank > nested->second->left_flank && second->second->right_flank < nested->second->right_flank)) nested = second;
            }
        }
        if (nested != group->end()) dependencies.emplace(first->first, nested->second->cluster_idx);
    }
    for (auto &dep : dependencies) {
        VariantCluster &parent = *group->at(dep.second);
        const VariantCluster &child = *group->at(dep.first);
        auto pos = parent.contained_clusters.begin();
        while (pos != parent.contained_clusters.end() && pos->left_flank < child.left_flank) ++pos;
        parent.contained_clusters.insert(pos, ContainedCluster{child.cluster_idx, child.left_flank, child.right_flank});
    }
    return dependencies;
}

// processVariantClusterGroups (:980-1001) + processVariantClusterGroupsCallback (:1044-1106) + VariantClusterGroup's constructor
void VariantFileParser::closeGroup(Open *open, std::vector<ClusterGroup> *groups) {
    if (!open->clusters.empty()) {
        mergeVariantClusters(&open->clusters, open->merge_sets);
        const auto dependencies = getVariantClusterGroupDependencies(&open->clusters);
        ClusterGroup g;
        g.chrom_name = open->clusters.begin()->second->chrom_name;
        g.start_position = 0xFFFFFFFFu;
        std::unordered_map<uint32_t, uint32_t> vertex_of;
        for (auto &entry : open->clusters) {   // vertex order = iteration order of the group's hash map
            const VariantCluster &c = *entry.second;
            if (dependencies.count(c.cluster_idx) < 1) g.source_vertices.push_back((uint32_t)g.clusters.size());
            vertex_of.emplace(c.cluster_idx, (uint32_t)g.clusters.size());
            g.start_position = std::min(g.start_position, c.left_flank + 1);
            g.end_position = std::max(g.end_position, c.right_flank + 1);
            g.num_variants += (uint32_t)c.variants.size();
            g.clusters.push_back(c);
        }
        g.out_edges.resize(g.clusters.size());
        for (auto &dep : dependencies) g.out_edges[vertex_of.at(dep.second)].push_back(vertex_of.at(dep.first));
        num_variant_clusters += (uint32_t)g.clusters.size();
        num_variant_cluster_groups++;
        groups->push_back(std::move(g));
        Group().swap(open->clusters);   // a fresh table: the bucket count of the next group's map starts from scratch, as a new map's does
    }
    open->merge_sets.clear();
    open->flanks.clear();
}

// ---------------------------------------------------------------------------------------------------------------
// intercluster regions
// ---------------------------------------------------------------------------------------------------------------
void VariantFileParser::sortInterclusterRegions() {
    std::sort(intercluster_regions.begin(), intercluster_regions.end(), [](const InterClusterRegion &a, const InterClusterRegion &b) {
        return (a.end_position - a.start_position) > (b.end_position - b.start_position);
    });
}

std::string VariantFileParser::interclusterRegionsText() const {
    std::ostringstream os;
    for (auto &r : intercluster_regions) os << r.chrom_name << "\t" << r.is_decoy << "\t" << r.start_position << "\t" << r.end_position << "\n";
    return os.str();
}

uint64_t VariantFileParser::getNumberOfInterclusterRegionKmers() const { return intercluster_regions_length - intercluster_regions.size() * (uint64_t)(kmer_size - 1); }

}  // namespace bthost
