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
//
// A unit is read record by record.  Every record passes three steps:
//   arrive   the record's contig / position against the previous record: contig change (the previous contig's tail becomes an
//            inter-cluster region), ordering errors, expiry of the reference stretches earlier alleles still cover, unit cut;
//   screen   which of its alternative alleles can be genotyped at all (decoy / unknown contig, REF that does not match the genome,
//            too close to a contig end, too long) — the reference's allele counters are kept here;
//   place    the surviving alleles join the open group (or close it first when the record lies k or more past everything the
//            group covers), and the gap since the previous variant becomes an inter-cluster region.
// Behaviour, counters and error texts are the reference's (VariantFileParser.cpp:185-545).
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct RecordAlleles {   // one VCF record's REF/ALT pairs after upper-casing and right-trimming, with their origin attributes
    std::string ref_field;                       // REF as written (upper case)
    std::vector<std::string> ref, alt, origin;   // per alternative allele
};
RecordAlleles readAlleles(const std::vector<std::string> &line) {
    RecordAlleles r;
    r.ref_field = upper(line[3]);
    r.alt = split(line[4], ',');
    const auto aco = getInfoAttributeString(line[5], "ACO");
    if (aco.second) r.origin = split(aco.first, ',');
    else r.origin.assign(r.alt.size(), "");
    if (r.origin.size() != r.alt.size()) throw std::runtime_error("ACO attribute does not have one entry per alternative allele (" + line[2] + ")");
    if (r.alt.back() == "*") {   // the missing allele is implied by has_dependency
        r.alt.pop_back();
        r.origin.pop_back();
    }
    if (r.alt.empty()) throw std::runtime_error("Variant without alternative allele (" + line[2] + ")");
    r.ref.assign(r.alt.size(), r.ref_field);
    for (size_t i = 0; i < r.alt.size(); i++) {
        r.alt[i] = upper(r.alt[i]);
        VariantFileParser::rightTrimAllele(&r.ref[i], &r.alt[i]);
    }
    return r;
}
}  // namespace

struct VariantFileParser::UnitScan {   // what a unit's scan carries from record to record
    const Chromosomes &chromosomes;
    std::vector<ClusterGroup> *groups;
    Open open;                          // the group under construction
    std::set<uint32_t> covered_until;   // last reference positions of the alleles of earlier records that are still open
    std::string contig;                 // contig of the current record
    int contig_index = -1;              // its index in the genome (-1: not in the genome)
    int group_reach = -1;               // last position the open group covers (alleles + their copy-number repeats)
    uint32_t records = 0;               // records taken into this unit
};

// step 1.  Returns false when the unit ends before this record (the record stays pending for the next unit).
bool VariantFileParser::arrive(UnitScan *u, int position, uint32_t min_unit_variants) {
    const std::string &contig = variant_line[0];
    if (contig != prev_chrom_name) {
        if (!prev_chrom_name.empty()) {
            const int left = u->chromosomes.find(prev_chrom_name);
            if (left >= 0) {   // the contig just left: its open group is complete, its tail is inter-cluster sequence
                closeGroup(&u->open, u->groups);
                addSequenceToInterclusterRegions(prev_chrom_name, u->chromosomes.isDecoy(prev_chrom_name), (uint32_t)(prev_var_end_position + 1), (uint32_t)u->chromosomes.sequence(left).size() - 1);
                if (!intercluster_chromosomes.insert(prev_chrom_name).second)
                    throw std::runtime_error("Variants need to be sorted by contig; variants on contig \"" + prev_chrom_name + "\" is unordered");
            }
        }
        u->contig_index = u->chromosomes.find(contig);
        prev_var_end_position = -1;
        u->group_reach = -1;
        u->covered_until.clear();
    } else if (prev_position > position) {
        throw std::runtime_error("Variants need to be sorted by position; \"" + std::to_string(prev_position) + "\" is before \"" + std::to_string(position) + "\" on contig \"" + contig + "\"");
    } else if (prev_position == position) {   // (the reference prints this message and then fails on the duplicate map key)
        throw std::runtime_error("Variants on the same position need to be multi-allelic; multiple variants observed on position \"" + std::to_string(prev_position) + "\" on contig \"" +
                                 contig + "\"");
    }
    u->contig = contig;
    prev_chrom_name = contig;
    while (!u->covered_until.empty() && (int)*u->covered_until.begin() < position) u->covered_until.erase(u->covered_until.begin());
    if (u->records >= min_unit_variants && (position - u->group_reach) >= (int)kmer_size) return false;
    prev_position = position;
    return true;
}

// step 2.  keep[i] = alternative allele i can be genotyped; returns false when the whole record is unsupported.
bool VariantFileParser::screen(UnitScan *u, int position, const std::vector<std::string> &ref, const std::vector<std::string> &alt, const std::string &ref_field, std::vector<bool> *keep) {
    const uint32_t n = (uint32_t)alt.size();
    allele_type_counter[Total] += n;
    keep->assign(n, false);
    if (u->contig_index >= 0 && u->chromosomes.isDecoy(u->contig)) {
        allele_type_counter[Excluded_decoy] += n;
        return false;
    }
    if (u->contig_index < 0) {
        allele_type_counter[Excluded_genome] += n;
        return false;
    }
    const std::string &sequence = u->chromosomes.sequence(u->contig_index);
    bool whole_record_out = false;
    if (ref_field != upper((size_t)position <= sequence.size() ? sequence.substr(position, ref_field.size()) : std::string())) {
        allele_type_counter[Excluded_match] += n;
        whole_record_out = true;
    }
    if (position < (int)kmer_size - 1) {
        allele_type_counter[Excluded_end] += n;
        whole_record_out = true;
    }
    if (whole_record_out) return false;
    uint32_t kept = 0;
    for (uint32_t i = 0; i < n; i++) {
        if ((uint64_t)position + ref[i].size() - 1 + kmer_size > sequence.size()) allele_type_counter[Excluded_end]++;
        else if (ref[i].size() > max_allele_length || alt[i].size() > max_allele_length) allele_type_counter[Excluded_length]++;
        else {
            (*keep)[i] = true;
            kept++;
            u->covered_until.insert((uint32_t)(position + ref[i].size() - 1));
        }
    }
    return kept > 0;
}

bool VariantFileParser::constructVariantClusterGroups(std::vector<ClusterGroup> *groups, uint32_t min_unit_variants, const Chromosomes &chromosomes) {
    UnitScan u{chromosomes, groups};
    u.contig_index = prev_chrom_name.empty() ? -1 : chromosomes.find(prev_chrom_name);
    u.contig = prev_chrom_name;
    u.group_reach = prev_var_end_position;
    // the first record of a unit is the one the previous unit stopped at (still in variant_line)
    for (bool pending = true; pending ? line_good : (line_good = updateVariantLine()); pending = false) {
        const int position = std::stoi(variant_line[1]) - 1;
        if (!arrive(&u, position, min_unit_variants)) break;
        const RecordAlleles rec = readAlleles(variant_line);
        Variant variant;
        variant.id = variant_line[2];
        variant.has_dependency = !u.covered_until.empty();
        num_variants += 1;
        u.records += 1;
        std::vector<bool> keep;
        if (!screen(&u, position, rec.ref, rec.alt, rec.ref_field, &keep)) {
            variant_type_counter[(size_t)VariantType::Unsupported]++;
            continue;
        }
        // step 3: place
        if ((position - u.group_reach) >= (int)kmer_size) closeGroup(&u.open, groups);
        if (position > prev_var_end_position + 1) addSequenceToInterclusterRegions(u.contig, false, (uint32_t)(prev_var_end_position + 1), (uint32_t)(position - 1));
        const std::string &sequence = chromosomes.sequence(u.contig_index);
        std::set<uint32_t> allele_ends;
        for (size_t i = 0; i < rec.alt.size(); i++) {
            if (!keep[i]) continue;
            addAlternativeAllele(&variant, rec.ref[i], rec.alt[i], rec.origin[i]);
            const uint32_t after = (uint32_t)(position + rec.ref[i].size());
            const uint32_t repeat = std::max(copyNumberVariantLength(rec.ref[i], sequence, after), copyNumberVariantLength(rec.alt[i], sequence, after));
            allele_ends.insert(after - 1);
            u.group_reach = std::max(u.group_reach, (int)(after - 1 + repeat));
        }
        prev_var_end_position = std::max(prev_var_end_position, (int)*allele_ends.rbegin());
        clusterVariants(variant, (uint32_t)position, allele_ends, u.contig, &u.open);
        variant_type_counter[(size_t)variant.type]++;
    }
    closeGroup(&u.open, groups);

    if (total_num_variants == num_variants) {   // the whole file has been read: the last contig's tail and every contig without variants (:519-544)
        if (u.contig_index >= 0) {
            addSequenceToInterclusterRegions(u.contig, chromosomes.isDecoy(u.contig), (uint32_t)(prev_var_end_position + 1), (uint32_t)chromosomes.sequence(u.contig_index).size() - 1);
            intercluster_chromosomes.insert(u.contig);
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
