// ORACLE (test infrastructure only — never linked into or called by the product): CPU restatement of the cluster stage's front
// end of BayesTyper: VariantFileParser (src/bayesTyper/VariantFileParser.cpp:185-1160), the constructor of VariantClusterGroup
// (src/bayesTyper/VariantClusterGroup.cpp:47-107) and the unit ordering of main.cpp:247.
//
// PARITY UNPINNED: VariantFileParser.cpp includes Boost headers (boost/algorithm/string.hpp, boost/iostreams), which this image
// lacks, so the reference TU cannot be compiled here and no reference-generated fixture exists; the reference holds no test
// vectors for it either.  This file follows the reference's control flow statement by statement (same container types, so the
// iteration orders of libstdc++'s unordered_map / unordered_set that decide cluster indices, vertex order and edge order are the
// reference's), single consumer thread.  One documented difference: the reference orders a variant's "second overlap" clusters
// by heap address (std::set<VariantCluster*>); here by cluster index, which is allocation order.
//
// Output: the same text dump bayestyper_amd/host/VariantFileParser.cpp produces (dumpClusterGroups, interclusterRegionsText).
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

typedef unsigned int uint;
typedef unsigned short ushort;
typedef unsigned char uchar;

bool doubleCompare(const double a, const double b) {   // Utils.hpp:81-90
    return ((a == b) or (std::abs(a - b) < std::abs(std::min(a, b)) * std::numeric_limits<double>::epsilon() * 100));
}

enum VariantType { SNV = 0, Insertion, Deletion, Complex, Mixture, Unsupported, VARIANT_TYPE_SIZE };
enum AlleleCount { Total = 0, Excluded_decoy, Excluded_genome, Excluded_match, Excluded_end, Excluded_length, ALLELE_COUNT_SIZE };

struct AlleleInfo {
    uint ref_length;
    std::string sequence, aco_att;
};
struct Variant {   // VariantCluster.hpp:56-71
    std::string id;
    bool has_dependency;
    int type = Unsupported;
    uint num_redundant_nucleotides = 0xFFFFFFFFu;
    std::vector<AlleleInfo> alt_alleles;
};
struct ContainedCluster {
    uint cluster_idx, left_flank, right_flank;
    bool operator<(const ContainedCluster &rhs) const { return left_flank < rhs.left_flank; }
};
struct VariantCluster {   // VariantCluster.hpp:99-108
    uint cluster_idx, left_flank, right_flank;
    std::string chrom_name;
    std::map<uint, Variant> variants;
    std::set<ContainedCluster> contained_clusters;
};
struct Group {   // VariantClusterGroup.cpp:47-107
    std::string chrom_name;
    uint start_position, end_position, num_variants;
    std::vector<VariantCluster> vertices;
    std::vector<uint> source_vertices;
    std::vector<std::vector<uint>> out_edges;
    std::string region() const { return chrom_name + ":" + std::to_string(start_position) + "-" + std::to_string(end_position); }
};
struct Region {
    std::string chrom_name;
    bool is_decoy;
    uint start_position, end_position;
};

struct Genome {
    std::vector<std::pair<std::string, std::string>> chromosomes;
    std::unordered_set<std::string> decoys;
    int find(const std::string &name) const {
        for (size_t i = 0; i < chromosomes.size(); i++)
            if (chromosomes[i].first == name) return (int)i;
        return -1;
    }
    bool isDecoy(const std::string &name) const { return decoys.count(name) > 0; }
};

// KmerPair::getLexicographicalLowestKmer on the window ending at every position (Kmer.tpp:182-255), as ASCII; "" where no full window ends
std::vector<std::string> canonicalWindows(const std::string &seq, size_t begin, size_t end, uint k) {
    std::vector<std::string> out(end - begin);
    std::string window;
    for (size_t i = begin; i < end; i++) {
        const char c = (char)std::toupper((unsigned char)seq[i]);
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') {
            window.clear();
            continue;
        }
        window.push_back(c);
        if (window.size() > k) window.erase(window.begin());
        if (window.size() == k) {
            std::string rc(window.rbegin(), window.rend());
            for (auto &x : rc) x = x == 'A' ? 'T' : x == 'C' ? 'G' : x == 'G' ? 'C' : 'A';
            out[i - begin] = std::min(window, rc);
        }
    }
    return out;
}

struct Parser {
    uint kmer_size, max_allele_length;
    float copy_number_variant_threshold;
    std::vector<uint> allele_type_counter = std::vector<uint>(ALLELE_COUNT_SIZE, 0), variant_type_counter = std::vector<uint>(VARIANT_TYPE_SIZE, 0);
    uint num_variants = 0, num_variant_clusters = 0, num_variant_cluster_groups = 0, total_num_variants = 0;
    std::vector<Region> intercluster_regions;
    unsigned long intercluster_regions_length = 0;
    std::unordered_set<std::string> intercluster_chromosomes;
    std::string prev_chrom_name = "";
    int prev_position = -1, prev_var_end_position = -1;
    std::vector<std::vector<std::string>> lines;   // data lines: CHROM POS ID REF ALT INFO
    size_t next_line = 0;
    std::vector<std::string> variant_line;
    std::string error;

    bool updateVariantLine() {   // :148-171
        if (next_line >= lines.size()) return false;
        variant_line = lines[next_line++];
        return true;
    }
    void addSequenceToInterclusterRegions(const std::string &chrom_name, const bool is_decoy, const uint start_position, const uint end_position) {   // :173-183
        intercluster_regions_length += end_position - start_position + 1;
        if ((end_position - start_position + 1) >= kmer_size) intercluster_regions.push_back(Region{chrom_name, is_decoy, start_position, end_position});
    }

    uint copyNumberVariantLength(const std::string &allele_sequence, const std::string &chrom_sequence, const uint chrom_start_position) {   // :649-733
        uint copy_number_variant_length = 0;
        if (allele_sequence.size() < kmer_size) return copy_number_variant_length;
        std::unordered_set<std::string> allele_kmers;
        for (auto &w : canonicalWindows(allele_sequence, 0, allele_sequence.size(), kmer_size))
            if (!w.empty()) allele_kmers.emplace(w);
        if (allele_kmers.empty()) return copy_number_variant_length;
        uint chrom_window_end_position = std::min(chrom_start_position + copy_number_variant_length + static_cast<uint>(allele_sequence.size()), static_cast<uint>(chrom_sequence.size()));
        while (true) {
            uint num_bases = 0;
            uint num_identical_kmers = 0;
            std::pair<double, uint> highest_scoring_window(0, 0);
            const uint from = chrom_start_position + copy_number_variant_length;
            const auto windows = canonicalWindows(chrom_sequence, from, std::max(from, chrom_window_end_position), kmer_size);   // the k-mer pair is reset at `from`
            for (uint chrom_position = from; chrom_position < chrom_window_end_position; chrom_position++) {
                const std::string &w = windows[chrom_position - from];
                if (!w.empty() && allele_kmers.count(w) > 0) num_identical_kmers++;
                num_bases++;
                if (num_identical_kmers > 0) {
                    double identical_kmer_fraction = num_identical_kmers / static_cast<double>(num_bases - kmer_size + 1);
                    if (doubleCompare(identical_kmer_fraction, highest_scoring_window.first) or (identical_kmer_fraction > highest_scoring_window.first)) {
                        highest_scoring_window.first = identical_kmer_fraction;
                        highest_scoring_window.second = num_bases;
                    }
                }
            }
            if (highest_scoring_window.first < copy_number_variant_threshold) break;
            copy_number_variant_length += highest_scoring_window.second;
            if (chrom_window_end_position == chrom_sequence.size()) break;
            chrom_window_end_position = std::min(chrom_start_position + copy_number_variant_length + static_cast<uint>(allele_sequence.size()), static_cast<uint>(chrom_sequence.size()));
        }
        return copy_number_variant_length;
    }

    static int classifyAllele(const int reference_size, const int allele_size) {   // :624-647
        if ((reference_size == 1) and (allele_size == 1)) return SNV;
        else if ((reference_size == 0) or (allele_size == 0)) return (allele_size - reference_size) > 0 ? Insertion : Deletion;
        return Complex;
    }
    static void addAlternativeAllele(Variant *cur_variant, const std::string &ref_allele, const std::string &alt_allele, const std::string &origin_att) {   // :582-622
        auto ref_allele_it = ref_allele.begin();
        auto alt_allele_it = alt_allele.begin();
        uint identical_left_nucleotides = 0;
        while ((ref_allele_it != ref_allele.end()) and (alt_allele_it != alt_allele.end())) {
            if (*alt_allele_it == *ref_allele_it) identical_left_nucleotides++;
            else break;
            ref_allele_it++;
            alt_allele_it++;
        }
        cur_variant->num_redundant_nucleotides = std::min(cur_variant->num_redundant_nucleotides, identical_left_nucleotides);
        cur_variant->alt_alleles.push_back(AlleleInfo{(uint)ref_allele.size(), alt_allele, origin_att});
        int variant_type = classifyAllele(ref_allele.size() - identical_left_nucleotides, alt_allele.size() - identical_left_nucleotides);
        if (cur_variant->type == Unsupported) cur_variant->type = variant_type;
        else if (cur_variant->type != variant_type) cur_variant->type = Mixture;
    }

    // :735-978
    void clusterVariants(Variant &cur_variant, const uint cur_position, const std::set<uint> cur_end_positions, const std::string &cur_chrom_name,
                         std::map<uint, VariantCluster *> *variant_cluster_group_flanks, std::unordered_map<uint, VariantCluster *> *variant_cluster_group,
                         std::list<std::unordered_set<uint>> *variant_cluster_group_merge_sets) {
        const int k = kmer_size;
        if (!variant_cluster_group_flanks->empty()) {
            auto lit = variant_cluster_group_flanks->begin();
            while (static_cast<int>(cur_position - lit->first) >= k) {
                lit = variant_cluster_group_flanks->erase(lit);
                if (lit == variant_cluster_group_flanks->end()) break;
            }
        }
        auto by_idx = [](VariantCluster *a, VariantCluster *b) { return a->cluster_idx < b->cluster_idx; };
        std::set<VariantCluster *, decltype(by_idx)> second_overlaps(by_idx);
        VariantCluster *variant_cluster_group_first = nullptr;
        auto vit = variant_cluster_group_flanks->begin();
        while (vit != variant_cluster_group_flanks->end()) {
            if ((std::abs(static_cast<int>(cur_position - vit->first)) + 1) <= k) {
                if (variant_cluster_group_first == nullptr) {
                    variant_cluster_group_first = vit->second;
                    if (cur_position >= vit->first) {
                        vit = variant_cluster_group_flanks->erase(vit);
                        continue;
                    }
                } else if (variant_cluster_group_first != vit->second) {
                    second_overlaps.insert(vit->second);
                }
            }
            for (auto &cit : cur_end_positions) {
                if ((std::abs(static_cast<int>(cit - vit->first)) + 1) <= k) {
                    if (variant_cluster_group_first == nullptr) {
                        variant_cluster_group_first = vit->second;
                        // (the reference erases the entry here when cur_position >= vit->first; that cannot hold: an entry at or
                        // behind cur_position that survived the pruning above is within k of it and was taken by the test before)
                        assert(cur_position < vit->first);
                    } else if (variant_cluster_group_first != vit->second) {
                        second_overlaps.insert(vit->second);
                    }
                } else if ((cur_position < vit->first) and (cit > vit->first)) {
                    if (variant_cluster_group_first == nullptr) variant_cluster_group_first = vit->second;
                    else if (variant_cluster_group_first != vit->second) second_overlaps.insert(vit->second);
                }
            }
            vit++;
        }
        if (variant_cluster_group_first == nullptr) {
            VariantCluster *variant_cluster = new VariantCluster();
            variant_cluster->cluster_idx = variant_cluster_group->size();
            variant_cluster->left_flank = cur_position;
            variant_cluster->right_flank = *cur_end_positions.rbegin();
            variant_cluster->chrom_name = cur_chrom_name;
            variant_cluster->variants.insert(std::pair<uint, Variant>(cur_position, cur_variant));
            for (auto &cit : cur_end_positions) variant_cluster_group_flanks->insert(std::pair<uint, VariantCluster *>(cit, variant_cluster));
            if (*cur_end_positions.rbegin() - cur_position >= kmer_size) variant_cluster_group_flanks->insert(std::pair<uint, VariantCluster *>(cur_position, variant_cluster));
            variant_cluster_group->insert({(uint)variant_cluster_group->size(), variant_cluster});
        } else {
            if (!variant_cluster_group_first->variants.insert(std::pair<uint, Variant>(cur_position, cur_variant)).second) error = "duplicate position";
            variant_cluster_group_first->right_flank = std::max(*cur_end_positions.rbegin(), variant_cluster_group_first->right_flank);
            for (auto &cit : cur_end_positions) variant_cluster_group_flanks->insert(std::pair<uint, VariantCluster *>(cit, variant_cluster_group_first));
            if (*cur_end_positions.rbegin() - cur_position >= kmer_size) variant_cluster_group_flanks->insert(std::pair<uint, VariantCluster *>(cur_position, variant_cluster_group_first));
        }
        if (!second_overlaps.empty()) {
            auto found_set = variant_cluster_group_merge_sets->end();
            auto sit = variant_cluster_group_merge_sets->begin();
            while (sit != variant_cluster_group_merge_sets->end()) {
                if (sit->count(variant_cluster_group_first->cluster_idx) > 0) {
                    if (found_set == variant_cluster_group_merge_sets->end()) found_set = sit;
                    else {
                        if (sit != found_set) {
                            found_set->insert(sit->begin(), sit->end());
                            sit = variant_cluster_group_merge_sets->erase(sit);
                            continue;
                        }
                    }
                }
                bool merged_cluster_merge_sets = false;
                for (auto &lit : second_overlaps) {
                    if (sit->count(lit->cluster_idx) > 0) {
                        if (found_set == variant_cluster_group_merge_sets->end()) found_set = sit;
                        else {
                            if (sit != found_set) {
                                found_set->insert(sit->begin(), sit->end());
                                sit = variant_cluster_group_merge_sets->erase(sit);
                                merged_cluster_merge_sets = true;
                                break;
                            }
                        }
                    }
                }
                if (!merged_cluster_merge_sets) sit++;
            }
            if (found_set != variant_cluster_group_merge_sets->end()) {
                found_set->insert(variant_cluster_group_first->cluster_idx);
                for (auto &lit : second_overlaps) found_set->insert(lit->cluster_idx);
            } else {
                variant_cluster_group_merge_sets->push_back(std::unordered_set<uint>());
                variant_cluster_group_merge_sets->back().insert(variant_cluster_group_first->cluster_idx);
                for (auto &lit : second_overlaps) variant_cluster_group_merge_sets->back().insert(lit->cluster_idx);
            }
        }
    }

    static void mergeVariantClusters(std::unordered_map<uint, VariantCluster *> *variant_cluster_group, std::list<std::unordered_set<uint>> &variant_cluster_group_merge_sets) {   // :1003-1042
        for (auto &cit : variant_cluster_group_merge_sets) {
            auto cur_cluster = cit.begin();
            uint first_cluster = *cit.begin();
            cur_cluster++;
            while (cur_cluster != cit.end()) {
                variant_cluster_group->at(first_cluster)->left_flank = std::min(variant_cluster_group->at(first_cluster)->left_flank, variant_cluster_group->at(*cur_cluster)->left_flank);
                variant_cluster_group->at(first_cluster)->right_flank = std::max(variant_cluster_group->at(first_cluster)->right_flank, variant_cluster_group->at(*cur_cluster)->right_flank);
                for (auto &variant : variant_cluster_group->at(*cur_cluster)->variants) variant_cluster_group->at(first_cluster)->variants.insert(variant);
                delete variant_cluster_group->at(*cur_cluster);
                variant_cluster_group->erase(*cur_cluster);
                cur_cluster++;
            }
        }
    }

    static std::unordered_map<uint, uint> getVariantClusterGroupDependencies(std::unordered_map<uint, VariantCluster *> *variant_cluster_group) {   // :1108-1160
        std::unordered_map<uint, uint> variant_cluster_depedencies;
        auto vit_first = variant_cluster_group->begin();
        while (vit_first != variant_cluster_group->end()) {
            auto vit_second = variant_cluster_group->begin();
            auto nested_variant_cluster = variant_cluster_group->end();
            while (vit_second != variant_cluster_group->end()) {
                if (vit_first != vit_second) {
                    if ((vit_first->second->left_flank > vit_second->second->left_flank) and (vit_first->second->right_flank < vit_second->second->right_flank)) {
                        if (nested_variant_cluster == variant_cluster_group->end()) nested_variant_cluster = vit_second;
                        else if ((vit_second->second->left_flank > nested_variant_cluster->second->left_flank) and (vit_second->second->right_flank < nested_variant_cluster->second->right_flank))
                            nested_variant_cluster = vit_second;
                    }
                }
                vit_second++;
            }
            if (nested_variant_cluster != variant_cluster_group->end()) variant_cluster_depedencies.emplace(vit_first->first, nested_variant_cluster->second->cluster_idx);
            vit_first++;
        }
        for (auto &vit : variant_cluster_depedencies)
            variant_cluster_group->at(vit.second)->contained_clusters.insert(
                ContainedCluster{variant_cluster_group->at(vit.first)->cluster_idx, variant_cluster_group->at(vit.first)->left_flank, variant_cluster_group->at(vit.first)->right_flank});
        return variant_cluster_depedencies;
    }

    // processVariantClusterGroups (:980-1001) with the consumer side (:1044-1106) and the VariantClusterGroup constructor inlined
    void processVariantClusterGroups(std::vector<Group> *unit_groups, std::unordered_map<uint, VariantCluster *> **variant_cluster_group,
                                     std::list<std::unordered_set<uint>> *variant_cluster_group_merge_sets, std::map<uint, VariantCluster *> *variant_cluster_group_flanks) {
        if (!(*variant_cluster_group)->empty()) {
            mergeVariantClusters(*variant_cluster_group, *variant_cluster_group_merge_sets);
            auto *cur = *variant_cluster_group;
            auto deps = getVariantClusterGroupDependencies(cur);
            Group g;
            g.chrom_name = cur->begin()->second->chrom_name;
            g.start_position = 0xFFFFFFFFu;
            g.end_position = 0;
            g.num_variants = 0;
            std::unordered_map<uint, uint> variant_cluster_idx_to_vertex_id;
            for (auto &variant_cluster : *cur) {
                if (deps.count(variant_cluster.second->cluster_idx) < 1) g.source_vertices.push_back(g.vertices.size());
                variant_cluster_idx_to_vertex_id.emplace(variant_cluster.second->cluster_idx, g.vertices.size());
                g.vertices.push_back(*variant_cluster.second);
                g.start_position = std::min(g.start_position, variant_cluster.second->left_flank + 1);
                g.end_position = std::max(g.end_position, variant_cluster.second->right_flank + 1);
                g.num_variants += variant_cluster.second->variants.size();
            }
            g.out_edges = std::vector<std::vector<uint>>(g.vertices.size());
            for (auto &dep : deps) g.out_edges.at(variant_cluster_idx_to_vertex_id.at(dep.second)).push_back(variant_cluster_idx_to_vertex_id.at(dep.first));
            num_variant_clusters += g.vertices.size();
            num_variant_cluster_groups++;
            unit_groups->push_back(g);
            for (auto &vit : *cur) delete vit.second;
            delete cur;
            *variant_cluster_group = new std::unordered_map<uint, VariantCluster *>();
        }
        variant_cluster_group_merge_sets->clear();
        variant_cluster_group_flanks->clear();
    }

    // parseVariants (:237-545); returns true when the whole file has been parsed (:185-235)
    bool unit(std::vector<Group> *unit_groups, const uint min_unit_variants, const Genome &chromosomes) {
        const int k = kmer_size;
        bool is_first_unit_variant = true;
        uint unit_variant_counter = 0;
        std::string cur_chrom_name = "";
        int chromosomes_it = -1;
        if (prev_chrom_name != "") chromosomes_it = chromosomes.find(prev_chrom_name);
        int cur_position = 0;
        int cur_group_end_position = prev_var_end_position;
        auto *variant_cluster_group = new std::unordered_map<uint, VariantCluster *>();
        std::map<uint, VariantCluster *> variant_cluster_group_flanks;
        std::list<std::unordered_set<uint>> variant_cluster_group_merge_sets;
        std::set<uint> variant_depedencies;
        // `while (is_first_unit_variant or updateVariantLine())`: the first line of a unit is the one the previous unit stopped at
        // (variant_line; empty when the previous read hit the end of the file)
        for (;;) {
            if (is_first_unit_variant) {
                if (variant_line.empty()) break;
            } else if (!updateVariantLine()) {
                variant_line.clear();
                break;
            }
            is_first_unit_variant = false;
            cur_chrom_name = variant_line.at(0);
            cur_position = std::stoi(variant_line.at(1)) - 1;
            if (cur_chrom_name != prev_chrom_name) {
                if (prev_chrom_name != "") {
                    int prev_chromosomes_it = chromosomes.find(prev_chrom_name);
                    if (prev_chromosomes_it >= 0) {
                        processVariantClusterGroups(unit_groups, &variant_cluster_group, &variant_cluster_group_merge_sets, &variant_cluster_group_flanks);
                        addSequenceToInterclusterRegions(prev_chrom_name, chromosomes.isDecoy(prev_chrom_name), prev_var_end_position + 1, chromosomes.chromosomes[prev_chromosomes_it].second.size() - 1);
                        if (!intercluster_chromosomes.insert(prev_chrom_name).second) {
                            error = "Variants need to be sorted by contig; variants on contig \"" + prev_chrom_name + "\" is unordered";
                            return false;
                        }
                    }
                }
                chromosomes_it = chromosomes.find(cur_chrom_name);
                prev_var_end_position = -1;
                cur_group_end_position = -1;
                variant_depedencies.clear();
            } else if (prev_position > cur_position) {
                error = "Variants need to be sorted by position";
                return false;
            } else if (prev_position == cur_position) {
                error = "Variants on the same position need to be multi-allelic";
                return false;
            }
            prev_chrom_name = cur_chrom_name;
            auto vit = variant_depedencies.begin();
            while (vit != variant_depedencies.end()) {
                if (static_cast<int>(*vit) >= cur_position) break;
                variant_depedencies.erase(vit);
                vit = variant_depedencies.begin();
            }
            if ((unit_variant_counter >= min_unit_variants) and ((cur_position - cur_group_end_position) >= k)) break;
            prev_position = cur_position;
            std::string var_ref_seq = variant_line.at(3);
            std::transform(var_ref_seq.begin(), var_ref_seq.end(), var_ref_seq.begin(), ::toupper);
            std::vector<std::string> alt_alleles = splitOn(variant_line.at(4), ',');
            std::vector<std::string> origin_allele_att;
            auto origin_att_str = getInfoAttributeString(variant_line.at(5), "ACO");
            if (origin_att_str.second) origin_allele_att = splitOn(origin_att_str.first, ',');
            else origin_allele_att = std::vector<std::string>(alt_alleles.size(), "");
            if (origin_allele_att.size() != alt_alleles.size()) {
                error = "ACO";
                return false;
            }
            Variant cur_variant;
            cur_variant.id = variant_line.at(2);
            cur_variant.has_dependency = !variant_depedencies.empty();
            num_variants += 1;
            unit_variant_counter += 1;
            if (alt_alleles.back() == "*") alt_alleles.pop_back();
            allele_type_counter.at(Total) += alt_alleles.size();
            if (chromosomes_it >= 0 and chromosomes.isDecoy(cur_chrom_name)) {
                allele_type_counter.at(Excluded_decoy) += alt_alleles.size();
                variant_type_counter.at(Unsupported)++;
                continue;
            }
            if (chromosomes_it < 0) {
                allele_type_counter.at(Excluded_genome) += alt_alleles.size();
                variant_type_counter.at(Unsupported)++;
                continue;
            }
            const std::string &chrom_sequence = chromosomes.chromosomes[chromosomes_it].second;
            std::vector<std::string> ref_alleles(alt_alleles.size(), var_ref_seq);
            for (ushort i = 0; i < alt_alleles.size(); i++) {
                std::transform(alt_alleles.at(i).begin(), alt_alleles.at(i).end(), alt_alleles.at(i).begin(), ::toupper);
                while ((ref_alleles.at(i).size() > 1) and (alt_alleles.at(i).size() > 1)) {   // rightTrimAllele :563-580
                    if (ref_alleles.at(i).back() == alt_alleles.at(i).back()) {
                        ref_alleles.at(i).pop_back();
                        alt_alleles.at(i).pop_back();
                    } else break;
                }
            }
            bool is_excluded = false;
            std::string gen_ref_seq = (size_t)cur_position <= chrom_sequence.size() ? chrom_sequence.substr(cur_position, variant_line.at(3).size()) : "";
            std::transform(gen_ref_seq.begin(), gen_ref_seq.end(), gen_ref_seq.begin(), ::toupper);
            if (var_ref_seq.compare(gen_ref_seq) != 0) {
                allele_type_counter.at(Excluded_match) += alt_alleles.size();
                is_excluded = true;
            }
            if (cur_position < (k - 1)) {
                allele_type_counter.at(Excluded_end) += alt_alleles.size();
                is_excluded = true;
            }
            std::unordered_set<ushort> excluded_alleles;
            if (!is_excluded) {
                for (ushort i = 0; i < alt_alleles.size(); i++) {
                    if ((cur_position + ref_alleles.at(i).size() - 1 + kmer_size) > chrom_sequence.size()) {
                        allele_type_counter.at(Excluded_end)++;
                        excluded_alleles.insert(i);
                    } else if ((ref_alleles.at(i).size() > max_allele_length) or (alt_alleles.at(i).size() > max_allele_length)) {
                        allele_type_counter.at(Excluded_length)++;
                        excluded_alleles.insert(i);
                    } else {
                        variant_depedencies.insert(cur_position + ref_alleles.at(i).size() - 1);
                    }
                }
            }
            if (is_excluded or excluded_alleles.size() == alt_alleles.size()) {
                variant_type_counter.at(Unsupported)++;
                continue;
            }
            if ((cur_position - cur_group_end_position) >= k)
                processVariantClusterGroups(unit_groups, &variant_cluster_group, &variant_cluster_group_merge_sets, &variant_cluster_group_flanks);
            if (cur_position > (prev_var_end_position + 1)) addSequenceToInterclusterRegions(cur_chrom_name, chromosomes.isDecoy(cur_chrom_name), prev_var_end_position + 1, cur_position - 1);
            std::set<uint> cur_end_positions;
            for (ushort alt_allele_idx = 0; alt_allele_idx < alt_alleles.size(); alt_allele_idx++) {
                if (excluded_alleles.count(alt_allele_idx) == 0) {
                    addAlternativeAllele(&cur_variant, ref_alleles.at(alt_allele_idx), alt_alleles.at(alt_allele_idx), origin_allele_att.at(alt_allele_idx));
                    uint ref_copy_number_variant_length = copyNumberVariantLength(ref_alleles.at(alt_allele_idx), chrom_sequence, cur_position + ref_alleles.at(alt_allele_idx).size());
                    uint alt_copy_number_variant_length = copyNumberVariantLength(alt_alleles.at(alt_allele_idx), chrom_sequence, cur_position + ref_alleles.at(alt_allele_idx).size());
                    cur_end_positions.insert(cur_position + ref_alleles.at(alt_allele_idx).size() - 1);
                    cur_group_end_position = std::max(cur_group_end_position, static_cast<int>(cur_position + ref_alleles.at(alt_allele_idx).size() - 1 + std::max(ref_copy_number_variant_length, alt_copy_number_variant_length)));
                }
            }
            prev_var_end_position = std::max(prev_var_end_position, static_cast<int>(*cur_end_positions.rbegin()));
            clusterVariants(cur_variant, cur_position, cur_end_positions, cur_chrom_name, &variant_cluster_group_flanks, variant_cluster_group, &variant_cluster_group_merge_sets);
            if (!error.empty()) return false;
            variant_type_counter.at(cur_variant.type)++;
        }
        processVariantClusterGroups(unit_groups, &variant_cluster_group, &variant_cluster_group_merge_sets, &variant_cluster_group_flanks);
        delete variant_cluster_group;
        if (total_num_variants == num_variants) {
            if (chromosomes_it >= 0) {
                addSequenceToInterclusterRegions(cur_chrom_name, chromosomes.isDecoy(cur_chrom_name), prev_var_end_position + 1, chromosomes.chromosomes[chromosomes_it].second.size() - 1);
                intercluster_chromosomes.insert(cur_chrom_name);
            }
            for (auto &c : chromosomes.chromosomes)
                if (intercluster_chromosomes.insert(c.first).second) addSequenceToInterclusterRegions(c.first, chromosomes.isDecoy(c.first), 0, c.second.size() - 1);
        }
        return (num_variants == total_num_variants);
    }

    static std::vector<std::string> splitOn(const std::string &s, char sep) {
        std::vector<std::string> out(1);
        for (char c : s) {
            if (c == sep) out.emplace_back();
            else out.back().push_back(c);
        }
        return out;
    }
    static std::pair<std::string, bool> getInfoAttributeString(const std::string &info_str, const std::string &att_name) {   // :547-561
        std::stringstream info_ss(info_str);
        std::string att_str = "";
        while (std::getline(info_ss, att_str, ';')) {
            if ((att_str.substr(0, att_name.size()) == att_name) and (att_str.size() > att_name.size()) and (att_str.substr(att_name.size(), 1) == "=")) return std::make_pair(att_str.substr(att_name.size() + 1), true);
        }
        return std::make_pair("", false);
    }
};

bool groupCompare(const Group &a, const Group &b) {   // VariantClusterGroup.cpp:277-292
    if (a.num_variants != b.num_variants) return a.num_variants > b.num_variants;
    return a.region() > b.region();
}

void dumpGroups(std::ostringstream &os, const std::vector<Group> &groups) {
    for (size_t g = 0; g < groups.size(); g++) {
        const Group &G = groups[g];
        os << "GROUP " << g << " region=" << G.region() << " nvar=" << G.num_variants << " sources=";
        for (size_t i = 0; i < G.source_vertices.size(); i++) os << (i ? "," : "") << G.source_vertices[i];
        os << "\n";
        for (size_t v = 0; v < G.vertices.size(); v++) {
            const VariantCluster &c = G.vertices[v];
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
                os << "  VAR pos=" << pv.first << " id=" << pv.second.id << " dep=" << (pv.second.has_dependency ? 1 : 0) << " type=" << pv.second.type << " red=" << pv.second.num_redundant_nucleotides
                   << " alts=";
                for (size_t a = 0; a < pv.second.alt_alleles.size(); a++)
                    os << (a ? "|" : "") << pv.second.alt_alleles[a].ref_length << ":" << pv.second.alt_alleles[a].sequence << ":" << pv.second.alt_alleles[a].aco_att;
                os << "\n";
            }
        }
    }
}

}  // namespace

extern "C" {

// Runs the whole cluster front end.  Output text: for every unit "UNIT <i>\n" + its group dump (groups sorted as main.cpp:247), then
// "REGIONS\n" + the intercluster regions in file order, "SORTED\n" + the same after sortInterclusterRegions, "COUNTERS\n" + one line.
// On a reference error path: "ERROR <message>\n".  Returns the text length (nothing written if cap is too small).
unsigned long long orc_cluster_stage(const char *vcf, unsigned long long vcf_len, unsigned num_chrom, const char *const *chrom_names, const char *const *chrom_seqs,
                                     const unsigned long long *chrom_lens, const unsigned char *chrom_decoy, unsigned k, unsigned max_allele_length, float cnv_threshold,
                                     unsigned min_unit_variants, char *out, unsigned long long cap) {
    Genome genome;
    for (unsigned c = 0; c < num_chrom; c++) {
        genome.chromosomes.emplace_back(chrom_names[c], std::string(chrom_seqs[c], chrom_seqs[c] + chrom_lens[c]));
        if (chrom_decoy[c]) genome.decoys.insert(chrom_names[c]);
    }
    Parser p;
    p.kmer_size = k;
    p.max_allele_length = max_allele_length;
    p.copy_number_variant_threshold = cnv_threshold;
    // the reader (:86-119,148-171): data lines are those not starting with '#'; fields CHROM POS ID REF ALT . . INFO
    {
        std::stringstream ss(std::string(vcf, vcf + vcf_len));
        for (std::string line; std::getline(ss, line);) {
            if (line.empty() || line[0] == '#') continue;
            std::vector<std::string> f = Parser::splitOn(line, '\t');
            f.resize(std::max<size_t>(f.size(), 8));
            p.lines.push_back({f[0], f[1], f[2], f[3], f[4], f[7]});
        }
        p.total_num_variants = p.lines.size();
    }
    std::ostringstream os;
    p.updateVariantLine();
    uint unit_idx = 1;
    bool done = p.lines.empty();
    while (!done) {
        std::vector<Group> groups;
        const uint parsed_before = p.num_variants;
        done = p.unit(&groups, min_unit_variants, genome);
        if (!p.error.empty()) {
            os << "ERROR " << p.error << "\n";
            break;
        }
        std::sort(groups.begin(), groups.end(), groupCompare);
        os << "UNIT " << unit_idx++ << "\n";
        dumpGroups(os, groups);
        if (p.num_variants == parsed_before && !done) break;   // no progress
    }
    os << "REGIONS\n";
    for (auto &r : p.intercluster_regions) os << r.chrom_name << "\t" << r.is_decoy << "\t" << r.start_position << "\t" << r.end_position << "\n";
    std::sort(p.intercluster_regions.begin(), p.intercluster_regions.end(),
              [](const Region &a, const Region &b) { return ((a.end_position - a.start_position) > (b.end_position - b.start_position)); });   // :59-65
    os << "SORTED\n";
    for (auto &r : p.intercluster_regions) os << r.chrom_name << "\t" << r.is_decoy << "\t" << r.start_position << "\t" << r.end_position << "\n";
    os << "COUNTERS\ntotal=" << p.total_num_variants << " parsed=" << p.num_variants << " clusters=" << p.num_variant_clusters << " groups=" << p.num_variant_cluster_groups
       << " region_length=" << p.intercluster_regions_length << " alleles=";
    for (size_t i = 0; i < p.allele_type_counter.size(); i++) os << (i ? "," : "") << p.allele_type_counter[i];
    os << " types=";
    for (size_t i = 0; i < p.variant_type_counter.size(); i++) os << (i ? "," : "") << p.variant_type_counter[i];
    os << "\n";
    const std::string s = os.str();
    if (out && cap >= s.size()) memcpy(out, s.data(), s.size());
    return s.size();
}

}  // extern "C"
