#include "Genotypes.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace bthost {

namespace {
constexpr uint16_t NONE = 0xFFFF;
// Utils::floatCompare / floatLess (include/bayesTyper/Utils.hpp:89-103)
inline bool floatCompare(float a, float b) { return a == b || std::abs(a - b) < std::abs(std::min(a, b)) * std::numeric_limits<float>::epsilon() * 100; }
inline bool floatLess(float a, float b) { return a < b && !floatCompare(a, b); }
}  // namespace

float Filters::minFractionObservedKmers(double genomic_mean) { return (float)(1 - std::exp(-(0.275f * genomic_mean))); }

std::vector<VariantGenotypes> getGenotypes(const ClusterResults &r, const Filters &filters) {
    if ((uint32_t)filters.min_fraction_observed_kmers.size() != r.S) throw std::invalid_argument("Filters: one min_fraction_observed_kmers per sample");
    std::vector<uint32_t> allele_base(r.V + 1, 0);
    for (uint32_t v = 0; v < r.V; ++v) allele_base[v + 1] = allele_base[v] + r.var_num_alleles[v];
    const uint32_t A_total = allele_base[r.V];
    std::vector<VariantGenotypes> out(r.V);
    for (uint32_t v = 0; v < r.V; ++v) {
        const uint32_t A = r.var_num_alleles[v];
        VariantGenotypes &g = out[v];
        // getNonCoveredAlleles (:221-247)
        std::vector<uint8_t> covered(A, 0);
        for (uint32_t h = 0; h < r.H; ++h) covered[r.hap_allele[(size_t)h * r.V + v]] = 1;
        if (r.var_has_dependency[v]) covered[A - 1] = 1;
        for (uint32_t a = 0; a < A; ++a)
            if (!covered[a]) g.non_covered_alleles.push_back((uint16_t)a);
        auto allele_of = [&](uint16_t h) -> uint16_t { return h != NONE ? r.hap_allele[(size_t)h * r.V + v] : (uint16_t)(A - 1); };   // haplotypeToAlleleIndex
        // getGenotypeSampleStats (:249-466)
        for (uint32_t s = 0; s < r.S; ++s) {
            g.sample_stats.emplace_back();
            SampleStats &st = g.sample_stats.back();
            const uint8_t ploidy = r.ploidy[s];
            const uint32_t num_genotypes = ploidy == 2 ? A * (A - 1) / 2 + A : (ploidy == 1 ? A : 0);
            st.genotype_posteriors.assign(num_genotypes, 0.f);
            st.allele_posteriors.assign(ploidy ? A : 0, 0.f);
            st.allele_filters.assign(ploidy ? A : 0, 0);
            uint32_t num_iterations = 0;
            std::vector<std::pair<uint16_t, uint16_t>> best;
            float best_value = 0;
            for (uint64_t e = 0; e < r.num_diplotypes; ++e) {
                const uint32_t f = r.freq[e * r.S + s];
                if (!f) continue;
                std::pair<uint16_t, uint16_t> gt(NONE, NONE);
                uint32_t gi = NONE;
                if (ploidy == 2) {
                    gt.first = allele_of(r.h1[e]);
                    gt.second = allele_of(r.h2[e]);
                    if (gt.first > gt.second) std::swap(gt.first, gt.second);
                    gi = (uint32_t)gt.second * (gt.second + 1) / 2 + gt.first;
                    st.genotype_posteriors[gi] += f;
                    st.allele_posteriors[gt.first] += f;
                    if (gt.first != gt.second) st.allele_posteriors[gt.second] += f;
                } else if (ploidy == 1) {
                    gt.first = allele_of(r.h1[e]);
                    gi = gt.first;
                    st.genotype_posteriors[gi] += f;
                    st.allele_posteriors[gt.first] += f;
                }
                num_iterations += f;
                if (ploidy) {   // running maximum over the accumulating genotype sums (:334-346); the final set is order-independent
                    if (floatCompare(best_value, st.genotype_posteriors[gi])) best.push_back(gt);
                    else if (best_value < st.genotype_posteriors[gi]) {
                        best.assign(1, gt);
                        best_value = st.genotype_posteriors[gi];
                    }
                }
            }
            best_value /= num_iterations;
            for (float &p : st.genotype_posteriors) p /= num_iterations;
            for (float &p : st.allele_posteriors) p /= num_iterations;
            const double *cell0 = r.stats + ((size_t)s * A_total + allele_base[v]) * 12;
            for (uint32_t a = 0; a < (uint32_t)st.allele_posteriors.size(); ++a) {
                if (floatCompare(st.allele_posteriors[a], 0)) continue;
                const double *cell = cell0 + (size_t)a * 12;
                const float count_mean = (float)cell[2];            // count_stats.getMean()
                if (floatLess(count_mean, filters.min_number_of_kmers)) st.allele_filters[a] += 1;
                if (!floatCompare(count_mean, 0)) {
                    const float fraction_mean = (float)cell[4 + 2];   // fraction_stats.getMean()
                    if (floatLess(fraction_mean, filters.min_fraction_observed_kmers[s])) st.allele_filters[a] += 2;
                }
            }
            if (floatCompare(best_value, 1)) st.genotype_quality = 99;
            else if (floatCompare(best_value, 0)) st.genotype_quality = 0;
            else st.genotype_quality = (uint32_t)(-10 * std::log10(1 - best_value));
            if (ploidy == 2) {
                st.genotype_estimate.assign(2, NONE);
                if (best.size() == 1 && !floatLess(best_value, filters.min_genotype_posterior) && st.allele_filters[best[0].first] == 0 &&
                    st.allele_filters[best[0].second] == 0) {
                    st.genotype_estimate[0] = best[0].first;
                    st.genotype_estimate[1] = best[0].second;
                }
            } else if (ploidy == 1) {
                st.genotype_estimate.assign(1, NONE);
                if (best.size() == 1 && !floatLess(best_value, filters.min_genotype_posterior) && st.allele_filters[best[0].first] == 0)
                    st.genotype_estimate[0] = best[0].first;
            }
        }
        // getGenotypeVariantStats (:468-527)
        VariantStats &vs = g.variant_stats;
        vs.alt_allele_counts.assign(A - 1, 0);
        vs.alt_allele_frequency.assign(A - 1, 0.f);
        vs.allele_call_probabilities.assign(A, 0.f);
        for (uint32_t s = 0; s < r.S; ++s) {
            const SampleStats &st = g.sample_stats[s];
            for (uint16_t a : st.genotype_estimate)
                if (a != NONE) {
                    vs.total_count++;
                    if (a > 0) vs.alt_allele_counts[a - 1]++;
                }
            for (uint32_t a = 0; a < (uint32_t)st.allele_posteriors.size(); ++a)
                if (st.allele_filters[a] == 0) vs.allele_call_probabilities[a] = std::max(vs.allele_call_probabilities[a], st.allele_posteriors[a]);
        }
        const uint32_t num_alt = A - 1 - (r.var_has_dependency[v] ? 1 : 0);   // alt_alleles.size(): the missing allele is not an alt allele
        for (uint32_t a = 0; a < num_alt; ++a) vs.max_alt_allele_call_probability = std::max(vs.max_alt_allele_call_probability, vs.allele_call_probabilities[a + 1]);
        if (vs.total_count > 0)
            for (uint32_t a = 0; a + 1 < A; ++a) vs.alt_allele_frequency[a] = vs.alt_allele_counts[a] / (float)vs.total_count;
    }
    return out;
}

namespace {
template <typename T>
void writeAlleleField(std::ostream &o, const std::vector<T> &v) {   // GenotypeWriter.cpp:130-143
    for (size_t i = 0; i < v.size(); ++i) {
        if (i) o << ",";
        o << v[i];
    }
}
}  // namespace

std::string formatQualityFilterAndStats(const VariantGenotypes &g) {
    std::ostringstream o;
    const VariantStats &vs = g.variant_stats;
    if (floatCompare(vs.max_alt_allele_call_probability, 1)) o << "99";
    else if (floatCompare(vs.max_alt_allele_call_probability, 0)) o << "0";
    else o << -10 * std::log10(1 - vs.max_alt_allele_call_probability);
    o << (vs.total_count == 0 ? "\tAN0" : "\tPASS");
    o << "\tAC=";
    writeAlleleField(o, vs.alt_allele_counts);
    o << ";AF=";
    writeAlleleField(o, vs.alt_allele_frequency);
    o << ";AN=" << vs.total_count << ";ACP=";
    writeAlleleField(o, vs.allele_call_probabilities);
    return o.str();
}

std::string formatAlleleCover(const VariantGenotypes &g) {
    if (g.non_covered_alleles.empty()) return "";
    std::ostringstream o;
    std::vector<uint16_t> nc = g.non_covered_alleles;
    std::sort(nc.begin(), nc.end());
    o << ";ANC=";
    writeAlleleField(o, nc);
    return o.str();
}

std::string formatVariantStatsColumns(const VariantGenotypes &g) { return formatQualityFilterAndStats(g) + formatAlleleCover(g); }

std::string formatSampleColumns(const ClusterResults &r, uint32_t variant, const VariantGenotypes &g) {
    std::ostringstream o;
    uint32_t allele_base = 0, A_total = 0;
    for (uint32_t v = 0; v < r.V; ++v) {
        if (v < variant) allele_base += r.var_num_alleles[v];
        A_total += r.var_num_alleles[v];
    }
    const uint32_t A = r.var_num_alleles[variant];
    for (uint32_t s = 0; s < r.S; ++s) {
        const SampleStats &st = g.sample_stats[s];
        o << "\t";
        if (st.genotype_estimate.empty()) {
            o << ":" << ".:.:.:.:.:.";   // empty_variant_sample (GenotypeWriter.cpp:58,319): the GT column itself stays empty
            continue;
        }
        for (size_t i = 0; i < st.genotype_estimate.size(); ++i) {
            if (i) o << "/";
            if (st.genotype_estimate[i] != NONE) o << st.genotype_estimate[i];
            else o << ".";
        }
        o << ":" << st.genotype_quality << ":";
        writeAlleleField(o, st.genotype_posteriors);
        o << ":";
        writeAlleleField(o, st.allele_posteriors);
        o << ":";
        {   // writeAlleleKmerStats: the means of the count / fraction / mean statistics per allele, -1 when nothing was added
            std::ostringstream counts, fractions, means;
            for (uint32_t a = 0; a < A; ++a) {
                const double *cell = r.stats + ((size_t)s * A_total + allele_base + a) * 12;
                auto mean_of = [](const double *ks) { return ks[0] == 0 ? -1.0 : ks[2]; };   // KmerStats::getMean (KmerStats.cpp:82-92)
                if (a) {
                    counts << ",";
                    fractions << ",";
                    means << ",";
                }
                counts << mean_of(cell);
                fractions << mean_of(cell + 4);
                means << mean_of(cell + 8);
            }
            o << counts.str() << ":" << fractions.str() << ":" << means.str();
        }
        o << ":";
        writeAlleleField(o, st.allele_filters);
    }
    return o.str();
}

}  // namespace bthost
