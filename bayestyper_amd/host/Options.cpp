#include "Options.hpp"

#include <algorithm>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace bthost {

std::string getLocalTime() {
    time_t now = time(nullptr);
    struct tm lt;
    localtime_r(&now, &lt);
    char buf[64];
    strftime(buf, sizeof(buf), "%d/%m/%Y %H:%M:%S", &lt);
    return buf;
}

OptionsContainer::OptionsContainer(const std::string &type_in, const std::string &version_in, const std::string &start_time_in, unsigned kmer_size_in)
    : type(type_in), version(version_in), start_time(start_time_in), kmer_size(kmer_size_in) {}

std::vector<OptionSpec> clusterOptionSpecs() {
    const std::string now = std::to_string((unsigned)time(nullptr));
    return {
        {"variant-file", 'v', "Required", true, false, "", "variant file (vcf format)."},
        {"samples-file", 's', "Required", true, false, "", "samples file (see github documentation for format specifications)."},
        {"genome-file", 'g', "Required", true, false, "", "reference genome file (fasta format)."},
        {"decoy-file", 'd', "General", false, false, "", "decoy sequences file (fasta format)."},
        {"output-prefix", 'o', "General", false, false, "bayestyper", "output prefix."},
        {"random-seed", 'r', "General", false, false, now, "seed for pseudo-random number generator.", 'u'},
        {"threads", 'p', "General", false, false, "1", "number of threads used (+= 2 I/O threads).", 'u'},
        {"min-number-of-unit-variants", 'u', "General", false, false, "5000000", "minimum number of variants per inference unit.", 'u'},
        {"max-allele-length", 0, "Cluster", false, false, "500000", "exclude alleles (reference and alternative) longer than <length>.", 'u'},
        {"copy-number-variant-threshold", 0, "Cluster", false, false, "0.5",
         "minimum fraction of identical kmers required between an allele and the downstream reference sequence in order for it to be classified as a copy number.", 'f'},
        {"max-number-of-sample-haplotypes", 0, "Cluster", false, false, "32", "maximum number of haplotype candidates per sample.", 'u'},
    };
}

std::vector<OptionSpec> genotypeOptionSpecs() {
    const std::string now = std::to_string((unsigned)time(nullptr));
    return {
        {"variant-clusters-file", 'v', "Required", true, false, "", "variant_clusters.bin file (BayesTyper cluster output)."},
        {"cluster-data-dir", 'c', "Required", true, false, "",
         "cluster data directory containing intercluster_regions.txt.gz, multigroup_kmers.bloom[Meta|Data] & parameter_kmers.fa.gz (BayesTyper cluster output)."},
        {"samples-file", 's', "Required", true, false, "", "samples file (see github documentation for format specifications)."},
        {"genome-file", 'g', "Required", true, false, "", "reference genome file (fasta format)."},
        {"decoy-file", 'd', "General", false, false, "", "decoy sequences file (fasta format)."},
        {"output-prefix", 'o', "General", false, false, "bayestyper", "output prefix."},
        {"gzip-output", 'z', "General", false, true, "0", "compress <output-prefix>.vcf using gzip.", 'b'},
        {"random-seed", 'r', "General", false, false, now, "seed for pseudo-random number generator.", 'u'},
        {"threads", 'p', "General", false, false, "1", "number of threads used (+= 2 I/O threads).", 'u'},
        {"chromosome-ploidy-file", 'y', "General", false, false, "",
         "chromosome gender ploidy file (see github documentation for format specifications). Human ploidy levels will be assumed if no file is given."},
        {"gibbs-burn-in", 0, "Genotyping", false, false, "100", "number of burn-in iterations.", 'u'},
        {"gibbs-samples", 0, "Genotyping", false, false, "250", "number of Gibbs iterations.", 'u'},
        {"number-of-gibbs-chains", 0, "Genotyping", false, false, "20", "number of independent Gibbs sampling chains.", 'u'},
        {"kmer-subsampling-rate", 0, "Genotyping", false, false, "0.1", "subsampling rate for subsetting kmers used for genotype inference.", 'f'},
        {"max-haplotype-variant-kmers", 0, "Genotyping", false, false, "500",
         "maximum number of kmers used for genotype inference after subsampling across a haplotype candidate for each variant.", 'u'},
        {"noise-genotyping", 0, "Genotyping", false, true, "0", "estimate noise model parameters and genotypes jointly (generally slower and uses more memory).", 'b'},
        {"noise-rate-prior", 0, "Genotyping", false, false, "1,0.01",
         "parameters for Poisson noise rate gamma prior (<shape>,<scale>). All samples will use the same parameters.", 'p'},
        {"min-genotype-posterior", 0, "Filter", false, false, "0.99", "filter genotypes with a posterior probability (GPP) below <value>.", 'f'},
        {"min-number-of-kmers", 0, "Filter", false, false, "1", "filter sampled alleles with less than <value> kmers (NAK).", 'f'},
        {"disable-observed-kmers", 0, "Filter", false, true, "0", "disable filtering of sampled alleles with a low fraction of observed kmers (FAK).", 'b'},
    };
}

static void printHelp(const std::vector<OptionSpec> &specs, const std::string &title) {
    std::cout << title << ":\n\n  -h [ --help ]                         produce help message for options\n";
    std::string group;
    for (auto &s : specs) {
        if (s.group != group) {
            group = s.group;
            std::cout << "\n== " << group << " ==:\n";
        }
        std::string left = "  ";
        if (s.short_name) left += std::string("-") + s.short_name + " [ --" + s.long_name + " ]";
        else left += "--" + s.long_name;
        if (s.is_flag) left += " [=arg(=1)]";
        else left += " arg";
        if (!s.required && !s.default_text.empty() && s.long_name != "random-seed") left += " (=" + s.default_text + ")";
        if (s.long_name == "random-seed") left += " (=unix time)";
        if (left.size() < 40) left.resize(40, ' ');
        else left += " ";
        std::cout << left << s.help << "\n";
    }
    std::cout << std::endl;
}

int OptionsContainer::parse(int argc, char *const argv[], const std::vector<OptionSpec> &specs, const std::string &title) {
    if (argc == 2) {
        printHelp(specs, title);
        return 1;
    }
    auto find = [&](const std::string &name, bool is_short) -> const OptionSpec * {
        for (auto &s : specs)
            if (is_short ? (name.size() == 1 && s.short_name == name[0]) : s.long_name == name) return &s;
        return nullptr;
    };
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            printHelp(specs, title);
            return 1;
        }
        const OptionSpec *spec = nullptr;
        std::string value;
        bool has_value = false;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            spec = find(a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2), false);
            if (eq != std::string::npos) {
                value = a.substr(eq + 1);
                has_value = true;
            }
        } else if (a.size() >= 2 && a[0] == '-') {
            spec = find(a.substr(1, 1), true);
            if (a.size() > 2) {
                value = a.substr(2);
                has_value = true;
            }
        }
        if (!spec) throw std::runtime_error("unrecognised option '" + a + "'");
        if (options.count(spec->long_name)) throw std::runtime_error("option '--" + spec->long_name + "' cannot be specified more than once");
        if (spec->is_flag) {
            if (!has_value) value = "1";   // implicit_value(true); an explicit value must be attached (--flag=false)
            else if (value == "true" || value == "1" || value == "on" || value == "yes") value = "1";
            else if (value == "false" || value == "0" || value == "off" || value == "no") value = "0";
            else throw std::runtime_error("the argument ('" + value + "') for option '--" + spec->long_name + "' is invalid");
        } else if (!has_value) {
            if (i + 1 >= argc) throw std::runtime_error("the required argument for option '--" + spec->long_name + "' is missing");
            value = argv[++i];
        }
        options[spec->long_name] = value;
    }
    // typed values are recorded as operator<< prints them (OptionsContainer::parseValue, OptionsContainer.tpp:52-62)
    for (auto &s : specs) {
        auto it = options.find(s.long_name);
        if (it == options.end()) continue;
        std::ostringstream os;
        if (s.kind == 'u') {
            os << getUInt(s.long_name);
            it->second = os.str();
        } else if (s.kind == 'f') {
            os << getFloat(s.long_name);
            it->second = os.str();
        }
    }
    for (auto &s : specs) {
        if (options.count(s.long_name)) continue;
        if (s.required) throw std::runtime_error("the option '--" + s.long_name + "' is required but missing");
        options[s.long_name] = s.default_text;
    }
    return 0;
}

const std::string &OptionsContainer::text(const std::string &option) const {
    auto it = options.find(option);
    if (it == options.end()) throw std::runtime_error("unknown option " + option);
    return it->second;
}
unsigned long OptionsContainer::getUInt(const std::string &option) const {
    const std::string &t = text(option);
    size_t used = 0;
    unsigned long v = 0;
    try {
        v = std::stoul(t, &used);
    } catch (...) {
        used = 0;
    }
    if (used != t.size() || t.empty() || t[0] == '-') throw std::runtime_error("the argument ('" + t + "') for option '--" + option + "' is invalid");
    return v;
}
float OptionsContainer::getFloat(const std::string &option) const {
    const std::string &t = text(option);
    size_t used = 0;
    float v = 0;
    try {
        v = std::stof(t, &used);
    } catch (...) {
        used = 0;
    }
    if (used != t.size() || t.empty()) throw std::runtime_error("the argument ('" + t + "') for option '--" + option + "' is invalid");
    return v;
}
bool OptionsContainer::getBool(const std::string &option) const { return text(option) == "1"; }
std::pair<float, float> OptionsContainer::getFloatPair(const std::string &option) const {   // OptionsContainer::parseValuePair (OptionsContainer.tpp:64-95)
    const std::string &t = text(option);
    const size_t comma = t.find(',');
    if (comma == std::string::npos || t.find(',', comma + 1) != std::string::npos)
        throw std::runtime_error("Argument to option \"" + option + "\" should be two values (comma-seperated)");
    return {std::stof(t.substr(0, comma)), std::stof(t.substr(comma + 1))};
}

std::string OptionsContainer::getHeader() const {
    std::ostringstream h;
    h << "##BayesTyperOptions=command:\"" << type << "\", version:\"" << version << "\", time:\"" << start_time << "\", kmer-size:\"" << kmer_size << "\"";
    for (auto &o : options) h << ", " << o.first << ":\"" << o.second << "\"";
    h << "\n";
    return h.str();
}

}  // namespace bthost
