// Command lines of `bayesTyper cluster` and `bayesTyper genotype`: the option tables of the reference's main
// (src/bayesTyper/main.cpp:114-138 cluster, :366-404 genotype), their defaults, the help screens (returns 1 like main.cpp:146-150,
// 410-414), and OptionsContainer's text record of the parsed values that both stages write into the output VCF's header
// (include/bayesTyper/OptionsContainer.tpp:138-155).
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace bthost {

struct OptionSpec {
    std::string long_name;   // "variant-file"
    char short_name;         // 'v' or 0
    std::string group;       // "Required", "General", ...
    bool required, is_flag;  // flags (bool_switch-like: --gzip-output [true|false]) take an optional value
    std::string default_text, help;
    char kind = 's';         // 's' string, 'u' unsigned, 'f' float, 'b' bool, 'p' pair: how the value is re-printed for the header (operator<< of the typed value)
};

class OptionsContainer {
  public:
    OptionsContainer(const std::string &type, const std::string &version, const std::string &start_time, unsigned kmer_size);
    // parses argv[2..]; returns 0 ok, 1 help was printed (caller returns 1), throws std::runtime_error with the message on bad input
    int parse(int argc, char *const argv[], const std::vector<OptionSpec> &specs, const std::string &title);
    const std::string &text(const std::string &option) const;
    std::string getString(const std::string &option) const { return text(option); }
    unsigned long getUInt(const std::string &option) const;
    float getFloat(const std::string &option) const;
    bool getBool(const std::string &option) const;
    std::pair<float, float> getFloatPair(const std::string &option) const;
    std::string getHeader() const;   // "##BayesTyperOptions=command:"..", version:"..", time:"..", kmer-size:"..", <option>:"<value>", ...\n" (options in name order)

  private:
    std::map<std::string, std::string> options;   // name -> value text (std::map: the header lists them in name order, as the reference's map does)
    std::string type, version, start_time;
    unsigned kmer_size;
};

std::vector<OptionSpec> clusterOptionSpecs();
std::vector<OptionSpec> genotypeOptionSpecs();
std::string getLocalTime();   // Utils::getLocalTime (include/bayesTyper/Utils.hpp): "dd/mm/yyyy hh:mm:ss"

}  // namespace bthost
