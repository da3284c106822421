// `bayesTyperTools makeBloom` — the one bayesTyperTools command on the path this build covers (src/bayesTyperTools/main.cpp:66-146,
// MakeBloom.cpp:39-51,200-295): the sample's KMC k-mer table -> <kmc-table-prefix>.bloomMeta / .bloomData, the sample Bloom filter
// `bayesTyper cluster` reads (Sample.cpp / KmerCounter.cpp:59-103).  Same options (-k/--kmc-table-prefix, -p/--num-threads,
// --false-positive-rate 0.001), same progress lines, byte-identical output files (insertion is an order-independent OR).  The records are
// streamed from the memory-mapped .kmc_suf through the GPU in chunks (bt_kmc_scan_make_bloom); there is no CPU path.
// The VCF utilities of bayesTyperTools (convertAllele, combine, filter, annotate, addAttributes) are not part of this build.
#include <cstring>
#include <iostream>
#include <stdexcept>

#include "../../include/btgpu.h"
#include "KmcFile.hpp"
#include "Options.hpp"

using namespace bthost;

namespace {
const char *const BT_VERSION = "v1.5 (MI355X build)";
std::string stamp() { return "[" + getLocalTime() + "] "; }
void check(int rc, const char *what) {
    if (rc != BT_OK) throw std::runtime_error(std::string(what) + ": " + bt_last_error());
}

int runMakeBloom(int argc, char *const argv[], unsigned kmer_size) {
    const std::vector<OptionSpec> specs = {
        {"kmc-table-prefix", 'k', "Required", true, false, "", "KMC kmer table prefix. Output is written as <kmc-table-prefix>.bloomMeta and <kmc-table-prefix>.bloomData.", 's'},
        {"num-threads", 'p', "General", false, false, "1", "number of threads used (+= 1 I/O thread).", 'u'},
        {"false-positive-rate", 0, "Parameters", false, false, "0.001", "bloom filter false positive rate.", 'f'},
    };
    OptionsContainer options("makeBloom", BT_VERSION, getLocalTime(), kmer_size);
    if (options.parse(argc, argv, specs, "## BayesTyperTools makeBloom ##")) return 1;
    const std::string prefix = options.getString("kmc-table-prefix");
    const float fpr = options.getFloat("false-positive-rate");
    if (!(fpr > 0) || !(fpr < 1)) throw std::runtime_error("--false-positive-rate must be in (0, 1)");
    std::cout << stamp() << "Running BayesTyperTools (" << BT_VERSION << ") makeBloom ...\n" << std::endl;
    KmcFile db(prefix);   // throws "Unable to open KMC table ..." like MakeBloom.cpp:206-210
    if (db.kmer_length != kmer_size) throw std::runtime_error("KMC table " + prefix + " holds " + std::to_string(db.kmer_length) + "-mers, not " + std::to_string(kmer_size) + "-mers");
    std::cout << stamp() << "Making bloom filter of " << db.total_kmers << " kmers with a false positive rate of " << fpr << " ...\n" << std::endl;
    bt_ctx *ctx = nullptr;
    const char *dev = getenv("BT_DEVICE");
    check(bt_ctx_create(dev ? atoi(dev) : 0, &ctx), "bt_ctx_create");
    bt_bloom *bloom = nullptr;
    bt_kmc_scan *scan = nullptr;
    void *d_chunk = nullptr;
    int rc = 0;
    try {
        check(bt_bloom_create(ctx, db.total_kmers, fpr, kmer_size, 0, &bloom), "bt_bloom_create");   // KmerBloom(total_kmers, fpr), MakeBloom.cpp:221
        check(bt_kmc_scan_create_bins(ctx, db.kmer_length, db.lut_prefix_length, db.counter_size, db.total_kmers, db.prefix_lut().data(), db.prefix_lut().size(), &scan), "bt_kmc_scan_create");
        bt_kmc_scan_set_count_range(scan, db.min_count, db.max_count);   // ReadNextKmer skips records outside the database's counter range (kmc_file.cpp:496-511)
        const uint64_t rec = (db.kmer_length - db.lut_prefix_length) / 4 + db.counter_size;
        const uint64_t chunk = 1ull << 24;   // records per transfer
        check(bt_malloc(ctx, chunk * rec + 32, &d_chunk), "bt_malloc");
        uint64_t next_report = 100000000;
        for (uint64_t first = 0; first < db.total_kmers; first += chunk) {
            const uint64_t n = std::min<uint64_t>(chunk, db.total_kmers - first);
            check(bt_memcpy_h2d(ctx, d_chunk, db.records() + first * rec, n * rec), "bt_memcpy_h2d");
            check(bt_kmc_scan_make_bloom(scan, bloom, (const uint8_t *)d_chunk, first, n), "bt_kmc_scan_make_bloom");
            check(bt_sync(ctx), "bt_sync");
            while (first + n >= next_report) {   // MakeBloom.cpp:262-265
                std::cout << stamp() << "Parsed " << next_report << " kmers" << std::endl;
                next_report += 100000000;
            }
        }
        std::cout << "\n" << stamp() << "Saving bloom filter to " << prefix << " ..." << std::endl;
        check(bt_bloom_save(bloom, prefix.c_str()), "bt_bloom_save");
        std::cout << stamp() << "Completed saving bloom filter\n" << std::endl;
    } catch (...) {
        if (d_chunk) bt_free(ctx, d_chunk);
        if (scan) bt_kmc_scan_destroy(scan);
        if (bloom) bt_bloom_destroy(bloom);
        bt_ctx_destroy(ctx);
        throw;
    }
    bt_free(ctx, d_chunk);
    bt_kmc_scan_destroy(scan);
    bt_bloom_destroy(bloom);
    bt_ctx_destroy(ctx);
    return rc;
}
}  // namespace

int main(int argc, char *const argv[]) {
    const unsigned kmer_size = getenv("BT_KMER_SIZE") ? (unsigned)atoi(getenv("BT_KMER_SIZE")) : 55u;
    std::cout << "\n[" << getLocalTime() << "] You are using BayesTyperTools (" << BT_VERSION << ")\n" << std::endl;
    const std::string command_info = "Usage: bayesTyperTools <command> [options]\n\nCommands:\n\n\tmakeBloom\t\tcreate kmer bloom filter\n";
    if (argc == 1) {
        std::cout << command_info << std::endl;
        return 0;
    }
    try {
        if (kmer_size < 1 || kmer_size > 64) throw std::runtime_error("BT_KMER_SIZE must be between 1 and 64");
        if (std::strcmp(argv[1], "makeBloom") == 0) return runMakeBloom(argc, argv, kmer_size);
        std::cout << command_info << std::endl;   // (the reference prints the command list for an unknown command, main.cpp:... and returns 0)
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "\nERROR: " << e.what() << "\n" << std::endl;
        return 1;
    }
}
