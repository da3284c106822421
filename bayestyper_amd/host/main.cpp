// `bayesTyper` — the two command lines of the reference (src/bayesTyper/main.cpp:80-655) over the MI355X path:
//
//   bayesTyper cluster   -v <candidates.vcf[.gz]> -s <samples.tsv> -g <genome.fa> [...]   (main.cpp:110-358)
//   bayesTyper genotype  -v <unit>/variant_clusters.bin -c <cluster_data dir> -s <samples.tsv> -g <genome.fa> [...]   (main.cpp:360-652)
//
// Same options, defaults, input files (VCF, FASTA, samples file, KMC databases, sample Bloom filters), stage sequence, progress lines
// and outputs (<o>_unit_<i>/variant_clusters.bin, <o>_cluster_data/{intercluster_regions.txt.gz, parameter_kmers.fa.gz,
// multigroup_kmers.bloom[Meta|Data]}, <o>.vcf[.gz], <o>_genomic_parameters.txt, <o>_noise_parameters.txt).  The k-mer passes and the
// Gibbs sampler run on the GPU through libbtgpu.so (KmerCounter.hpp, InferenceEngine.hpp); there is no CPU fallback.
// variant_clusters.bin is this build's own format (InferenceUnit.hpp), not the reference's Boost archive.
// The k-mer size is a build constant in the reference (BT_KMER_SIZE, 55 in the released binaries); here BT_KMER_SIZE in the environment overrides 55.
//
// Several GPUs of one node (`genotype` only; the reference's -p/--threads has no meaning here): one process per GPU (Comm.hpp).
//   BT_GPUS=N bayesTyper genotype ...      this process becomes rank 0 and starts ranks 1..N-1 itself (GPU r for rank r)
//   BT_WORLD=N BT_RANK=r BT_COMM_ID_FILE=<path> bayesTyper genotype ...   ranks started by a launcher of one's own
// Every rank reads the inputs; the KMC scan of every sample is split by record range and the matched counts are merged (all-gather); the
// unit's variant-cluster groups are dealt to the ranks (every group keeps its unit-wide index, from which its seeds derive); the noise
// drivers add up their histograms with one all-reduce per iteration; rank 0 gathers the collected samples and writes the outputs —
// byte for byte the files of a one-GPU run.
#include <sys/stat.h>
#include <signal.h>
#include <fcntl.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <map>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>

#include "Comm.hpp"
#include "CountDistribution.hpp"
#include "GenotypeWriter.hpp"
#include "Genotypes.hpp"
#include "InferenceEngine.hpp"
#include "InferenceUnit.hpp"
#include "KmerCounter.hpp"
#include "KmerHashOrder.hpp"
#include "Options.hpp"
#include "Parallel.hpp"
#include "Sample.hpp"
#include "StageTimes.hpp"

using namespace bthost;

namespace {

const char *const BT_VERSION = "v1.5 (MI355X build)";
const unsigned max_parameter_kmers = 1000000;   // main.cpp:73
const char *const intercluster_regions_file_prefix = "intercluster_regions";
const char *const multigroup_kmers_file_prefix = "multigroup_kmers";
const char *const parameter_kmers_file_prefix = "parameter_kmers";

std::string stamp() { return "[" + getLocalTime() + "] "; }
const std::chrono::steady_clock::time_point g_main_start = std::chrono::steady_clock::now();   // (static initialisation: just before main())

void check(int rc, const char *what) {
    if (rc != BT_OK) throw std::runtime_error(std::string(what) + ": " + bt_last_error());
}

struct Context {
    bt_ctx *h = nullptr;
    Context() {   // BT_DEVICE, else the rank (one GPU per rank of a multi-GPU run), else GPU 0
        const char *dev = getenv("BT_DEVICE");
        // several ranks told to share one GPU (tests): the resident launches of their noise chains would wait for each other's wavefront slots
        if (dev && getenv("BT_WORLD") && atoi(getenv("BT_WORLD")) > 1) setenv("BT_NOISE_CHAIN_OFF", "1", 0);
        check(bt_ctx_create(dev ? atoi(dev) : Comm::envRank(), &h), "bt_ctx_create");
    }
    ~Context() { bt_ctx_destroy(h); }
};

void makeDirectory(const std::string &dir, const char *what) {
    if (mkdir(dir.c_str(), 0777) != 0) throw std::runtime_error(std::string(what) + " directory " + dir + "/ already exist");
}

Chromosomes readGenome(const OptionsContainer &options) {
    StageScope stage("read reference genome");
    std::cout << "\n" << stamp() << "Parsing reference genome ..." << std::endl;
    Chromosomes chromosomes;
    chromosomes.addFasta(options.getString("genome-file"), false);
    std::cout << stamp() << "Parsed " << chromosomes.size() << " reference genome chromosomes(s) (" << chromosomes.getTotalLength() << " nucleotides)" << std::endl;
    std::cout << "\n" << stamp() << "Parsing decoy sequence(s) ..." << std::endl;
    const size_t before = chromosomes.size();
    if (!options.getString("decoy-file").empty()) chromosomes.addFasta(options.getString("decoy-file"), true);
    std::cout << stamp() << "Parsed " << chromosomes.size() - before << " decoy sequence(s) (" << chromosomes.getDecoyLength() << " nucleotides)" << std::endl;
    chromosomes.convertToUpper();
    return chromosomes;
}

std::string kmerToString(uint64_t lo, uint64_t hi, unsigned k) {   // Nucleotide::bitToNt (Nucleotide.hpp:101-129)
    std::string s(k, 'A');
    for (unsigned i = 0; i < k; i++) s[i] = "ACGT"[((i < 32 ? lo >> (2 * i) : hi >> (2 * (i - 32))) & 3u)];
    return s;
}
// every stored k-mer of a table with its flag byte
void exportTable(bt_table *t, std::vector<uint64_t> *kmers, std::vector<uint8_t> *flags) {
    uint64_t n = 0;
    check(bt_table_status(t, &n, nullptr, nullptr), "bt_table_status");
    kmers->assign(std::max<uint64_t>(2 * n, 2), 0);
    std::vector<uint8_t> meta(std::max<uint64_t>(4 * n, 4));
    uint64_t written = 0;
    check(bt_table_export(t, kmers->data(), nullptr, meta.data(), std::max<uint64_t>(n, 1), &written), "bt_table_export");
    kmers->resize(2 * written);
    flags->resize(written);
    for (uint64_t i = 0; i < written; i++) (*flags)[i] = meta[4 * i];
}

// ------------------------------------------------------------------------------------------------------------------------------------
int runCluster(int argc, char *const argv[], unsigned kmer_size) {
    OptionsContainer options("cluster", BT_VERSION, getLocalTime(), kmer_size);
    if (options.parse(argc, argv, clusterOptionSpecs(), "## BayesTyper cluster options ##")) return 1;
    setenv("BT_HOST_THREADS", std::to_string(clampThreads(options.getUInt("threads"))).c_str(), 0);   // the library's host-side assembly steps (bt_paths_candidates) take -p too
    const uint32_t min_unit_variants = (uint32_t)options.getUInt("min-number-of-unit-variants");
    const float cnv_threshold = options.getFloat("copy-number-variant-threshold");
    const uint16_t max_sample_haplotypes = (uint16_t)options.getUInt("max-number-of-sample-haplotypes");
    if (min_unit_variants == 0) throw std::runtime_error("--min-number-of-unit-variants must be positive");
    if (cnv_threshold < 0 || cnv_threshold > 1) throw std::runtime_error("--copy-number-variant-threshold must be between zero and one");
    if (max_sample_haplotypes == 0 || (uint32_t)max_sample_haplotypes * 30 >= 65535) throw std::runtime_error("--max-number-of-sample-haplotypes is out of range");
    const unsigned seed = (unsigned)options.getUInt("random-seed");
    std::cout << stamp() << "Seeding pseudo-random number generator with " << seed << " ..." << std::endl;
    std::cout << stamp() << "Setting the kmer size to " << kmer_size << " ..." << std::endl;
    const std::vector<Sample> samples = readSamples(options.getString("samples-file"));
    const std::string output_prefix = options.getString("output-prefix");
    std::cout << "\n" << stamp() << "Parsed information for " << samples.size() << " sample(s)" << std::endl;
    const Chromosomes chromosomes = readGenome(options);

    std::unique_ptr<StageScope> st_ctx(new StageScope("start-up: HIP runtime + context"));
    Context ctx;
    st_ctx.reset();
    KmerCounter kmer_counter(ctx.h, samples, kmer_size, seed);
    std::unique_ptr<StageScope> st_read(new StageScope("read variant file"));
    VariantFileParser variant_file_parser(VariantFileParser::readVariantFile(options.getString("variant-file")), kmer_size, (uint32_t)options.getUInt("max-allele-length"), cnv_threshold);
    st_read.reset();
    const uint32_t num_variants = variant_file_parser.getNumberOfVariants();
    const uint32_t num_units = std::max<uint32_t>(1, (uint32_t)std::floor(num_variants / (float)min_unit_variants));
    std::cout << "\n" << stamp() << "Setting the number of inference units to " << num_units << " across " << num_variants << " variants ..." << std::endl;

    const uint64_t expected_num_path_kmers = (uint64_t)std::ceil((chromosomes.getTotalLength() - chromosomes.getDecoyLength()) * (1 + (0.05 * 2 * samples.size())));
    uint64_t num_path_kmers = 0;
    BloomHandle path_kmer_bloom;   // ThreadedKmerBloom(expected_num_path_kmers, 0.0001)
    check(bt_bloom_create(ctx.h, std::max<uint64_t>(expected_num_path_kmers, 1), 0.0001f, kmer_size, 1, &path_kmer_bloom.h), "bt_bloom_create");
    TableHandle multigroup_kmer_hash;   // KmerHash<bool>(expected * 0.01)
    check(bt_table_create(ctx.h, (uint64_t)std::ceil(expected_num_path_kmers * 0.01) + 1024, 1, kmer_size, &multigroup_kmer_hash.h), "bt_table_create");

    bool variant_file_parsed = false;
    for (uint32_t unit_idx = 1; unit_idx < num_units + 1; unit_idx++) {
        std::cout << "\n" << std::endl;
        if (variant_file_parsed) throw std::runtime_error("the variant file ended before the last inference unit");
        InferenceUnit unit;
        unit.index = unit_idx;
        unit.cluster_options_header = options.getHeader();
        const uint32_t parsed_before = variant_file_parser.numParsedVariants(), clusters_before = variant_file_parser.numVariantClusters();
        {
            StageScope stage("parse variants -> clusters -> groups");
            variant_file_parsed = variant_file_parser.constructVariantClusterGroups(&unit.variant_cluster_groups, (uint32_t)std::ceil(num_variants / (float)num_units), chromosomes);
        }
        unit.num_variants = variant_file_parser.numParsedVariants() - parsed_before;
        unit.num_variant_clusters = variant_file_parser.numVariantClusters() - clusters_before;
        std::sort(unit.variant_cluster_groups.begin(), unit.variant_cluster_groups.end(), ClusterGroupCompare);
        std::cout << stamp() << "Parsed unit " << unit_idx << ": " << unit.num_variants << " variants in " << unit.num_variant_clusters << " clusters and "
                  << unit.variant_cluster_groups.size() << " groups\n" << std::endl;
        {
            std::unique_ptr<StageScope> st(new StageScope("construct variant cluster graphs (host)"));
            const UnitGraphs graphs(unit, chromosomes, kmer_size, clampThreads(options.getUInt("threads")));
            st.reset(new StageScope("find sample paths (sample Bloom filters + search)"));
            kmer_counter.findVariantClusterPaths(&unit, graphs, max_sample_haplotypes);
            st.reset(new StageScope("count path / multigroup k-mers"));
            kmer_counter.countPathMultigroupKmers(multigroup_kmer_hash.h, path_kmer_bloom.h, &unit, graphs);
        }
        num_path_kmers += unit.num_path_kmers;
        const std::string unit_dir = output_prefix + "_unit_" + std::to_string(unit.index);
        makeDirectory(unit_dir, "Unit");
        {
            StageScope stage("write variant_clusters.bin");
            unit.write(unit_dir + "/variant_clusters.bin");
        }
        std::cout << "\n" << stamp() << "Wrote unit " << unit.index << " variant clusters to " << unit_dir << "/variant_clusters.bin" << std::endl;
    }
    if (!variant_file_parsed) throw std::runtime_error("variants remain after the last inference unit");
    if (expected_num_path_kmers < num_path_kmers)
        std::cout << "\nWARNING: Multigroup kmer estimate might be inflated due to the number of kmers being higher than expected.\n" << std::endl;

    const std::string cluster_data_dir = output_prefix + "_cluster_data";
    makeDirectory(cluster_data_dir, "Cluster data");

    std::unique_ptr<StageScope> st_tail(new StageScope("inter-cluster regions, parameter k-mers, multigroup filter"));
    std::cout << "\n\n" << stamp() << "Writing inter-cluster regions ..." << std::endl;
    const std::string intercluster_regions_dir_prefix = cluster_data_dir + "/" + intercluster_regions_file_prefix;
    variant_file_parser.sortInterclusterRegions();
    writeGzFile(intercluster_regions_dir_prefix + ".txt.gz", variant_file_parser.interclusterRegionsText());
    std::cout << stamp() << "Wrote " << variant_file_parser.getInterclusterRegions().size() << " regions to " << intercluster_regions_dir_prefix << ".txt.gz\n" << std::endl;

    // parameter k-mers: non-path k-mers of the inter-cluster regions, a Bernoulli(fraction) sample of them (main.cpp:319-341)
    const uint32_t max_intercluster_kmers = 3 * max_parameter_kmers;
    const uint64_t num_region_kmers = variant_file_parser.getNumberOfInterclusterRegionKmers();
    const float parameter_kmer_fraction = std::min(1.0f, (float)max_intercluster_kmers / (float)num_region_kmers);
    uint64_t num_parameter_kmers = 0;
    {
        TableHandle parameter_kmer_hash;
        check(bt_table_create(ctx.h, (uint64_t)max_intercluster_kmers + chromosomes.getDecoyLength(), 1, kmer_size, &parameter_kmer_hash.h), "bt_table_create");
        std::unique_ptr<StageScope> sub(new StageScope("  parameter k-mers: count in inter-cluster regions (GPU)"));
        kmer_counter.countInterclusterParameterKmers(parameter_kmer_hash.h, variant_file_parser.getInterclusterRegions(), chromosomes, path_kmer_bloom.h, parameter_kmer_fraction);
        std::vector<uint64_t> kmers;
        std::vector<uint8_t> flags;
        sub.reset(new StageScope("  parameter k-mers: export + HybridHash shuffled order (host)"));
        exportTable(parameter_kmer_hash.h, &kmers, &flags);
        // KmerHash::shuffle(seed) then writeKmersToFasta: the first <= 10^6 k-mers with value true (accepted in a non-decoy region and in no decoy)
        const std::vector<uint32_t> order = hybridHashShuffledOrder(kmers.data(), flags.size(), kmer_size, seed);
        sub.reset(new StageScope("  parameter k-mers: write fasta.gz"));
        std::string fasta = ">k" + std::to_string(kmer_size) + "\n";
        for (uint32_t i : order) {
            if (!(flags[i] & BT_KC_PARAMETER) || (flags[i] & BT_KC_DECOY_OCC)) continue;
            fasta += kmerToString(kmers[2 * i], kmers[2 * i + 1], kmer_size);
            fasta += "\n";
            if (++num_parameter_kmers == max_parameter_kmers) break;
        }
        writeGzFile(cluster_data_dir + "/" + parameter_kmers_file_prefix + ".fa.gz", fasta, clampThreads(options.getUInt("threads")));
    }
    std::cout << stamp() << "Wrote " << num_parameter_kmers << " kmers to " << cluster_data_dir << "/" << parameter_kmers_file_prefix << ".fa.gz" << std::endl;

    std::cout << "\n" << stamp() << "Creating multigroup kmers bloom filter ..." << std::endl;
    uint64_t num_multigroup_kmers = 0;
    {
        std::vector<uint64_t> kmers;
        std::vector<uint8_t> flags;
        exportTable(multigroup_kmer_hash.h, &kmers, &flags);
        num_multigroup_kmers = flags.size();
        BloomHandle multigroup_kmer_bloom;   // KmerBloom(num_multigroup_kmers, 0.0001)
        check(bt_bloom_create(ctx.h, std::max<uint64_t>(num_multigroup_kmers, 1), 0.0001f, kmer_size, 0, &multigroup_kmer_bloom.h), "bt_bloom_create");
        if (num_multigroup_kmers) {
            void *d = nullptr;
            check(bt_malloc(ctx.h, kmers.size() * 8, &d), "bt_malloc");
            int rc = bt_memcpy_h2d(ctx.h, d, kmers.data(), kmers.size() * 8);
            if (rc == BT_OK) rc = bt_bloom_insert_batch(multigroup_kmer_bloom.h, (const uint64_t *)d, num_multigroup_kmers);
            if (rc == BT_OK) rc = bt_sync(ctx.h);
            bt_free(ctx.h, d);
            check(rc, "multigroup k-mer Bloom filter");
        }
        check(bt_bloom_save(multigroup_kmer_bloom.h, (cluster_data_dir + "/" + multigroup_kmers_file_prefix).c_str()), "bt_bloom_save");
    }
    std::cout << stamp() << "Wrote " << num_multigroup_kmers << " kmers to " << cluster_data_dir << "/" << multigroup_kmers_file_prefix << ".bloom[Meta|Data]" << std::endl;
    st_tail.reset();
    StageTimes::get().add("wall since main()", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_main_start).count());
    StageTimes::get().print("bayesTyper cluster");
    std::cout << "\n\n" << stamp() << "BayesTyper cluster completed succesfully!\n" << std::endl;
    return 0;
}

// CountDistribution::setGenomicCountDistributions (CountDistribution.cpp:66-141) from the table's parameter k-mer statistics
void setGenomicCountDistributions(CountDistribution *cd, bt_table *table, const std::vector<Sample> &samples, const std::string &output_prefix) {
    const unsigned max_nb_kmer_multiplicity = 32, min_nb_kmer_count = 10000;   // CountDistribution.cpp:42-43
    const size_t S = samples.size();
    std::vector<uint8_t> gender(S);
    for (size_t s = 0; s < S; s++) gender[s] = samples[s].gender;
    std::vector<uint64_t> class_counts(7), n(S * 256), nonzero(S * 256), sum(S * 256), sumsq(S * 256);
    check(bt_table_kmer_stats(table, gender.data(), class_counts.data(), n.data(), nonzero.data(), sum.data(), sumsq.data()), "bt_table_kmer_stats");
    // ObservedKmerCountsHash::calculateKmerStats' report (KmerHash.cpp:326-336)
    std::cout << stamp() << "Out of " << class_counts[0] << " kmers:\n" << std::endl;
    std::cout << "\t- " << class_counts[1] << " have a match to a single variant cluster" << std::endl;
    std::cout << "\t- " << class_counts[2] << " have a match to single variant cluster group and multiple variant clusters" << std::endl;
    std::cout << "\n\t- " << class_counts[3] << " have match to at least one variant cluster and has match to a decoy sequence (not used for inference)" << std::endl;
    std::cout << "\t- " << class_counts[4] << " have match to at least one variant cluster and has a maximum haploid multiplicity higher than 127 (not used for inference)" << std::endl;
    std::cout << "\t- " << class_counts[5] << " have matches to multiple variant cluster groups within or across inference units (not used for inference)" << std::endl;
    std::cout << "\n\t- " << class_counts[6] << " have no match to a variant cluster (includes parameter kmers)" << std::endl;

    std::cout << "\n\n" << stamp() << "Estimating genomic haploid kmer count distribution(s) from parameter kmers ...\n" << std::endl;
    std::ofstream out(output_prefix + ".txt");
    if (!out.is_open()) throw std::runtime_error("Unable to write file " + output_prefix + ".txt");
    out << "Sample\tMean\tVariance" << std::endl;
    for (size_t s = 0; s < S; s++) {
        uint64_t max_count = 0;
        unsigned max_multiplicity = 0;
        for (unsigned m = 1; m <= max_nb_kmer_multiplicity; m++)
            if (n[s * 256 + m] > max_count) {
                max_count = n[s * 256 + m];
                max_multiplicity = m;
            }
        if (max_count < min_nb_kmer_count) {
            std::cout << "\nWARNING: Low number of kmers used for negative binomial parameters estimation for sample " << samples[s].name << " (" << max_count << " < " << min_nb_kmer_count << ")" << std::endl;
            std::cout << "WARNING: The mean and variance estimates might be biased due to the genome used being too small, too variant dense and/or too repetitive\n" << std::endl;
        }
        if (max_multiplicity == 0 || max_count < 2) throw std::runtime_error("no parameter kmers to estimate the genomic kmer count distribution of sample " + samples[s].name + " from");
        // KmerStats' running mean / M2 (KmerStats.cpp:51-63) from the exact integer moments: mean = sum / n, M2 = sumsq - sum^2 / n
        const long double cnt = (long double)max_count, sm = (long double)sum[s * 256 + max_multiplicity], sq = (long double)sumsq[s * 256 + max_multiplicity];
        const double mean = (double)(sm / cnt), var = (double)((sq - sm * sm / cnt) / (cnt - 1));
        cd->setGenomicFromMoments((unsigned short)s, mean, var, max_multiplicity);
        const NegativeBinomialDistribution &nb = cd->getGenomicCountDistributions()[s];
        std::cout << stamp() << "Estimated negative binomial (mean = " << nb.mean() << ", var = " << nb.var() << ") for sample " << samples[s].name << " using " << max_count
                  << " parameter kmers (multiplicity = " << max_multiplicity << ")" << std::endl;
        out << samples[s].name << "\t" << nb.mean() << "\t" << nb.var() << std::endl;
    }
    std::cout << "\n" << stamp() << "Wrote genomic parameters to " << output_prefix << ".txt" << std::endl;
}

// ------------------------------------------------------------------------------------------------------------------------------------
int runGenotype(int argc, char *const argv[], unsigned kmer_size) {
    OptionsContainer options("genotype", BT_VERSION, getLocalTime(), kmer_size);
    if (options.parse(argc, argv, genotypeOptionSpecs(), "## BayesTyper genotype options ##")) return 1;
    setenv("BT_HOST_THREADS", std::to_string(clampThreads(options.getUInt("threads"))).c_str(), 0);   // the library's host-side assembly steps (bt_paths_candidates) take -p too
    GibbsOptions gibbs;
    gibbs.seed = (unsigned)options.getUInt("random-seed");
    gibbs.burn_in = (uint32_t)options.getUInt("gibbs-burn-in");
    gibbs.samples = (uint32_t)options.getUInt("gibbs-samples");
    gibbs.chains = (uint32_t)options.getUInt("number-of-gibbs-chains");
    gibbs.kmer_subsampling_rate = options.getFloat("kmer-subsampling-rate");
    gibbs.max_haplotype_variant_kmers = (uint32_t)options.getUInt("max-haplotype-variant-kmers");
    if (const char *e = getenv("BT_MAX_GROUPS_PER_LAUNCH")) gibbs.max_groups_per_launch = (uint32_t)strtoul(e, nullptr, 0);   // (tests: several launches on a small unit)
    if (gibbs.burn_in == 0 || gibbs.samples == 0 || gibbs.chains == 0) throw std::runtime_error("--gibbs-burn-in, --gibbs-samples and --number-of-gibbs-chains must be positive");
    if (!(gibbs.kmer_subsampling_rate > 0) || gibbs.kmer_subsampling_rate > 1) throw std::runtime_error("--kmer-subsampling-rate must be in (0, 1]");
    const std::pair<float, float> noise_rate_prior = options.getFloatPair("noise-rate-prior");
    if (!(noise_rate_prior.first > 0) || !(noise_rate_prior.second > 0)) throw std::runtime_error("--noise-rate-prior values must be positive");
    std::cout << stamp() << "Seeding pseudo-random number generator with " << gibbs.seed << " ..." << std::endl;
    std::cout << stamp() << "Setting the kmer size to " << kmer_size << " ..." << std::endl;
    const std::vector<Sample> samples = readSamples(options.getString("samples-file"));
    const size_t S = samples.size();
    const std::string output_prefix = options.getString("output-prefix");
    std::cout << "\n" << stamp() << "Parsed information for " << S << " sample(s)" << std::endl;
    const Chromosomes chromosomes = readGenome(options);

    std::cout << "\n\n" << stamp() << "Parsing variant clusters ..." << std::endl;
    std::unique_ptr<StageScope> st(new StageScope("read variant_clusters.bin"));
    InferenceUnit unit = InferenceUnit::read(options.getString("variant-clusters-file"));
    st.reset();
    std::cout << stamp() << "Parsed " << unit.num_variant_clusters << " variant clusters (" << unit.num_variants << " variants)" << std::endl;
    const std::string cluster_data_dir = options.getString("cluster-data-dir");
    const std::string intercluster_regions_dir_prefix = cluster_data_dir + "/" + intercluster_regions_file_prefix;
    const std::string parameter_kmers_dir_prefix = cluster_data_dir + "/" + parameter_kmers_file_prefix;
    const std::string multigroup_kmers_dir_prefix = cluster_data_dir + "/" + multigroup_kmers_file_prefix;

    std::unique_ptr<StageScope> st_ctx(new StageScope("start-up: HIP runtime + context"));
    Context ctx;
    st_ctx.reset();
    std::unique_ptr<Comm> comm = Comm::fromEnvironment(ctx.h);   // nullptr: one rank
    const int rank = comm ? comm->rank() : 0, world = comm ? comm->world() : 1;
    if (comm) std::cout << stamp() << "Rank " << rank << " of " << world << " (one GPU per rank)" << std::endl;
    KmerCounter kmer_counter(ctx.h, samples, kmer_size, gibbs.seed);
    std::unique_ptr<BloomHandle> path_kmer_bloom(new BloomHandle());   // ThreadedKmerBloom(num_path_kmers + max_parameter_kmers, 0.0001)
    check(bt_bloom_create(ctx.h, unit.num_path_kmers + max_parameter_kmers, 0.0001f, kmer_size, 1, &path_kmer_bloom->h), "bt_bloom_create");
    TableHandle kmer_hash;   // ObservedKmerCountsHash<N>(num_path_kmers + max_parameter_kmers)
    check(bt_table_create(ctx.h, unit.num_path_kmers + max_parameter_kmers, (uint32_t)S, kmer_size, &kmer_hash.h), "bt_table_create");

    std::cout << "\n" << stamp() << "Parsing parameter kmers ..." << std::endl;
    uint64_t num_parameter_kmers = 0;
    {
        StageScope stage("parameter k-mers");
        const std::string text = readGzFile(parameter_kmers_dir_prefix + ".fa.gz");
        const size_t header_end = std::min(text.find('\n'), text.size());
        const std::string header = text.substr(0, header_end);
        if (header != ">k" + std::to_string(kmer_size)) throw std::runtime_error(parameter_kmers_dir_prefix + ".fa.gz was written for another kmer size (" + header + ")");
        std::vector<uint64_t> kmers;
        try {   // (the lines are parsed by the -p host threads: a million k-mers were 0.3 s of a chr20-sized run on one)
            kmers = parseKmerLines(text, header_end + 1, kmer_size, clampThreads(options.getUInt("threads")));
        } catch (const std::runtime_error &e) {
            throw std::runtime_error(std::string(e.what()) + " in " + parameter_kmers_dir_prefix + ".fa.gz");
        }
        num_parameter_kmers = kmers.size() / 2;
        if (num_parameter_kmers > max_parameter_kmers) throw std::runtime_error("more than " + std::to_string(max_parameter_kmers) + " parameter kmers");
        if (num_parameter_kmers) {   // path_kmer_bloom->addKmer + kmer_hash->addKmer + isParameter(true) (main.cpp:560-577)
            void *d = nullptr;
            check(bt_malloc(ctx.h, kmers.size() * 8, &d), "bt_malloc");
            int rc = bt_memcpy_h2d(ctx.h, d, kmers.data(), kmers.size() * 8);
            if (rc == BT_OK) rc = bt_bloom_insert_batch(path_kmer_bloom->h, (const uint64_t *)d, num_parameter_kmers);
            if (rc == BT_OK) rc = bt_table_insert_batch(kmer_hash.h, (const uint64_t *)d, num_parameter_kmers, 1);
            if (rc == BT_OK) rc = bt_sync(ctx.h);
            bt_free(ctx.h, d);
            check(rc, "parameter kmers");
        }
    }
    std::cout << stamp() << "Parsed " << num_parameter_kmers << " kmers" << std::endl;
    std::cout << "\n" << std::endl;

    const ChromosomePloidy chrom_ploidy(options.getString("chromosome-ploidy-file"), chromosomes, samples);
    st.reset(new StageScope("construct variant cluster graphs (host)"));
    const UnitGraphs graphs(unit, chromosomes, kmer_size, clampThreads(options.getUInt("threads")));
    st.reset(new StageScope("count path k-mers (enumerate + Bloom insert)"));
    kmer_counter.countPathKmers(path_kmer_bloom->h, unit, graphs);
    st.reset(new StageScope("count inter-cluster k-mers"));
    kmer_counter.countInterclusterKmers(kmer_hash.h, path_kmer_bloom->h, intercluster_regions_dir_prefix, chromosomes, chrom_ploidy);
    std::cout << std::endl;
    st.reset(new StageScope("parse sample k-mers (KMC scan incl. H2D)"));
    kmer_counter.parseSampleKmers(kmer_hash.h, path_kmer_bloom->h, comm.get());
    path_kmer_bloom.reset();
    std::cout << std::endl;
    st.reset(new StageScope("classify path k-mers + haplotype candidates"));
    const GibbsBatchData batch = kmer_counter.classifyPathKmers(kmer_hash.h, unit, graphs, multigroup_kmers_dir_prefix, chrom_ploidy);
    if (getenv("BT_STAGE_TIMES")) {   // the unit's shape beside the stage table: what tools/mixture_shape.py prints for the bench's synthetic mixture
        std::map<uint32_t, uint64_t> by_h, by_k, by_n, by_v;
        auto bucket = [](uint32_t x) {   // 1, 2, 3-4, 5-8, 9-16, ...
            uint32_t b = 1;
            while (b < x) b *= 2;
            return b;
        };
        uint64_t kh = 0;
        for (uint32_t c = 0; c + 1 < batch.kmer_off.size(); c++) {
            by_h[bucket(batch.num_haplotypes[c])] += 1;
            by_k[bucket(batch.kmer_off[c + 1] - batch.kmer_off[c])] += 1;
            by_v[bucket(batch.num_variants[c])] += 1;
            kh += (uint64_t)(batch.kmer_off[c + 1] - batch.kmer_off[c]) * batch.num_haplotypes[c];
        }
        for (uint32_t g = 0; g < batch.numGroups(); g++) by_n[bucket(batch.group_cluster_off[g + 1] - batch.group_cluster_off[g])] += 1;
        auto line = [](const char *what, const std::map<uint32_t, uint64_t> &m) {
            std::cerr << "unit shape: " << what << " (upper bucket bound: count)";
            for (auto &kv : m) std::cerr << " " << kv.first << ":" << kv.second;
            std::cerr << std::endl;
        };
        std::cerr << "unit shape: " << batch.numGroups() << " groups, " << batch.kmer_off.size() - 1 << " clusters, sum over clusters of k-mers x haplotype candidates " << kh << std::endl;
        line("haplotype candidates per cluster", by_h);
        line("path k-mers per cluster", by_k);
        line("variants per cluster", by_v);
        line("clusters per group", by_n);
    }
    std::cout << "\n" << std::endl;
    st.reset(new StageScope("k-mer statistics -> count model"));

    // (every rank holds the whole table: the same moments, the same count model; only rank 0's files are kept)
    const std::string rank_suffix = rank == 0 ? "" : ".rank" + std::to_string(rank);
    CountDistribution count_distribution((unsigned short)S, noise_rate_prior, gibbs.seed);
    setGenomicCountDistributions(&count_distribution, kmer_hash.h, samples, output_prefix + "_genomic_parameters" + rank_suffix);
    bt_table_destroy(kmer_hash.h);   // the bundles hold everything the sampler needs
    kmer_hash.h = nullptr;
    st.reset();

    const bool noise_genotyping = options.getBool("noise-genotyping");
    std::vector<uint8_t> gender(S);
    std::vector<std::string> names(S);
    for (size_t s = 0; s < S; s++) {
        gender[s] = samples[s].gender;
        names[s] = samples[s].name;
    }
    InferenceEngine inference_engine(ctx.h, gender, names, gibbs);
    // this rank's groups: the whole unit, or its share (ascending unit-wide group indices; LPT on a cost proxy, the same on every rank)
    std::vector<std::vector<uint32_t>> rank_groups;
    std::vector<uint32_t> unit_clusters, unit_variants;   // per group of the WHOLE unit (estimateNoise selects its groups from them on every rank alike)
    GibbsBatchData shard;
    if (comm) {
        inference_engine.setHistReducer([&](uint64_t *hist, size_t n) { comm->allreduceHist(hist, n); }, /*uses_device_stream=*/comm->deviceReduction());
        if (comm->deviceReduction()) inference_engine.setDeviceHistReducer([&](uint64_t *d_hist, size_t n) { comm->allreduceDeviceAsync(d_hist, n); });
        rank_groups = assignGroups(batch, world);
        shard = batch.take(rank_groups[rank]);
        for (uint32_t g = 0; g < batch.numGroups(); g++) {
            unit_clusters.push_back(batch.group_cluster_off[g + 1] - batch.group_cluster_off[g]);
            uint32_t nv = 0;
            for (uint32_t c = batch.group_cluster_off[g]; c < batch.group_cluster_off[g + 1]; c++) nv += batch.num_variants[c];
            unit_variants.push_back(nv);
        }
    }
    const GibbsBatchData &my_batch = comm ? shard : batch;
    const std::string noise_prefix = output_prefix + "_noise_parameters" + rank_suffix;
    if (!noise_genotyping) {
        std::cout << "\n" << std::endl;
        StageScope stage("estimate noise (Gibbs, noise driver)");
        inference_engine.estimateNoise(&count_distribution, my_batch, noise_prefix, NoiseGroupSelector::noise_variants_batch_size, comm ? &unit_clusters : nullptr,
                                       comm ? &unit_variants : nullptr);
    }
    std::cout << "\n" << std::endl;

    Filters filters;   // Filters.cpp:33-54
    filters.min_genotype_posterior = options.getFloat("min-genotype-posterior");
    filters.min_number_of_kmers = options.getFloat("min-number-of-kmers");
    filters.min_fraction_observed_kmers.assign(S, 0.0f);
    if (!options.getBool("disable-observed-kmers"))
        for (size_t s = 0; s < S; s++) filters.min_fraction_observed_kmers[s] = Filters::minFractionObservedKmers(count_distribution.getGenomicCountDistributions()[s].mean());
    GenotypeWriter genotype_writer(names, chromosomes);
    const unsigned host_threads = clampThreads(options.getUInt("threads"));

    // collectGenotypes of every cluster of a launch (VariantClusterGenotyper::getGenotypes) -> GenotypeWriter
    const InferenceEngine::Collector collect = [&](const GibbsBatchData &b, const BatchResults &r) {
        StageScope stage("genotypes (getGenotypes + VCF lines, -p host threads)");
        std::vector<uint64_t> hapvar_off(b.numClusters() + 1, 0), var_base(b.numClusters() + 1, 0);
        for (uint32_t c = 0; c < b.numClusters(); c++) {
            hapvar_off[c + 1] = hapvar_off[c] + (uint64_t)b.num_haplotypes[c] * b.num_variants[c];
            var_base[c + 1] = var_base[c] + b.num_variants[c];
        }
        // the groups are shared out among the host threads (-p; the reference's worker threads collect their own groups' genotypes,
        // InferenceEngine.cpp:292-310); every thread formats its range's VCF lines, which are appended range after range: the order of one thread
        std::vector<std::vector<std::pair<const std::string *, GenotypeWriter::GenotypedVariant>>> lines(host_threads);
        parallelFor(b.numGroups(), host_threads, [&](size_t g_begin, size_t g_end, unsigned part) {
        auto &mine = lines[part];
        for (uint32_t g = (uint32_t)g_begin; g < (uint32_t)g_end; g++) {
            const ClusterGroup &grp = unit.variant_cluster_groups[b.group_index[g]];
            for (uint32_t c = b.group_cluster_off[g]; c < b.group_cluster_off[g + 1]; c++) {
                const VariantCluster &cluster = grp.clusters[c - b.group_cluster_off[g]];
                const std::vector<VariantInfo> info = variantClusterInfo(cluster);
                ClusterResults cr;
                cr.S = (uint32_t)S;
                cr.H = b.num_haplotypes[c];
                cr.V = b.num_variants[c];
                cr.hap_allele = b.hap_allele.data() + hapvar_off[c];
                cr.var_num_alleles = b.var_num_alleles.data() + var_base[c];
                cr.var_has_dependency = b.var_has_dependency.data() + var_base[c];
                cr.num_diplotypes = r.dip_off[c + 1] - r.dip_off[c];
                cr.h1 = r.h1.data() + r.dip_off[c];
                cr.h2 = r.h2.data() + r.dip_off[c];
                cr.freq = r.freq.data() + r.dip_off[c] * S;
                cr.stats = r.stats.data() + r.cell_off[c] * 12;
                cr.ploidy = b.group_ploidy.data() + (uint64_t)g * S;
                const std::vector<VariantGenotypes> res = getGenotypes(cr, filters);
                const ClusterAnnotation where{cluster.chrom_name, (uint32_t)info.size(), variantClusterRegion(cluster.chrom_name, info), (uint32_t)grp.clusters.size(), grp.region(), cr.H};
                for (size_t v = 0; v < info.size(); v++)
                    mine.emplace_back(&cluster.chrom_name, genotype_writer.formatGenotypes(where, info[v], res[v], formatSampleColumns(cr, (uint32_t)v, res[v])));
            }
        }
        });
        for (auto &part : lines)
            for (auto &l : part) genotype_writer.append(*l.first, std::move(l.second));
    };
    if (!comm) {
        if (!noise_genotyping) inference_engine.estimateGenotypes(batch, count_distribution, collect);
        else inference_engine.estimateNoiseAndGenotypes(batch, &count_distribution, collect, noise_prefix);
    } else {
        // every rank samples its groups; the collected samples are gathered to rank 0, which turns them into genotypes in the unit's order
        BatchResults mine;
        mine.dip_off.push_back(0);
        mine.cell_off.push_back(0);
        const InferenceEngine::Collector keep = [&](const GibbsBatchData &b, const BatchResults &r) {   // launches arrive in group order
            const uint32_t C = b.numClusters();
            const uint64_t e0 = mine.dip_off.back(), k0 = mine.cell_off.back(), nd = r.dip_off[C], nc = r.cell_off[C];
            for (uint32_t c = 0; c < C; c++) {
                mine.dip_off.push_back(e0 + r.dip_off[c + 1]);
                mine.cell_off.push_back(k0 + r.cell_off[c + 1]);
            }
            mine.h1.insert(mine.h1.end(), r.h1.begin(), r.h1.begin() + nd);
            mine.h2.insert(mine.h2.end(), r.h2.begin(), r.h2.begin() + nd);
            mine.freq.insert(mine.freq.end(), r.freq.begin(), r.freq.begin() + nd * S);
            mine.stats.insert(mine.stats.end(), r.stats.begin(), r.stats.begin() + nc * 12);
        };
        // the product's path: every launch's results are packed into one word string ON THE DEVICE (bt_gibbs_result_words), the strings wait in
        // device memory and go into the gather from there; `keep` only serves a sampler without such a string (BT_GATHER_FROM_HOST=1: the old path)
        DeviceWords on_device(ctx.h);
        uint64_t device_clusters = 0;
        if (!getenv("BT_GATHER_FROM_HOST"))
            inference_engine.setWireCollector([&](const GibbsBatchData &b, const uint32_t *d_words, uint64_t n) {
                on_device.append(d_words, n);
                device_clusters += b.numClusters();
            });
        if (!noise_genotyping) inference_engine.estimateGenotypes(my_batch, count_distribution, keep);
        else inference_engine.estimateNoiseAndGenotypes(my_batch, &count_distribution, keep, noise_prefix);
        bool from_host = mine.dip_off.size() > 1;
        const bool covered = (from_host ? mine.dip_off.size() - 1 : device_clusters) == (size_t)my_batch.numClusters() && !(from_host && device_clusters);
        {   // the ranks AGREE on the path (and on a failure) before the first collective of the gather: the two paths happen to issue the same collectives
            // today, but nothing else keeps a rank without clusters — which has neither kind of result — on its peers' path
            uint64_t votes[3] = {from_host ? 1u : 0u, !from_host && device_clusters ? 1u : 0u, covered ? 0u : 1u};
            comm->allreduceHist(votes, 3);
            if (votes[2]) throw std::runtime_error("rank " + std::to_string(rank) + ": collected samples do not cover " + (covered ? "another rank's" : "the rank's") + " clusters");
            if (votes[0] && votes[1]) throw std::runtime_error("the ranks collected their samples in different ways (BT_GATHER_FROM_HOST set on some ranks only?)");
            from_host = votes[0] != 0;
        }
        std::unique_ptr<StageScope> gather_stage(new StageScope("gather of the collected samples to rank 0"));
        const BatchResults all = from_host ? gatherResults(*comm, batch, rank_groups, mine, (uint32_t)S) : gatherResults(*comm, batch, rank_groups, on_device, (uint32_t)S);
        gather_stage.reset();
        std::cout << "[" << rank << "] gather: " << (from_host ? "from the host" : "from the device") << ", " << (from_host ? 1u : on_device.parts()) << " launch(es), "
                  << (from_host ? (uint64_t)0 : on_device.size()) * 4 << " bytes of result strings on this rank" << std::endl;
        if (rank == 0) collect(batch, all);
    }
    if (rank != 0) {   // rank 0 writes the outputs; the other ranks' parameter files are copies of its own
        std::remove((output_prefix + "_genomic_parameters" + rank_suffix + ".txt").c_str());
        std::remove((noise_prefix + ".txt").c_str());
        comm->barrier();
        return 0;
    }

    st.reset(new StageScope("write VCF (sort + output)"));
    const uint32_t num_genotyped_variants = genotype_writer.finalise(output_prefix, options.getBool("gzip-output"), options.getString("genome-file"), unit.cluster_options_header, options.getHeader());
    std::cout << "\n" << stamp() << "Out of " << unit.num_variants << " variants:\n" << std::endl;
    std::cout << "\t- " << num_genotyped_variants << " were genotyped" << std::endl;
    std::cout << "\t- " << unit.num_variants - num_genotyped_variants << " were skipped (unsupported)" << std::endl;
    st.reset();
    // (what the table does not list: option / sample parsing before the first stage, and — after this line — the release of the unit's host arrays and of the device
    //  allocations when the process ends; the caller's wall clock also holds the loading of the executable and its libraries)
    StageTimes::get().add("wall since main()", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_main_start).count());
    StageTimes::get().print("bayesTyper genotype");
    std::cout << "\n\n" << stamp() << "BayesTyper genotype completed succesfully!\n" << std::endl;
    if (comm) comm->barrier();
    return 0;
}

// BT_GPUS=N: this process becomes rank 0 of N and starts the other ranks as copies of itself (before anything touches the GPU); their
// output goes to <output-prefix>.rank<r>.log.  Returns the children's pids (empty when the ranks were started by someone else).
std::vector<pid_t> startRanks(int argc, char *const argv[]) {
    std::vector<pid_t> children;
    const char *gpus = getenv("BT_GPUS");
    const int n = gpus ? atoi(gpus) : 1;
    if (n <= 1 || getenv("BT_WORLD")) return children;
    if (argc < 2 || std::strcmp(argv[1], "genotype") != 0) return children;
    std::string prefix = "bayestyper";   // -o / --output-prefix (main.cpp:379)
    for (int i = 2; i + 1 < argc; i++)
        if (std::strcmp(argv[i], "-o") == 0 || std::strcmp(argv[i], "--output-prefix") == 0) prefix = argv[i + 1];
    const std::string id_file = prefix + ".comm_id." + std::to_string((long)getpid());
    std::remove(id_file.c_str());
    setenv("BT_WORLD", std::to_string(n).c_str(), 1);
    setenv("BT_COMM_ID_FILE", id_file.c_str(), 1);
    // the run's nonce: every rank only accepts a rendezvous file that carries it (host/Comm.cpp)
    setenv("BT_COMM_NONCE", (std::to_string((long)getpid()) + "." + std::to_string((long long)std::chrono::steady_clock::now().time_since_epoch().count())).c_str(), 1);
    for (int r = 1; r < n; r++) {
        const pid_t pid = fork();
        if (pid < 0) throw std::runtime_error("cannot start rank " + std::to_string(r));
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGTERM);   // (a rank does not outlive the process that started it)
            setenv("BT_RANK", std::to_string(r).c_str(), 1);
            const std::string log = prefix + ".rank" + std::to_string(r) + ".log";
            // ONE open file description for both streams (O_APPEND): two independent ones would overwrite each other's output
            const int fd = open(log.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_APPEND, 0644);
            if (fd < 0 || dup2(fd, 1) < 0 || dup2(fd, 2) < 0) _exit(1);
            if (fd > 2) close(fd);
            execv("/proc/self/exe", argv);
            _exit(127);
        }
        children.push_back(pid);
    }
    setenv("BT_RANK", "0", 1);
    return children;
}

}  // namespace

int main(int argc, char *const argv[]) {
    const unsigned kmer_size = getenv("BT_KMER_SIZE") ? (unsigned)atoi(getenv("BT_KMER_SIZE")) : 55u;
    std::cout << "\n[" << getLocalTime() << "] You are using BayesTyper (" << BT_VERSION << ")\n" << std::endl;
    const std::string command_info = "Usage: bayesTyper <command> [options]\n\nCommands:\n\n\tcluster\t\tcreate variant clusters\n\tgenotype\tgenotype variant clusters\n";
    if (argc == 1) {
        std::cout << command_info << std::endl;
        return 0;
    }
    std::vector<pid_t> children;
    std::atomic<bool> finished(false);
    std::thread watchdog;
    int rc = 0;
    try {
        if (kmer_size < 1 || kmer_size > 64) throw std::runtime_error("BT_KMER_SIZE must be between 1 and 64");
        if (std::strcmp(argv[1], "cluster") == 0) return runCluster(argc, argv, kmer_size);
        if (std::strcmp(argv[1], "genotype") == 0) {
            children = startRanks(argc, argv);
            if (!children.empty())   // a rank that dies would leave the others waiting in a collective: end the run instead
                watchdog = std::thread([&]() {
                    while (!finished.load()) {
                        for (pid_t &pid : children) {
                            if (pid <= 0) continue;   // reaped: its pid may belong to somebody else by now
                            int status = 0;
                            const pid_t got = waitpid(pid, &status, WNOHANG);
                            if (got != pid) continue;
                            pid = 0;
                            if (!(WIFEXITED(status) && WEXITSTATUS(status) == 0)) {
                                std::cerr << "\nERROR: a rank of this run failed (see <output-prefix>.rank<r>.log)\n" << std::endl;
                                for (pid_t other : children)
                                    if (other > 0) kill(other, SIGTERM);
                                if (getenv("BT_COMM_ID_FILE")) std::remove(getenv("BT_COMM_ID_FILE"));
                                _exit(1);
                            }
                        }
                        std::this_thread::sleep_for(std::chrono::milliseconds(200));
                    }
                });
            rc = runGenotype(argc, argv, kmer_size);
        } else {
            std::cout << command_info << std::endl;
            return 0;
        }
    } catch (const std::exception &e) {   // the reference prints "\nERROR: ...\n" and exits with 1
        std::cerr << "\nERROR: " << e.what() << "\n" << std::endl;
        Comm::markFailed();
        rc = 1;
    }
    finished.store(true);
    if (watchdog.joinable()) watchdog.join();
    for (pid_t pid : children) {   // the ranks this process started (those the watchdog has reaped ended well)
        if (pid <= 0) continue;
        int status = 0;
        if (rc != 0) kill(pid, SIGTERM);   // rank 0 failed: the others would wait for it forever
        if (waitpid(pid, &status, 0) == pid && !(WIFEXITED(status) && WEXITSTATUS(status) == 0)) {
            if (rc == 0) std::cerr << "\nERROR: a rank of this run failed (see <output-prefix>.rank<r>.log)\n" << std::endl;
            rc = 1;
        }
    }
    if (!children.empty() && getenv("BT_COMM_ID_FILE")) {
        std::remove(getenv("BT_COMM_ID_FILE"));
        std::remove((std::string(getenv("BT_COMM_ID_FILE")) + ".failed").c_str());
    }
    return rc;
}
