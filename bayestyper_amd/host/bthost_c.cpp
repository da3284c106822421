// C entry points of libbthost.so used by the Python tests and bench.py (ctypes).  The C++ classes in this directory
// are the host layer proper; this file only exposes them.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <sstream>
#include <memory>

#include "Comm.hpp"
#include "CountDistribution.hpp"
#include "GenotypeWriter.hpp"
#include "Genotypes.hpp"
#include "InferenceEngine.hpp"
#include "KmcFile.hpp"
#include "KmerHashOrder.hpp"
#include "Parallel.hpp"
#include "InferenceUnit.hpp"
#include "KmerCounter.hpp"
#include "Sample.hpp"
#include "VariantClusterGraph.hpp"
#include "VariantFileParser.hpp"

using namespace bthost;

extern "C" {

// writeGzFile (the cluster stage's parameter_kmers.fa.gz / intercluster_regions.txt.gz): for the test that the bytes do not depend on the thread count
int bth_write_gz(const char *filename, const char *data, unsigned long long len, unsigned threads) {
    try {
        writeGzFile(filename, std::string(data, (size_t)len), threads);
    } catch (const std::exception &) {
        return 1;
    }
    return 0;
}

// parseKmerLines (the genotype stage's reading of parameter_kmers.fa.gz): out[2 * i] = lo, out[2 * i + 1] = hi; returns 0, or 1 with the message in err
int bth_parse_kmer_lines(const char *text, unsigned long long len, unsigned long long begin, unsigned k, unsigned threads, unsigned long long *out, unsigned long long capacity,
                         unsigned long long *num_kmers, char *err, unsigned err_len) {
    try {
        const std::vector<uint64_t> v = parseKmerLines(std::string(text, (size_t)len), (size_t)begin, k, threads);
        if (v.size() > capacity) throw std::runtime_error("output buffer too small");
        if (!v.empty()) std::memcpy(out, v.data(), v.size() * 8);
        *num_kmers = v.size() / 2;
        return 0;
    } catch (const std::exception &e) {
        if (err && err_len) {
            std::strncpy(err, e.what(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return 1;
    }
}

// LUTs for S samples from per-sample (mean, var, multiplicity) of the parameter k-mers and explicit noise rates
int bth_build_luts(unsigned S, const double *mean, const double *var, const unsigned *multiplicity, const double *noise_rates, double *genomic, double *noise) {
    try {
        CountDistribution cd((unsigned short)S, std::make_pair(1.0f, 0.01f), 0);
        for (unsigned s = 0; s < S; s++) cd.setGenomicFromMoments((unsigned short)s, mean[s], var[s], multiplicity ? multiplicity[s] : 1);
        cd.setNoiseRates(std::vector<double>(noise_rates, noise_rates + S));
        std::memcpy(genomic, cd.genomicTable().data(), (size_t)S * 65536 * 8);
        std::memcpy(noise, cd.noiseTable().data(), (size_t)S * 256 * 8);
        return 0;
    } catch (...) {
        return 1;
    }
}

void *bth_count_distribution_new(unsigned S, float prior_shape, float prior_scale, unsigned seed) {
    return new CountDistribution((unsigned short)S, std::make_pair(prior_shape, prior_scale), seed);
}
void bth_count_distribution_free(void *h) { delete (CountDistribution *)h; }
void bth_count_distribution_set_genomic(void *h, unsigned s, double mean, double var, unsigned multiplicity) {
    ((CountDistribution *)h)->setGenomicFromMoments((unsigned short)s, mean, var, multiplicity);
}
void bth_count_distribution_noise_rates(void *h, double *out) {
    auto &r = ((CountDistribution *)h)->getNoiseRates();
    std::memcpy(out, r.data(), r.size() * 8);
}
void bth_count_distribution_set_noise_rates(void *h, const double *rates, unsigned S) { ((CountDistribution *)h)->setNoiseRates(std::vector<double>(rates, rates + S)); }
void bth_count_distribution_reset_noise_rates(void *h) { ((CountDistribution *)h)->resetNoiseRates(); }
// the generator state as bt_noise_rng lays it out: 624 words, position, saved_available, then the saved variate
void bth_count_distribution_export_generator(void *h, uint32_t *words626, double *saved) {
    ((CountDistribution *)h)->exportGenerator(words626, words626 + 624, words626 + 625, saved);
}
void bth_count_distribution_import_generator(void *h, const uint32_t *words626, double saved) {
    ((CountDistribution *)h)->importGenerator(words626, words626[624], words626[625], saved);
}
// sampleNoiseParameters from a [S*256] u64 histogram
void bth_count_distribution_sample_noise(void *h, const unsigned long long *hist, unsigned S) {
    CountAllocation ca((unsigned short)S);
    for (unsigned s = 0; s < S; s++)
        for (unsigned i = 0; i < 256; i++) ca.counts()[s][i] = hist[s * 256 + i];
    ((CountDistribution *)h)->sampleNoiseParameters(ca);
}
// ---- InferenceEngine (the three drivers) behind a handle: what bayestyper_amd/host/inference_engine.py binds ----------------------
// A sampler supplied by the caller (tests: the oracle's; see InferenceEngine.hpp: GibbsSampler).  All callbacks get `user` first.
struct bth_sampler_vtable {
    void *(*create)(void *user, const bt_gibbs_params *params, const bt_gibbs_batch *batch);
    void (*destroy)(void *user, void *h);
    void (*set_lut)(void *user, void *h, const double *genomic, const double *noise);
    void (*set_noise_lut)(void *user, void *h, const double *noise);
    void (*init_chain)(void *user, void *h, uint32_t chain);
    void (*sweep)(void *user, void *h, uint32_t n, int collect);
    void (*run)(void *user, void *h);
    void (*noise_counts)(void *user, void *h, uint64_t *hist /* [S*256] */);
    void (*result_sizes)(void *user, void *h, uint64_t *num_entries, uint64_t *num_cells);
    void (*result_fetch)(void *user, void *h, uint64_t *dip_off, uint16_t *h1, uint16_t *h2, uint32_t *freq, uint64_t *cell_off, double *stats);
};
namespace {
struct CallbackSampler : GibbsSampler {
    bth_sampler_vtable vt;
    void *user, *h;
    uint32_t S;
    CallbackSampler(const bth_sampler_vtable &v, void *u, const bt_gibbs_params &p, const GibbsBatchData &batch) : vt(v), user(u), S(p.num_samples) {
        const bt_gibbs_batch b = batch.view();
        h = vt.create(user, &p, &b);
        if (!h) throw std::runtime_error("sampler callback: create failed");
    }
    ~CallbackSampler() override { vt.destroy(user, h); }
    void setLut(const double *g, const double *n) override { vt.set_lut(user, h, g, n); }
    void setNoiseLut(const double *n) override { vt.set_noise_lut(user, h, n); }
    void initChain(uint32_t c) override { vt.init_chain(user, h, c); }
    void sweep(uint32_t n, bool collect) override { vt.sweep(user, h, n, collect ? 1 : 0); }
    void run() override { vt.run(user, h); }
    std::vector<uint64_t> noiseCounts() override {
        std::vector<uint64_t> hist((size_t)S * 256, 0);
        vt.noise_counts(user, h, hist.data());
        return hist;
    }
    BatchResults results(uint32_t num_clusters) override {
        BatchResults r;
        uint64_t nd = 0, nc = 0;
        vt.result_sizes(user, h, &nd, &nc);
        r.dip_off.resize(num_clusters + 1);
        r.cell_off.resize(num_clusters + 1);
        r.h1.resize(std::max<uint64_t>(nd, 1));
        r.h2.resize(std::max<uint64_t>(nd, 1));
        r.freq.resize(std::max<uint64_t>(nd * S, 1));
        r.stats.resize(std::max<uint64_t>(nc * 12, 1));
        vt.result_fetch(user, h, r.dip_off.data(), r.h1.data(), r.h2.data(), r.freq.data(), r.cell_off.data(), r.stats.data());
        return r;
    }
};
struct EngineHandle {
    std::unique_ptr<InferenceEngine> engine;
    uint32_t S = 0;
    // collected samples of the last estimateGenotypes / estimateNoiseAndGenotypes: the launches' results appended in call order
    BatchResults collected;
    uint64_t num_clusters = 0, num_entries = 0, num_cells = 0;
    std::vector<uint32_t> launch_groups;   // groups per launch
    void reset() {
        collected = BatchResults();
        collected.dip_off.push_back(0);
        collected.cell_off.push_back(0);
        num_clusters = num_entries = num_cells = 0;
        launch_groups.clear();
    }
    InferenceEngine::Collector collector() {
        return [this](const GibbsBatchData &batch, const BatchResults &r) {
            const uint32_t C = batch.numClusters();
            const uint64_t nd = r.dip_off[C], nc = r.cell_off[C];
            for (uint32_t c = 0; c < C; c++) {
                collected.dip_off.push_back(num_entries + r.dip_off[c + 1]);
                collected.cell_off.push_back(num_cells + r.cell_off[c + 1]);
            }
            collected.h1.insert(collected.h1.end(), r.h1.begin(), r.h1.begin() + nd);
            collected.h2.insert(collected.h2.end(), r.h2.begin(), r.h2.begin() + nd);
            collected.freq.insert(collected.freq.end(), r.freq.begin(), r.freq.begin() + nd * S);
            collected.stats.insert(collected.stats.end(), r.stats.begin(), r.stats.begin() + nc * 12);
            num_clusters += C;
            num_entries += nd;
            num_cells += nc;
            launch_groups.push_back(batch.numGroups());
        };
    }
};
int engine_error(const std::exception &e, char *err, unsigned err_len) {
    if (err && err_len) {
        std::strncpy(err, e.what(), err_len - 1);
        err[err_len - 1] = 0;
    }
    return 1;
}
}  // namespace

void *bth_engine_new(bt_ctx *ctx, unsigned S, const uint8_t *gender, const char *sample_names /* tab-separated */, unsigned seed, uint32_t burn_in, uint32_t samples, uint32_t chains, float kmer_subsampling_rate,
                     uint32_t max_haplotype_variant_kmers, uint32_t max_groups_per_launch) {
    GibbsOptions o;
    o.seed = seed;
    o.burn_in = burn_in;
    o.samples = samples;
    o.chains = chains;
    o.kmer_subsampling_rate = kmer_subsampling_rate;
    o.max_haplotype_variant_kmers = max_haplotype_variant_kmers;
    o.max_groups_per_launch = max_groups_per_launch;
    std::vector<std::string> names;
    {
        std::stringstream ss(sample_names ? sample_names : "");
        std::string n;
        while (std::getline(ss, n, '\t')) names.push_back(n);
        for (unsigned s = (unsigned)names.size(); s < S; s++) names.push_back("sample_" + std::to_string(s));
    }
    auto *h = new EngineHandle();
    h->S = S;
    h->engine.reset(new InferenceEngine(ctx, std::vector<uint8_t>(gender, gender + S), names, o));
    h->engine->recordNoiseRows(true);
    h->engine->setQuiet(true);
    h->reset();
    return h;
}
void bth_engine_free(void *h) { delete (EngineHandle *)h; }
void bth_engine_set_sampler(void *h, const bth_sampler_vtable *vt, void *user) {
    const bth_sampler_vtable v = *vt;
    ((EngineHandle *)h)->engine->setSamplerFactory([v, user](const bt_gibbs_params &p, const GibbsBatchData &b) { return std::unique_ptr<GibbsSampler>(new CallbackSampler(v, user, p, b)); });
}
void bth_engine_set_hist_reducer(void *h, void (*fn)(uint64_t *hist, uint64_t n, void *user), void *user) {
    ((EngineHandle *)h)->engine->setHistReducer([fn, user](uint64_t *hist, size_t n) { fn(hist, (uint64_t)n, user); });
}
int bth_engine_estimate_noise(void *h, void *count_distribution, const bt_gibbs_batch *batch, const char *output_prefix, uint32_t variants_batch_size,
                              const uint32_t *unit_clusters_per_group, const uint32_t *unit_variants_per_group, uint32_t unit_groups, int *low_variant_warning, char *err,
                              unsigned err_len) {
    auto *e = (EngineHandle *)h;
    try {
        const GibbsBatchData unit = GibbsBatchData::fromView(*batch, e->S);
        std::vector<uint32_t> cl, va;
        if (unit_clusters_per_group && unit_variants_per_group) {
            cl.assign(unit_clusters_per_group, unit_clusters_per_group + unit_groups);
            va.assign(unit_variants_per_group, unit_variants_per_group + unit_groups);
        }
        e->engine->estimateNoise((CountDistribution *)count_distribution, unit, output_prefix, variants_batch_size, cl.empty() ? nullptr : &cl, va.empty() ? nullptr : &va);
        if (low_variant_warning) *low_variant_warning = e->engine->lowNoiseVariantWarning() ? 1 : 0;
    } catch (const std::exception &ex) {
        return engine_error(ex, err, err_len);
    }
    return 0;
}
int bth_engine_estimate_genotypes(void *h, const bt_gibbs_batch *batch, void *count_distribution, char *err, unsigned err_len) {
    auto *e = (EngineHandle *)h;
    try {
        e->reset();
        e->engine->estimateGenotypes(GibbsBatchData::fromView(*batch, e->S), *(CountDistribution *)count_distribution, e->collector());
    } catch (const std::exception &ex) {
        return engine_error(ex, err, err_len);
    }
    return 0;
}
int bth_engine_estimate_noise_and_genotypes(void *h, const bt_gibbs_batch *batch, void *count_distribution, const char *output_prefix, char *err, unsigned err_len) {
    auto *e = (EngineHandle *)h;
    try {
        e->reset();
        e->engine->estimateNoiseAndGenotypes(GibbsBatchData::fromView(*batch, e->S), (CountDistribution *)count_distribution, e->collector(), output_prefix);
    } catch (const std::exception &ex) {
        return engine_error(ex, err, err_len);
    }
    return 0;
}
// sizes[4] = clusters, diplotype entries, allele cells, launches
void bth_engine_result_sizes(void *h, uint64_t *sizes) {
    auto *e = (EngineHandle *)h;
    sizes[0] = e->num_clusters;
    sizes[1] = e->num_entries;
    sizes[2] = e->num_cells;
    sizes[3] = e->launch_groups.size();
}
void bth_engine_result_fetch(void *h, uint64_t *dip_off, uint16_t *h1, uint16_t *h2, uint32_t *freq, uint64_t *cell_off, double *stats, uint32_t *launch_groups) {
    auto *e = (EngineHandle *)h;
    const BatchResults &r = e->collected;
    std::memcpy(dip_off, r.dip_off.data(), r.dip_off.size() * 8);
    std::memcpy(cell_off, r.cell_off.data(), r.cell_off.size() * 8);
    if (!r.h1.empty()) std::memcpy(h1, r.h1.data(), r.h1.size() * 2);
    if (!r.h2.empty()) std::memcpy(h2, r.h2.data(), r.h2.size() * 2);
    if (!r.freq.empty()) std::memcpy(freq, r.freq.data(), r.freq.size() * 4);
    if (!r.stats.empty()) std::memcpy(stats, r.stats.data(), r.stats.size() * 8);
    if (launch_groups && !e->launch_groups.empty()) std::memcpy(launch_groups, e->launch_groups.data(), e->launch_groups.size() * 4);
}
// rows of the noise parameter file in full precision: (chain, iteration, rate_0 ..) per row; returns the number of doubles
uint64_t bth_engine_noise_rows(void *h, double *out, uint64_t cap) {
    const std::vector<double> &r = ((EngineHandle *)h)->engine->noiseRows();
    if (out && cap >= r.size() && !r.empty()) std::memcpy(out, r.data(), r.size() * 8);
    return r.size();
}

// bthost::Comm over the files transport (no GPU needed): `rounds` rounds of all-reduce, byte all-gather and word gather with checked contents.
// Called by every rank of a test run (BT_WORLD / BT_RANK / BT_COMM_ID_FILE / BT_COMM_TRANSPORT=files in the environment); 0 = all exchanges correct.
int bth_comm_selftest(unsigned rounds, char *err, unsigned err_len) {
    try {
        std::unique_ptr<Comm> c = Comm::fromEnvironment(nullptr);
        if (!c) throw std::runtime_error("bth_comm_selftest: BT_WORLD is 1");
        const int W = c->world(), R = c->rank();
        for (unsigned i = 0; i < rounds; i++) {
            std::vector<uint64_t> h(5);
            for (size_t j = 0; j < h.size(); j++) h[j] = (uint64_t)(R + 1) * (j + 1) + i;
            c->allreduceHist(h.data(), h.size());
            for (size_t j = 0; j < h.size(); j++)
                if (h[j] != (uint64_t)W * (W + 1) / 2 * (j + 1) + (uint64_t)W * i) throw std::runtime_error("all-reduce: wrong sum");
            std::vector<uint8_t> mine((size_t)(3 + R + i % 2), (uint8_t)(10 * R + i));
            std::vector<uint64_t> off;
            const std::vector<uint8_t> all = c->allgatherBytes(mine, &off);
            for (int r = 0; r < W; r++) {
                if (off[r + 1] - off[r] != (uint64_t)(3 + r + i % 2)) throw std::runtime_error("all-gather: wrong part size");
                for (uint64_t b = off[r]; b < off[r + 1]; b++)
                    if (all[b] != (uint8_t)(10 * r + i)) throw std::runtime_error("all-gather: wrong byte");
            }
            std::vector<uint32_t> words((size_t)(R == 1 ? 0 : 2 + R), (uint32_t)(1000 * R + i));   // (one rank contributes nothing)
            const std::vector<uint32_t> g = c->gatherWords(words, &off);
            if (R == 0) {
                for (int r = 0; r < W; r++)
                    for (uint64_t b = off[r]; b < off[r + 1]; b++)
                        if (g[b] != (uint32_t)(1000 * r + i)) throw std::runtime_error("gather: wrong word");
                if (off[W] != g.size()) throw std::runtime_error("gather: wrong total");
            } else if (!g.empty())
                throw std::runtime_error("gather: a rank other than 0 received data");
        }
        {   // gatherResults: a unit of 11 groups dealt to the ranks by index; cluster c has 1 + c % 3 entries and 2 + c % 2 cells whose values name it
            const uint32_t S = 2, G = 11;
            GibbsBatchData unit;
            unit.group_cluster_off.push_back(0);
            for (uint32_t g = 0; g < G; g++) {
                unit.group_index.push_back(g);
                for (uint32_t k = 0; k < 1 + g % 2; k++) unit.cluster_idx.push_back((uint32_t)unit.cluster_idx.size());
                unit.group_cluster_off.push_back((uint32_t)unit.cluster_idx.size());
            }
            std::vector<std::vector<uint32_t>> ids((size_t)W);
            for (uint32_t g = 0; g < G; g++)
                if (W < 3 || g % W != 1 || g % 2) ids[g % W].push_back(g);
                else ids[0].push_back(g);   // (uneven shares)
            for (auto &v : ids) std::sort(v.begin(), v.end());
            auto fill = [&](const std::vector<uint32_t> &clusters) {
                BatchResults r;
                r.dip_off.push_back(0);
                r.cell_off.push_back(0);
                for (uint32_t cl : clusters) {
                    for (uint32_t e = 0; e < 1 + cl % 3; e++) {
                        r.h1.push_back((uint16_t)(cl + e));
                        r.h2.push_back((uint16_t)(e ? 0xFFFFu : cl));
                        for (uint32_t smp = 0; smp < S; smp++) r.freq.push_back(1000 * cl + 10 * e + smp);
                    }
                    for (uint32_t k = 0; k < 2 + cl % 2; k++)
                        for (uint32_t j = 0; j < 12; j++) r.stats.push_back(cl + k / 8.0 + j / 1024.0);
                    r.dip_off.push_back(r.h1.size());
                    r.cell_off.push_back(r.stats.size() / 12);
                }
                return r;
            };
            std::vector<uint32_t> my_clusters, all_clusters;
            for (uint32_t g : ids[R])
                for (uint32_t cl = unit.group_cluster_off[g]; cl < unit.group_cluster_off[g + 1]; cl++) my_clusters.push_back(cl);
            for (uint32_t cl = 0; cl < unit.numClusters(); cl++) all_clusters.push_back(cl);
            const BatchResults got = gatherResults(*c, unit, ids, fill(my_clusters), S), want = fill(all_clusters);
            if (R == 0) {
                const uint64_t nd = want.dip_off.back(), nc = want.cell_off.back();
                if (got.dip_off != want.dip_off || got.cell_off != want.cell_off) throw std::runtime_error("gatherResults: wrong offsets");
                if (!std::equal(want.h1.begin(), want.h1.end(), got.h1.begin()) || !std::equal(want.h2.begin(), want.h2.end(), got.h2.begin())) throw std::runtime_error("gatherResults: wrong keys");
                if (!std::equal(want.freq.begin(), want.freq.begin() + nd * S, got.freq.begin())) throw std::runtime_error("gatherResults: wrong counts");
                if (!std::equal(want.stats.begin(), want.stats.begin() + nc * 12, got.stats.begin())) throw std::runtime_error("gatherResults: wrong statistics");
            } else if (!got.dip_off.empty())
                throw std::runtime_error("gatherResults: a rank other than 0 received results");
        }
        c->barrier();
    } catch (const std::exception &e) {
        Comm::markFailed();
        return engine_error(e, err, err_len);
    }
    return 0;
}

// NoiseGroupSelector (estimateNoise's per-chain choice of groups)
void *bth_noise_selector_new(const uint32_t *clusters_per_group, const uint32_t *variants_per_group, uint32_t num_groups, unsigned seed, uint32_t batch_size) {
    return new NoiseGroupSelector(clusters_per_group, variants_per_group, num_groups, seed, batch_size);
}
void bth_noise_selector_free(void *h) { delete (NoiseGroupSelector *)h; }
// out must hold num_groups entries; returns the number of selected groups, *num_variants = variants they cover
uint32_t bth_noise_selector_next_chain(void *h, uint32_t *out, uint32_t *num_variants) {
    auto *sel = (NoiseGroupSelector *)h;
    auto v = sel->nextChain();
    std::memcpy(out, v.data(), v.size() * 4);
    if (num_variants) *num_variants = sel->lastNumVariants();
    return (uint32_t)v.size();
}
// row of the noise parameter file into buf (NUL-terminated, truncated to cap); returns the full length
unsigned bth_noise_parameter_row(unsigned chain, unsigned iteration, const double *rates, unsigned S, char *buf, unsigned cap) {
    std::string r = noiseParameterRow(chain, iteration, std::vector<double>(rates, rates + S));
    if (cap) {
        unsigned n = (unsigned)std::min<size_t>(r.size(), cap - 1);
        std::memcpy(buf, r.data(), n);
        buf[n] = 0;
    }
    return (unsigned)r.size();
}

void bth_count_distribution_tables(void *h, double *genomic, double *noise, unsigned S) {
    auto *cd = (CountDistribution *)h;
    if (genomic) std::memcpy(genomic, cd->genomicTable().data(), (size_t)S * 65536 * 8);
    if (noise) std::memcpy(noise, cd->noiseTable().data(), (size_t)S * 256 * 8);
}


// getGenotypes for one cluster into flat arrays.  Strides: Amax alleles, Gmax = Amax*(Amax+1)/2 genotypes.
//   gpp[(v*S+s)*Gmax + g], app[(v*S+s)*Amax + a], filters[(v*S+s)*Amax + a], estimate[(v*S+s)*2 + i] (0xFFFF none / unused),
//   gq[v*S+s], total_count[v], alt_counts[v*Amax + a], alt_freq[v*Amax + a], acp[v*Amax + a], max_alt_acp[v], non_covered[v*Amax + a] (0/1)
int bth_cluster_genotypes(unsigned S, unsigned H, unsigned V, const uint16_t *hap_allele, const uint16_t *var_num_alleles, const uint8_t *var_has_dependency,
                          unsigned long long num_diplotypes, const uint16_t *h1, const uint16_t *h2, const uint32_t *freq, const double *stats, const uint8_t *ploidy,
                          float min_gpp, float min_kmers, const float *min_fraction, unsigned Amax, float *gpp, float *app, uint16_t *filters, uint16_t *estimate,
                          uint32_t *gq, uint32_t *total_count, uint32_t *alt_counts, float *alt_freq, float *acp, float *max_alt_acp, uint8_t *non_covered) {
    try {
        ClusterResults r;
        r.S = S; r.H = H; r.V = V;
        r.hap_allele = hap_allele; r.var_num_alleles = var_num_alleles; r.var_has_dependency = var_has_dependency;
        r.num_diplotypes = num_diplotypes; r.h1 = h1; r.h2 = h2; r.freq = freq; r.stats = stats; r.ploidy = ploidy;
        Filters f;
        f.min_genotype_posterior = min_gpp;
        f.min_number_of_kmers = min_kmers;
        f.min_fraction_observed_kmers.assign(min_fraction, min_fraction + S);
        const auto res = getGenotypes(r, f);
        const unsigned Gmax = Amax * (Amax + 1) / 2;
        for (unsigned v = 0; v < V; v++) {
            const auto &g = res[v];
            for (unsigned s = 0; s < S; s++) {
                const auto &st = g.sample_stats[s];
                const size_t vs = (size_t)v * S + s;
                for (size_t i = 0; i < st.genotype_posteriors.size(); i++) gpp[vs * Gmax + i] = st.genotype_posteriors[i];
                for (size_t i = 0; i < st.allele_posteriors.size(); i++) {
                    app[vs * Amax + i] = st.allele_posteriors[i];
                    filters[vs * Amax + i] = st.allele_filters[i];
                }
                estimate[vs * 2] = estimate[vs * 2 + 1] = 0xFFFF;
                for (size_t i = 0; i < st.genotype_estimate.size(); i++) estimate[vs * 2 + i] = st.genotype_estimate[i];
                gq[vs] = st.genotype_quality;
            }
            total_count[v] = g.variant_stats.total_count;
            max_alt_acp[v] = g.variant_stats.max_alt_allele_call_probability;
            for (size_t a = 0; a < g.variant_stats.alt_allele_counts.size(); a++) {
                alt_counts[(size_t)v * Amax + a] = g.variant_stats.alt_allele_counts[a];
                alt_freq[(size_t)v * Amax + a] = g.variant_stats.alt_allele_frequency[a];
            }
            for (size_t a = 0; a < g.variant_stats.allele_call_probabilities.size(); a++) acp[(size_t)v * Amax + a] = g.variant_stats.allele_call_probabilities[a];
            for (uint16_t a : g.non_covered_alleles) non_covered[(size_t)v * Amax + a] = 1;
        }
        return 0;
    } catch (...) {
        return 1;
    }
}


// One cluster's graph from flat variant arrays (tests: cross-check against the Python restatement in synth_graphs.py).
//   variants: var_pos[nvar], var_nalt[nvar], var_redundant[nvar], var_dep[nvar]; alternative alleles concatenated: alt_ref_len[], alt_off[] (+1), alt_seq (ASCII)
//   contained clusters: cont_lf[], cont_rf[], cont_idx[]
void *bth_graph_build(unsigned k, const char *chrom, unsigned long long chrom_len, unsigned nvar, const uint32_t *var_pos, const uint32_t *var_nalt,
                      const uint32_t *var_redundant, const uint8_t *var_dep, const uint32_t *alt_ref_len, const uint32_t *alt_off, const char *alt_seq, unsigned ncont,
                      const uint32_t *cont_lf, const uint32_t *cont_rf, const uint32_t *cont_idx) {
    try {
        VariantCluster vc;
        unsigned a = 0;
        for (unsigned v = 0; v < nvar; v++) {
            Variant var;
            var.has_dependency = var_dep[v] != 0;
            var.num_redundant_nucleotides = var_redundant[v];
            for (unsigned i = 0; i < var_nalt[v]; i++, a++) var.alt_alleles.push_back(AlleleInfo{alt_ref_len[a], std::string(alt_seq + alt_off[a], alt_seq + alt_off[a + 1])});
            vc.variants.emplace(var_pos[v], var);
        }
        for (unsigned i = 0; i < ncont; i++) vc.contained_clusters.push_back(ContainedCluster{cont_idx[i], cont_lf[i], cont_rf[i]});
        return new VariantClusterGraph(vc, std::string(chrom, chrom + chrom_len), k);
    } catch (...) {
        return nullptr;
    }
}
void bth_graph_free(void *h) { delete (VariantClusterGraph *)h; }
// sizes: vertices, edges, total nucleotides, total reference_variant_indices
void bth_graph_sizes(void *h, uint64_t *sizes) {
    auto *g = (VariantClusterGraph *)h;
    sizes[0] = g->vertices.size();
    sizes[1] = g->edges.size();
    sizes[2] = sizes[3] = 0;
    for (auto &v : g->vertices) {
        sizes[2] += v.sequence.size();
        sizes[3] += v.reference_variant_indices.size();
    }
}
void bth_graph_fetch(void *h, uint64_t *seq_off, uint8_t *seq, uint16_t *vvar, uint16_t *vall, uint8_t *vflags, uint32_t *vnested, uint32_t *refvar_off, uint16_t *refvar,
                     uint32_t *edges, uint16_t *var_num_alleles, uint8_t *var_dep) {
    auto *g = (VariantClusterGraph *)h;
    uint64_t so = 0;
    uint32_t ro = 0;
    seq_off[0] = 0;
    refvar_off[0] = 0;
    for (size_t v = 0; v < g->vertices.size(); v++) {
        auto &x = g->vertices[v];
        for (uint8_t c : x.sequence) seq[so++] = c;
        seq_off[v + 1] = so;
        vvar[v] = x.variant;
        vall[v] = x.allele;
        vflags[v] = (uint8_t)((x.is_disconnected ? 1 : 0) | (x.is_first_nucleotides_redundant ? 2 : 0));
        vnested[v] = x.nested_variant_cluster_index;
        for (uint16_t r : x.reference_variant_indices) refvar[ro++] = r;
        refvar_off[v + 1] = ro;
    }
    for (size_t e = 0; e < g->edges.size(); e++) {
        edges[2 * e] = g->edges[e].first;
        edges[2 * e + 1] = g->edges[e].second;
    }
    for (size_t v = 0; v < g->var_num_alleles.size(); v++) {
        var_num_alleles[v] = g->var_num_alleles[v];
        var_dep[v] = g->var_has_dependency[v];
    }
}


// ---- cluster stage front end: VCF + genome -> units of variant-cluster groups (VariantFileParser.hpp) ----
namespace {
struct ClusterStage {
    unsigned k;
    uint32_t max_allele_length;
    float cnv_threshold;
    Chromosomes chromosomes;
    std::unique_ptr<VariantFileParser> parser;
    std::vector<ClusterGroup> unit;   // groups of the unit parsed last
};
int stage_error(const std::exception &e, char *err, unsigned err_len) {
    if (err && err_len) {
        std::strncpy(err, e.what(), err_len - 1);
        err[err_len - 1] = 0;
    }
    return -1;
}
uint64_t copy_out(const std::string &s, char *buf, uint64_t cap) {
    if (buf && cap >= s.size()) std::memcpy(buf, s.data(), s.size());
    return s.size();
}
}  // namespace

void *bth_cluster_stage_new(unsigned k, uint32_t max_allele_length, float cnv_threshold) { return new ClusterStage{k, max_allele_length, cnv_threshold, {}, nullptr, {}}; }
void bth_cluster_stage_free(void *h) { delete (ClusterStage *)h; }
int bth_cluster_stage_add_sequence(void *h, const char *name, const char *seq, unsigned long long len, int is_decoy, char *err, unsigned err_len) {
    try {
        ((ClusterStage *)h)->chromosomes.addSequence(name, std::string(seq, seq + len), is_decoy != 0);
        return 0;
    } catch (const std::exception &e) {
        return stage_error(e, err, err_len);
    }
}
// the candidate variants: VCF text (vcf_len > 0) or the name of a .vcf / .vcf.gz file (vcf_len == 0)
int bth_cluster_stage_set_variants(void *h, const char *vcf, unsigned long long vcf_len, char *err, unsigned err_len) {
    auto *st = (ClusterStage *)h;
    try {
        std::string text = vcf_len ? std::string(vcf, vcf + vcf_len) : VariantFileParser::readVariantFile(vcf);
        st->parser.reset(new VariantFileParser(std::move(text), st->k, st->max_allele_length, st->cnv_threshold));
        return 0;
    } catch (const std::exception &e) {
        return stage_error(e, err, err_len);
    }
}
// next unit: 1 = the file is exhausted, 0 = more units follow, -1 = error.  The unit's groups are ordered as main.cpp:247 orders them.
int bth_cluster_stage_next_unit(void *h, uint32_t min_unit_variants, char *err, unsigned err_len) {
    auto *st = (ClusterStage *)h;
    try {
        if (!st->parser) throw std::runtime_error("no variants set");
        st->unit.clear();
        const bool done = st->parser->constructVariantClusterGroups(&st->unit, min_unit_variants, st->chromosomes);
        std::sort(st->unit.begin(), st->unit.end(), ClusterGroupCompare);
        return done ? 1 : 0;
    } catch (const std::exception &e) {
        return stage_error(e, err, err_len);
    }
}
// text views (return the size needed; nothing is written when cap is too small): 0 groups of the last unit, 1 intercluster regions in
// their current order (file order until bth_cluster_stage_sort_regions), 3 counters
unsigned long long bth_cluster_stage_dump(void *h, int what, char *buf, unsigned long long cap) {
    auto *st = (ClusterStage *)h;
    if (what == 0) return copy_out(dumpClusterGroups(st->unit), buf, cap);
    if (!st->parser) return 0;
    if (what == 1) return copy_out(st->parser->interclusterRegionsText(), buf, cap);
    std::ostringstream os;
    os << "total=" << st->parser->getNumberOfVariants() << " parsed=" << st->parser->numParsedVariants() << " clusters=" << st->parser->numVariantClusters()
       << " groups=" << st->parser->numVariantClusterGroups() << " region_length=" << st->parser->getInterclusterRegionLength() << " alleles=";
    for (size_t i = 0; i < st->parser->alleleTypeCounter().size(); i++) os << (i ? "," : "") << st->parser->alleleTypeCounter()[i];
    os << " types=";
    for (size_t i = 0; i < st->parser->variantTypeCounter().size(); i++) os << (i ? "," : "") << st->parser->variantTypeCounter()[i];
    os << "\n";
    return copy_out(os.str(), buf, cap);
}
// VariantFileParser::sortInterclusterRegions (main.cpp:306): by length, descending; call once, after the last unit
void bth_cluster_stage_sort_regions(void *h) {
    auto *st = (ClusterStage *)h;
    if (st->parser) st->parser->sortInterclusterRegions();
}
// out[0] = groups of the last unit; out[1 + g] = clusters of group g (capacity: cap entries)
void bth_cluster_stage_unit_sizes(void *h, uint32_t *out, unsigned cap) {
    auto *st = (ClusterStage *)h;
    if (cap) out[0] = (uint32_t)st->unit.size();
    for (size_t g = 0; g < st->unit.size() && g + 1 < cap; g++) out[1 + g] = (uint32_t)st->unit[g].clusters.size();
}
// the graph of one cluster of the last unit (VariantClusterGraph.hpp; read it with bth_graph_sizes / bth_graph_fetch, release it with bth_graph_free)
void *bth_cluster_stage_graph(void *h, uint32_t group, uint32_t vertex) {
    auto *st = (ClusterStage *)h;
    try {
        const VariantCluster &c = st->unit.at(group).clusters.at(vertex);
        return new VariantClusterGraph(c, st->chromosomes.sequence((size_t)st->chromosomes.find(c.chrom_name)), st->k);
    } catch (...) {
        return nullptr;
    }
}


// ---- GenotypeWriter over the clusters of a cluster stage ----
namespace {
struct WriterHandle {
    ClusterStage *stage;
    std::unique_ptr<GenotypeWriter> writer;
};
}  // namespace
// sample_names: tab-separated
void *bth_writer_new(void *stage, const char *sample_names) {
    auto *st = (ClusterStage *)stage;
    std::vector<std::string> names;
    std::stringstream ss(sample_names);
    for (std::string n; std::getline(ss, n, '\t');) names.push_back(n);
    auto *w = new WriterHandle{st, nullptr};
    w->writer.reset(new GenotypeWriter(names, st->chromosomes));
    return w;
}
void bth_writer_free(void *h) { delete (WriterHandle *)h; }
// every variant of cluster (group, vertex) of the stage's last unit, genotyped from the sampler's results of that cluster (the arrays of
// bth_cluster_genotypes); H = number of haplotype candidates.  0 on success.
int bth_writer_add_cluster(void *h, uint32_t group, uint32_t vertex, unsigned S, unsigned H, const uint16_t *hap_allele, unsigned long long num_diplotypes, const uint16_t *h1,
                           const uint16_t *h2, const uint32_t *freq, const double *stats, const uint8_t *ploidy, float min_gpp, float min_kmers, const float *min_fraction, char *err,
                           unsigned err_len) {
    auto *w = (WriterHandle *)h;
    try {
        const ClusterGroup &g = w->stage->unit.at(group);
        const VariantCluster &c = g.clusters.at(vertex);
        const std::vector<VariantInfo> info = variantClusterInfo(c);
        std::vector<uint16_t> var_num_alleles;
        std::vector<uint8_t> var_has_dependency;
        for (auto &vi : info) {
            var_num_alleles.push_back(vi.numberOfAlleles());
            var_has_dependency.push_back(vi.has_dependency ? 1 : 0);
        }
        ClusterResults r;
        r.S = S; r.H = H; r.V = (uint32_t)info.size();
        r.hap_allele = hap_allele; r.var_num_alleles = var_num_alleles.data(); r.var_has_dependency = var_has_dependency.data();
        r.num_diplotypes = num_diplotypes; r.h1 = h1; r.h2 = h2; r.freq = freq; r.stats = stats; r.ploidy = ploidy;
        Filters f;
        f.min_genotype_posterior = min_gpp;
        f.min_number_of_kmers = min_kmers;
        f.min_fraction_observed_kmers.assign(min_fraction, min_fraction + S);
        const auto res = getGenotypes(r, f);
        ClusterAnnotation where{c.chrom_name, (uint32_t)info.size(), variantClusterRegion(c.chrom_name, info), (uint32_t)g.clusters.size(), g.region(), H};
        for (size_t v = 0; v < info.size(); v++) w->writer->addGenotypes(where, info[v], res[v], formatSampleColumns(r, (uint32_t)v, res[v]));
        return 0;
    } catch (const std::exception &e) {
        return stage_error(e, err, err_len);
    }
}
// the VCF (header + sorted lines); returns the size needed
unsigned long long bth_writer_text(void *h, const char *genome_filename, const char *graph_options_header, const char *genotype_options_header, char *buf, unsigned long long cap) {
    return copy_out(((WriterHandle *)h)->writer->vcfText(genome_filename, graph_options_header, genotype_options_header), buf, cap);
}
// writes <output_prefix>.vcf[.gz]; returns the number of genotyped variants or -1
long long bth_writer_finalise(void *h, const char *output_prefix, int gzip_output, const char *genome_filename, const char *graph_options_header,
                              const char *genotype_options_header, char *err, unsigned err_len) {
    try {
        return ((WriterHandle *)h)->writer->finalise(output_prefix, gzip_output != 0, genome_filename, graph_options_header, genotype_options_header);
    } catch (const std::exception &e) {
        return stage_error(e, err, err_len);
    }
}


// KmerCounter::parseSampleKmers for one sample from a KMC database on disk; returns the number of Bloom hits, or -1 (error text in *err)
long long bth_parse_sample_kmers(void *ctx, const char *kmc_prefix, void *path_bloom, void *table, unsigned sample_idx, unsigned long long chunk_records, char *err,
                                 unsigned err_len) {
    try {
        KmcFile db(kmc_prefix);
        return (long long)parseSampleKmers((bt_ctx *)ctx, db, (bt_bloom *)path_bloom, (bt_table *)table, sample_idx, chunk_records ? chunk_records : (1ull << 24));
    } catch (const std::exception &e) {
        if (err && err_len) {
            std::strncpy(err, e.what(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return -1;
    }
}
// header fields of a KMC database: out[0..5] = k, mode, counter_size, lut_prefix_length, total_kmers, record_size; 0 on success
int bth_kmc_info(const char *kmc_prefix, unsigned long long *out) {
    try {
        KmcFile db(kmc_prefix);
        out[0] = db.kmer_length; out[1] = db.mode; out[2] = db.counter_size; out[3] = db.lut_prefix_length; out[4] = db.total_kmers; out[5] = db.record_size();
        return 0;
    } catch (...) {
        return 1;
    }
}


// the genotype-derived output columns of every variant of one cluster, one line per variant ("<stats columns><sample columns>\n");
// returns the number of bytes needed (call with out = NULL first)
long long bth_cluster_output_columns(unsigned S, unsigned H, unsigned V, const uint16_t *hap_allele, const uint16_t *var_num_alleles, const uint8_t *var_has_dependency,
                                     unsigned long long num_diplotypes, const uint16_t *h1, const uint16_t *h2, const uint32_t *freq, const double *stats,
                                     const uint8_t *ploidy, float min_gpp, float min_kmers, const float *min_fraction, char *out, unsigned long long out_len) {
    try {
        ClusterResults r;
        r.S = S; r.H = H; r.V = V;
        r.hap_allele = hap_allele; r.var_num_alleles = var_num_alleles; r.var_has_dependency = var_has_dependency;
        r.num_diplotypes = num_diplotypes; r.h1 = h1; r.h2 = h2; r.freq = freq; r.stats = stats; r.ploidy = ploidy;
        Filters f;
        f.min_genotype_posterior = min_gpp;
        f.min_number_of_kmers = min_kmers;
        f.min_fraction_observed_kmers.assign(min_fraction, min_fraction + S);
        const auto res = getGenotypes(r, f);
        std::string all;
        for (unsigned v = 0; v < V; v++) all += formatVariantStatsColumns(res[v]) + formatSampleColumns(r, v, res[v]) + "\n";
        if (out && out_len >= all.size()) std::memcpy(out, all.data(), all.size());
        return (long long)all.size();
    } catch (...) {
        return -1;
    }
}


// The genotype collection of a whole launch on `threads` host threads (what `bayesTyper genotype -p` does per launch, main.cpp: getGenotypes + the
// formatted sample columns of every variant of every cluster; no VCF coordinates, so the synthetic batches of bench.py can be timed): clusters in batch
// order, group_cluster_off[G + 1] maps groups to their clusters, group_ploidy[G][S].  Returns the number of output bytes formatted (a checksum of the
// lengths: the work cannot be optimised away) or -1.
long long bth_batch_output_columns(unsigned S, unsigned long long G, const uint32_t *group_cluster_off, const uint8_t *group_ploidy, const uint32_t *num_haplotypes,
                                   const uint32_t *num_variants, const uint16_t *hap_allele, const uint16_t *var_num_alleles, const uint8_t *var_has_dependency,
                                   const uint64_t *dip_off, const uint16_t *h1, const uint16_t *h2, const uint32_t *freq, const uint64_t *cell_off, const double *stats, float min_gpp,
                                   float min_kmers, const float *min_fraction, unsigned threads) {
    try {
        const uint64_t C = group_cluster_off[G];
        std::vector<uint64_t> hapvar_off(C + 1, 0), var_base(C + 1, 0);
        for (uint64_t c = 0; c < C; c++) {
            hapvar_off[c + 1] = hapvar_off[c] + (uint64_t)num_haplotypes[c] * num_variants[c];
            var_base[c + 1] = var_base[c] + num_variants[c];
        }
        Filters f;
        f.min_genotype_posterior = min_gpp;
        f.min_number_of_kmers = min_kmers;
        f.min_fraction_observed_kmers.assign(min_fraction, min_fraction + S);
        const unsigned T = clampThreads(threads);
        std::vector<long long> bytes(T, 0);
        parallelFor((size_t)G, T, [&](size_t g0, size_t g1, unsigned part) {
            long long n = 0;
            for (size_t g = g0; g < g1; g++)
                for (uint64_t c = group_cluster_off[g]; c < group_cluster_off[g + 1]; c++) {
                    ClusterResults r;
                    r.S = S; r.H = num_haplotypes[c]; r.V = num_variants[c];
                    r.hap_allele = hap_allele + hapvar_off[c]; r.var_num_alleles = var_num_alleles + var_base[c]; r.var_has_dependency = var_has_dependency + var_base[c];
                    r.num_diplotypes = dip_off[c + 1] - dip_off[c]; r.h1 = h1 + dip_off[c]; r.h2 = h2 + dip_off[c]; r.freq = freq + dip_off[c] * S;
                    r.stats = stats + cell_off[c] * 12; r.ploidy = group_ploidy + g * S;
                    const auto res = getGenotypes(r, f);
                    for (unsigned v = 0; v < r.V; v++) n += (long long)(formatVariantStatsColumns(res[v]).size() + formatSampleColumns(r, v, res[v]).size());
                }
            bytes[part] = n;
        });
        long long total = 0;
        for (long long b : bytes) total += b;
        return total;
    } catch (...) {
        return -1;
    }
}


// ---- pieces of the command lines (Options / Sample / ChromosomePloidy / InferenceUnit / KmerHashOrder), for the CPU tests ----
uint64_t bth_bitset_hash(uint64_t lo, uint64_t hi, unsigned kmer_size) { return bitsetHash(lo, hi, kmer_size); }
void bth_hybrid_hash_order(const uint64_t *kmers, uint64_t n, unsigned kmer_size, unsigned seed, uint64_t root_hash_size, uint32_t *order_out) {
    const std::vector<uint32_t> o = hybridHashShuffledOrder(kmers, n, kmer_size, seed, root_hash_size);
    std::copy(o.begin(), o.end(), order_out);
}
// CountAllocation: counts of a (sample, count) list merged with those of a second list -> out[S*256]
void bth_count_allocation(unsigned short S, const unsigned short *s1, const unsigned char *c1, uint64_t n1, const unsigned short *s2, const unsigned char *c2, uint64_t n2,
                          unsigned long *out) {
    CountAllocation a(S), b(S);
    for (uint64_t i = 0; i < n1; i++) a.addCount(s1[i], c1[i]);
    for (uint64_t i = 0; i < n2; i++) b.addCount(s2[i], c2[i]);
    a.mergeInCountAllocations(b);
    for (unsigned short s = 0; s < S; s++)
        for (unsigned c = 0; c < 256; c++) out[(size_t)s * 256 + c] = a.getCounts()[s][c];
}
// unit file round trip: the cluster stage's last unit (every cluster with a few constant best paths over its graph's vertices)
// written to `filename`, read back, and dumped as text; returns the text's size
unsigned long long bth_unit_roundtrip(void *stage, const char *filename, uint32_t num_paths, char *buf, unsigned long long cap, char *err, unsigned err_len) {
    auto *st = (ClusterStage *)stage;
    try {
        InferenceUnit u;
        u.index = 7;
        u.cluster_options_header = "##BayesTyperOptions=test\n";
        u.variant_cluster_groups = st->unit;
        u.num_variants = 123;
        u.num_variant_clusters = 45;
        u.num_path_kmers = 1ull << 40;
        const UnitGraphs ug(u, st->chromosomes, st->k);
        u.best_paths.resize(u.variant_cluster_groups.size());
        for (size_t c = 0; c < ug.graphs.size(); c++) {
            auto &dst = u.best_paths[ug.cluster_group[c]];
            dst.resize(u.variant_cluster_groups[ug.cluster_group[c]].clusters.size());
            dst[ug.cluster_vertex[c]].assign(num_paths + c % 3, std::vector<uint8_t>(ug.graphs[c].vertices.size(), (uint8_t)(c & 1)));
        }
        u.write(filename);
        const InferenceUnit r = InferenceUnit::read(filename);
        std::ostringstream os;
        os << r.index << " " << r.cluster_options_header << r.num_variants << " " << r.num_variant_clusters << " " << r.num_path_kmers << "\n" << dumpClusterGroups(r.variant_cluster_groups);
        bool same = dumpClusterGroups(r.variant_cluster_groups) == dumpClusterGroups(u.variant_cluster_groups) && r.best_paths == u.best_paths;
        for (size_t g = 0; same && g < u.variant_cluster_groups.size(); g++)
            for (size_t v = 0; same && v < u.variant_cluster_groups[g].clusters.size(); v++) {
                const VariantCluster &a = u.variant_cluster_groups[g].clusters[v], &b = r.variant_cluster_groups[g].clusters[v];
                same = a.variants.size() == b.variants.size();
                for (auto ia = a.variants.begin(), ib = b.variants.begin(); same && ia != a.variants.end(); ++ia, ++ib) {
                    same = ia->first == ib->first && ia->second.id == ib->second.id && ia->second.has_dependency == ib->second.has_dependency && ia->second.type == ib->second.type &&
                           ia->second.num_redundant_nucleotides == ib->second.num_redundant_nucleotides && ia->second.alt_alleles.size() == ib->second.alt_alleles.size();
                    for (size_t i = 0; same && i < ia->second.alt_alleles.size(); i++)
                        same = ia->second.alt_alleles[i].ref_length == ib->second.alt_alleles[i].ref_length && ia->second.alt_alleles[i].sequence == ib->second.alt_alleles[i].sequence &&
                               ia->second.alt_alleles[i].aco_att == ib->second.alt_alleles[i].aco_att;
                }
            }
        os << (same ? "IDENTICAL\n" : "DIFFERENT\n");
        return copy_out(os.str(), buf, cap);
    } catch (const std::exception &e) {
        stage_error(e, err, err_len);
        return 0;
    }
}
// ChromosomePloidy of the stage's genome for samples given as a gender string ("FM.."): rows "<chromosome>\t<female>\t<male>\t<per-sample ploidies>"
unsigned long long bth_chromosome_ploidy(void *stage, const char *ploidy_filename, const char *genders, char *buf, unsigned long long cap, char *err, unsigned err_len) {
    auto *st = (ClusterStage *)stage;
    try {
        std::vector<Sample> samples;
        for (const char *g = genders; *g; g++) samples.emplace_back(std::string("s") + std::to_string(samples.size()) + "\t" + *g + "\tprefix");
        const ChromosomePloidy cp(ploidy_filename, st->chromosomes, samples);
        std::ostringstream os;
        for (size_t i = 0; i < st->chromosomes.size(); i++) {
            const std::string &name = st->chromosomes.name(i);
            if (st->chromosomes.isDecoy(name)) continue;
            os << name << "\t" << (int)cp.getGenderPloidy(name)[0] << "\t" << (int)cp.getGenderPloidy(name)[1] << "\t";
            for (uint8_t p : cp.getSamplePloidy(name)) os << (int)p;
            os << "\n";
        }
        return copy_out(os.str(), buf, cap);
    } catch (const std::exception &e) {
        stage_error(e, err, err_len);
        return 0;
    }
}

}  // extern "C"
