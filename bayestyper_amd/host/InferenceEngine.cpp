#include "InferenceEngine.hpp"
#include "StageTimes.hpp"

#include <algorithm>
#include <fstream>
#include <future>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "Options.hpp"

namespace bthost {

NoiseGroupSelector::NoiseGroupSelector(const uint32_t *clusters_per_group, const uint32_t *variants_per_group, uint32_t num_groups, unsigned prng_seed,
                                       uint32_t variants_batch_size)
    : variants(variants_per_group, variants_per_group + num_groups), prng(prng_seed), batch_size(variants_batch_size) {
    for (uint32_t g = 0; g < num_groups; g++)
        if (clusters_per_group[g] == 1) noise_group_indices.push_back(g);
}

std::vector<uint32_t> NoiseGroupSelector::nextChain() {
    std::shuffle(noise_group_indices.begin(), noise_group_indices.end(), prng);
    size_t end = 0;
    num_noise_variants = 0;
    while (num_noise_variants < batch_size && end < noise_group_indices.size()) num_noise_variants += variants[noise_group_indices[end++]];
    std::sort(noise_group_indices.begin(), noise_group_indices.begin() + end);
    return std::vector<uint32_t>(noise_group_indices.begin(), noise_group_indices.begin() + end);
}

std::string noiseParameterHeader(const std::vector<std::string> &sample_names) {
    std::string s = "Chain\tIteration";
    for (auto &n : sample_names) s += "\t" + n;
    return s + "\n";
}

std::string noiseParameterRow(unsigned chain, unsigned iteration, const std::vector<double> &rates) {
    std::ostringstream os;
    os << chain << "\t" << iteration;
    for (double r : rates) os << "\t" << r;
    os << "\n";
    return os.str();
}

// ---- the drivers ------------------------------------------------------------------------------------------------------------------

namespace {
void check(int rc, const char *what) {
    if (rc != BT_OK) throw std::runtime_error(std::string(what) + ": " + bt_last_error());
}
}  // namespace

// one bt_gibbs over a batch of groups: the product's sampler
namespace {
struct GpuSampler : GibbsSampler {
    bt_ctx *ctx;
    bt_gibbs *g = nullptr;
    uint64_t *d_hist = nullptr;
    uint32_t S;
    GpuSampler(bt_ctx *ctx_in, const bt_gibbs_params &p, const GibbsBatchData &batch) : ctx(ctx_in), S(p.num_samples) {
        if (!ctx) throw std::runtime_error("InferenceEngine: no GPU context (there is no CPU path)");
        const bt_gibbs_batch b = batch.view();
        check(bt_gibbs_create(ctx, &p, &b, &g), "bt_gibbs_create");
    }
    // over a selection of the groups of a batch that is on the device already
    GpuSampler(bt_ctx *ctx_in, const bt_gibbs_params &p, bt_gibbs_source *source, const std::vector<uint32_t> &ids) : ctx(ctx_in), S(p.num_samples) {
        if (!ctx) throw std::runtime_error("InferenceEngine: no GPU context (there is no CPU path)");
        check(bt_gibbs_create_from_source(source, ctx, &p, ids.data(), (uint32_t)ids.size(), &g), "bt_gibbs_create_from_source");
    }
    ~GpuSampler() override {
        if (resident_chain) bt_gibbs_noise_chain_end(g);
        if (noise_model) bt_noise_model_destroy(noise_model);
        if (d_hist) bt_free(ctx, d_hist);
        bt_gibbs_destroy(g);
    }
    void setLut(const double *genomic, const double *noise) override { check(bt_gibbs_set_lut(g, genomic, noise), "bt_gibbs_set_lut"); }
    void setNoiseLut(const double *noise) override { check(bt_gibbs_set_noise_lut(g, noise), "bt_gibbs_set_noise_lut"); }
    void initChain(uint32_t chain) override { check(bt_gibbs_init_chain(g, chain), "bt_gibbs_init_chain"); }
    bool resetGroups() override {
        check(bt_gibbs_reset_groups(g), "bt_gibbs_reset_groups");
        return true;
    }
    void sweep(uint32_t n, bool collect) override { check(bt_gibbs_sweep(g, n, collect ? 1 : 0), "bt_gibbs_sweep"); }
    void run() override { check(bt_gibbs_run(g), "bt_gibbs_run"); }
    void sync() override {
        if (getenv("BT_STAGE_TIMES")) check(bt_sync(ctx), "bt_sync");
    }
    std::vector<uint64_t> noiseCounts() override {   // VariantClusterGroup::getNoiseCounts of every group + clearGenotyperCache (InferenceEngine.cpp:90-92)
        if (!d_hist) check(bt_malloc(ctx, (size_t)S * 256 * 8, (void **)&d_hist), "bt_malloc");
        check(bt_gibbs_noise_counts(g, d_hist, 1), "bt_gibbs_noise_counts");
        std::vector<uint64_t> h((size_t)S * 256);
        check(bt_sync(ctx), "bt_sync");
        check(bt_memcpy_d2h(ctx, h.data(), d_hist, h.size() * 8), "bt_memcpy_d2h");
        return h;
    }
    bt_noise_model *noise_model = nullptr;
    bool noiseChain(CountDistribution *cd, uint32_t n_iterations, uint32_t first_collect, const DeviceReducer &reduce, std::vector<double> *rows) override {
        // Default: the per-iteration host loop (bt_gibbs_noise_iteration: one synchronisation per iteration) — the rates are drawn by libstdc++'s own
        // gamma distribution and the Poisson table is built by glibc, bit for bit the reference's, whatever the number of ranks and the transport.
        // BT_NOISE_ON_DEVICE=1 runs a whole chain on the device instead (ocml log / pow / lgamma: last-bit differences in the rates, which a rejection
        // test of a later draw can amplify; every rank of a run must then take this path).
        if (!getenv("BT_NOISE_ON_DEVICE")) return false;
        if (!noise_model) {
            std::vector<float> prior;
            for (auto &p : cd->noiseRatePriors()) {
                prior.push_back(p.first);
                prior.push_back(p.second);
            }
            check(bt_noise_model_create(ctx, S, prior.data(), &noise_model), "bt_noise_model_create");
        }
        bt_noise_rng rng;
        cd->exportGenerator(rng.mt, &rng.mt_pos, &rng.saved_available, &rng.saved);
        check(bt_noise_model_set_rng(noise_model, &rng), "bt_noise_model_set_rng");
        rows->assign((size_t)n_iterations * S, 0.0);
        struct Hook {
            const DeviceReducer *r;
            static int call(void *user, uint64_t *d_hist, uint64_t n) {
                try {
                    (*((Hook *)user)->r)(d_hist, (size_t)n);
                } catch (...) {
                    return 1;
                }
                return 0;
            }
        } hook{&reduce};
        check(bt_gibbs_noise_chain(g, noise_model, n_iterations, first_collect, reduce ? &Hook::call : nullptr, &hook, rows->data()), "bt_gibbs_noise_chain");
        check(bt_noise_model_get_rng(noise_model, &rng), "bt_noise_model_get_rng");
        cd->importGenerator(rng.mt, rng.mt_pos, rng.saved_available, rng.saved);
        cd->setNoiseRates(std::vector<double>(rows->end() - S, rows->end()));
        return true;
    }
    uint64_t deviceBytes() override {
        uint64_t b = 0;
        check(bt_gibbs_device_bytes(g, &b), "bt_gibbs_device_bytes");
        return b;
    }
    bool resident_chain = false;
    bool beginResidentChain(uint32_t n_iterations, uint32_t first_collect) override {
        int resident = 0;
        check(bt_gibbs_noise_chain_begin(g, n_iterations, first_collect, &resident), "bt_gibbs_noise_chain_begin");
        resident_chain = resident != 0;
        return resident_chain;
    }
    void endResidentChain() override {
        if (!resident_chain) return;
        resident_chain = false;
        check(bt_gibbs_noise_chain_end(g), "bt_gibbs_noise_chain_end");
    }
    std::vector<uint64_t> noiseIteration(const double *noise, bool collect) override {
        std::vector<uint64_t> h((size_t)S * 256);
        if (resident_chain) check(bt_gibbs_noise_chain_step(g, noise, h.data()), "bt_gibbs_noise_chain_step");   // (the chain knows from which iteration on it collects)
        else check(bt_gibbs_noise_iteration(g, noise, collect ? 1 : 0, h.data()), "bt_gibbs_noise_iteration");
        return h;
    }
    bool resultWords(const uint32_t **d_words, uint64_t *num_words) override {
        check(bt_gibbs_result_words(g, d_words, num_words), "bt_gibbs_result_words");
        return true;
    }
    BatchResults results(uint32_t num_clusters) override {
        BatchResults r;
        uint64_t nd = 0, nc = 0;
        check(bt_gibbs_result_sizes(g, &nd, &nc), "bt_gibbs_result_sizes");
        r.dip_off.resize(num_clusters + 1);
        r.cell_off.resize(num_clusters + 1);
        r.h1.resize(std::max<uint64_t>(nd, 1));
        r.h2.resize(std::max<uint64_t>(nd, 1));
        r.freq.resize(std::max<uint64_t>(nd * S, 1));
        r.stats.resize(std::max<uint64_t>(nc * 12, 1));
        check(bt_gibbs_result_fetch(g, r.dip_off.data(), r.h1.data(), r.h2.data(), r.freq.data(), r.cell_off.data(), r.stats.data()), "bt_gibbs_result_fetch");
        return r;
    }
};
}  // namespace

std::unique_ptr<GibbsSampler> InferenceEngine::newSampler(uint32_t noise_seeding, const GibbsBatchData &batch, bt_ctx *on_ctx) {
    const bt_gibbs_params p = params(noise_seeding);
    if (make_sampler) return make_sampler(p, batch);
    return std::unique_ptr<GibbsSampler>(new GpuSampler(on_ctx ? on_ctx : ctx, p, batch));
}

void InferenceEngine::logRow(std::ostream &out, unsigned chain, unsigned iteration, const std::vector<double> &rates) {
    out << noiseParameterRow(chain, iteration, rates);
    if (record_rows) {
        noise_rows.push_back((double)chain);
        noise_rows.push_back((double)iteration);
        noise_rows.insert(noise_rows.end(), rates.begin(), rates.end());
    }
}

InferenceEngine::InferenceEngine(bt_ctx *ctx_in, std::vector<uint8_t> gender_in, std::vector<std::string> sample_names_in, const GibbsOptions &options, HistReducer reduce)
    : ctx(ctx_in), gender(std::move(gender_in)), sample_names(std::move(sample_names_in)), opt(options), reduce_hist(std::move(reduce)) {}

bt_gibbs_params InferenceEngine::params(uint32_t noise_seeding) const {
    bt_gibbs_params p{};
    p.num_samples = (uint32_t)gender.size();
    p.seed = opt.seed;
    p.num_chains = opt.chains;
    p.burn_in = opt.burn_in;
    p.num_iterations = opt.samples;
    p.kmer_subsampling_rate = opt.kmer_subsampling_rate;
    p.max_haplotype_variant_kmers = opt.max_haplotype_variant_kmers;
    p.noise_seeding = noise_seeding;
    p.gender = gender.data();
    return p;
}

// sampleGenotypesCallback + sampleNoiseParameters of one iteration (InferenceEngine.cpp:77-98, CountDistribution.cpp:173-186)
void InferenceEngine::iteration(Sampler *sampler, CountDistribution *cd, bool collect) {
    const size_t S = gender.size();
    std::vector<uint64_t> hist(S * 256, 0);
    // the table drawn at the end of the previous iteration travels with this iteration's sweep (pending_noise); the first iteration of a
    // chain finds the table set by the driver
    {
        StageScope stage("  noise iterations: sweep + noise counts (device, one synchronisation each)");
        if (sampler) hist = sampler->noiseIteration(pending_noise ? cd->noiseTable().data() : nullptr, collect);   // (a rank without groups in this chain still takes part in the reduction)
    }
    if (reduce_hist) reduce_hist(hist.data(), hist.size());
    StageScope stage("  noise iterations: rates + Poisson table (host)");
    CountAllocation counts((unsigned short)S);
    for (size_t s = 0; s < S; s++)
        for (size_t c = 0; c < 256; c++) counts.counts()[s][c] = hist[s * 256 + c];
    cd->sampleNoiseParameters(counts);
    pending_noise = true;
}

void InferenceEngine::runNoiseChain(Sampler *sampler, CountDistribution *cd, uint32_t chain, uint32_t first_collect_iteration, std::ostream &out,
                                    const std::function<void(uint32_t, const std::vector<double> &)> &each) {
    const uint32_t n = opt.burn_in + opt.samples;
    const size_t S = gender.size();
    // On the device: a single rank, or ranks whose histograms are reduced on the device and that ALL hold a sampler in this chain (a rank
    // without one would have to draw on the host, and host and device arithmetic may differ in the last bit — the ranks' generators must not)
    bool on_device = sampler != nullptr && (!reduce_hist || device_reduce);
    if (reduce_hist && device_reduce) {
        uint64_t have = sampler ? 1 : 0;
        uint64_t all[2] = {have, 1};
        reduce_hist(all, 2);
        on_device = all[0] == all[1];
    }
    std::vector<double> rows;
    if (on_device && sampler->noiseChain(cd, n, first_collect_iteration - 1, reduce_hist ? device_reduce : GibbsSampler::DeviceReducer(), &rows)) {
        for (uint32_t it = 1; it <= n; it++) {
            const std::vector<double> rates(rows.begin() + (size_t)(it - 1) * S, rows.begin() + (size_t)it * S);
            logRow(out, chain + 1, it, rates);
            if (each) each(it, rates);
        }
        pending_noise = false;   // (the sampler holds the table of the last rates)
        return;
    }
    // The iterations: with the sampler's groups resident for the whole chain when it can (one launch per chain; the histogram and the table of an
    // iteration cross through pinned memory), else a sweep + tally launch and a synchronisation per iteration.  The host side of an iteration — the
    // reduction over the ranks, the exact draws — is the same either way.
    struct ResidentGuard {   // (an exception on the host side must not leave the launch waiting for a table)
        Sampler *s;
        ~ResidentGuard() {
            if (s) try { s->endResidentChain(); } catch (...) {}
        }
    } guard{nullptr};
    {
        StageScope stage("  noise chains: resident launch set up");
        const bool reducer_blocks = reduce_hist && reduce_on_stream;   // (host <-> device deadlock until the chain's deadline)
        if (sampler && pending_noise == false && !reducer_blocks && sampler->beginResidentChain(n, first_collect_iteration - 1)) guard.s = sampler;
    }
    for (uint32_t it = 1; it <= n; it++) {
        iteration(sampler, cd, it >= first_collect_iteration);
        logRow(out, chain + 1, it, cd->getNoiseRates());
        if (each) each(it, cd->getNoiseRates());
    }
    if (guard.s) {
        StageScope stage("  noise chains: resident launch ended");
        guard.s = nullptr;
        sampler->endResidentChain();   // (errors of a completed chain are reported)
    }
}

void InferenceEngine::estimateNoise(CountDistribution *cd, const GibbsBatchData &unit, const std::string &output_prefix, uint32_t variants_batch_size,
                                    const std::vector<uint32_t> *unit_clusters, const std::vector<uint32_t> *unit_variants) {
    if (!quiet)
        std::cout << "[" << getLocalTime() << "] Estimating noise model parameters using " << opt.chains << " parallel gibbs sampling chains each with " << (opt.burn_in + opt.samples)
                  << " iterations (" << opt.burn_in << " burn-in) ..." << std::endl;
    const size_t S = gender.size();
    noise_rows.clear();
    // clusters / variants per group of the whole unit (default: `unit` is the whole unit)
    std::vector<uint32_t> clusters, variants;
    if (unit_clusters && unit_variants) {
        clusters = *unit_clusters;
        variants = *unit_variants;
    } else {
        for (uint32_t g = 0; g < unit.numGroups(); g++) {
            if (unit.group_index[g] != g) throw std::runtime_error("estimateNoise: pass the unit's group shape when the batch is a shard of the unit");
            clusters.push_back(unit.group_cluster_off[g + 1] - unit.group_cluster_off[g]);
            uint32_t nv = 0;
            for (uint32_t c = unit.group_cluster_off[g]; c < unit.group_cluster_off[g + 1]; c++) nv += unit.num_variants[c];
            variants.push_back(nv);
        }
    }
    NoiseGroupSelector selector(clusters.data(), variants.data(), (uint32_t)clusters.size(), opt.seed, variants_batch_size);
    std::vector<int64_t> local(clusters.size(), -1);   // unit-wide group index -> position in this rank's batch
    for (uint32_t g = 0; g < unit.numGroups(); g++) local[unit.group_index[g]] = g;
    std::ofstream out(output_prefix + ".txt");
    if (!out.is_open()) throw std::runtime_error("Unable to write file " + output_prefix + ".txt");
    out << noiseParameterHeader(sample_names);
    std::vector<double> mean(S, 0.0);
    // The helper's sampler is built on a second context of the same GPU (a stream of its own: a chain's launch stays on its stream until the chain ends, and
    // the construction's copies and kernels must not queue behind it); chains alternate between the two contexts.
    struct AltCtx {
        bt_ctx *c = nullptr;
        ~AltCtx() {
            if (c) bt_ctx_destroy(c);
        }
    } alt;   // (declared before the samplers that may live on it)
    if (!make_sampler && ctx && !getenv("BT_NOISE_SAMPLER_ON_MAIN_THREAD")) check(bt_ctx_clone(ctx, &alt.c), "bt_ctx_clone");
    // The unit's flat arrays go to the device ONCE; every chain's sampler is laid out from the per-cluster dimensions and filled on the device from them
    // (bt_gibbs_create_from_source) — no per-chain subset copy on the host, no per-chain upload.
    struct Source {
        bt_gibbs_source *s = nullptr;
        ~Source() {
            if (s) bt_gibbs_source_destroy(s);
        }
    } source;
    if (!make_sampler && ctx && unit.numGroups() && !getenv("BT_NOISE_NO_SOURCE")) {
        StageScope stage("  noise chains: the unit's clusters to the device (once)");
        const bt_gibbs_batch view = unit.view();
        check(bt_gibbs_source_create(ctx, (uint32_t)S, &view, &source.s), "bt_gibbs_source_create");
    }
    const bt_gibbs_params noise_params = params(1);
    auto sampler_over = [this, &unit, &source, &noise_params](const std::vector<uint32_t> &ids, bt_ctx *on_ctx) -> std::unique_ptr<Sampler> {
        if (source.s) return std::unique_ptr<Sampler>(new GpuSampler(on_ctx ? on_ctx : ctx, noise_params, source.s, ids));
        return newSampler(1, unit.take(ids), on_ctx);
    };
    std::unique_ptr<Sampler> sampler;
    std::vector<uint32_t> sampler_groups;
    // this rank's groups of the next chain (the selection depends on the selector's generator only, not on the chains' results)
    auto select = [&]() {
        std::vector<uint32_t> mine;
        for (uint32_t g : selector.nextChain())
            if (local[g] >= 0) mine.push_back((uint32_t)local[g]);
        return mine;
    };
    std::vector<uint32_t> mine = select();
    // what a helper thread prepares for the next chain while the current one runs: the subset copy, and — for the product's own (GPU) sampler, whose
    // construction only enqueues on the context's stream — the sampler itself; a caller-supplied sampler factory is only ever called from this thread
    struct Prepared {
        std::vector<uint32_t> groups;   // the next chain's selection (the shuffle + sort of the unit's groups: 6 ms per chain at chr20 size, off the calling thread)
        std::unique_ptr<Sampler> sampler;
        std::string why_not;   // the helper could not build the sampler (not enough free HBM next to the running chain's, or an error): the calling thread does, after freeing the previous one
    };
    std::future<Prepared> prepared;
    bt_ctx *sampler_ctx = ctx;   // the context the current chain's sampler lives on
    for (uint32_t chain = 0; chain < opt.chains; chain++) {
        // The genotypers of a chain are constructed with seed + (i+1)(chain+1) (:70) and deleted afterwards (resetGroupsCallback, :240-251).  A unit
        // with fewer variants than the batch size selects the same (sorted) groups in every chain: the previous chain's sampler then only has its
        // groups reset — the reference's own sequence.  Otherwise (a unit of 100 000 variants or more: another random subset per chain), and for a
        // sampler that cannot reset, a fresh sampler over a copy of the subset; the NEXT chain's subset copy and sampler are made by a helper thread
        // while this chain's iterations run (the twenty copies + constructions were 3.8 of the 7.9 s of this stage at chr20 size).
        Prepared ready;
        if (prepared.valid()) {
            StageScope stage("  noise chains: waiting for the helper thread (next chain's selection + sampler)");
            ready = prepared.get();
            mine = std::move(ready.groups);
        }
        const bool again = sampler && !mine.empty() && mine == sampler_groups && sampler->resetGroups();
        if (again) {
            sampler->setNoiseLut(cd->noiseTable().data());
            sampler->initChain(chain);
        } else {
            {
                StageScope stage("  noise chains: previous sampler released");
                sampler.reset();   // (before anything is built on this thread: two samplers' state at once is the helper's privilege, and only when it fits)
            }
            sampler_groups.clear();
            if (!mine.empty()) {
                StageScope stage("  noise chains: what the helper thread had not prepared (first chain: subset copy + sampler construction)");
                if (ready.sampler) sampler_ctx = sampler_ctx == ctx && alt.c ? alt.c : ctx;
                else ready.sampler = sampler_over(mine, sampler_ctx);
                sampler = std::move(ready.sampler);
                StageScope stage2("  noise chains: count tables to the sampler + chain start (enqueued)");
                sampler->setLut(cd->genomicTable().data(), cd->noiseTable().data());
                sampler->initChain(chain);
                sampler_groups = mine;
            }
        }
        if (chain + 1 < opt.chains) {
            bt_ctx *helper_ctx = alt.c ? (sampler_ctx == ctx ? alt.c : ctx) : nullptr;
            const uint64_t like = sampler ? sampler->deviceBytes() : 0;   // the next chain's sampler is over as many groups of the same unit: about as large
            // (the selector is touched by one thread at a time: this one before the first chain, then the helper of each chain, whose result is taken before the next is started)
            prepared = std::async(std::launch::async, [this, &source, &noise_params, &sampler_over, &select, current = sampler_groups, helper_ctx, like]() {
                Prepared r;
                r.groups = select();
                const std::vector<uint32_t> &next = r.groups;
                if (!helper_ctx || next.empty() || next == current) return r;
                StageScope stage("  noise chains (helper thread, overlapped): sampler construction");
                try {   // next to the running chain's sampler only when its state fits the free HBM with room to spare
                    uint64_t need = 0, total = 0, free_bytes = 0;
                    int num_cu = 0;
                    need = like + like / 4;
                    if (source.s && ((need == 0 && bt_gibbs_state_bytes_from_source(source.s, &noise_params, next.data(), (uint32_t)next.size(), &need) != BT_OK) ||
                                     bt_ctx_info(helper_ctx, &num_cu, &total, &free_bytes, nullptr, 0) != BT_OK))
                        r.why_not = bt_last_error();
                    else if (source.s && (double)need > 0.7 * (double)free_bytes) r.why_not = "sampler state does not fit next to the running chain's";
                    else r.sampler = sampler_over(next, helper_ctx);
                } catch (const std::exception &e) {
                    r.sampler.reset();
                    r.why_not = e.what();
                }
                return r;
            });
        }
        pending_noise = false;   // (the chain's sampler starts with the current table)
        logRow(out, chain + 1, 0, cd->getNoiseRates());
        runNoiseChain(sampler.get(), cd, chain, opt.burn_in + opt.samples + 1 /* never collects */, out, [&](uint32_t it, const std::vector<double> &rates) {
            if (opt.burn_in < it)
                for (size_t s = 0; s < S; s++) mean[s] += rates[s];
        });
        cd->resetNoiseRates();
    }
    sampler.reset();
    for (auto &m : mean) m /= (double)opt.samples * opt.chains;
    cd->setNoiseRates(mean);
    logRow(out, 0, 0, cd->getNoiseRates());
    low_variant_warning = selector.lastNumVariants() < variants_batch_size;
    if (low_variant_warning && !quiet) {
        std::cout << "\nWARNING: Low number of variants used for noise model parameter estimation (" << selector.lastNumVariants() << " < " << variants_batch_size << ")" << std::endl;
        std::cout << "WARNING: The noise estimates might be biased\n" << std::endl;
    }
    if (!quiet) std::cout << "[" << getLocalTime() << "] Wrote noise parameters to " << output_prefix << ".txt" << std::endl;
}

// the default schedule for a batch; a batch whose sampler state does not fit the GPU is run as two consecutive halves (groups are
// independent and keep their unit-wide index, so the split changes nothing but the peak memory; InferenceEngine.cpp:335-382 hands
// groups to its threads in batches the same way)
void InferenceEngine::runDefault(const GibbsBatchData &batch, const CountDistribution &cd, const Collector &collect) {
    if (opt.max_groups_per_launch && batch.numGroups() > opt.max_groups_per_launch) {   // consecutive group ranges of the unit, in order
        for (uint32_t a = 0; a < batch.numGroups(); a += opt.max_groups_per_launch) {
            std::vector<uint32_t> ids;
            for (uint32_t g = a; g < std::min<uint32_t>(a + opt.max_groups_per_launch, batch.numGroups()); g++) ids.push_back(g);
            runDefault(batch.take(ids), cd, collect);
        }
        return;
    }
    // The product's sampler: a unit larger than the GPU is genotyped as consecutive group ranges sized from the free HBM
    // (bt_gibbs_state_bytes lays a range out without allocating); groups are independent and keep their unit-wide index.
    if (!make_sampler && ctx && batch.numGroups() > 1) {
        const bt_gibbs_params p = params(0);
        const bt_gibbs_batch view = batch.view();
        uint64_t need = 0, total = 0, free_bytes = 0;
        int num_cu = 0;
        int have = bt_gibbs_state_bytes(ctx, &p, &view, &need) == BT_OK && bt_ctx_info(ctx, &num_cu, &total, &free_bytes, nullptr, 0) == BT_OK;
        if (const char *e = getenv("BT_GIBBS_FREE_BYTES")) free_bytes = strtoull(e, nullptr, 0);   // (tests: pretend a smaller GPU)
        if (have && free_bytes && (double)need > 0.9 * (double)free_bytes) {
            const uint32_t parts = (uint32_t)std::min<uint64_t>(batch.numGroups(), (uint64_t)((double)need / (0.75 * (double)free_bytes)) + 1);
            if (!quiet) std::cout << "[" << getLocalTime() << "] " << batch.numGroups() << " groups need " << need / 1000000000.0 << " GB of sampler state (" << free_bytes / 1000000000.0
                                  << " GB free): " << parts << " launches" << std::endl;
            for (uint32_t q = 0; q < parts; q++) {   // (a range that still does not fit is cut again by the recursion)
                std::vector<uint32_t> ids;
                for (uint32_t g = (uint32_t)((uint64_t)batch.numGroups() * q / parts); g < (uint32_t)((uint64_t)batch.numGroups() * (q + 1) / parts); g++) ids.push_back(g);
                if (!ids.empty()) runDefault(batch.take(ids), cd, collect);
            }
            return;
        }
    }
    std::unique_ptr<Sampler> sampler;
    std::unique_ptr<StageScope> stage(new StageScope("Gibbs: sampler construction (tiles)"));
    try {
        sampler = newSampler(0, batch);
    } catch (const std::runtime_error &e) {
        stage.reset();
        if (batch.numGroups() < 2 || std::string(e.what()).find("state pool") == std::string::npos) throw;
        std::vector<uint32_t> a, b;
        for (uint32_t g = 0; g < batch.numGroups(); g++) (g < batch.numGroups() / 2 ? a : b).push_back(g);
        runDefault(batch.take(a), cd, collect);
        runDefault(batch.take(b), cd, collect);
        return;
    }
    num_launches += 1;
    sampler->setLut(cd.genomicTable().data(), cd.noiseTable().data());
    stage.reset(new StageScope("Gibbs: sampling launch (20 x 350 sweeps)"));
    sampler->run();
    sampler->sync();
    stage.reset();
    handOver(sampler, batch, collect);
}

// a finished launch's collected samples -> the caller; the sampler is released (its HBM is free before the next one is built)
void InferenceEngine::handOver(std::unique_ptr<Sampler> &sampler, const GibbsBatchData &batch, const Collector &collect) {
    const uint32_t *d_words = nullptr;
    uint64_t num_words = 0;
    if (wire_collect) {
        StageScope stage("Gibbs: result string (device pack, stays on the device)");
        if (sampler->resultWords(&d_words, &num_words)) {
            wire_collect(batch, d_words, num_words);
            sampler.reset();
            return;
        }
    }
    std::unique_ptr<StageScope> stage(new StageScope("Gibbs: result fetch (device pack + copy)"));
    const BatchResults r = sampler->results(batch.numClusters());
    sampler.reset();
    stage.reset();
    collect(batch, r);
}

void InferenceEngine::estimateGenotypes(const GibbsBatchData &unit, const CountDistribution &cd, const Collector &collect) {
    uint64_t num_variants = 0;
    for (uint32_t v : unit.num_variants) num_variants += v;
    if (!quiet)
        std::cout << "[" << getLocalTime() << "] Estimating genotypes on " << num_variants << " variants using " << opt.chains << " parallel gibbs sampling chains each with "
                  << (opt.burn_in + opt.samples) << " iterations (" << opt.burn_in << " burn-in) ..." << std::endl;
    num_launches = 0;
    if (unit.numGroups()) runDefault(unit, cd, collect);
    if (!quiet) std::cout << "[" << getLocalTime() << "] Finished genotyping" << std::endl;
}

void InferenceEngine::estimateNoiseAndGenotypes(const GibbsBatchData &unit, CountDistribution *cd, const Collector &collect, const std::string &output_prefix) {
    uint64_t num_variants = 0;
    for (uint32_t v : unit.num_variants) num_variants += v;
    if (!quiet)
        std::cout << "[" << getLocalTime() << "] Estimating noise model parameters and genotypes on " << num_variants << " variants using " << opt.chains
                  << " parallel gibbs sampling chains each with " << (opt.burn_in + opt.samples) << " iterations (" << opt.burn_in << " burn-in) ..." << std::endl;
    noise_rows.clear();
    std::ofstream out(output_prefix + ".txt");
    if (!out.is_open()) throw std::runtime_error("Unable to write file " + output_prefix + ".txt");
    out << noiseParameterHeader(sample_names);
    std::unique_ptr<Sampler> sampler;
    if (unit.numGroups()) {
        sampler = newSampler(1, unit);
        sampler->setLut(cd->genomicTable().data(), cd->noiseTable().data());
    }
    for (uint32_t chain = 0; chain < opt.chains; chain++) {
        if (sampler) {
            sampler->setNoiseLut(cd->noiseTable().data());
            sampler->initChain(chain);
        }
        pending_noise = false;
        logRow(out, chain + 1, 0, cd->getNoiseRates());
        runNoiseChain(sampler.get(), cd, chain, opt.burn_in + 1, out, nullptr);
        cd->resetNoiseRates();
    }
    if (sampler) handOver(sampler, unit, collect);
    if (!quiet) std::cout << "[" << getLocalTime() << "] Wrote noise parameters to " << output_prefix << ".txt" << std::endl;
}

}  // namespace bthost
