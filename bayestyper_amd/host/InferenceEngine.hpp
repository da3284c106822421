// InferenceEngine (include/bayesTyper/InferenceEngine.hpp:60-62, src/bayesTyper/InferenceEngine.cpp): the three drivers of the
// genotyping stage over the C ABI of libbtgpu.so —
//   estimateNoise              (:135-276)  noise rates from single-cluster groups, chain by chain
//   estimateGenotypes          (:278-382)  default mode: the whole schedule of every group in one launch (or a few, see below)
//   estimateNoiseAndGenotypes  (:384-472)  --noise-genotyping
// In the two noise drivers an iteration is: one sweep of all (selected) groups on the GPU, their noise-count histograms added up
// (CountAllocation; across ranks: `reduce_hist`, one all-reduce of S x 256 counters), one gamma draw per sample from the run's
// CountDistribution generator on the host, the rebuilt noise table uploaded.  Every rank draws from an identically seeded generator,
// so all ranks hold the same rates without a broadcast.  (bayestyper_amd/host/inference_engine.py is the ctypes binding of THIS class
// for the tests and the bench: there is one implementation of the drivers.)
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/btgpu.h"
#include "CountDistribution.hpp"
#include "KmerCounter.hpp"

namespace bthost {

// InferenceEngine.cpp:141-151 (candidates: groups with exactly one variant cluster, in group order) and :172-189 (per chain:
// std::shuffle of the WHOLE candidate vector with one mt19937(seed) that lives across chains, then the shortest prefix with
// >= noise_variants_batch_size variants, sorted in place — the partly sorted vector is what the next chain shuffles).
class NoiseGroupSelector {
  public:
    static constexpr uint32_t noise_variants_batch_size = 100000;   // InferenceEngine.cpp:50
    NoiseGroupSelector(const uint32_t *clusters_per_group, const uint32_t *variants_per_group, uint32_t num_groups, unsigned prng_seed,
                       uint32_t variants_batch_size = noise_variants_batch_size);
    // group indices (ascending) the next chain runs on
    std::vector<uint32_t> nextChain();
    uint32_t numCandidates() const { return (uint32_t)noise_group_indices.size(); }
    uint32_t lastNumVariants() const { return num_noise_variants; }   // < batch size => the reference prints its "low number of variants" warning (:270-274)

  private:
    std::vector<uint32_t> noise_group_indices;
    std::vector<uint32_t> variants;   // per group of the unit
    std::mt19937 prng;
    uint32_t batch_size, num_noise_variants = 0;
};

// one row of the noise parameter file: "<chain>\t<iteration>\t<rate_0>\t...\n" with the default ostream formatting (6 significant
// digits) the reference writes with (InferenceEngine.cpp:205,236,267; Utils.hpp:209-224)
std::string noiseParameterHeader(const std::vector<std::string> &sample_names);
std::string noiseParameterRow(unsigned chain, unsigned iteration, const std::vector<double> &rates);

// what bt_gibbs_result_fetch returns for the clusters of a batch
struct BatchResults {
    std::vector<uint64_t> dip_off, cell_off;   // [C+1]
    std::vector<uint16_t> h1, h2;
    std::vector<uint32_t> freq;                // [entries * S]
    std::vector<double> stats;                 // [cells * 12]
};

struct GibbsOptions {   // main.cpp:389-393 + --random-seed
    unsigned seed = 0;
    uint32_t burn_in = 100, samples = 250, chains = 20, max_haplotype_variant_kmers = 500;
    float kmer_subsampling_rate = 0.1f;
    uint32_t max_groups_per_launch = 0;   // default mode: a unit is genotyped in consecutive launches of at most this many groups (0: as many as fit the free HBM)
};

// What the drivers need from the object that samples a batch of groups.  The product's sampler is bt_gibbs on the engine's GPU
// (InferenceEngine.cpp: GpuSampler, the default); the interface exists so that the drivers' own logic — group selection, sharding,
// the histogram reduction, launch splitting — can be exercised on a machine without a GPU by handing the engine another sampler
// (tests/: the oracle's, through bth_engine_set_sampler; nothing in the product provides one).
struct GibbsSampler {
    virtual ~GibbsSampler() {}
    virtual void setLut(const double *genomic, const double *noise) = 0;
    virtual void setNoiseLut(const double *noise) = 0;
    virtual void initChain(uint32_t chain) = 0;
    // resetGroup of every group (InferenceEngine.cpp:100-113: the genotypers are deleted, the next initChain constructs them anew with the chain's
    // seeds).  false: not supported — the caller then builds a fresh sampler instead.
    virtual bool resetGroups() { return false; }
    virtual void sweep(uint32_t n, bool collect) = 0;
    virtual void run() = 0;                                   // the whole default schedule
    virtual uint64_t deviceBytes() { return 0; }              // device memory the sampler holds (0: unknown)
    virtual void sync() {}                                    // wait for what run() enqueued (stage timing; results() waits anyway)
    virtual std::vector<uint64_t> noiseCounts() = 0;          // [S*256], VariantClusterGroup::getNoiseCounts of every group + clearGenotyperCache
    // one iteration of the noise drivers: (the noise table drawn in the previous iteration, or nullptr;) one sweep; the noise counts.
    // The GPU sampler does it with one synchronisation (bt_gibbs_noise_iteration).
    virtual std::vector<uint64_t> noiseIteration(const double *noise, bool collect) {
        if (noise) setNoiseLut(noise);
        sweep(1, collect);
        return noiseCounts();
    }
    // A whole chain of a noise driver on the sampler's side, without a host round trip per iteration: num_iterations x { sweep
    // (collecting from iteration first_collect on); noise counts; reduction over the ranks (device_reduce, may be empty); the rates drawn
    // from count_distribution's generator; the rebuilt table }.  rows receives the rates of every iteration ([it * S + s]); afterwards
    // count_distribution holds the last rates and the advanced generator.  false: not supported (the driver then iterates itself).
    // A chain's iterations with the sampler's groups RESIDENT on its side (bt_gibbs_noise_chain_begin / _step / _end: one launch per chain, the
    // per-iteration exchange through a mailbox in pinned memory): begin returns false when the sampler cannot (the driver then iterates with
    // noiseIteration as before); while a chain is resident noiseIteration is a step of it.  first_collect is 0-based.
    virtual bool beginResidentChain(uint32_t /*num_iterations*/, uint32_t /*first_collect*/) { return false; }
    virtual void endResidentChain() {}
    typedef std::function<void(uint64_t *d_hist, size_t n)> DeviceReducer;   // enqueues the all-reduce of a DEVICE histogram on the context's stream
    virtual bool noiseChain(CountDistribution *, uint32_t, uint32_t, const DeviceReducer &, std::vector<double> *) { return false; }
    virtual BatchResults results(uint32_t num_clusters) = 0;
    // the same results as one word string in the sampler's DEVICE memory (bt_gibbs_result_words; valid while the sampler lives); false: not supported
    virtual bool resultWords(const uint32_t ** /*d_words*/, uint64_t * /*num_words*/) { return false; }
};
typedef std::function<std::unique_ptr<GibbsSampler>(const bt_gibbs_params &, const GibbsBatchData &)> SamplerFactory;

class InferenceEngine {
  public:
    // collected samples of a launch: `batch` holds the launched groups (group_index = index in the unit), results in its cluster order
    typedef std::function<void(const GibbsBatchData &batch, const BatchResults &results)> Collector;
    // the same hand-over for a run of several ranks: the launch's result string in device memory (GibbsSampler::resultWords), which the caller
    // keeps on the device for the gather to rank 0.  Used instead of the Collector whenever it is set and the sampler has the string.
    typedef std::function<void(const GibbsBatchData &batch, const uint32_t *d_words, uint64_t num_words)> WireCollector;
    void setWireCollector(WireCollector w) { wire_collect = std::move(w); }
    typedef std::function<void(uint64_t *hist, size_t n)> HistReducer;   // sums the S*256 counters over all ranks in place

    InferenceEngine(bt_ctx *ctx, std::vector<uint8_t> gender, std::vector<std::string> sample_names, const GibbsOptions &options, HistReducer reduce_hist = nullptr);

    // unit = this rank's groups; unit_shape (optional) = clusters / variants per group of the WHOLE unit when `unit` is a shard of it
    void estimateNoise(CountDistribution *count_distribution, const GibbsBatchData &unit, const std::string &output_prefix, uint32_t variants_batch_size = NoiseGroupSelector::noise_variants_batch_size,
                       const std::vector<uint32_t> *unit_clusters_per_group = nullptr, const std::vector<uint32_t> *unit_variants_per_group = nullptr);
    void estimateGenotypes(const GibbsBatchData &unit, const CountDistribution &count_distribution, const Collector &collect);
    void estimateNoiseAndGenotypes(const GibbsBatchData &unit, CountDistribution *count_distribution, const Collector &collect, const std::string &output_prefix);

    bool lowNoiseVariantWarning() const { return low_variant_warning; }
    uint32_t numLaunches() const { return num_launches; }   // of the last estimateGenotypes
    void setSamplerFactory(SamplerFactory f) { make_sampler = std::move(f); }
    // uses_device_stream: the reducer copies / reduces on the context's stream (Comm over RCCL) — it would queue behind a resident chain launch that is itself
    // waiting for the reduced histogram, so with such a reducer the chains run launch by launch (the files transport and the tests' reducers stay on the host)
    void setHistReducer(HistReducer r, bool uses_device_stream = false) {
        reduce_hist = std::move(r);
        reduce_on_stream = uses_device_stream;
    }
    // with several ranks the noise chains stay on the device only if the histogram can be reduced there (Comm over RCCL: bt_comm_allreduce_hist)
    void setDeviceHistReducer(GibbsSampler::DeviceReducer r) { device_reduce = std::move(r); }
    // every row of the noise parameter file in full precision: (chain, iteration, rate_0 .. rate_{S-1}) per row (tests compare these, the file has 6 digits)
    void recordNoiseRows(bool on) { record_rows = on; }
    const std::vector<double> &noiseRows() const { return noise_rows; }
    void setQuiet(bool q) { quiet = q; }

  private:
    typedef GibbsSampler Sampler;
    void iteration(Sampler *sampler, CountDistribution *count_distribution, bool collect);
    void logRow(std::ostream &out, unsigned chain, unsigned iteration, const std::vector<double> &rates);
    std::unique_ptr<Sampler> newSampler(uint32_t noise_seeding, const GibbsBatchData &batch, bt_ctx *on_ctx = nullptr);
    void runDefault(const GibbsBatchData &batch, const CountDistribution &count_distribution, const Collector &collect);
    bt_gibbs_params params(uint32_t noise_seeding) const;

    void handOver(std::unique_ptr<Sampler> &sampler, const GibbsBatchData &batch, const Collector &collect);
    WireCollector wire_collect;
    bt_ctx *ctx;
    std::vector<uint8_t> gender;
    std::vector<std::string> sample_names;
    GibbsOptions opt;
    HistReducer reduce_hist;
    bool reduce_on_stream = false;
    bool low_variant_warning = false;
    uint32_t num_launches = 0;
    SamplerFactory make_sampler;
    GibbsSampler::DeviceReducer device_reduce;
    // one chain of a noise driver (iterations 1..n of the reference's loop): on the device when the sampler can, else iteration by iteration;
    // every iteration's rates are logged (row) and handed to `each`
    void runNoiseChain(Sampler *sampler, CountDistribution *cd, uint32_t chain, uint32_t first_collect_iteration, std::ostream &out,
                       const std::function<void(uint32_t iteration, const std::vector<double> &rates)> &each);
    bool record_rows = false, quiet = false;
    bool pending_noise = false;   // the count distribution holds a noise table the sampler has not been given yet
    std::vector<double> noise_rows;
};

}  // namespace bthost
