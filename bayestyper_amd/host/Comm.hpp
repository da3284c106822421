// One rank of a multi-GPU `bayesTyper genotype` run (one process per GPU): the rank layout read from the environment and the exchange
// steps of include/btcomm.h (libbtcomm.so, RCCL over xGMI) as the host layer uses them.  The reference is one process; what is sharded
// here is what its worker threads share out — KMC records (KmerCounter.cpp:431-524) and variant-cluster groups (InferenceEngine.cpp:335-382).
//
//   BT_WORLD / BT_RANK / BT_COMM_ID_FILE   set by whoever starts the ranks (rank 0 writes the communicator id to the file, the others wait for it)
//   BT_GPUS=N                              a single invocation becomes rank 0 and forks ranks 1..N-1 itself (host/main.cpp)
//
// libbtcomm.so is opened at run time and only when BT_WORLD > 1: a one-GPU run does not load RCCL.
//
//   BT_COMM_TRANSPORT=files   (tests) the ranks exchange through files next to BT_COMM_ID_FILE instead of RCCL: RCCL refuses a communicator
//                             whose ranks share a GPU, and a one-GPU box is where the sharded run's LOGIC — record ranges, count merge, group
//                             shares, histogram reduction, result gather — gets tested (tests/test_cli_gpu.py, BT_DEVICE=0 for every rank).
//                             Host-side only: no kernel depends on the transport.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/btgpu.h"

struct bt_comm;

namespace bthost {

struct GibbsBatchData;
struct BatchResults;

class Comm {
  public:
    // nullptr when the run has one rank (BT_WORLD unset or 1)
    static std::unique_ptr<Comm> fromEnvironment(bt_ctx *ctx);
    static int envRank();    // BT_RANK (0 when unset)
    // a rank that fails says so (a marker next to BT_COMM_ID_FILE): ranks waiting in a files-transport exchange give up instead of waiting
    // for it; with RCCL the process that started the ranks ends the run (host/main.cpp)
    static void markFailed();
    static int envWorld();   // BT_WORLD (1 when unset)
    ~Comm();
    int rank() const { return rank_; }
    int world() const { return world_; }
    // in-place sum over all ranks of n counters held on the host (the S x 256 noise-count histogram of an iteration)
    void allreduceHist(uint64_t *hist, size_t n);
    // the same for counters on the device, only ENQUEUED on the context's stream (RCCL transport only: deviceReduction() says whether)
    bool deviceReduction() const { return dir.empty(); }
    void allreduceDeviceAsync(uint64_t *d_hist, size_t n);
    // every rank's byte string, concatenated in rank order, on every rank (device round trip inside); offsets[world + 1]
    std::vector<uint8_t> allgatherBytes(const std::vector<uint8_t> &mine, std::vector<uint64_t> *offsets);
    // the same for device buffers: d_out (capacity bytes) receives all parts; returns the offsets
    std::vector<uint64_t> allgatherDevice(const uint8_t *d_local, uint64_t local_bytes, uint8_t *d_out, uint64_t capacity);
    // rank 0 receives every rank's words in rank order (offsets[world + 1]); the other ranks get an empty vector
    std::vector<uint32_t> gatherWords(const std::vector<uint32_t> &mine, std::vector<uint64_t> *offsets);
    // the same for a word string that is already in device memory (bt_gibbs_result_words): nothing is staged through the host on the
    // sending side, rank 0 copies the gathered string to the host once
    std::vector<uint32_t> gatherWordsDevice(const uint32_t *d_mine, uint64_t num_words, std::vector<uint64_t> *offsets);
    void barrier();

  private:
    Comm();
    bt_ctx *ctx = nullptr;
    bt_comm *comm = nullptr;
    void *dl = nullptr;
    int rank_ = 0, world_ = 1;
    std::string id_file_;   // BT_COMM_ID_FILE: removed by rank 0 when the run is over
    struct Api;
    std::unique_ptr<Api> api;
    // files transport
    std::string dir;
    uint64_t seq = 0;
    std::vector<std::vector<uint8_t>> exchangeFiles(const void *mine, size_t bytes);
};

// Longest-processing-time assignment of a unit's groups to `world` ranks on a cost proxy (per cluster H(H+1)/2 + 8 + K/16; groups in
// descending cost dealt in serpentine order): deterministic, identical on every rank.  ids[r] = ascending group indices of rank r.
std::vector<std::vector<uint32_t>> assignGroups(const GibbsBatchData &unit, int world);

// a rank's launches' result strings (bt_gibbs_result_words), one after the other in device memory in the order of the launches
class DeviceWords {
  public:
    explicit DeviceWords(bt_ctx *c) : ctx(c) {}
    ~DeviceWords();
    DeviceWords(const DeviceWords &) = delete;
    DeviceWords &operator=(const DeviceWords &) = delete;
    void append(const uint32_t *d_words, uint64_t num_words);
    const uint32_t *data() const { return (const uint32_t *)p; }
    uint64_t size() const { return n; }
    uint32_t parts() const { return k; }

  private:
    bt_ctx *ctx;
    void *p = nullptr;
    uint64_t n = 0, cap = 0;
    uint32_t k = 0;
};

// the collected samples of all ranks' groups, rebuilt on rank 0 in the unit's cluster order (so that what follows — getGenotypes,
// GenotypeWriter — sees exactly what a one-rank run hands it); every rank calls it, the others get an empty result
BatchResults gatherResults(Comm &comm, const GibbsBatchData &unit, const std::vector<std::vector<uint32_t>> &ids, const BatchResults &mine, uint32_t num_samples);
// the same from the rank's result strings in device memory (the product's path: the launches' results never visit the sending rank's host)
BatchResults gatherResults(Comm &comm, const GibbsBatchData &unit, const std::vector<std::vector<uint32_t>> &ids, const DeviceWords &mine, uint32_t num_samples);

}  // namespace bthost
