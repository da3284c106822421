// Internal (non-ABI) definitions shared by the libbtgpu translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/btgpu.h"
#include "bt_kmer_device.hpp"

namespace bt {

void set_error(const std::string &msg);
int fail(const std::string &msg);   // set_error + return BT_ERR

#define BT_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess)                                                               \
            return bt::fail(std::string(#call) + ": " + hipGetErrorString(_e));             \
    } while (0)

#define BT_CHECK_LAUNCH()                                                                   \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess)                                                               \
            return bt::fail(std::string("kernel launch: ") + hipGetErrorString(_e));        \
    } while (0)

inline unsigned grid_for(uint64_t n, unsigned block, unsigned max_blocks = 1u << 30) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (unsigned)g;
}

}  // namespace bt

struct bt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;     // the stream work is submitted to
    int num_cu = 0;
    // pinned staging arena of this context's small uploads (bt_gibbs.hip: staged_upload), reused from sampler to sampler
    uint8_t *pin = nullptr;
    size_t pin_bytes = 0, pin_used = 0;
    // staging slots of the KMC scans that stream records from the host (bt_table.hip: bt_kmc_scan_run_host / _run_file): kept from scan to scan — a database
    // per sample means a scan handle per sample, and 2 x 218 MB of pinned memory allocated and released for each of them
    uint8_t *kmc_pin[2] = {nullptr, nullptr}, *kmc_dev[2] = {nullptr, nullptr};
    size_t kmc_stage_bytes = 0;
    // the state pool of the last noise-driver sampler released on this context (bt_gibbs.hip): the next chain's sampler is over as many groups of the same
    // unit and takes it over instead of a hipFree + hipMalloc of gigabytes per chain; released when it does not fit the next sampler, and with the context
    void *pool_cache = nullptr;
    size_t pool_cache_bytes = 0;
    // pinned host buffers handed back by released samplers (the mailbox of a resident noise chain, the per-iteration staging words): hipHostMalloc /
    // hipHostFree cost milliseconds each, a noise driver builds a sampler per chain
    struct HostBuf {
        void *p;
        size_t bytes;
        unsigned flags;
    };
    HostBuf host_cache[8] = {};
    // The streams the launch classes of this context's samplers run on (bt_gibbs.hip): created once, each PROBED to run concurrently with the context's
    // stream and with one another (ctx_class_streams) — the runtime deals a handful of hardware queues to the process's streams round robin, and launch
    // classes whose streams share a queue run one after the other (the same ten-sample schedule: 12.9 or 17.1 s, a chr20 unit 0.41 or 0.73 s, depending on
    // which streams a sampler happened to get).  Samplers borrow them; they live as long as the context.
    std::vector<hipStream_t> class_streams;
    std::vector<hipStream_t> retired_streams;   // class streams of an earlier context stream: samplers built then may still hold them
    hipStream_t class_streams_for = nullptr;   // the context stream they were probed against (bt_ctx_set_stream may change it)
    int class_streams_prio = 0;
    bool class_streams_probed = false;
    unsigned class_streams_concurrent = 0;   // how many of class_streams proved to run concurrently with the context's stream and with one another (the rest share hardware queues)
};

namespace bt {
// a pinned host buffer of exactly `bytes` with `flags` from the context's cache, or a new one
inline hipError_t ctx_host_take(bt_ctx *ctx, void **out, size_t bytes, unsigned flags) {
    for (auto &h : ctx->host_cache)
        if (h.p && h.bytes == bytes && h.flags == flags) {
            *out = h.p;
            h.p = nullptr;
            return hipSuccess;
        }
    return hipHostMalloc(out, bytes, flags);
}
inline bool ctx_caches_off() {
    static const bool off = getenv("BT_CTX_NO_CACHE") != nullptr;   // (diagnosis: every sampler allocates and releases its own)
    return off;
}
// n streams of priority prio that run concurrently with ctx->stream and with each other (bt_ctx.hip); fewer than n concurrent ones found: the rest share
hipError_t ctx_class_streams(bt_ctx *ctx, unsigned n, int prio, hipStream_t *out);
inline void ctx_host_give(bt_ctx *ctx, void *p, size_t bytes, unsigned flags) {
    if (!p) return;
    for (auto &h : ctx->host_cache)
        if (!h.p && !ctx_caches_off()) {
            h = bt_ctx::HostBuf{p, bytes, flags};
            return;
        }
    (void)hipHostFree(p);
}
}  // namespace bt

struct bt_timer {
    bt_ctx *ctx = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
};

struct bt_bloom {
    bt_ctx *ctx = nullptr;
    uint64_t num_kmers = 0;       // per (sub-)filter
    uint64_t num_bits = 0;        // per (sub-)filter
    uint32_t num_hashes = 0;
    uint32_t num_sub = 1;
    uint32_t k = 0;
    uint64_t stride = 0;          // bytes per sub-filter, multiple of 4
    uint64_t bytes = 0;           // total device bytes
    uint32_t *d_words = nullptr;
    bt::BloomView view() const {
        bt::BloomView v;
        v.words = d_words;
        v.stride = stride;
        v.bits = bt::make_fastmod(num_bits);
        v.num_hashes = num_hashes;
        v.num_sub = num_sub;
        v.k = k;
        return v;
    }
};

namespace bt {
// Open-addressing table in HBM, one packed record per slot: a probe touches ONE 64-byte sector (round 3 kept state / key_lo / key_hi / meta /
// counts in five arrays: four sectors and four address translations per probed slot).  A slot is slot_words 32-bit words, a multiple of four
// (32 bytes up to 8 samples, 48 up to 20, 64 up to 30), 16-byte aligned:
//   word 0 state (0 empty, 1 being written, 2 ready) | word 1 meta (byte0 flags, byte1 max_haploid_multiplicity, byte2 female ic, byte3 male ic)
//   words 2-3 key_lo | words 4-5 key_hi | words 6.. the samples' count bytes (spad bytes, a multiple of 4) | padding
struct TableView {
    uint32_t *slots;
    uint64_t mask;        // capacity - 1
    uint32_t slot_words;  // words per slot
    uint32_t spad;        // bytes of counts per slot (multiple of 4)
    uint32_t k;
    uint32_t flags;       // bit 0: publish a new key with a release store (BT_TABLE_RELEASE_PUBLISH=1) instead of ordered write-through stores
    unsigned long long *num_keys;   // device counter
    uint32_t *overflow;             // device flag
    __host__ __device__ inline uint32_t *slot(uint64_t i) const { return slots + i * slot_words; }
    __host__ __device__ inline uint32_t *state(uint64_t i) const { return slot(i); }
    __host__ __device__ inline uint32_t *meta(uint64_t i) const { return slot(i) + 1; }
    __host__ __device__ inline uint64_t *key_lo(uint64_t i) const { return reinterpret_cast<uint64_t *>(slot(i) + 2); }
    __host__ __device__ inline uint64_t *key_hi(uint64_t i) const { return reinterpret_cast<uint64_t *>(slot(i) + 4); }
    __host__ __device__ inline uint32_t *counts(uint64_t i) const { return slot(i) + 6; }   // spad / 4 words
    __host__ __device__ inline const uint8_t *count_bytes(uint64_t i) const { return reinterpret_cast<const uint8_t *>(slot(i) + 6); }
    __host__ __device__ static inline uint32_t slot_words_for(uint32_t spad) { return (6u + spad / 4u + 3u) & ~3u; }
};
}  // namespace bt

struct bt_table {
    bt_ctx *ctx = nullptr;
    uint64_t capacity = 0;
    uint32_t num_samples = 0;
    uint32_t spad = 0;
    uint32_t k = 0;
    bt::TableView v{};
};

struct bt_kmc_scan {
    bt_ctx *ctx = nullptr;
    uint32_t k = 0, p = 0, counter_size = 0, suffix_bytes = 0, rec_size = 0;
    uint64_t total = 0;
    uint32_t min_count = 0, max_count = 0xFFFFFFFFu;   // bt_kmc_scan_set_count_range
    uint64_t lut_entries = 0;   // 4^p + 1
    uint64_t *d_lut = nullptr;
    uint32_t *d_hint = nullptr;   // prefix of every 4096th record (bt_table.hip: KmcView::hint)
    // staging of bt_kmc_scan_run_host (created on first use, kept for the life of the handle): two pinned host buffers, two device
    // buffers, a copy stream and the events that order copy and scan
    uint8_t *h_pin[2] = {nullptr, nullptr}, *d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr}, scanned[2] = {nullptr, nullptr};
    unsigned long long *d_host_hits = nullptr;
    // buffers of the partitioned scan (bt_table.hip: kmc_partition_kernel / kmc_probe_bucket_kernel), created on first use: [0] the stripe regions, [1] the hit list
    void *d_route_vals[2] = {nullptr, nullptr};
    uint64_t routed_cap = 0;
    unsigned int *d_part_cursor = nullptr;   // partitioned scan: fill of the 256 bucket regions
    uint32_t part_cap = 0;                   // records per bucket region
    unsigned int *d_num_hits = nullptr;
    // round 6: a SECOND set of those buffers, so that the partition of chunk i + 1 (LDS-bound) runs on a class stream of the context while chunk i is probed
    // (VALU-bound) and applied (latency-bound) on the context's stream; absent (single-buffered scan) when the allocation fails
    void *d_route_vals2[2] = {nullptr, nullptr};
    unsigned int *d_part_cursor2 = nullptr, *d_num_hits2 = nullptr;
    hipEvent_t part_done[2] = {nullptr, nullptr}, part_free[2] = {nullptr, nullptr}, scan_begin = nullptr;
};
