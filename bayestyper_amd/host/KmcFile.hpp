// KMC database access for the count-table scan and the scan driver itself:
//   CKMCFile::OpenForListing / ReadParamsFrom_prefix_file_buf (external/kmc_api/kmc_file.cpp:96-292) — header + prefix LUT(s) of a KMC1
//   ("version 0") or KMC2 ("0x200", one prefix table per signature bin) database; the .kmc_suf payload is memory-mapped and streamed to the GPU as it lies on disk;
//   KmerCounter::parseSampleKmers (src/bayesTyper/KmerCounter.cpp:431-524) — one sample's records through
//   bt_kmc_scan_run in double-buffered chunks.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/btgpu.h"

namespace bthost {

class KmcFile {
  public:
    explicit KmcFile(const std::string &prefix);   // throws std::runtime_error with the reason
    ~KmcFile();
    KmcFile(const KmcFile &) = delete;
    KmcFile &operator=(const KmcFile &) = delete;

    uint32_t kmer_length = 0, mode = 0, counter_size = 0, lut_prefix_length = 0, min_count = 0;
    uint64_t max_count = 0, total_kmers = 0;
    uint32_t record_size() const { return (kmer_length - lut_prefix_length) / 4 + counter_size; }
    const std::vector<uint64_t> &prefix_lut() const { return lut; }   // bins * 4^p + 1 entries (bins = 1 for KMC1), last = total_kmers
    const uint8_t *records() const { return payload; }                 // total_kmers * record_size() bytes
    const std::string &suffix_file() const { return suf_path; }        // the .kmc_suf file: records() start 4 bytes into it

  private:
    std::vector<uint64_t> lut;
    std::string suf_path;
    void *map = nullptr;
    size_t map_bytes = 0;
    const uint8_t *payload = nullptr;
};

// KmerCounter::parseSampleKmers for ONE sample: every record of the database through decode -> path-Bloom lookup -> (hit)
// addKmer + addSampleCount(sample_idx).  Returns the number of Bloom hits.  chunk_records: records per host-to-device copy.
// [first_record, first_record + num_records): the part of the database this rank scans (default: all of it)
uint64_t parseSampleKmers(bt_ctx *ctx, const KmcFile &db, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, uint64_t chunk_records = 1ull << 24, uint64_t first_record = 0,
                          uint64_t num_records = ~0ull);

}  // namespace bthost
