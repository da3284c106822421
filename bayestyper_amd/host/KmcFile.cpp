#include "KmcFile.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace bthost {

KmcFile::KmcFile(const std::string &prefix) {
    // ---- .kmc_pre: "KMCP" | prefix LUT | header | header_offset (4 B) | "KMCP"  (kmc_file.cpp:96-140,177-292) ----
    std::ifstream pre(prefix + ".kmc_pre", std::ios::binary | std::ios::ate);
    if (!pre.is_open()) throw std::runtime_error("cannot open " + prefix + ".kmc_pre");
    const uint64_t size = (uint64_t)pre.tellg();
    if (size < 4 + 4 + 4 + 40) throw std::runtime_error(prefix + ".kmc_pre: too short");
    std::vector<char> buf(size);
    pre.seekg(0);
    pre.read(buf.data(), (std::streamsize)size);
    if (std::memcmp(buf.data(), "KMCP", 4) != 0 || std::memcmp(buf.data() + size - 4, "KMCP", 4) != 0) throw std::runtime_error(prefix + ".kmc_pre: missing KMCP markers");
    uint32_t kmc_version;
    std::memcpy(&kmc_version, buf.data() + size - 12, 4);
    const uint64_t header_offset = (uint8_t)buf[size - 8];
    const uint64_t body = size - 8;                       // without the two markers
    if (header_offset < 40 || header_offset + 4 > body) throw std::runtime_error(prefix + ".kmc_pre: bad header offset");
    uint64_t lut_entries = 0;                             // prefix-table entries in the file (bins x 4^p for KMC2)
    if (kmc_version == 0x200) {
        // KMC2: "KMCP" | per-bin prefix tables | guard word | signature map | header | header_offset | "KMCP" (kmc_file.cpp:186-238)
        const char *h = buf.data() + size - 8 - header_offset;
        uint32_t f[7];
        std::memcpy(f, h, sizeof(f));
        kmer_length = f[0];
        mode = f[1];
        counter_size = f[2];
        lut_prefix_length = f[3];
        const uint32_t signature_len = f[4];
        min_count = f[5];
        max_count = f[6];
        std::memcpy(&total_kmers, h + 28, 8);
        if (signature_len > 11) throw std::runtime_error(prefix + ".kmc_pre: unsupported signature length");
        const uint64_t signature_map_bytes = ((1ull << (2 * signature_len)) + 1) * 4;
        if (body < 4 + signature_map_bytes + header_offset + 8) throw std::runtime_error(prefix + ".kmc_pre: too short for its signature map");
        lut_entries = (body - 4 - signature_map_bytes - header_offset - 8) / 8;
    } else if (kmc_version == 0) {
        const uint64_t header_at = 4 + (body - 4 - header_offset);   // byte offset of the header in the file
        uint64_t h[5];
        std::memcpy(h, buf.data() + header_at, sizeof(h));
        kmer_length = (uint32_t)h[0];
        mode = (uint32_t)(h[0] >> 32);
        counter_size = (uint32_t)h[1];
        lut_prefix_length = (uint32_t)(h[1] >> 32);
        min_count = (uint32_t)h[2];
        max_count = (h[2] >> 32) + (h[4] & 0xFFFFFFFF00000000ull);
        total_kmers = h[3];
        lut_entries = (header_at - 4) / 8;
    } else
        throw std::runtime_error(prefix + ".kmc_pre: unknown KMC database version");
    if (mode != 0) throw std::runtime_error(prefix + ": KMC databases with quality-weighted counters (mode 1) are not supported (KmerCounter.cpp:449)");
    if (kmer_length == 0 || kmer_length > 64 || lut_prefix_length > 15 || lut_prefix_length > kmer_length || (kmer_length - lut_prefix_length) % 4 != 0 || counter_size < 1 ||
        counter_size > 4)
        throw std::runtime_error(prefix + ".kmc_pre: unsupported parameters");
    const uint64_t nlut = 1ull << (2 * lut_prefix_length);
    if (lut_entries == 0 || lut_entries % nlut != 0 || (kmc_version == 0 && lut_entries != nlut)) throw std::runtime_error(prefix + ".kmc_pre: prefix table is not a multiple of 4^p entries");
    lut.resize(lut_entries + 1);
    std::memcpy(lut.data(), buf.data() + 4, lut_entries * 8);
    lut[lut_entries] = total_kmers;
    // ---- .kmc_suf: "KMCS" | records | "KMCS" ----
    const std::string suf = prefix + ".kmc_suf";
    suf_path = suf;
    const int fd = ::open(suf.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + suf);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        ::close(fd);
        throw std::runtime_error("cannot stat " + suf);
    }
    map_bytes = (size_t)st.st_size;
    if (map_bytes != 8 + total_kmers * record_size()) {
        ::close(fd);
        throw std::runtime_error(suf + ": size does not match total_kmers x record size");
    }
    map = mmap(nullptr, map_bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) {
        map = nullptr;
        throw std::runtime_error("cannot map " + suf);
    }
    if (std::memcmp(map, "KMCS", 4) != 0) throw std::runtime_error(suf + ": missing KMCS marker");
    payload = (const uint8_t *)map + 4;
}

KmcFile::~KmcFile() {
    if (map) munmap(map, map_bytes);
}

uint64_t parseSampleKmers(bt_ctx *ctx, const KmcFile &db, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, uint64_t chunk_records, uint64_t first_record,
                          uint64_t num_records) {
    if (first_record > db.total_kmers) first_record = db.total_kmers;
    if (num_records > db.total_kmers - first_record) num_records = db.total_kmers - first_record;
    if (num_records == 0) return 0;
    bt_kmc_scan *scan = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (bt_kmc_scan_create_bins(ctx, db.kmer_length, db.lut_prefix_length, db.counter_size, db.total_kmers, db.prefix_lut().data(), db.prefix_lut().size(), &scan) != BT_OK)
        throw std::runtime_error(std::string("parseSampleKmers: ") + bt_last_error());
    bt_kmc_scan_set_count_range(scan, db.min_count, db.max_count);   // ReadNextKmer's counter filter (kmc_file.cpp:496-511)
    uint64_t hits = 0;
    const auto t1 = std::chrono::steady_clock::now();
    // copies from the page cache (mmap) into pinned staging, H2D transfers and scan kernels of consecutive chunks overlap inside the library
    const uint64_t rec_bytes = (db.kmer_length - db.lut_prefix_length) / 4 + db.counter_size;
    // (the file is read with pread() by several threads straight into the pinned slots; BT_KMC_MMAP=1: copied out of the memory mapping instead)
    const int rc = getenv("BT_KMC_MMAP") ? bt_kmc_scan_run_host(scan, path_bloom, table, sample_idx, db.records() + first_record * rec_bytes, first_record, num_records, chunk_records, &hits)
                                         : bt_kmc_scan_run_file(scan, path_bloom, table, sample_idx, db.suffix_file().c_str(), 4, first_record, num_records, chunk_records, &hits);
    const auto t2 = std::chrono::steady_clock::now();
    bt_kmc_scan_destroy(scan);
    if (getenv("BT_STAGE_TIMES")) {
        const auto t3 = std::chrono::steady_clock::now();
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        fprintf(stderr, "  parse sample k-mers: scan set-up %.3f s, stream + scan %.3f s, release %.3f s\n", sec(t0, t1), sec(t1, t2), sec(t2, t3));
    }
    if (rc != BT_OK) throw std::runtime_error(std::string("parseSampleKmers: ") + bt_last_error());
    return hits;
}

}  // namespace bthost
