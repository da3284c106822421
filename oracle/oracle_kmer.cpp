// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library, and only as the checker / CPU baseline.
//
// Plain scalar C++ restatement of the reference's k-mer side of the hot path (BayesTyper v1.5).
// Each function cites the reference lines it follows (paths relative to the reference root).
// Parity status: PINNED — validated bit-for-bit against the reference's own translation units
// compiled unmodified into oracle/_ref/libbtref.so (tests/test_oracle_vs_ref.py) and against the
// known answers of SURVEY.md Appendix A.2 (tests/golden/known_answers.json).
//
// The oracle deliberately works on ASCII k-mers and byte-addressed filters like the reference does
// (the product works on 2-bit packed words), so the two implementations share no code.
#include <algorithm>
#include <bitset>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

// ---- ntHash: external/ntHash/nthash.hpp:18-28 (seeds), :30-118 (msTab = rol(seed, j)), :262-282 ----
const uint64_t kSeedA = 0x3c8bfbb395c60474ULL, kSeedC = 0x3193c18562a02b4cULL, kSeedG = 0x20323ed082572324ULL,
               kSeedT = 0x295549f54be24456ULL;
const uint64_t kMultiSeed = 0x90b45d39fb6da1faULL;
const int kMultiShift = 27;

inline uint64_t rol(uint64_t x, unsigned r) { r %= 64; return r ? (x << r) | (x >> (64 - r)) : x; }

// msTab[c][j]: rows for A/a C/c G/g T/t are the rotated seeds, every other character is the zero row
// (nthash.hpp:85-118)
inline uint64_t ms_tab(unsigned char c, unsigned j) {
    switch (c) {
        case 'A': case 'a': return rol(kSeedA, j);
        case 'C': case 'c': return rol(kSeedC, j);
        case 'G': case 'g': return rol(kSeedG, j);
        case 'T': case 't': return rol(kSeedT, j);
        default: return 0;
    }
}

uint64_t ntp64(const char *kmer, unsigned k) {   // nthash.hpp:262-267
    uint64_t h = 0;
    for (unsigned i = 0; i < k; i++) h ^= ms_tab((unsigned char)kmer[i], (k - 1 - i) % 64);
    return h;
}

uint64_t ntp64_seed(const char *kmer, unsigned k, unsigned seed) {   // nthash.hpp:275-282
    uint64_t h = ntp64(kmer, k);
    h *= seed ^ k * kMultiSeed;   // same C precedence as the reference: seed ^ (k * multiSeed)
    h ^= h >> kMultiShift;
    return h;
}

// ---- BloomFilter: external/ntHash/BloomFilter.hpp:40-66,149-161 ----
struct FlatBloom {
    uint64_t m_size = 0;
    unsigned m_hashNum = 0, m_kmerSize = 0;
    std::vector<unsigned char> filter;
    FlatBloom(uint64_t size, unsigned hashes, unsigned k) : m_size(size), m_hashNum(hashes), m_kmerSize(k), filter((size + 7) / 8, 0) {}
    void insertF(const char *kmer) {
        uint64_t h = ntp64(kmer, m_kmerSize);
        uint64_t loc = h % m_size;
        filter[loc / 8] |= (1 << (7 - loc % 8));
        for (unsigned i = 1; i < m_hashNum; i++) {
            uint64_t mh = h * (i ^ m_kmerSize * kMultiSeed);
            mh ^= mh >> kMultiShift;
            uint64_t l = mh % m_size;
            filter[l / 8] |= (1 << (7 - l % 8));
        }
    }
    bool containsF(const char *kmer) const {
        uint64_t h = ntp64(kmer, m_kmerSize);
        uint64_t loc = h % m_size;
        if ((filter[loc / 8] & (1 << (7 - loc % 8))) == 0) return false;
        for (unsigned i = 1; i < m_hashNum; i++) {
            uint64_t mh = h * (i ^ m_kmerSize * kMultiSeed);
            mh ^= mh >> kMultiShift;
            uint64_t l = mh % m_size;
            if ((filter[l / 8] & (1 << (7 - l % 8))) == 0) return false;
        }
        return true;
    }
};

// ---- KmerBloom sizing: src/kmerBloom/KmerBloom.cpp:134-146 (types kept: float fpr, uint64 n) ----
uint64_t calcOptNumBloomBits(const float fpr, const uint64_t num_kmers) {
    auto ln2 = std::log(2);
    return std::ceil(-(num_kmers * std::log(fpr) / ln2 / ln2));
}
unsigned calcOptNumHashes(const uint64_t num_bloom_bits, const uint64_t num_kmers) {
    auto frac = static_cast<double>(num_bloom_bits) / static_cast<double>(num_kmers);
    return std::ceil(frac * std::log(2));
}

// KmerBloom (KmerBloom.cpp:54-60) or ThreadedKmerBloom (KmerBloom.cpp:204-215,277-280)
struct OrcBloom {
    unsigned k = 0;
    uint64_t num_kmers = 0, num_bits = 0;
    unsigned num_hashes = 0;
    std::vector<FlatBloom> subs;   // 1 or 65536
    unsigned route(const char *kmer) const { return subs.size() == 1 ? 0 : (unsigned)(ntp64_seed(kmer, k, 1029283129) % subs.size()); }
};

// ---- 2-bit packing of the reference's bitset<2k>: Nucleotide.hpp:40-70 (A=00 C=01 G=10 T=11, bit 2i = low bit) ----
inline int nt_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}
inline void pack(const char *kmer, unsigned k, uint64_t *out2) {
    out2[0] = out2[1] = 0;
    for (unsigned i = 0; i < k; i++) {
        uint64_t v = (uint64_t)nt_code(kmer[i]);
        out2[(2 * i) / 64] |= v << ((2 * i) % 64);
    }
}
inline void unpack(const uint64_t *in2, unsigned k, char *out) {
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    for (unsigned i = 0; i < k; i++) out[i] = nt[(in2[(2 * i) / 64] >> ((2 * i) % 64)) & 3];
}

// ---- KmerCounts record: include/bayesTyper/KmerCounts.hpp:70-75, src/bayesTyper/KmerCounts.cpp ----
struct KC {
    uint8_t flags = 0;   // bit0 cluster, 1 multicluster, 2 multigroup, 3 decoy, 4 maxmult, 5 parameter
    uint8_t max_haploid = 0, fem = 0, male = 0;
    uint8_t counts[30] = {0};
    static uint8_t upd(uint8_t cur, uint8_t in) { return ((255 - cur) <= in) ? 255 : cur + in; }   // KmerCounts.cpp:178-188
    void addInterclusterMultiplicity(bool is_decoy, unsigned fp, unsigned mp) {                      // KmerCounts.cpp:98-118
        max_haploid = upd(max_haploid, 1);
        if (max_haploid > 127) flags |= 0x10;
        if (is_decoy) flags |= 0x08;
        else {
            fem = upd(fem, (uint8_t)fp);
            male = upd(male, (uint8_t)mp);
        }
    }
    void addClusterMultiplicity(uint8_t mult, bool is_multigroup) {   // KmerCounts.cpp:137-159
        if (flags & 0x01) flags |= 0x02;
        flags |= 0x01;
        if (is_multigroup) flags |= 0x04;
        max_haploid = upd(max_haploid, mult);
        if (max_haploid > 127) flags |= 0x10;
    }
    void addSampleCount(unsigned s, uint8_t c) { counts[s] = upd(counts[s], c); }   // KmerCounts.cpp:161-171
    bool isExcluded() const { return flags & (0x08 | 0x10 | 0x04); }                  // KmerCounts.cpp:93-96
};

struct OrcTable {
    unsigned k = 0, num_samples = 0;
    std::unordered_map<std::string, KC> map;   // keyed by the ASCII canonical k-mer
};

// sliding canonical k-mers: Kmer.tpp:44-81 (forward), :116-153 (reverse complement), :225-255 (lowest)
inline char comp(char c) {
    switch (c) {
        case 'A': case 'a': return 'T';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'A';
    }
}
inline char up(char c) {
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    return nt[nt_code(c)];
}
// calls f(end_position, canonical_ascii) for every complete window
template <typename F>
void slide_canonical(const char *seq, uint64_t len, unsigned k, F f) {
    std::string fw, rc;
    uint64_t run = 0;   // number of consecutive valid nucleotides ending here (window "reset" on anything else)
    for (uint64_t i = 0; i < len; i++) {
        if (nt_code(seq[i]) < 0) {
            run = 0;
            continue;
        }
        run++;
        if (run >= k) {
            fw.assign(k, 'A');
            rc.assign(k, 'A');
            for (unsigned j = 0; j < k; j++) {
                fw[j] = up(seq[i - k + 1 + j]);
                rc[j] = comp(seq[i - j]);
            }
            // lexicographically lowest, ties -> forward (Kmer.tpp:253)
            f(i, (rc < fw) ? rc : fw);
        }
    }
}

// ---- KMC database (format: SURVEY Appendix C.1, external/kmc_api/kmc_file.cpp:177-292,428-494) ----
struct KmcDb {
    unsigned k = 0, p = 0, counter_size = 0;
    uint64_t total = 0;
    std::vector<uint64_t> lut;        // 4^p + 1
    std::vector<unsigned char> suf;   // payload of .kmc_suf without the two markers
};

bool read_file(const std::string &fn, std::vector<unsigned char> &out) {
    std::ifstream f(fn, std::ios::binary);
    if (!f.is_open()) return false;
    f.seekg(0, std::ios::end);
    size_t n = (size_t)f.tellg();
    f.seekg(0);
    out.resize(n);
    f.read((char *)out.data(), (std::streamsize)n);
    return true;
}

bool kmc_open(const std::string &prefix, KmcDb &db) {
    std::vector<unsigned char> pre, suf;
    if (!read_file(prefix + ".kmc_pre", pre) || !read_file(prefix + ".kmc_suf", suf)) return false;
    if (pre.size() < 8 + 12 || memcmp(pre.data(), "KMCP", 4) || memcmp(pre.data() + pre.size() - 4, "KMCP", 4)) return false;
    if (suf.size() < 8 || memcmp(suf.data(), "KMCS", 4) || memcmp(suf.data() + suf.size() - 4, "KMCS", 4)) return false;
    uint32_t version;
    memcpy(&version, pre.data() + pre.size() - 12, 4);   // kmc_file.cpp:180-185
    if (version == 0x200) {   // KMC2: "KMCP" | per-bin LUTs | guard word | signature map | header | header_offset | "KMCP" (kmc_file.cpp:186-238)
        const uint64_t header_offset = pre[pre.size() - 8];
        const unsigned char *h = pre.data() + pre.size() - 8 - header_offset;
        auto u32 = [&](unsigned i) { uint32_t w; memcpy(&w, h + 4 * i, 4); return w; };
        db.k = u32(0);
        if (u32(1) != 0) return false;   // mode 0 only
        db.counter_size = u32(2);
        db.p = u32(3);
        const uint32_t signature_len = u32(4);
        memcpy(&db.total, h + 28, 8);
        const uint64_t signature_map_size = (1ULL << (2 * signature_len)) + 1;
        const uint64_t size = pre.size() - 8 - 4;   // without markers and header_offset
        const uint64_t lut_area = size - (signature_map_size * 4 + header_offset + 8);
        const uint64_t m = lut_area / 8;            // number of LUT entries: bins x 4^p
        if (m == 0 || (m % (1ULL << (2 * db.p))) != 0) return false;
        db.lut.assign(m + 1, 0);
        memcpy(db.lut.data(), pre.data() + 4, m * 8);
        db.lut[m] = db.total;
        db.suf.assign(suf.begin() + 4, suf.end() - 4);
        return true;
    }
    if (version != 0) return false;
    uint64_t size = pre.size() - 8;                        // without the two markers
    uint64_t header_offset = pre[pre.size() - 8];          // kmc_file.cpp:245-246
    size -= 4;
    uint64_t header_index = (size - header_offset) / 8;
    const unsigned char *words = pre.data() + 4;
    auto word = [&](uint64_t i) { uint64_t w; memcpy(&w, words + 8 * i, 8); return w; };
    uint64_t d = word(header_index);
    db.k = (uint32_t)d;
    if ((d >> 32) != 0) return false;   // mode 0 only (KmerCounter.cpp:449)
    db.counter_size = (uint32_t)word(header_index + 1);
    db.p = (uint32_t)(word(header_index + 1) >> 32);
    db.total = word(header_index + 3);
    if (header_index != (1ULL << (2 * db.p))) return false;
    db.lut.assign(header_index + 1, 0);
    for (uint64_t i = 0; i < header_index; i++) db.lut[i] = word(i);
    db.lut[header_index] = db.total;
    db.suf.assign(suf.begin() + 4, suf.end() - 4);
    return true;
}

// record n -> (ascii k-mer, count): kmc_file.cpp:428-494
void kmc_record(const KmcDb &db, uint64_t n, uint64_t prefix, std::string &kmer, uint32_t &count) {
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    kmer.resize(db.k);
    prefix &= (1ULL << (2 * db.p)) - 1;   // KMC2: the LUT index runs over all bins ("& prefix_mask", kmc_file.cpp:430,449)
    for (unsigned i = 0; i < db.p; i++) kmer[i] = nt[(prefix >> (2 * (db.p - 1 - i))) & 3];
    const unsigned sb = (db.k - db.p) / 4;
    const unsigned char *rec = db.suf.data() + n * (sb + db.counter_size);
    for (unsigned b = 0; b < sb; b++)
        for (unsigned j = 0; j < 4; j++) kmer[db.p + 4 * b + j] = nt[(rec[b] >> (6 - 2 * j)) & 3];
    count = 0;
    for (unsigned b = 0; b < db.counter_size; b++) count |= (uint32_t)rec[sb + b] << (8 * b);
}

}  // namespace

extern "C" {

uint64_t orc_ntp64(const char *kmer, unsigned k) { return ntp64(kmer, k); }
uint64_t orc_ntp64_seed(const char *kmer, unsigned k, unsigned seed) { return ntp64_seed(kmer, k, seed); }
void orc_ntp64_batch(const char *kmers, uint64_t n, unsigned k, int seeded, unsigned seed, uint64_t *out) {
    for (uint64_t i = 0; i < n; i++) out[i] = seeded ? ntp64_seed(kmers + i * k, k, seed) : ntp64(kmers + i * k, k);
}

void orc_pack_batch(const char *kmers, uint64_t n, unsigned k, uint64_t *out) {
    for (uint64_t i = 0; i < n; i++) pack(kmers + i * k, k, out + 2 * i);
}
void orc_unpack_batch(const uint64_t *packed, uint64_t n, unsigned k, char *out) {
    for (uint64_t i = 0; i < n; i++) unpack(packed + 2 * i, k, out + i * k);
}

void orc_bloom_sizing(uint64_t num_kmers, float fpr, uint64_t *bits, unsigned *hashes) {
    *bits = calcOptNumBloomBits(fpr, num_kmers);
    *hashes = calcOptNumHashes(*bits, num_kmers);
}

void *orc_bloom_new(uint64_t num_kmers, float fpr, unsigned k, int threaded) {
    OrcBloom *b = new OrcBloom();
    b->k = k;
    unsigned nsub = threaded ? 65536u : 1u;
    // ThreadedKmerBloom: KmerBloom(ceil(num_kmers / float(root_size)), fpr) (KmerBloom.cpp:213)
    uint64_t per = threaded ? (uint64_t)std::ceil(num_kmers / static_cast<float>(nsub)) : num_kmers;
    b->num_kmers = std::max(per, static_cast<uint64_t>(1));   // KmerBloom.cpp:54
    b->num_bits = calcOptNumBloomBits(fpr, b->num_kmers);
    b->num_hashes = calcOptNumHashes(b->num_bits, b->num_kmers);
    b->subs.assign(nsub, FlatBloom(b->num_bits, b->num_hashes, k));
    return b;
}
void *orc_bloom_load(const char *prefix, unsigned k) {   // KmerBloom.cpp:63-89
    std::ifstream meta(std::string(prefix) + ".bloomMeta");
    if (!meta.is_open()) return nullptr;
    std::string line;
    std::getline(meta, line);
    std::stringstream ss(line);
    std::vector<std::string> tok;
    for (std::string item; std::getline(ss, item, '\t');) tok.push_back(item);
    if (tok.size() != 3 || (unsigned)std::stoi(tok[2]) != k) return nullptr;
    OrcBloom *b = new OrcBloom();
    b->k = k;
    b->num_kmers = std::stol(tok[0]);
    b->num_bits = std::stol(tok[1]);
    b->num_hashes = calcOptNumHashes(b->num_bits, b->num_kmers);
    b->subs.assign(1, FlatBloom(b->num_bits, b->num_hashes, k));
    std::ifstream data(std::string(prefix) + ".bloomData", std::ios::binary);
    data.read((char *)b->subs[0].filter.data(), (std::streamsize)b->subs[0].filter.size());
    return b;
}
int orc_bloom_save(void *h, const char *prefix) {   // KmerBloom.cpp:149-164
    OrcBloom *b = (OrcBloom *)h;
    if (b->subs.size() != 1) return 1;
    std::ofstream meta(std::string(prefix) + ".bloomMeta");
    meta << std::to_string(b->num_kmers) << "\t" << std::to_string(b->num_bits) << "\t" << std::to_string(b->k) << std::endl;
    std::ofstream data(std::string(prefix) + ".bloomData", std::ios::binary);
    data.write((const char *)b->subs[0].filter.data(), (std::streamsize)b->subs[0].filter.size());
    return 0;
}
void orc_bloom_free(void *h) { delete (OrcBloom *)h; }
void orc_bloom_info(void *h, uint64_t *num_kmers, uint64_t *bits, unsigned *hashes, unsigned *nsub) {
    OrcBloom *b = (OrcBloom *)h;
    *num_kmers = b->num_kmers;
    *bits = b->num_bits;
    *hashes = b->num_hashes;
    *nsub = (unsigned)b->subs.size();
}
void orc_bloom_insert(void *h, const char *kmers, uint64_t n) {
    OrcBloom *b = (OrcBloom *)h;
    for (uint64_t i = 0; i < n; i++) b->subs[b->route(kmers + i * b->k)].insertF(kmers + i * b->k);
}
void orc_bloom_contains(void *h, const char *kmers, uint64_t n, uint8_t *hits) {
    OrcBloom *b = (OrcBloom *)h;
    for (uint64_t i = 0; i < n; i++) hits[i] = b->subs[b->route(kmers + i * b->k)].containsF(kmers + i * b->k) ? 1 : 0;
}
void orc_bloom_bits(void *h, unsigned sub, uint8_t *out) {
    OrcBloom *b = (OrcBloom *)h;
    memcpy(out, b->subs[sub].filter.data(), b->subs[sub].filter.size());
}
unsigned orc_bloom_route(void *h, const char *kmer) { return ((OrcBloom *)h)->route(kmer); }

// canonical k-mer per position, packed; valid[i] = 1 when a full window ends at i
void orc_kmers_from_sequence(const char *seq, uint64_t len, unsigned k, uint64_t *kmers, uint8_t *valid) {
    memset(valid, 0, len);
    memset(kmers, 0, len * 16);
    slide_canonical(seq, len, k, [&](uint64_t i, const std::string &can) {
        pack(can.data(), k, kmers + 2 * i);
        valid[i] = 1;
    });
}

// ---- table ----
void *orc_table_new(unsigned num_samples, unsigned k) {
    OrcTable *t = new OrcTable();
    t->k = k;
    t->num_samples = num_samples;
    return t;
}
void orc_table_free(void *h) { delete (OrcTable *)h; }
uint64_t orc_table_size(void *h) { return ((OrcTable *)h)->map.size(); }
void orc_table_insert(void *h, const char *kmers, uint64_t n, int mark_parameter) {   // main.cpp:571-577
    OrcTable *t = (OrcTable *)h;
    for (uint64_t i = 0; i < n; i++) {
        KC &kc = t->map[std::string(kmers + i * t->k, t->k)];
        if (mark_parameter) kc.flags |= 0x20;
    }
}
// KmerCounter::countInterclusterKmersCallback for one region (KmerCounter.cpp:291-338)
void orc_table_count_intercluster(void *h, void *bloom, const char *seq, uint64_t len, int is_decoy, unsigned fp, unsigned mp) {
    OrcTable *t = (OrcTable *)h;
    OrcBloom *b = (OrcBloom *)bloom;
    slide_canonical(seq, len, t->k, [&](uint64_t, const std::string &can) {
        if (b->subs[b->route(can.data())].containsF(can.data())) t->map[can].addInterclusterMultiplicity(is_decoy != 0, fp, mp);
    });
}
// KmerCounter::countInterclusterParameterKmersCallback for ONE region (KmerCounter.cpp:161-231): seed = prng_seed + region index;
// the table plays KmerHash<bool>: decoy -> value false (flag 0x08 here), accepted -> value true unless already present (flag 0x20)
void orc_table_count_parameter_kmers(void *h, void *bloom, const char *seq, uint64_t len, int is_decoy, unsigned seed, float fraction) {
    OrcTable *t = (OrcTable *)h;
    OrcBloom *b = (OrcBloom *)bloom;
    std::bernoulli_distribution bernoulli_dist(fraction);
    std::mt19937 prng;
    prng.seed(seed);
    slide_canonical(seq, len, t->k, [&](uint64_t, const std::string &can) {
        if (!b->subs[b->route(can.data())].containsF(can.data())) {
            if (is_decoy) t->map[can].flags |= 0x08;
            else if (bernoulli_dist(prng)) t->map[can].flags |= 0x20;
        }
    });
}
// table half of VariantClusterGraph::classifyPathKmers (VariantClusterGraph.cpp:902-938)
void orc_table_classify(void *h, void *mg_bloom, const char *kmers, const uint8_t *mult, uint64_t n, uint8_t *excluded) {
    OrcTable *t = (OrcTable *)h;
    OrcBloom *b = (OrcBloom *)mg_bloom;
    for (uint64_t i = 0; i < n; i++) {
        std::string key(kmers + i * t->k, t->k);
        auto it = t->map.find(key);
        if (it == t->map.end() && mult[i] > 127) it = t->map.emplace(key, KC()).first;
        excluded[i] = 0;
        if (it != t->map.end()) {
            it->second.addClusterMultiplicity(mult[i], b->subs[b->route(key.data())].containsF(key.data()));
            excluded[i] = it->second.isExcluded() ? 1 : 0;
        }
    }
}
// ObservedKmerCountsHash<N>::calculateKmerStats (KmerHash.cpp:256-340) with KmerStats::addValue (KmerStats.cpp:51-63) as the
// reference runs it: a running Welford update per (sample, intercluster multiplicity) in table iteration order.
// class_counts[7] = {total, unique, multicluster, decoy, max_multiplicity, multigroup, non_cluster};
// stats[(s*256+m)*4 + {count, fraction, mean, M2}]
void orc_table_kmer_stats(void *h, const uint8_t *gender, uint64_t *class_counts, double *stats) {
    OrcTable *t = (OrcTable *)h;
    for (int i = 0; i < 7; i++) class_counts[i] = 0;
    for (size_t i = 0; i < (size_t)t->num_samples * 256 * 4; i++) stats[i] = 0;
    for (auto &e : t->map) {
        const KC &kc = e.second;
        class_counts[0]++;
        if (kc.flags & 0x01) {
            if (kc.isExcluded()) {
                if (kc.flags & 0x08) class_counts[3]++;
                else if (kc.flags & 0x10) class_counts[4]++;
                else class_counts[5]++;
            } else if (kc.flags & 0x02) class_counts[2]++;
            else class_counts[1]++;
        } else {
            class_counts[6]++;
            if (kc.flags & 0x20) {
                for (unsigned s = 0; s < t->num_samples; s++) {
                    double *ks = stats + ((size_t)s * 256 + (gender[s] ? kc.male : kc.fem)) * 4;
                    const double value = kc.counts[s];
                    ks[0] += 1;
                    ks[1] += ((value != 0 ? 1.0 : 0.0) - ks[1]) / ks[0];
                    const double delta = value - ks[2];
                    ks[2] += delta / ks[0];
                    ks[3] += delta * (value - ks[2]);
                }
            }
        }
    }
}
// export sorted by ASCII k-mer: kmers (n*k chars), counts (n*num_samples), meta (n*4)
uint64_t orc_table_export(void *h, char *kmers, uint8_t *counts, uint8_t *meta) {
    OrcTable *t = (OrcTable *)h;
    std::vector<const std::pair<const std::string, KC> *> v;
    for (auto &e : t->map) v.push_back(&e);
    std::sort(v.begin(), v.end(), [](const std::pair<const std::string, KC> *a, const std::pair<const std::string, KC> *b) { return a->first < b->first; });
    uint64_t i = 0;
    for (auto e : v) {
        memcpy(kmers + i * t->k, e->first.data(), t->k);
        for (unsigned s = 0; s < t->num_samples; s++) counts[i * t->num_samples + s] = e->second.counts[s];
        meta[4 * i] = e->second.flags;
        meta[4 * i + 1] = e->second.max_haploid;
        meta[4 * i + 2] = e->second.fem;
        meta[4 * i + 3] = e->second.male;
        i++;
    }
    return i;
}

// ---- KMC ----
// Writer for test fixtures (KMC1 "version 0" layout).  kmers must be sorted ascending
// (ASCII order == KMC order) and unique.
int orc_kmc_write(const char *prefix, const char *kmers, const uint32_t *counts, uint64_t n, unsigned k, unsigned p, unsigned counter_size) {
    if ((k - p) % 4) return 1;
    const uint64_t nlut = 1ULL << (2 * p);
    std::vector<uint64_t> lut(nlut, 0);
    const unsigned sb = (k - p) / 4;
    std::vector<unsigned char> suf;
    suf.reserve(n * (sb + counter_size));
    std::vector<uint64_t> per(nlut, 0);
    for (uint64_t i = 0; i < n; i++) {
        const char *km = kmers + i * k;
        uint64_t pre = 0;
        for (unsigned j = 0; j < p; j++) pre = (pre << 2) | (uint64_t)nt_code(km[j]);
        per[pre]++;
        for (unsigned b = 0; b < sb; b++) {
            unsigned char byte = 0;
            for (unsigned j = 0; j < 4; j++) byte = (unsigned char)((byte << 2) | nt_code(km[p + 4 * b + j]));
            suf.push_back(byte);
        }
        for (unsigned b = 0; b < counter_size; b++) suf.push_back((unsigned char)((counts[i] >> (8 * b)) & 0xFF));
    }
    uint64_t acc = 0;
    for (uint64_t j = 0; j < nlut; j++) {
        lut[j] = acc;
        acc += per[j];
    }
    std::ofstream fp(std::string(prefix) + ".kmc_pre", std::ios::binary);
    std::ofstream fs(std::string(prefix) + ".kmc_suf", std::ios::binary);
    if (!fp.is_open() || !fs.is_open()) return 2;
    fp.write("KMCP", 4);
    fp.write((const char *)lut.data(), (std::streamsize)(nlut * 8));
    uint64_t header[8] = {0};
    header[0] = (uint64_t)k;                                    // k | mode << 32 (mode 0)
    header[1] = (uint64_t)counter_size | ((uint64_t)p << 32);   // counter_size | lut_prefix_len << 32
    header[2] = 1ULL | (255ULL << 32);                          // min | max << 32
    header[3] = n;                                              // total k-mers
    header[4] = 0;                                              // both-strands flag word (0 => canonical counting)
    fp.write((const char *)header, 64);
    uint32_t header_offset = 64;
    fp.write((const char *)&header_offset, 4);
    fp.write("KMCP", 4);
    fs.write("KMCS", 4);
    fs.write((const char *)suf.data(), (std::streamsize)suf.size());
    fs.write("KMCS", 4);
    return 0;
}

// KMC2 ("0x200") writer for tests: the k-mers (sorted ascending, canonical) are dealt into `nbins` bins (bin = a hash of the k-mer, as
// KMC's signature binning does by minimiser), each bin sorted; per-bin prefix LUTs hold ABSOLUTE record offsets (kmc_file.cpp:428-447)
int orc_kmc2_write(const char *prefix, const char *kmers, const uint32_t *counts, uint64_t n, unsigned k, unsigned p, unsigned counter_size, unsigned nbins) {
    if ((k - p) % 4 || nbins == 0) return 1;
    const uint64_t nlut = 1ULL << (2 * p);
    const unsigned sb = (k - p) / 4;
    std::vector<std::vector<uint64_t>> bin(nbins);
    for (uint64_t i = 0; i < n; i++) bin[ntp64(kmers + i * k, k) % nbins].push_back(i);   // input order is sorted, so every bin stays sorted
    std::vector<uint64_t> lut;
    std::vector<unsigned char> suf;
    uint64_t at = 0;
    for (unsigned b = 0; b < nbins; b++) {
        std::vector<uint64_t> per(nlut, 0);
        for (uint64_t i : bin[b]) {
            const char *km = kmers + i * k;
            uint64_t pre = 0;
            for (unsigned j = 0; j < p; j++) pre = (pre << 2) | (uint64_t)nt_code(km[j]);
            per[pre]++;
            for (unsigned x = 0; x < sb; x++) {
                unsigned char byte = 0;
                for (unsigned j = 0; j < 4; j++) byte = (unsigned char)((byte << 2) | nt_code(km[p + 4 * x + j]));
                suf.push_back(byte);
            }
            for (unsigned x = 0; x < counter_size; x++) suf.push_back((unsigned char)((counts[i] >> (8 * x)) & 0xFF));
        }
        for (uint64_t j = 0; j < nlut; j++) {
            lut.push_back(at);
            at += per[j];
        }
    }
    lut.push_back(at);   // guard word (the reader overwrites it with total + 1)
    const uint32_t signature_len = 5;
    std::vector<uint32_t> sigmap((1u << (2 * signature_len)) + 1, 0);
    unsigned char header[48] = {0};
    const uint32_t f[7] = {k, 0, counter_size, p, signature_len, 1, 255};
    memcpy(header, f, 28);
    memcpy(header + 28, &n, 8);
    header[36] = 0;   // both-strands byte: 0 => canonical counting
    const uint32_t version = 0x200;
    memcpy(header + 44, &version, 4);
    std::ofstream fp(std::string(prefix) + ".kmc_pre", std::ios::binary);
    std::ofstream fs(std::string(prefix) + ".kmc_suf", std::ios::binary);
    if (!fp.is_open() || !fs.is_open()) return 2;
    fp.write("KMCP", 4);
    fp.write((const char *)lut.data(), (std::streamsize)(lut.size() * 8));
    fp.write((const char *)sigmap.data(), (std::streamsize)(sigmap.size() * 4));
    fp.write((const char *)header, 48);
    const uint32_t header_offset = 48;
    fp.write((const char *)&header_offset, 4);
    fp.write("KMCP", 4);
    fs.write("KMCS", 4);
    fs.write((const char *)suf.data(), (std::streamsize)suf.size());
    fs.write("KMCS", 4);
    return 0;
}

void *orc_kmc_open(const char *prefix) {
    KmcDb *db = new KmcDb();
    if (!kmc_open(prefix, *db)) {
        delete db;
        return nullptr;
    }
    return db;
}
void orc_kmc_free(void *h) { delete (KmcDb *)h; }
void orc_kmc_info(void *h, unsigned *k, unsigned *p, unsigned *counter_size, uint64_t *total) {
    KmcDb *db = (KmcDb *)h;
    *k = db->k;
    *p = db->p;
    *counter_size = db->counter_size;
    *total = db->total;
}
uint64_t orc_kmc_lut_entries(void *h) { return ((KmcDb *)h)->lut.size(); }
void orc_kmc_lut(void *h, uint64_t *out) { memcpy(out, ((KmcDb *)h)->lut.data(), ((KmcDb *)h)->lut.size() * 8); }
uint64_t orc_kmc_payload_size(void *h) { return ((KmcDb *)h)->suf.size(); }
void orc_kmc_payload(void *h, uint8_t *out) { memcpy(out, ((KmcDb *)h)->suf.data(), ((KmcDb *)h)->suf.size()); }
// list all records in order (CKMCFile::ReadNextKmer semantics, kmc_file.cpp:428-494)
void orc_kmc_list(void *h, char *kmers, uint32_t *counts) {
    KmcDb *db = (KmcDb *)h;
    uint64_t prefix = 0;
    std::string km;
    for (uint64_t n = 0; n < db->total; n++) {
        while (db->lut[prefix + 1] <= n) prefix++;
        uint32_t c;
        kmc_record(*db, n, prefix, km, c);
        memcpy(kmers + n * db->k, km.data(), db->k);
        counts[n] = c;
    }
}

// KmerCounter::parseSampleKmers for one sample (KmerCounter.cpp:388-429): every record -> path-Bloom lookup
// -> on hit addKmer + addSampleCount.  Returns the number of Bloom hits.
uint64_t orc_parse_sample_kmers(void *table, void *bloom, void *kmc, unsigned sample_idx, uint64_t first, uint64_t n) {
    OrcTable *t = (OrcTable *)table;
    OrcBloom *b = (OrcBloom *)bloom;
    KmcDb *db = (KmcDb *)kmc;
    uint64_t prefix = 0, hits = 0;
    std::string km;
    for (uint64_t r = first; r < first + n; r++) {
        while (db->lut[prefix + 1] <= r) prefix++;
        uint32_t c;
        kmc_record(*db, r, prefix, km, c);
        assert(c <= 255);   // KmerCounter.cpp:401
        if (b->subs[b->route(km.data())].containsF(km.data())) {
            hits++;
            t->map[km].addSampleCount(sample_idx, (uint8_t)c);
        }
    }
    return hits;
}

// lookup-only variant used as the CPU baseline of "k-mer matches/sec" when no table growth is wanted
uint64_t orc_match_only(void *bloom, void *kmc, uint64_t first, uint64_t n) {
    OrcBloom *b = (OrcBloom *)bloom;
    KmcDb *db = (KmcDb *)kmc;
    uint64_t prefix = 0, hits = 0;
    std::string km;
    for (uint64_t r = first; r < first + n; r++) {
        while (db->lut[prefix + 1] <= r) prefix++;
        uint32_t c;
        kmc_record(*db, r, prefix, km, c);
        hits += b->subs[b->route(km.data())].containsF(km.data()) ? 1 : 0;
    }
    return hits;
}


// =====================================================================================================================
// Path k-mer enumeration over variant-cluster graphs (VariantClusterGraph.cpp:800-1184), restated on the flattened graph
// arrays of include/btgpu.h's bt_paths_batch (plain arrays here; the oracle includes no product header).
// =====================================================================================================================
struct OrcGraphs {
    unsigned k;
    uint32_t C;
    const uint32_t *vertex_off, *num_paths;
    const uint64_t *seq_off;
    const uint8_t *seq;
    const uint16_t *vvar, *vall;
    const uint8_t *vflags;
    const uint32_t *vnested, *refvar_off;
    const uint16_t *refvar;
    const uint64_t *path_off;
    const uint8_t *paths;
    const uint32_t *var_off;
    const uint16_t *var_na;
    const uint8_t *var_dep;
    const uint32_t *in_off = nullptr, *in_src = nullptr;   // in-edges per vertex (CSR over global vertices, local source indices, insertion order)
};
void *orc_graphs_new(unsigned k, uint32_t C, const uint32_t *vertex_off, const uint32_t *num_paths, const uint64_t *seq_off, const uint8_t *seq,
                     const uint16_t *vvar, const uint16_t *vall, const uint8_t *vflags, const uint32_t *vnested, const uint32_t *refvar_off,
                     const uint16_t *refvar, const uint64_t *path_off, const uint8_t *paths, const uint32_t *var_off, const uint16_t *var_na,
                     const uint8_t *var_dep) {
    return new OrcGraphs{k, C, vertex_off, num_paths, seq_off, seq, vvar, vall, vflags, vnested, refvar_off, refvar, path_off, paths, var_off, var_na, var_dep};
}
void orc_graphs_free(void *h) { delete (OrcGraphs *)h; }
void orc_graphs_set_edges(void *h, const uint32_t *in_off, const uint32_t *in_src) {
    ((OrcGraphs *)h)->in_off = in_off;
    ((OrcGraphs *)h)->in_src = in_src;
}

}  // extern "C" (closed around the template below)
// walks path `p` of cluster `c` nucleotide by nucleotide like the three reference loops do; f_vertex(v) is called when a path
// vertex is entered (before its nucleotides), f_kmer(canonical_ascii) for every completed window; f_nt() after every nucleotide
template <typename FV, typename FK, typename FN>
static void walk_path(const OrcGraphs &g, uint32_t c, uint32_t p, FV f_vertex, FK f_kmer, FN f_nt) {
    static const char nt[4] = {'A', 'C', 'G', 'T'};
    const uint32_t v0 = g.vertex_off[c], nv = g.vertex_off[c + 1] - v0;
    const uint8_t *row = g.paths + g.path_off[c] + (uint64_t)p * nv;
    std::string window;   // KmerPair: the last <= k nucleotides since the last reset (Kmer.tpp:182-255)
    for (uint32_t vi = 0; vi < nv; vi++) {
        if (!row[vi]) continue;
        const uint32_t v = v0 + vi;
        f_vertex(v);
        if (g.vflags[v] & 1) window.clear();   // is_disconnected -> kmer_pair.reset()
        for (uint64_t i = g.seq_off[v]; i < g.seq_off[v + 1]; i++) {
            window.push_back(nt[g.seq[i] & 3]);
            if (window.size() > g.k) window.erase(0, 1);
            if (window.size() == g.k) {
                std::string rc(g.k, 'A');
                for (unsigned j = 0; j < g.k; j++) rc[j] = comp(window[g.k - 1 - j]);
                f_kmer((rc < window) ? rc : window);
            }
            f_nt();
        }
    }
}

extern "C" {
// VariantClusterGraph::countPathKmers (:800-846) for all clusters + KmerCounter::countPathKmersCallback (KmerCounter.cpp:252-289):
// the set of path k-mers goes into the path Bloom filter.  Returns the number of k-mer windows.
uint64_t orc_paths_count_kmers(void *gh, void *bloom) {
    const OrcGraphs &g = *(OrcGraphs *)gh;
    OrcBloom *b = (OrcBloom *)bloom;
    uint64_t windows = 0;
    std::unordered_set<std::string> path_kmers;
    for (uint32_t c = 0; c < g.C; c++)
        for (uint32_t p = 0; p < g.num_paths[c]; p++)
            walk_path(g, c, p, [](uint32_t) {}, [&](const std::string &km) { path_kmers.insert(km); windows++; }, [] {});
    if (b)
        for (auto &km : path_kmers) b->subs[b->route(km.data())].insertF(km.data());
    return windows;
}

// KmerCounter::countPathMultigroupKmersCallback (KmerCounter.cpp:105-145), single thread (the only thread count for which the
// reference's outcome is defined): groups in index order; the group's path k-mers are collected in ONE
// std::unordered_set<std::bitset<2k>> that lives across the groups and is clear()ed between them (its bucket count survives), and are
// visited in that container's iteration order; a path k-mer the filter already reports goes into the multigroup table, otherwise it
// is added to the filter.  Returns num_path_kmers.  (k = 55 as the reference is built: bitset<110>; any other k falls back to a set
// of strings, whose order is not the reference's.)
uint64_t orc_paths_count_multigroup(void *gh, const uint32_t *cluster_group, void *bloom, void *table) {
    const OrcGraphs &g = *(OrcGraphs *)gh;
    OrcBloom *b = (OrcBloom *)bloom;
    OrcTable *t = (OrcTable *)table;
    uint32_t num_groups = 0;
    for (uint32_t c = 0; c < g.C; c++) num_groups = std::max(num_groups, cluster_group[c] + 1);
    uint64_t num_kmers = 0;
    auto visit = [&](const std::string &km) {
        FlatBloom &f = b->subs[b->route(km.data())];
        if (f.containsF(km.data())) t->map[km];
        else f.insertF(km.data());
    };
    if (g.k == 55) {
        static const char nt[4] = {'A', 'C', 'G', 'T'};
        std::unordered_set<std::bitset<110>> group_kmers;
        for (uint32_t grp = 0; grp < num_groups; grp++) {
            group_kmers.clear();
            for (uint32_t c = 0; c < g.C; c++)
                if (cluster_group[c] == grp)
                    for (uint32_t p = 0; p < g.num_paths[c]; p++)
                        walk_path(g, c, p, [](uint32_t) {}, [&](const std::string &km) {
                            uint64_t w[2];
                            pack(km.data(), 55, w);
                            std::bitset<110> bits(w[0]);
                            bits |= std::bitset<110>(w[1]) << 64;
                            group_kmers.emplace(bits);
                        }, [] {});
            num_kmers += group_kmers.size();
            for (auto &bits : group_kmers) {
                std::string km(55, 'A');
                for (unsigned i = 0; i < 55; i++) km[i] = nt[(bits[2 * i] ? 1 : 0) | (bits[2 * i + 1] ? 2 : 0)];
                visit(km);
            }
        }
        return num_kmers;
    }
    for (uint32_t grp = 0; grp < num_groups; grp++) {
        std::unordered_set<std::string> group_kmers;
        for (uint32_t c = 0; c < g.C; c++)
            if (cluster_group[c] == grp)
                for (uint32_t p = 0; p < g.num_paths[c]; p++) walk_path(g, c, p, [](uint32_t) {}, [&](const std::string &km) { group_kmers.insert(km); }, [] {});
        num_kmers += group_kmers.size();
        for (auto &km : group_kmers) visit(km);
    }
    return num_kmers;
}

// VariantClusterGraph::classifyPathKmers (:848-939) for every cluster, clusters in index order
void orc_paths_classify(void *gh, void *table, void *mg_bloom, uint32_t *num_path_kmers, uint8_t *has_excluded) {
    const OrcGraphs &g = *(OrcGraphs *)gh;
    OrcTable *t = (OrcTable *)table;
    OrcBloom *b = (OrcBloom *)mg_bloom;
    for (uint32_t c = 0; c < g.C; c++) {
        const uint32_t P = g.num_paths[c];
        std::unordered_map<std::string, std::vector<uint8_t>> mult;
        for (uint32_t p = 0; p < P; p++)
            walk_path(g, c, p, [](uint32_t) {}, [&](const std::string &km) {
                auto it = mult.emplace(km, std::vector<uint8_t>(P, 0)).first;
                if (it->second[p] < 255) it->second[p]++;
            }, [] {});
        num_path_kmers[c] = (uint32_t)mult.size();
        has_excluded[c] = 0;
        for (auto &e : mult) {
            const uint8_t mx = *std::max_element(e.second.begin(), e.second.end());
            auto it = t->map.find(e.first);
            if (it == t->map.end() && mx > 127) it = t->map.emplace(e.first, KC()).first;
            if (it != t->map.end()) {
                it->second.addClusterMultiplicity(mx, b->subs[b->route(e.first.data())].containsF(e.first.data()));
                if (it->second.isExcluded()) has_excluded[c] = 1;
            }
        }
    }
}

// VariantClusterGraph::getHaplotypeCandidates (:941-1135) + updateVariantPathIndices (:1137-1184) for every cluster.
// Results are kept in the handle and copied out by orc_paths_candidates_fetch (same arrays as bt_paths_candidates_out).
struct OrcCandidates {
    std::vector<uint32_t> kmer_off, kv_off, unique_off, unique_idx, multi_off, multi_idx, hapnest_off, hapnest_idx, nestdep_off, nestdep_cluster, nestdep_var_off,
        kv_bits;
    std::vector<uint8_t> mult, has_counts, counts, ic;
    std::vector<uint64_t> key;
    std::vector<uint16_t> kv_var, hap_allele, nestdep_var;
};
void *orc_paths_candidates(void *gh, void *table, uint64_t *sizes /* 11 values, order of bt_paths_candidates_sizes */) {
    const OrcGraphs &g = *(OrcGraphs *)gh;
    OrcTable *t = (OrcTable *)table;
    const unsigned S = t->num_samples;
    OrcCandidates *o = new OrcCandidates();
    o->kmer_off.push_back(0);
    o->kv_off.push_back(0);
    o->unique_off.push_back(0);
    o->multi_off.push_back(0);
    o->hapnest_off.push_back(0);
    o->nestdep_off.push_back(0);
    o->nestdep_var_off.push_back(0);
    uint64_t num_hap = 0;
    for (uint32_t c = 0; c < g.C; c++) {
        const uint32_t P = g.num_paths[c], V = g.var_off[c + 1] - g.var_off[c], HW = (P + 31) / 32;
        const uint32_t v0 = g.vertex_off[c], nv = g.vertex_off[c + 1] - v0;
        std::unordered_map<std::string, uint32_t> row_of;
        std::vector<std::string> row_key;
        std::vector<std::vector<uint8_t>> M;                                    // rows x P
        std::vector<std::vector<std::pair<uint16_t, std::vector<uint32_t>>>> kv;   // per row: (variant, bitset words)
        std::vector<uint32_t> uniq, multi;
        for (uint32_t p = 0; p < P; p++) {
            std::vector<uint16_t> alleles(V, 0xFFFF);
            std::vector<uint32_t> nested;
            uint32_t num_nucleotides = 0;
            std::map<std::pair<uint16_t, uint16_t>, std::pair<uint32_t, uint32_t>> running;   // (variant, allele) -> [first, second)
            walk_path(
                g, c, p,
                [&](uint32_t v) {
                    if (g.vvar[v] != 0xFFFF) {
                        if (!(g.vflags[v] & 1)) alleles[g.vvar[v]] = g.vall[v];
                        auto it = running.emplace(std::make_pair(g.vvar[v], g.vall[v]),
                                                  std::make_pair(num_nucleotides + ((g.vflags[v] >> 1) & 1u), num_nucleotides + g.k - 1)).first;
                        it->second.second += (uint32_t)(g.seq_off[v + 1] - g.seq_off[v]);
                    }
                    for (uint32_t r = g.refvar_off[v]; r < g.refvar_off[v + 1]; r++) {
                        auto it = running.find(std::make_pair(g.refvar[r], (uint16_t)0));
                        if (it != running.end()) it->second.second += (uint32_t)(g.seq_off[v + 1] - g.seq_off[v]);
                    }
                    if (g.vnested[v] != 0xFFFFFFFFu) nested.push_back(g.vnested[v]);
                },
                [&](const std::string &km) {
                    auto kc = t->map.find(km);
                    bool is_multi = false;
                    if (kc != t->map.end()) {
                        if (kc->second.isExcluded()) return;
                        is_multi = kc->second.flags & 0x02;
                    }
                    auto ins = row_of.emplace(km, (uint32_t)row_of.size());
                    const uint32_t row = ins.first->second;
                    if (ins.second) {
                        row_key.push_back(km);
                        M.emplace_back(P, 0);
                        kv.emplace_back();
                        (is_multi ? multi : uniq).push_back(row);
                    }
                    M[row][p]++;
                    // updateVariantPathIndices: every running variant whose interval covers this nucleotide
                    for (auto it = running.begin(); it != running.end();) {
                        if (it->second.second <= num_nucleotides) {
                            it = running.erase(it);
                            continue;
                        }
                        if (it->second.first <= num_nucleotides) {
                            auto &lst = kv[row];
                            auto e = std::find_if(lst.begin(), lst.end(), [&](const std::pair<uint16_t, std::vector<uint32_t>> &x) { return x.first == it->first.first; });
                            if (e == lst.end()) {
                                lst.emplace_back(it->first.first, std::vector<uint32_t>(HW, 0));
                                e = lst.end() - 1;
                            }
                            e->second[p >> 5] |= 1u << (p & 31);
                        }
                        ++it;
                    }
                },
                [&] { num_nucleotides++; });
            std::sort(nested.begin(), nested.end());
            for (uint32_t var = 0; var < V; var++)
                if (alleles[var] == 0xFFFF) alleles[var] = (uint16_t)(g.var_na[g.var_off[c] + var] - 1);   // missing allele (:1095-1102)
            o->hap_allele.insert(o->hap_allele.end(), alleles.begin(), alleles.end());
            o->hapnest_idx.insert(o->hapnest_idx.end(), nested.begin(), nested.end());
            o->hapnest_off.push_back((uint32_t)o->hapnest_idx.size());
            num_hap++;
        }
        const uint32_t K = (uint32_t)row_key.size();
        for (uint32_t r = 0; r < K; r++) {
            o->mult.insert(o->mult.end(), M[r].begin(), M[r].end());
            uint64_t pk[2];
            pack(row_key[r].data(), g.k, pk);
            o->key.push_back(pk[0]);
            o->key.push_back(pk[1]);
            auto kc = t->map.find(row_key[r]);
            o->has_counts.push_back(kc != t->map.end());
            for (unsigned s = 0; s < S; s++) o->counts.push_back(kc != t->map.end() ? kc->second.counts[s] : 0);
            o->ic.push_back(kc != t->map.end() ? kc->second.fem : 0);
            o->ic.push_back(kc != t->map.end() ? kc->second.male : 0);
            std::sort(kv[r].begin(), kv[r].end(), [](const std::pair<uint16_t, std::vector<uint32_t>> &a, const std::pair<uint16_t, std::vector<uint32_t>> &b) { return a.first < b.first; });
            for (auto &e : kv[r]) {
                o->kv_var.push_back(e.first);
                o->kv_bits.insert(o->kv_bits.end(), e.second.begin(), e.second.end());
            }
            o->kv_off.push_back((uint32_t)o->kv_var.size());
        }
        o->kmer_off.push_back(o->kmer_off.back() + K);
        o->unique_idx.insert(o->unique_idx.end(), uniq.begin(), uniq.end());
        o->unique_off.push_back((uint32_t)o->unique_idx.size());
        o->multi_idx.insert(o->multi_idx.end(), multi.begin(), multi.end());
        o->multi_off.push_back((uint32_t)o->multi_idx.size());
        // nested_variant_cluster_dependency (:1112-1132); std::map iteration = ascending child index here, the reference's
        // unordered_map order is irrelevant downstream (lookups by key)
        std::map<uint32_t, std::vector<uint16_t>> dep;
        for (uint32_t vi = 0; vi < nv; vi++) {
            const uint32_t v = v0 + vi;
            if (g.vnested[v] == 0xFFFFFFFFu) continue;
            auto &lst = dep[g.vnested[v]];
            if (g.vvar[v] != 0xFFFF) lst.push_back(g.vvar[v]);
            for (uint32_t r = g.refvar_off[v]; r < g.refvar_off[v + 1]; r++) lst.push_back(g.refvar[r]);
            std::sort(lst.begin(), lst.end(), std::greater<uint16_t>());
        }
        for (auto &e : dep) {
            o->nestdep_cluster.push_back(e.first);
            o->nestdep_var.insert(o->nestdep_var.end(), e.second.begin(), e.second.end());
            o->nestdep_var_off.push_back((uint32_t)o->nestdep_var.size());
        }
        o->nestdep_off.push_back((uint32_t)o->nestdep_cluster.size());
    }
    const uint64_t sz[11] = {o->kmer_off.back(), o->mult.size(), o->kv_var.size(), o->kv_bits.size(), o->unique_idx.size(), o->multi_idx.size(),
                             o->hap_allele.size(), num_hap, o->hapnest_idx.size(), o->nestdep_cluster.size(), o->nestdep_var.size()};
    memcpy(sizes, sz, sizeof(sz));
    return o;
}
void orc_paths_candidates_fetch(void *h, uint32_t *kmer_off, uint8_t *mult, uint64_t *key, uint8_t *has_counts, uint8_t *counts, uint8_t *ic, uint32_t *kv_off,
                                uint16_t *kv_var, uint32_t *kv_bits, uint32_t *unique_off, uint32_t *unique_idx, uint32_t *multi_off, uint32_t *multi_idx,
                                uint16_t *hap_allele, uint32_t *hapnest_off, uint32_t *hapnest_idx, uint32_t *nestdep_off, uint32_t *nestdep_cluster,
                                uint32_t *nestdep_var_off, uint16_t *nestdep_var) {
    OrcCandidates *o = (OrcCandidates *)h;
    auto cp = [](auto &v, auto *dst) { if (!v.empty()) memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(o->kmer_off, kmer_off); cp(o->mult, mult); cp(o->key, key); cp(o->has_counts, has_counts); cp(o->counts, counts); cp(o->ic, ic);
    cp(o->kv_off, kv_off); cp(o->kv_var, kv_var); cp(o->kv_bits, kv_bits); cp(o->unique_off, unique_off); cp(o->unique_idx, unique_idx);
    cp(o->multi_off, multi_off); cp(o->multi_idx, multi_idx); cp(o->hap_allele, hap_allele); cp(o->hapnest_off, hapnest_off);
    cp(o->hapnest_idx, hapnest_idx); cp(o->nestdep_off, nestdep_off); cp(o->nestdep_cluster, nestdep_cluster); cp(o->nestdep_var_off, nestdep_var_off);
    cp(o->nestdep_var, nestdep_var);
    delete o;
}


}  // extern "C" (closed around the path-search classes)
// =====================================================================================================================
// VariantClusterGraph::findSamplePaths (VariantClusterGraph.cpp:389-798) with VariantClusterGraphPath
// (VariantClusterGraphPath.cpp:38-225): the per-sample best-path search.  Restated object by object.
// =====================================================================================================================
namespace {
constexpr uint8_t MIN_OBSERVED_KMERS = 2;     // VariantClusterGraphPath.cpp:36
constexpr unsigned MIN_NUM_SAMPLE_PATHS = 1;  // VariantClusterGraph.cpp:60

struct FPVertex {
    uint32_t index;
    uint8_t num_observed_kmers;
};
struct FPath {
    std::vector<FPVertex> path;
    std::string window;   // KmerPair: nucleotides since the last reset, at most k kept
    uint32_t score_first = 0, score_second = 0;
};
struct FindCtx {
    const OrcGraphs &g;
    uint32_t c, v0, nv;
    OrcBloom *bloom;
    uint32_t len(uint32_t vi) const { return (uint32_t)(g.seq_off[v0 + vi + 1] - g.seq_off[v0 + vi]); }
    uint8_t nt(uint32_t vi, uint32_t i) const { return g.seq[g.seq_off[v0 + vi] + i] & 3; }
    bool disconnected(uint32_t vi) const { return g.vflags[v0 + vi] & 1; }
    bool redundant(uint32_t vi) const { return g.vflags[v0 + vi] & 2; }
};

void update_score(const FindCtx &x, FPath &p, bool observed, uint32_t cur_sequence_length) {   // VariantClusterGraphPath.cpp:87-129
    if (observed) {
        p.score_first++;
        auto rit = p.path.rbegin();
        if ((cur_sequence_length > 1) or !x.redundant(rit->index)) {
            if (rit->num_observed_kmers < MIN_OBSERVED_KMERS) rit->num_observed_kmers++;
        }
        rit++;
        while (rit != p.path.rend()) {
            if ((x.g.k <= cur_sequence_length) or (rit->num_observed_kmers == MIN_OBSERVED_KMERS)) break;
            if (rit->num_observed_kmers < MIN_OBSERVED_KMERS) rit->num_observed_kmers++;
            cur_sequence_length += x.len(rit->index);
            rit++;
        }
    }
    p.score_second++;
}
void add_vertex(const FindCtx &x, FPath &p, uint32_t vi) {   // VariantClusterGraphPath.cpp:46-85
    static const char ntc[4] = {'A', 'C', 'G', 'T'};
    p.path.push_back(FPVertex{vi, 0});
    if (x.disconnected(vi)) {
        if (x.len(vi) == 0) p.path.back().num_observed_kmers = MIN_OBSERVED_KMERS;
        p.window.clear();
    }
    for (uint32_t i = 0; i < x.len(vi); i++) {
        p.window.push_back(ntc[x.nt(vi, i)]);
        if (p.window.size() > x.g.k) p.window.erase(0, 1);
        if (p.window.size() == x.g.k) {
            std::string rc(x.g.k, 'A');
            for (unsigned j = 0; j < x.g.k; j++) rc[j] = comp(p.window[x.g.k - 1 - j]);
            const std::string &low = (rc < p.window) ? rc : p.window;
            update_score(x, p, x.bloom->subs[x.bloom->route(low.data())].containsF(low.data()), i + 1);
        }
    }
}
double kmer_score(const FPath &p) { return p.score_second > 0 ? p.score_first / static_cast<double>(p.score_second) : 1; }   // :136-148
uint32_t vertex_score(const FindCtx &x, const FPath &p, const std::vector<bool> &covered, bool is_complete) {            // :150-188
    uint32_t score = 0;
    for (auto &v : p.path)
        if ((v.num_observed_kmers == MIN_OBSERVED_KMERS) and !covered.at(v.index)) score++;
    if (!is_complete) {
        uint32_t cur = 0;
        for (auto rit = p.path.rbegin(); rit != p.path.rend(); rit++) {
            if (((x.g.k - 1) <= cur) or (rit->num_observed_kmers == MIN_OBSERVED_KMERS)) break;
            if (!covered.at(rit->index)) score++;
            cur += x.len(rit->index);
        }
    }
    return score;
}
void update_covered(const FindCtx &x, const FPath &p, std::vector<bool> &covered, bool is_complete) {   // :190-225
    for (auto &v : p.path)
        if (v.num_observed_kmers == MIN_OBSERVED_KMERS) covered.at(v.index) = true;
    if (!is_complete) {
        uint32_t cur = 0;
        for (auto rit = p.path.rbegin(); rit != p.path.rend(); rit++) {
            if (((x.g.k - 1) <= cur) or (rit->num_observed_kmers == MIN_OBSERVED_KMERS)) break;
            covered.at(rit->index) = true;
            cur += x.len(rit->index);
        }
    }
}
// isPathsRedundant (VariantClusterGraph.cpp:525-626): the two vertex lists spell the same nucleotides and separators, read backwards
template <typename P1, typename P2>
bool paths_redundant(const FindCtx &x, const P1 &p1, const P2 &p2) {
    // positions as (index into the path from the back, nucleotides still unread in that vertex)
    long i1 = (long)p1.size() - 1, i2 = (long)p2.size() - 1;
    uint32_t r1 = x.len(p1[i1].index), r2 = x.len(p2[i2].index);
    bool d1 = false, d2 = false;
    while (true) {
        while (r1 == 0) {
            if (x.disconnected(p1[i1].index)) d1 = true;
            i1--;
            if (i1 >= 0) r1 = x.len(p1[i1].index);
            else break;
        }
        while (r2 == 0) {
            if (x.disconnected(p2[i2].index)) d2 = true;
            i2--;
            if (i2 >= 0) r2 = x.len(p2[i2].index);
            else break;
        }
        if (d1 != d2) return false;
        d1 = false;
        d2 = false;
        if ((i1 < 0) or (i2 < 0)) break;
        if (p1[i1].index == p2[i2].index and r1 == r2) {   // the same position of the same vertex: the rest of this vertex is shared
            r1 = 0;
            r2 = 0;
        }
        while ((r1 != 0) and (r2 != 0)) {
            if (x.nt(p1[i1].index, r1 - 1) != x.nt(p2[i2].index, r2 - 1)) return false;
            r1--;
            r2--;
        }
    }
    if ((i1 >= 0) or (i2 >= 0)) return false;
    return true;
}
void merge_paths(const FindCtx &x, std::vector<FPath> &main_paths, const std::vector<FPath> &input_paths) {   // :474-523 (copies; the move is an optimisation)
    const size_t main_size = main_paths.size();
    for (auto &in : input_paths) {
        bool is_redundant = false;
        for (size_t m = 0; m < main_size; m++) {
            if (paths_redundant(x, main_paths[m].path, in.path)) {
                if (main_paths[m].path.size() < in.path.size()) main_paths[m] = in;
                is_redundant = true;
                break;
            }
        }
        if (!is_redundant) main_paths.push_back(in);
    }
}
bool double_compare(double a, double b) { return ((a == b) or (std::abs(a - b) < std::abs(std::min(a, b)) * std::numeric_limits<double>::epsilon() * 100)); }
void filter_paths(const FindCtx &x, std::vector<FPath> &paths, uint32_t max_paths, bool is_complete) {   // :628-724
    if (!((paths.size() > max_paths) or (is_complete and (paths.size() > MIN_NUM_SAMPLE_PATHS)))) return;
    bool is_first_pass = true;
    std::vector<bool> covered(x.nv, false);
    size_t sorted_end = 0;
    while (sorted_end != paths.size()) {
        size_t best = sorted_end;
        double best_kmer = kmer_score(paths[best]);
        uint32_t best_vertex = vertex_score(x, paths[best], covered, is_complete);
        for (size_t it = sorted_end + 1; it < paths.size(); it++) {
            const double cur_kmer = kmer_score(paths[it]);
            const uint32_t cur_vertex = vertex_score(x, paths[it], covered, is_complete);
            if (is_first_pass) {
                if (cur_vertex > 0) {
                    if ((double_compare(cur_kmer, best_kmer) and (cur_vertex > best_vertex)) or (cur_kmer > best_kmer) or (best_vertex == 0)) {
                        best = it;
                        best_kmer = cur_kmer;
                        best_vertex = cur_vertex;
                    }
                }
            } else if (!is_complete or (cur_vertex == paths[it].path.size())) {
                if (cur_kmer > best_kmer) {
                    best = it;
                    best_kmer = cur_kmer;
                    best_vertex = cur_vertex;
                }
            }
        }
        if (is_first_pass) {
            update_covered(x, paths[best], covered, is_complete);
        } else if (is_complete and (sorted_end >= MIN_NUM_SAMPLE_PATHS) and (best_vertex < paths[best].path.size())) {
            break;
        }
        if (sorted_end != best) std::swap(paths[sorted_end], paths[best]);
        if (is_first_pass and (best_vertex == 0)) {
            is_first_pass = false;
            std::fill(covered.begin(), covered.end(), false);
        } else {
            sorted_end++;
            if (sorted_end == max_paths) break;
        }
    }
    paths.resize(sorted_end);
}
struct OrcFind {
    std::vector<std::vector<std::vector<uint8_t>>> best;   // per cluster: rows of |V| flags
};
}  // namespace

extern "C" {
void *orc_find_new(void *gh) {
    OrcFind *f = new OrcFind();
    f->best.resize(((OrcGraphs *)gh)->C);
    return f;
}
void orc_find_free(void *h) { delete (OrcFind *)h; }
// one sample: findSamplePaths for every cluster with seeds[c] (= prng_seed + (group_idx+1)*(sample_idx+1) + variant_cluster_idx,
// KmerCounter.cpp:59-69 + VariantClusterGroup.cpp:138-144), then addPathIndices
void orc_find_sample_paths(void *fh, void *gh, void *bloom, const uint32_t *seeds, uint32_t max_sample_haplotypes) {
    OrcFind *F = (OrcFind *)fh;
    const OrcGraphs &g = *(OrcGraphs *)gh;
    for (uint32_t c = 0; c < g.C; c++) {
        FindCtx x{g, c, g.vertex_off[c], g.vertex_off[c + 1] - g.vertex_off[c], (OrcBloom *)bloom};
        std::mt19937 prng(seeds[c]);
        std::vector<std::vector<FPath>> vertex_paths(x.nv);
        for (uint32_t vi = 0; vi < x.nv; vi++) {
            auto &cur = vertex_paths[vi];
            const uint32_t e0 = g.in_off[x.v0 + vi], e1 = g.in_off[x.v0 + vi + 1];
            if (e0 == e1) cur.emplace_back();
            else
                for (uint32_t e = e0; e < e1; e++) merge_paths(x, cur, vertex_paths[g.in_src[e]]);
            std::shuffle(cur.begin(), cur.end(), prng);
            for (auto &p : cur) add_vertex(x, p, vi);
            filter_paths(x, cur, max_sample_haplotypes, false);
        }
        auto &final_paths = vertex_paths[x.nv - 1];
        filter_paths(x, final_paths, max_sample_haplotypes, true);
        // addPathIndices (:726-798)
        auto &best = F->best[c];
        std::vector<bool> redundant(final_paths.size(), false);
        for (auto &row : best) {
            std::vector<FPVertex> cur_best;
            for (uint32_t vi = 0; vi < x.nv; vi++)
                if (row[vi]) cur_best.push_back(FPVertex{vi, 0});
            for (size_t pi = 0; pi < final_paths.size(); pi++) {
                if (redundant[pi]) continue;
                if (paths_redundant(x, cur_best, final_paths[pi].path)) {
                    if (cur_best.size() < final_paths[pi].path.size()) {
                        std::fill(row.begin(), row.end(), 0);
                        for (auto &v : final_paths[pi].path) row[v.index] = 1;
                    }
                    redundant[pi] = true;
                    break;
                }
            }
        }
        for (size_t pi = 0; pi < final_paths.size(); pi++)
            if (!redundant[pi]) {
                std::vector<uint8_t> row(x.nv, 0);
                for (auto &v : final_paths[pi].path) row[v.index] = 1;
                best.push_back(row);
            }
    }
}
void orc_find_sizes(void *fh, uint32_t *num_paths) {
    OrcFind *F = (OrcFind *)fh;
    for (size_t c = 0; c < F->best.size(); c++) num_paths[c] = (uint32_t)F->best[c].size();
}
void orc_find_fetch(void *fh, uint8_t *out) {
    OrcFind *F = (OrcFind *)fh;
    for (auto &b : F->best)
        for (auto &row : b) {
            memcpy(out, row.data(), row.size());
            out += row.size();
        }
}
}  // extern "C"
