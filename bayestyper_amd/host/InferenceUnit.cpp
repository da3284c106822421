#include "InferenceUnit.hpp"
#include <atomic>
#include <vector>
#include <fstream>
#include "Parallel.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace bthost {

// one piece of a raw deflate stream (no gzip wrapper): ended with a sync flush — byte-aligned, not final — or, the last piece, with the final block
static std::string deflatePiece(const char *data, size_t len, bool last) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    std::string out(deflateBound(&zs, (uLong)len) + 64, '\0');
    zs.next_in = (Bytef *)data;
    zs.avail_in = (uInt)len;
    zs.next_out = (Bytef *)&out[0];
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    const size_t n = out.size() - zs.avail_out;
    const bool ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0);
    deflateEnd(&zs);
    if (!ok) throw std::runtime_error("deflate failed");
    out.resize(n);
    return out;
}

// Large contents (the parameter k-mer FASTA: 56 MB, four seconds of single-thread deflate at a chr20-sized unit) are compressed in pieces of 4 MB of input on
// `threads` host threads and written as ONE gzip member: every piece is a stretch of the same raw deflate stream, closed with a sync flush (what pigz does), the
// CRC-32 of the whole content in the trailer.  Any gzip reader reads it, and the bytes do not depend on the number of threads (the pieces are cut by size).
// Small contents go through gzwrite like the reference's single stream.
void writeGzFile(const std::string &filename, const std::string &content, unsigned threads) {
    const size_t piece = 4u << 20;
    if (content.size() > 2 * piece) {
        const size_t parts = (content.size() + piece - 1) / piece;
        std::vector<std::string> blocks(parts);
        std::vector<uLong> crcs(parts);
        parallelFor(parts, std::max(1u, threads), [&](size_t a, size_t b, unsigned) {
            for (size_t i = a; i < b; i++) {
                const size_t len = std::min(piece, content.size() - i * piece);
                blocks[i] = deflatePiece(content.data() + i * piece, len, i + 1 == parts);
                crcs[i] = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)content.data() + i * piece, (uInt)len);
            }
        });
        uLong crc = crc32(0L, Z_NULL, 0);
        for (size_t i = 0; i < parts; i++) crc = crc32_combine(crc, crcs[i], (z_off_t)std::min(piece, content.size() - i * piece));
        std::ofstream f(filename, std::ios::binary);
        if (!f.is_open()) throw std::runtime_error("Unable to write file " + filename);
        const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};   // deflate, no flags, no time stamp, unix
        f.write((const char *)header, 10);
        for (auto &m : blocks) f.write(m.data(), (std::streamsize)m.size());
        const uint32_t trailer[2] = {(uint32_t)crc, (uint32_t)(content.size() & 0xFFFFFFFFu)};
        f.write((const char *)trailer, 8);
        f.close();
        if (!f) throw std::runtime_error("Error while writing " + filename);
        return;
    }
    gzFile f = gzopen(filename.c_str(), "wb");
    if (!f) throw std::runtime_error("Unable to write file " + filename);
    size_t at = 0;
    bool ok = true;
    while (ok && at < content.size()) {
        const unsigned n = (unsigned)std::min<size_t>(content.size() - at, 1u << 30);
        ok = gzwrite(f, content.data() + at, n) == (int)n;
        at += n;
    }
    if (gzclose(f) != Z_OK || !ok) throw std::runtime_error("Error while writing " + filename);
}

std::vector<uint64_t> parseKmerLines(const std::string &text, size_t begin, uint32_t k, unsigned threads) {
    auto parse = [&](const char *line, size_t len, uint64_t *out) {
        uint64_t lo = 0, hi = 0;
        bool ok = len == k;
        for (size_t i = 0; ok && i < len; i++) {
            uint64_t c = 0;
            switch (line[i]) {
                case 'A': c = 0; break;
                case 'C': c = 1; break;
                case 'G': c = 2; break;
                case 'T': c = 3; break;
                default: ok = false;
            }
            if (i < 32) lo |= c << (2 * i);
            else hi |= c << (2 * (i - 32));
        }
        if (!ok) throw std::runtime_error("malformed kmer line: " + std::string(line, std::min<size_t>(len, 200)));
        out[0] = lo;
        out[1] = hi;
    };
    std::vector<uint64_t> kmers;
    if (begin >= text.size()) return kmers;
    const size_t body = text.size() - begin, w = (size_t)k + 1;
    // regular: every line k symbols + '\n' (the last newline may be missing)
    const size_t n_reg = (body + 1) / w;
    bool regular = k > 0 && (body % w == 0 || body % w == k);
    if (regular) {
        std::atomic<bool> all(true);
        parallelFor(n_reg, std::max(1u, threads), [&](size_t a, size_t b, unsigned) {
            for (size_t i = a; i < b; i++)
                if (begin + i * w + k < text.size() && text[begin + i * w + k] != '\n') {
                    all.store(false);
                    return;
                }
        });
        regular = all.load();
    }
    if (regular) {
        kmers.resize(2 * n_reg);
        parallelFor(n_reg, std::max(1u, threads), [&](size_t a, size_t b, unsigned) {
            for (size_t i = a; i < b; i++) parse(text.data() + begin + i * w, k, &kmers[2 * i]);
        });
        return kmers;
    }
    size_t at = begin;   // lines of any length: one after the other (the first malformed one is reported)
    while (at < text.size()) {
        size_t e = text.find('\n', at);
        if (e == std::string::npos) e = text.size();
        kmers.resize(kmers.size() + 2);
        parse(text.data() + at, e - at, &kmers[kmers.size() - 2]);
        at = e + 1;
    }
    return kmers;
}

std::string readGzFile(const std::string &filename) {
    gzFile f = gzopen(filename.c_str(), "rb");
    if (!f) throw std::runtime_error("Unable to open file " + filename);
    std::string out;
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
    const bool bad = n < 0;
    gzclose(f);
    if (bad) throw std::runtime_error("Error while reading " + filename);
    return out;
}

namespace {
const char kMagic[] = "BTAMDUNIT1";

struct Writer {
    std::string b;
    void u8(uint8_t v) { b.push_back((char)v); }
    void u32(uint32_t v) { b.append((const char *)&v, 4); }
    void u64(uint64_t v) { b.append((const char *)&v, 8); }
    void str(const std::string &s) {
        u64(s.size());
        b.append(s);
    }
};
struct Reader {
    const std::string &b;
    size_t at = 0;
    void need(size_t n) const {
        if (at + n > b.size()) throw std::runtime_error("variant clusters file is truncated");
    }
    uint8_t u8() {
        need(1);
        return (uint8_t)b[at++];
    }
    uint32_t u32() {
        need(4);
        uint32_t v;
        std::memcpy(&v, b.data() + at, 4);
        at += 4;
        return v;
    }
    uint64_t u64() {
        need(8);
        uint64_t v;
        std::memcpy(&v, b.data() + at, 8);
        at += 8;
        return v;
    }
    std::string str() {
        const uint64_t n = u64();
        need(n);
        std::string s = b.substr(at, n);
        at += n;
        return s;
    }
};
}  // namespace

void InferenceUnit::write(const std::string &filename) const {
    Writer w;
    w.b.append(kMagic, sizeof(kMagic));
    w.u32(index);
    w.str(cluster_options_header);
    w.u32(num_variants);
    w.u32(num_variant_clusters);
    w.u64(num_path_kmers);
    w.u64(variant_cluster_groups.size());
    for (size_t g = 0; g < variant_cluster_groups.size(); g++) {
        const ClusterGroup &grp = variant_cluster_groups[g];
        w.str(grp.chrom_name);
        w.u32(grp.start_position);
        w.u32(grp.end_position);
        w.u32(grp.num_variants);
        w.u64(grp.source_vertices.size());
        for (uint32_t s : grp.source_vertices) w.u32(s);
        w.u64(grp.clusters.size());
        for (size_t v = 0; v < grp.clusters.size(); v++) {
            const VariantCluster &c = grp.clusters[v];
            w.u32(c.cluster_idx);
            w.u32(c.left_flank);
            w.u32(c.right_flank);
            w.str(c.chrom_name);
            w.u64(c.variants.size());
            for (auto &pv : c.variants) {
                w.u32(pv.first);
                w.str(pv.second.id);
                w.u8(pv.second.has_dependency ? 1 : 0);
                w.u8((uint8_t)pv.second.type);
                w.u32(pv.second.num_redundant_nucleotides);
                w.u64(pv.second.alt_alleles.size());
                for (auto &a : pv.second.alt_alleles) {
                    w.u32(a.ref_length);
                    w.str(a.sequence);
                    w.str(a.aco_att);
                }
            }
            w.u64(c.contained_clusters.size());
            for (auto &cc : c.contained_clusters) {
                w.u32(cc.cluster_idx);
                w.u32(cc.left_flank);
                w.u32(cc.right_flank);
            }
            w.u64(grp.out_edges[v].size());
            for (uint32_t e : grp.out_edges[v]) w.u32(e);
            const auto &paths = best_paths.at(g).at(v);
            w.u64(paths.size());
            w.u64(paths.empty() ? 0 : paths[0].size());
            for (auto &row : paths) w.b.append((const char *)row.data(), row.size());
        }
    }
    writeGzFile(filename, w.b);
}

InferenceUnit InferenceUnit::read(const std::string &filename) {
    const std::string data = readGzFile(filename);
    if (data.size() < sizeof(kMagic) || std::memcmp(data.data(), kMagic, sizeof(kMagic)) != 0)
        throw std::runtime_error(filename + " is not a variant clusters file of this build (a file written by the reference's Boost archive cannot be read: the formats differ)");
    Reader r{data, sizeof(kMagic)};
    InferenceUnit u;
    u.index = r.u32();
    u.cluster_options_header = r.str();
    u.num_variants = r.u32();
    u.num_variant_clusters = r.u32();
    u.num_path_kmers = r.u64();
    const uint64_t G = r.u64();
    u.variant_cluster_groups.resize(G);
    u.best_paths.resize(G);
    for (uint64_t g = 0; g < G; g++) {
        ClusterGroup &grp = u.variant_cluster_groups[g];
        grp.chrom_name = r.str();
        grp.start_position = r.u32();
        grp.end_position = r.u32();
        grp.num_variants = r.u32();
        grp.source_vertices.resize(r.u64());
        for (auto &s : grp.source_vertices) s = r.u32();
        const uint64_t C = r.u64();
        grp.clusters.resize(C);
        grp.out_edges.resize(C);
        u.best_paths[g].resize(C);
        for (uint64_t v = 0; v < C; v++) {
            VariantCluster &c = grp.clusters[v];
            c.cluster_idx = r.u32();
            c.left_flank = r.u32();
            c.right_flank = r.u32();
            c.chrom_name = r.str();
            for (uint64_t i = 0, n = r.u64(); i < n; i++) {
                const uint32_t pos = r.u32();
                Variant var;
                var.id = r.str();
                var.has_dependency = r.u8() != 0;
                var.type = (VariantType)r.u8();
                var.num_redundant_nucleotides = r.u32();
                var.alt_alleles.resize(r.u64());
                for (auto &a : var.alt_alleles) {
                    a.ref_length = r.u32();
                    a.sequence = r.str();
                    a.aco_att = r.str();
                }
                c.variants.emplace(pos, std::move(var));
            }
            for (uint64_t i = 0, n = r.u64(); i < n; i++) {
                ContainedCluster cc;
                cc.cluster_idx = r.u32();
                cc.left_flank = r.u32();
                cc.right_flank = r.u32();
                c.contained_clusters.push_back(cc);
            }
            grp.out_edges[v].resize(r.u64());
            for (auto &e : grp.out_edges[v]) e = r.u32();
            const uint64_t P = r.u64(), NV = r.u64();
            u.best_paths[g][v].assign(P, std::vector<uint8_t>(NV));
            for (auto &row : u.best_paths[g][v]) {
                r.need(NV);
                std::memcpy(row.data(), data.data() + r.at, NV);
                r.at += NV;
            }
        }
    }
    return u;
}

}  // namespace bthost
