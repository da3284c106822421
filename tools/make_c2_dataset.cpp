// A chr20-sized synthetic data set for end-to-end stage timings of `bayesTyper cluster` + `genotype` (BASELINE.json configs[1]: "GRCh38 chr20, 1 sample,
// SNV+indel candidate VCF (~200k vars), k=55"): reference, candidate VCF, samples file and a KMC1 database per sample — everything the executables read
// from disk except the sample Bloom filters, which `bayesTyperTools makeBloom` makes from the databases (tools/e2e_c2.sh).
//
//   reference   genome_len nt uniform ACGT, one chromosome "chr20s"
//   candidates  num_variants SNVs (80 %), insertions and deletions of 1..8 nt (10 % each), at least 16 nt apart (gaps 16 + geometric), so that variants within
//               k of each other form multi-variant clusters as in a real call set
//   samples     genotype per variant {0/0: 0.25, 0/1: 0.5, 1/1: 0.25}; every canonical 55-mer of the two haplotypes with count Poisson(15 x multiplicity),
//               plus error_kmers random 55-mers with count 1; KMC1 layout (prefix length 7, one counter byte: 13-byte records)
//
// build: g++ -O2 -std=c++17 -fopenmp tools/make_c2_dataset.cpp -o scratch/make_c2_dataset
//               sv_per_mille > 0: that share of the candidates are deletions of 300..3000 nt; the candidates that follow inside a deleted stretch are nested in it
//               (a haplotype that carries the deletion carries none of them): groups of several variant clusters, as structural variants make them
// usage: make_c2_dataset <out dir> [genome_len 64000000] [num_variants 200000] [num_samples 1] [error_kmers 140000000] [sv_per_mille 0]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <parallel/algorithm>
#include <random>
#include <string>
#include <vector>

typedef unsigned __int128 u128;
static const unsigned K = 55, P = 7;

struct Variant {
    uint32_t pos;          // 0-based position of the REF allele's first nucleotide
    uint32_t ref_len;      // nucleotides of the reference the ALT allele replaces
    std::vector<uint8_t> alt;
};

static void kmers_of(const std::vector<uint8_t> &seq, std::vector<u128> *out) {
    if (seq.size() < K) return;
    const u128 mask = (((u128)1) << (2 * K)) - 1;
    u128 fwd = 0, rc = 0;
    for (size_t i = 0; i < seq.size(); i++) {
        fwd = ((fwd << 2) | seq[i]) & mask;                         // first nucleotide in the most significant bits: numeric order = KMC's (ASCII) order
        rc = (rc >> 2) | ((u128)(3 - seq[i]) << (2 * (K - 1)));
        if (i + 1 >= K) out->push_back(fwd < rc ? fwd : rc);
    }
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: make_c2_dataset <out dir> [genome_len] [num_variants] [num_samples] [error_kmers] [sv_per_mille]\n");
        return 2;
    }
    const std::string dir = argv[1];
    const uint64_t L = argc > 2 ? strtoull(argv[2], nullptr, 10) : 64000000ull;
    const uint32_t NV = argc > 3 ? (uint32_t)atoi(argv[3]) : 200000u;
    const unsigned NS = argc > 4 ? (unsigned)atoi(argv[4]) : 1u;
    const uint64_t NE = argc > 5 ? strtoull(argv[5], nullptr, 10) : 140000000ull;
    const unsigned SV_PM = argc > 6 ? (unsigned)atoi(argv[6]) : 0u;
    std::mt19937_64 rng(1);
    std::vector<uint8_t> genome(L);
    for (uint64_t i = 0; i < L; i += 32) {
        uint64_t r = rng();
        for (uint64_t j = i; j < std::min(L, i + 32); j++, r >>= 2) genome[j] = (uint8_t)(r & 3);
    }
    {
        std::ofstream f(dir + "/genome.fa");
        f << ">chr20s synthetic\n";
        std::string line;
        for (uint64_t i = 0; i < L; i += 60) {
            line.clear();
            for (uint64_t j = i; j < std::min(L, i + 60); j++) line += "ACGT"[genome[j]];
            f << line << "\n";
        }
    }
    // ---- candidates ----
    std::vector<Variant> vars;
    {
        rng.seed(2);
        const double mean_gap = std::max(1.0, (double)(L - 400) / NV - 16.0);
        std::geometric_distribution<uint64_t> gap(1.0 / mean_gap);
        uint64_t pos = 100;
        std::ofstream f(dir + "/candidates.vcf");
        f << "##fileformat=VCFv4.2\n##contig=<ID=chr20s,length=" << L << ">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
        for (uint32_t i = 0; i < NV; i++) {
            pos += 16 + gap(rng);
            if (pos + 200 >= L) break;
            Variant v;
            v.pos = (uint32_t)pos;
            unsigned kind = (unsigned)(rng() % 10);
            if (SV_PM && rng() % 1000 < SV_PM) kind = 10;
            std::string ref(1, "ACGT"[genome[pos]]), alt;
            if (kind == 10) {   // structural variant: a long deletion after the anchor nucleotide (the next candidates fall inside it)
                const unsigned n = 300 + (unsigned)(rng() % 2701);
                if (pos + n + 200 >= L) break;
                v.ref_len = 1 + n;
                v.alt = {genome[pos]};
                for (unsigned j = 1; j <= n; j++) ref += "ACGT"[genome[pos + j]];
                alt = ref.substr(0, 1);
            } else if (kind < 8) {   // SNV
                v.ref_len = 1;
                v.alt = {(uint8_t)((genome[pos] + 1 + rng() % 3) % 4)};
                alt = std::string(1, "ACGT"[v.alt[0]]);
            } else if (kind == 8) {   // insertion after the anchor nucleotide
                const unsigned n = 1 + (unsigned)(rng() % 8);
                v.ref_len = 1;
                v.alt = {genome[pos]};
                for (unsigned j = 0; j < n; j++) v.alt.push_back((uint8_t)(rng() & 3));
                for (uint8_t c : v.alt) alt += "ACGT"[c];
            } else {   // deletion of the nucleotides after the anchor
                const unsigned n = 1 + (unsigned)(rng() % 8);
                v.ref_len = 1 + n;
                v.alt = {genome[pos]};
                for (unsigned j = 1; j <= n; j++) ref += "ACGT"[genome[pos + j]];
                alt = ref.substr(0, 1);
            }
            f << "chr20s\t" << pos + 1 << "\tv" << i << "\t" << ref << "\t" << alt << "\t.\t.\t.\n";
            vars.push_back(std::move(v));
        }
    }
    std::fprintf(stderr, "%zu variants over %llu nt\n", vars.size(), (unsigned long long)L);
    // ---- samples ----
    std::ofstream sf(dir + "/samples.tsv");
    for (unsigned s = 0; s < NS; s++) {
        rng.seed(3 + s);
        std::vector<uint8_t> copies(vars.size()), which(vars.size());
        for (size_t i = 0; i < vars.size(); i++) {
            const unsigned r = (unsigned)(rng() % 4);
            copies[i] = r == 0 ? 0 : (r == 3 ? 2 : 1);
            which[i] = (uint8_t)(rng() & 1);
        }
        std::vector<u128> km;
        km.reserve(2 * L + NE);
        for (unsigned h = 0; h < 2; h++) {
            std::vector<uint8_t> hap;
            hap.reserve(L + 1024);
            uint64_t at = 0;
            for (size_t i = 0; i < vars.size(); i++) {
                if (!(copies[i] == 2 || (copies[i] == 1 && which[i] == h))) continue;
                if (vars[i].pos < at) continue;   // inside a deletion this haplotype carries
                hap.insert(hap.end(), genome.begin() + at, genome.begin() + vars[i].pos);
                hap.insert(hap.end(), vars[i].alt.begin(), vars[i].alt.end());
                at = (uint64_t)vars[i].pos + vars[i].ref_len;
            }
            hap.insert(hap.end(), genome.begin() + at, genome.end());
            kmers_of(hap, &km);
        }
        const uint64_t n_genome = km.size();
        __gnu_parallel::sort(km.begin(), km.end());
        // distinct genome k-mers with their multiplicity -> (k-mer, count)
        std::vector<u128> keys;
        std::vector<uint8_t> counts;
        keys.reserve(n_genome + NE);
        counts.reserve(n_genome + NE);
        rng.seed(4 + s);
        for (uint64_t i = 0; i < n_genome;) {
            uint64_t j = i;
            while (j < n_genome && km[j] == km[i]) j++;
            std::poisson_distribution<unsigned> cnt(15.0 * (double)(j - i));
            const unsigned c = cnt(rng);
            if (c) {
                keys.push_back(km[i]);
                counts.push_back((uint8_t)std::min(c, 255u));
            }
            i = j;
        }
        km.clear();
        km.shrink_to_fit();
        // sequencing-error k-mers: random canonical 55-mers, count 1 (merged where they hit a present k-mer)
        {
            const u128 mask = (((u128)1) << (2 * K)) - 1;
            std::vector<u128> err(NE);
            for (uint64_t i = 0; i < NE; i++) {
                const u128 f = ((((u128)rng()) << 64) | rng()) & mask;
                u128 rc = 0, t = f;
                for (unsigned j = 0; j < K; j++, t >>= 2) rc = (rc << 2) | (3 - (unsigned)(t & 3));
                err[i] = f < rc ? f : rc;
            }
            __gnu_parallel::sort(err.begin(), err.end());
            err.erase(std::unique(err.begin(), err.end()), err.end());
            std::vector<u128> mk(keys.size() + err.size());
            std::vector<uint8_t> mc(mk.size());
            size_t a = 0, b = 0, o = 0;
            while (a < keys.size() || b < err.size()) {
                if (b == err.size() || (a < keys.size() && keys[a] < err[b])) mk[o] = keys[a], mc[o++] = counts[a++];
                else if (a == keys.size() || err[b] < keys[a]) mk[o] = err[b++], mc[o++] = 1;
                else mk[o] = keys[a], mc[o++] = (uint8_t)std::min(255, counts[a++] + 1), b++;
            }
            mk.resize(o);
            mc.resize(o);
            keys.swap(mk);
            counts.swap(mc);
        }
        // ---- KMC1 database: "KMCP" | prefix table (4^P x u64: first record of every prefix) | header (64 B) | header offset | "KMCP";  "KMCS" | records | "KMCS" ----
        const std::string prefix = dir + "/sample" + std::to_string(s + 1);
        const uint64_t n = keys.size(), nlut = 1ull << (2 * P);
        std::vector<uint64_t> lut(nlut, 0), per(nlut, 0);
        const unsigned SB = (K - P) / 4;
        std::vector<uint8_t> suf(n * (SB + 1));
        for (uint64_t i = 0; i < n; i++) {
            per[(uint64_t)(keys[i] >> (2 * (K - P)))]++;
            for (unsigned b = 0; b < SB; b++) suf[i * (SB + 1) + b] = (uint8_t)(keys[i] >> (8 * (SB - 1 - b)));
            suf[i * (SB + 1) + SB] = counts[i];
        }
        uint64_t acc = 0;
        for (uint64_t j = 0; j < nlut; j++) {
            lut[j] = acc;
            acc += per[j];
        }
        std::ofstream fp(prefix + ".kmc_pre", std::ios::binary), fs(prefix + ".kmc_suf", std::ios::binary);
        fp.write("KMCP", 4);
        fp.write((const char *)lut.data(), (std::streamsize)(nlut * 8));
        uint64_t header[8] = {0};
        header[0] = K;                                  // k | mode << 32
        header[1] = 1ull | ((uint64_t)P << 32);         // counter size | prefix length << 32
        header[2] = 1ull | (255ull << 32);              // min | max << 32
        header[3] = n;
        fp.write((const char *)header, 64);
        const uint32_t header_offset = 64;
        fp.write((const char *)&header_offset, 4);
        fp.write("KMCP", 4);
        fs.write("KMCS", 4);
        fs.write((const char *)suf.data(), (std::streamsize)suf.size());
        fs.write("KMCS", 4);
        sf << "sample" << s + 1 << "\tF\t" << prefix << "\n";
        std::fprintf(stderr, "sample%u: %llu KMC records (%llu genome k-mer occurrences)\n", s + 1, (unsigned long long)n, (unsigned long long)n_genome);
    }
    return 0;
}
