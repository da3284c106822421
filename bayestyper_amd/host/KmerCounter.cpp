#include "KmerCounter.hpp"

#include <chrono>
#include "StageTimes.hpp"
#include "Parallel.hpp"

#include <algorithm>
#include <functional>
#include <future>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "Comm.hpp"
#include "KmcFile.hpp"
#include "Options.hpp"

namespace bthost {

namespace {
void check(int rc, const char *what) {
    if (rc != BT_OK) throw std::runtime_error(std::string(what) + ": " + bt_last_error());
}
// a host buffer copied to the device for the duration of a pass
struct DeviceCopy {
    bt_ctx *ctx;
    void *d = nullptr;
    DeviceCopy(bt_ctx *ctx_in, const void *h, size_t bytes) : ctx(ctx_in) {
        check(bt_malloc(ctx, std::max<size_t>(bytes, 16), &d), "bt_malloc");
        if (bytes) check(bt_memcpy_h2d(ctx, d, h, bytes), "bt_memcpy_h2d");
    }
    ~DeviceCopy() { bt_free(ctx, d); }
};
}  // namespace

UnitGraphs::UnitGraphs(const InferenceUnit &unit, const Chromosomes &chromosomes, unsigned kmer_size, unsigned threads) {
    std::vector<const VariantCluster *> clusters;
    std::vector<const std::string *> sequences;
    for (uint32_t g = 0; g < unit.variant_cluster_groups.size(); g++) {
        const ClusterGroup &grp = unit.variant_cluster_groups[g];
        group_first.push_back((uint32_t)clusters.size());
        for (uint32_t v = 0; v < grp.clusters.size(); v++) {
            const VariantCluster &c = grp.clusters[v];
            const int chrom = chromosomes.find(c.chrom_name);
            if (chrom < 0) throw std::runtime_error("chromosome " + c.chrom_name + " of a variant cluster is not in the genome");
            clusters.push_back(&c);
            sequences.push_back(&chromosomes.sequence((size_t)chrom));
            cluster_group.push_back(g);
            cluster_vertex.push_back(v);
        }
    }
    group_first.push_back((uint32_t)clusters.size());
    // every range of clusters into its own vector, the vectors joined in order
    const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, clusters.size() / 1024 + 1));
    std::vector<std::vector<VariantClusterGraph>> built(parts);
    parallelFor(clusters.size(), parts, [&](size_t a, size_t b, unsigned part) {
        built[part].reserve(b - a);
        for (size_t i = a; i < b; i++) built[part].emplace_back(*clusters[i], *sequences[i], kmer_size);
    });
    graphs.reserve(clusters.size());
    for (auto &part : built)
        for (auto &g : part) graphs.push_back(std::move(g));
}

bt_gibbs_batch GibbsBatchData::view() const {
    bt_gibbs_batch b{};
    b.num_groups = numGroups();
    b.num_clusters = numClusters();
    static const uint32_t zero32 = 0;
    static const uint16_t zero16 = 0;
    static const uint8_t zero8 = 0;
    static const int32_t zeroi = 0;
    auto p32 = [](const std::vector<uint32_t> &v) { return v.empty() ? &zero32 : v.data(); };
    auto p16 = [](const std::vector<uint16_t> &v) { return v.empty() ? &zero16 : v.data(); };
    auto p8 = [](const std::vector<uint8_t> &v) { return v.empty() ? &zero8 : v.data(); };
    b.group_index = p32(group_index);
    b.group_cluster_off = p32(group_cluster_off);
    b.group_ploidy = p8(group_ploidy);
    b.group_source_off = p32(group_source_off);
    b.group_sources = p32(group_sources);
    b.group_num_shared = p32(group_num_shared);
    b.cluster_idx = p32(cluster_idx);
    b.edge_off = p32(edge_off);
    b.edges = p32(edges);
    b.num_haplotypes = p32(num_haplotypes);
    b.num_variants = p32(num_variants);
    b.kmer_off = p32(kmer_off);
    b.hap_kmer_mult = p8(hap_kmer_mult);
    b.kmer_has_counts = p8(kmer_has_counts);
    b.kmer_counts = p8(kmer_counts);
    b.kmer_ic_mult = p8(kmer_ic_mult);
    b.kmer_shared = kmer_shared.empty() ? &zeroi : kmer_shared.data();
    b.kv_off = p32(kv_off);
    b.kv_var = p16(kv_var);
    b.kv_bits = p32(kv_bits);
    b.unique_off = p32(unique_off);
    b.unique_idx = p32(unique_idx);
    b.multi_off = p32(multi_off);
    b.multi_idx = p32(multi_idx);
    b.hap_allele = p16(hap_allele);
    b.hapnest_off = p32(hapnest_off);
    b.hapnest_idx = p32(hapnest_idx);
    b.var_num_alleles = p16(var_num_alleles);
    b.var_has_dependency = p8(var_has_dependency);
    b.nestdep_off = p32(nestdep_off);
    b.nestdep_cluster = p32(nestdep_cluster);
    b.nestdep_var_off = p32(nestdep_var_off);
    b.nestdep_var = p16(nestdep_var);
    return b;
}

namespace {
// appends the slice [off[i], off[i+1]) * width of `src` to `dst` and extends the offsets `dst_off`
template <typename T>
void appendSlice(std::vector<T> &dst, const std::vector<T> &src, uint64_t a, uint64_t b) {
    dst.insert(dst.end(), src.begin() + (std::ptrdiff_t)a, src.begin() + (std::ptrdiff_t)b);
}
}  // namespace

GibbsBatchData GibbsBatchData::fromView(const bt_gibbs_batch &b, uint32_t S) {
    GibbsBatchData o;
    o.S = S;
    const uint64_t G = b.num_groups, C = b.num_clusters;
    auto cp = [](auto &dst, const auto *src, uint64_t n) { dst.assign(src, src + n); };
    cp(o.group_index, b.group_index, G);
    cp(o.group_cluster_off, b.group_cluster_off, G + 1);
    cp(o.group_ploidy, b.group_ploidy, G * S);
    cp(o.group_source_off, b.group_source_off, G + 1);
    cp(o.group_sources, b.group_sources, b.group_source_off[G]);
    cp(o.group_num_shared, b.group_num_shared, G);
    cp(o.cluster_idx, b.cluster_idx, C);
    cp(o.edge_off, b.edge_off, C + 1);
    cp(o.edges, b.edges, b.edge_off[C]);
    cp(o.num_haplotypes, b.num_haplotypes, C);
    cp(o.num_variants, b.num_variants, C);
    cp(o.kmer_off, b.kmer_off, C + 1);
    const uint64_t R = b.kmer_off[C];
    uint64_t mult = 0, hv = 0, hsum = 0, vsum = 0, kvb = 0;
    for (uint64_t c = 0; c < C; c++) {
        const uint64_t H = b.num_haplotypes[c], V = b.num_variants[c], K = b.kmer_off[c + 1] - b.kmer_off[c];
        mult += K * H;
        hv += H * V;
        hsum += H;
        vsum += V;
        kvb += (uint64_t)(b.kv_off[b.kmer_off[c + 1]] - b.kv_off[b.kmer_off[c]]) * ((H + 31) / 32);
    }
    cp(o.hap_kmer_mult, b.hap_kmer_mult, mult);
    cp(o.kmer_has_counts, b.kmer_has_counts, R);
    cp(o.kmer_counts, b.kmer_counts, R * S);
    cp(o.kmer_ic_mult, b.kmer_ic_mult, R * 2);
    cp(o.kmer_shared, b.kmer_shared, R);
    cp(o.kv_off, b.kv_off, R + 1);
    cp(o.kv_var, b.kv_var, b.kv_off[R]);
    cp(o.kv_bits, b.kv_bits, kvb);
    cp(o.unique_off, b.unique_off, C + 1);
    cp(o.unique_idx, b.unique_idx, b.unique_off[C]);
    cp(o.multi_off, b.multi_off, C + 1);
    cp(o.multi_idx, b.multi_idx, b.multi_off[C]);
    cp(o.hap_allele, b.hap_allele, hv);
    cp(o.hapnest_off, b.hapnest_off, hsum + 1);
    cp(o.hapnest_idx, b.hapnest_idx, b.hapnest_off[hsum]);
    cp(o.var_num_alleles, b.var_num_alleles, vsum);
    cp(o.var_has_dependency, b.var_has_dependency, vsum);
    cp(o.nestdep_off, b.nestdep_off, C + 1);
    const uint64_t ND = b.nestdep_off[C];
    cp(o.nestdep_cluster, b.nestdep_cluster, ND);
    cp(o.nestdep_var_off, b.nestdep_var_off, ND + 1);
    cp(o.nestdep_var, b.nestdep_var, b.nestdep_var_off[ND]);
    return o;
}

GibbsBatchData GibbsBatchData::take(const std::vector<uint32_t> &ids) const {
    GibbsBatchData o;
    o.S = S;
    // prefix sums over the per-cluster quantities that have no explicit offsets
    const uint32_t C = numClusters();
    std::vector<uint64_t> mult_off(C + 1, 0), hapvar_off(C + 1, 0), hap_base(C + 1, 0), var_base(C + 1, 0), kvb_off(C + 1, 0);
    for (uint32_t c = 0; c < C; c++) {
        const uint64_t H = num_haplotypes[c], V = num_variants[c], K = kmer_off[c + 1] - kmer_off[c];
        mult_off[c + 1] = mult_off[c] + K * H;
        hapvar_off[c + 1] = hapvar_off[c] + H * V;
        hap_base[c + 1] = hap_base[c] + H;
        var_base[c + 1] = var_base[c] + V;
        kvb_off[c + 1] = kvb_off[c] + (uint64_t)(kv_off[kmer_off[c + 1]] - kv_off[kmer_off[c]]) * ((H + 31) / 32);
    }
    {   // the large arrays get their final capacity up front (appending slice by slice reallocated — and re-touched — them a dozen times over)
        uint64_t rows = 0, mult = 0, kv = 0, kvb = 0, uniq = 0, multi = 0, nclus = 0;
        for (uint32_t g : ids)
            for (uint32_t c = group_cluster_off[g]; c < group_cluster_off[g + 1]; c++) {
                rows += kmer_off[c + 1] - kmer_off[c];
                mult += mult_off[c + 1] - mult_off[c];
                kv += kv_off[kmer_off[c + 1]] - kv_off[kmer_off[c]];
                kvb += kvb_off[c + 1] - kvb_off[c];
                uniq += unique_off[c + 1] - unique_off[c];
                multi += multi_off[c + 1] - multi_off[c];
                nclus++;
            }
        o.hap_kmer_mult.reserve(mult);
        o.kmer_has_counts.reserve(rows);
        o.kmer_counts.reserve(rows * S);
        o.kmer_ic_mult.reserve(rows * 2);
        o.kmer_shared.reserve(rows);
        o.kv_off.reserve(rows + 1);
        o.kv_var.reserve(kv);
        o.kv_bits.reserve(kvb);
        o.unique_idx.reserve(uniq);
        o.multi_idx.reserve(multi);
        for (auto *v : {&o.cluster_idx, &o.num_haplotypes, &o.num_variants}) v->reserve(nclus);
        for (auto *v : {&o.kmer_off, &o.unique_off, &o.multi_off, &o.edge_off, &o.nestdep_off}) v->reserve(nclus + 1);
        o.group_index.reserve(ids.size());
        o.group_cluster_off.reserve(ids.size() + 1);
    }
    o.group_cluster_off.push_back(0);
    o.group_source_off.push_back(0);
    o.edge_off.push_back(0);
    o.kmer_off.push_back(0);
    o.kv_off.push_back(0);
    o.unique_off.push_back(0);
    o.multi_off.push_back(0);
    o.hapnest_off.push_back(0);
    o.nestdep_off.push_back(0);
    o.nestdep_var_off.push_back(0);
    for (uint32_t g : ids) {
        o.group_index.push_back(group_index[g]);
        appendSlice(o.group_ploidy, group_ploidy, (uint64_t)g * S, (uint64_t)(g + 1) * S);
        appendSlice(o.group_sources, group_sources, group_source_off[g], group_source_off[g + 1]);
        o.group_source_off.push_back((uint32_t)o.group_sources.size());
        o.group_num_shared.push_back(group_num_shared[g]);
        for (uint32_t c = group_cluster_off[g]; c < group_cluster_off[g + 1]; c++) {
            o.cluster_idx.push_back(cluster_idx[c]);
            appendSlice(o.edges, edges, edge_off[c], edge_off[c + 1]);
            o.edge_off.push_back((uint32_t)o.edges.size());
            o.num_haplotypes.push_back(num_haplotypes[c]);
            o.num_variants.push_back(num_variants[c]);
            const uint32_t r0 = kmer_off[c], r1 = kmer_off[c + 1];
            appendSlice(o.hap_kmer_mult, hap_kmer_mult, mult_off[c], mult_off[c + 1]);
            appendSlice(o.kmer_has_counts, kmer_has_counts, r0, r1);
            appendSlice(o.kmer_counts, kmer_counts, (uint64_t)r0 * S, (uint64_t)r1 * S);
            appendSlice(o.kmer_ic_mult, kmer_ic_mult, (uint64_t)r0 * 2, (uint64_t)r1 * 2);
            appendSlice(o.kmer_shared, kmer_shared, r0, r1);
            const uint32_t e0 = kv_off[r0], base = o.kv_off.back();
            for (uint32_t r = r0; r < r1; r++) o.kv_off.push_back(base + (kv_off[r + 1] - e0));
            appendSlice(o.kv_var, kv_var, e0, kv_off[r1]);
            appendSlice(o.kv_bits, kv_bits, kvb_off[c], kvb_off[c + 1]);
            o.kmer_off.push_back(o.kmer_off.back() + (r1 - r0));
            appendSlice(o.unique_idx, unique_idx, unique_off[c], unique_off[c + 1]);
            o.unique_off.push_back((uint32_t)o.unique_idx.size());
            appendSlice(o.multi_idx, multi_idx, multi_off[c], multi_off[c + 1]);
            o.multi_off.push_back((uint32_t)o.multi_idx.size());
            appendSlice(o.hap_allele, hap_allele, hapvar_off[c], hapvar_off[c + 1]);
            const uint32_t hn0 = hapnest_off[hap_base[c]], hbase = o.hapnest_off.back();
            for (uint64_t h = hap_base[c]; h < hap_base[c + 1]; h++) o.hapnest_off.push_back(hbase + (hapnest_off[h + 1] - hn0));
            appendSlice(o.hapnest_idx, hapnest_idx, hn0, hapnest_off[hap_base[c + 1]]);
            appendSlice(o.var_num_alleles, var_num_alleles, var_base[c], var_base[c + 1]);
            appendSlice(o.var_has_dependency, var_has_dependency, var_base[c], var_base[c + 1]);
            const uint32_t nd0 = nestdep_off[c], nd1 = nestdep_off[c + 1], ndv0 = nestdep_var_off[nd0], vbase = o.nestdep_var_off.back();
            appendSlice(o.nestdep_cluster, nestdep_cluster, nd0, nd1);
            for (uint32_t i = nd0; i < nd1; i++) o.nestdep_var_off.push_back(vbase + (nestdep_var_off[i + 1] - ndv0));
            appendSlice(o.nestdep_var, nestdep_var, ndv0, nestdep_var_off[nd1]);
            o.nestdep_off.push_back((uint32_t)o.nestdep_cluster.size());
        }
        o.group_cluster_off.push_back((uint32_t)o.cluster_idx.size());
    }
    return o;
}

KmerCounter::KmerCounter(bt_ctx *ctx_in, const std::vector<Sample> &samples_in, unsigned kmer_size_in, unsigned prng_seed_in)
    : ctx(ctx_in), samples(samples_in), kmer_size(kmer_size_in), prng_seed(prng_seed_in) {}

void KmerCounter::checkTable(bt_table *table, const char *stage) {
    int overflowed = 0;
    check(bt_table_status(table, nullptr, nullptr, &overflowed), "bt_table_status");
    if (overflowed) throw std::runtime_error(std::string(stage) + ": the k-mer table is full and dropped k-mers (more k-mers than it was sized for)");
}

// ---- cluster stage ----------------------------------------------------------------------------------------------------------------

void KmerCounter::findVariantClusterPaths(InferenceUnit *unit, const UnitGraphs &ug, uint16_t max_sample_haplotypes) {
    std::cout << "[" << getLocalTime() << "] Finding variant cluster paths for " << samples.size() << " sample(s) ..." << std::endl;
    PathsBatchBuilder builder;
    for (auto &g : ug.graphs) builder.add(g);
    const uint32_t C = (uint32_t)ug.graphs.size();
    bt_find_paths *fp = nullptr;
    check(bt_find_paths_create(ctx, &builder.batch(), kmer_size, max_sample_haplotypes, (uint32_t)samples.size(), &fp), "bt_find_paths_create");
    try {
        std::vector<uint32_t> seeds(C);
        // The NEXT sample's Bloom filter (KmerBloom<k>(samples[s].file): <file>.bloomMeta / .bloomData — gigabytes read and uploaded) is loaded by a helper thread on
        // a clone of the context while the current sample's search runs (BT_FIND_PATHS_NO_PREFETCH=1: one after the other, as before round 6).
        struct CtxClone {
            bt_ctx *c = nullptr;
            ~CtxClone() {
                if (c) bt_ctx_destroy(c);
            }
        } loader_ctx;
        const bool prefetch = samples.size() > 1 && !getenv("BT_FIND_PATHS_NO_PREFETCH");
        if (prefetch) check(bt_ctx_clone(ctx, &loader_ctx.c), "bt_ctx_clone");
        auto load = [&](size_t s, bt_ctx *on) {
            bt_bloom *h = nullptr;
            check(bt_bloom_load(on, samples[s].file.c_str(), kmer_size, &h), "bt_bloom_load");
            check(bt_sync(on), "bt_sync");
            return h;
        };
        std::future<bt_bloom *> coming;
        double load_wait_s = 0, search_s = 0;
        for (size_t s = 0; s < samples.size(); s++) {
            const auto t0 = std::chrono::steady_clock::now();
            BloomHandle sample_bloom;
            sample_bloom.h = coming.valid() ? coming.get() : load(s, ctx);
            if (prefetch && s + 1 < samples.size()) coming = std::async(std::launch::async, load, s + 1, loader_ctx.c);
            const auto t1 = std::chrono::steady_clock::now();
            // prng_seed + (group index + 1) * (sample index + 1) (KmerCounter.cpp:65) + variant_cluster_idx (VariantClusterGroup.cpp:142)
            for (uint32_t c = 0; c < C; c++)
                seeds[c] = prng_seed + (ug.cluster_group[c] + 1u) * (uint32_t)(s + 1) + unit->variant_cluster_groups[ug.cluster_group[c]].clusters[ug.cluster_vertex[c]].cluster_idx;
            try {
                check(bt_find_paths_sample(fp, sample_bloom.h, seeds.data()), "bt_find_paths_sample");
                check(bt_sync(ctx), "bt_sync");
            } catch (...) {
                if (coming.valid()) {   // (the loader's filter must not outlive the contexts)
                    try {
                        BloomHandle drop;
                        drop.h = coming.get();
                    } catch (...) {
                    }
                }
                throw;
            }
            const auto t2 = std::chrono::steady_clock::now();
            load_wait_s += std::chrono::duration<double>(t1 - t0).count();
            search_s += std::chrono::duration<double>(t2 - t1).count();
        }
        StageTimes::get().add("  sample Bloom filters: load, or wait for the loader thread", load_wait_s);
        StageTimes::get().add("  best-path search of every sample (device)", search_s);
        std::vector<uint32_t> num_paths(C);
        uint64_t total = 0;
        check(bt_find_paths_sizes(fp, num_paths.data(), &total), "bt_find_paths_sizes");
        std::vector<uint8_t> rows(std::max<uint64_t>(total, 1));
        check(bt_find_paths_fetch(fp, rows.data()), "bt_find_paths_fetch");
        unit->best_paths.assign(unit->variant_cluster_groups.size(), {});
        uint64_t at = 0;
        for (uint32_t c = 0; c < C; c++) {
            auto &dst = unit->best_paths[ug.cluster_group[c]];
            if (dst.size() <= ug.cluster_vertex[c]) dst.resize(unit->variant_cluster_groups[ug.cluster_group[c]].clusters.size());
            const size_t nv = ug.graphs[c].vertices.size();
            auto &paths = dst[ug.cluster_vertex[c]];
            paths.assign(num_paths[c], std::vector<uint8_t>(nv));
            for (auto &row : paths) {
                std::copy(rows.begin() + (std::ptrdiff_t)at, rows.begin() + (std::ptrdiff_t)(at + nv), row.begin());
                at += nv;
            }
        }
    } catch (...) {
        bt_find_paths_destroy(fp);
        throw;
    }
    bt_find_paths_destroy(fp);
}

static void buildPathsBatch(PathsBatchBuilder *builder, const InferenceUnit &unit, const UnitGraphs &ug) {
    for (size_t c = 0; c < ug.graphs.size(); c++) builder->add(ug.graphs[c], unit.best_paths.at(ug.cluster_group[c]).at(ug.cluster_vertex[c]));
}

void KmerCounter::countPathMultigroupKmers(bt_table *multigroup_table, bt_bloom *path_bloom, InferenceUnit *unit, const UnitGraphs &ug) {
    std::cout << "[" << getLocalTime() << "] Counting multigroup kmers in variant cluster paths ..." << std::endl;
    PathsBatchBuilder builder;
    buildPathsBatch(&builder, *unit, ug);
    PathsHandle paths;
    check(bt_paths_create(ctx, &builder.batch(), kmer_size, &paths.h, nullptr), "bt_paths_create");
    uint64_t num_kmers = 0;
    check(bt_paths_count_multigroup(paths.h, ug.cluster_group.data(), path_bloom, multigroup_table, &num_kmers), "bt_paths_count_multigroup");
    check(bt_sync(ctx), "bt_sync");
    checkTable(multigroup_table, "countPathMultigroupKmers");
    unit->num_path_kmers = num_kmers;
}

void KmerCounter::countInterclusterParameterKmers(bt_table *parameter_table, const std::vector<InterClusterRegion> &regions, const Chromosomes &chromosomes, bt_bloom *path_bloom,
                                                  float parameter_kmer_fraction) {
    std::cout << "[" << getLocalTime() << "] Counting parameter kmers in inter-cluster regions ..." << std::endl;
    // regions grouped by chromosome: one upload per chromosome; the seed of a region is prng_seed + its index in the (sorted) region vector
    std::map<std::string, std::vector<uint32_t>> by_chrom;
    for (uint32_t i = 0; i < regions.size(); i++) by_chrom[regions[i].chrom_name].push_back(i);
    for (auto &entry : by_chrom) {
        const int chrom = chromosomes.find(entry.first);
        if (chrom < 0) throw std::runtime_error("chromosome " + entry.first + " of an inter-cluster region is not in the genome");
        const std::string &seq = chromosomes.sequence((size_t)chrom);
        DeviceCopy d_seq(ctx, seq.data(), seq.size());
        std::vector<uint64_t> start, len;
        std::vector<uint8_t> decoy;
        std::vector<uint32_t> seed;
        for (uint32_t i : entry.second) {
            start.push_back(regions[i].start_position);
            len.push_back((uint64_t)regions[i].end_position - regions[i].start_position + 1);
            decoy.push_back(regions[i].is_decoy ? 1 : 0);
            seed.push_back(prng_seed + i);
        }
        check(bt_table_count_parameter_kmers(parameter_table, path_bloom, (const char *)d_seq.d, (uint32_t)start.size(), start.data(), len.data(), decoy.data(), seed.data(),
                                             parameter_kmer_fraction),
              "bt_table_count_parameter_kmers");
        check(bt_sync(ctx), "bt_sync");
    }
    checkTable(parameter_table, "countInterclusterParameterKmers");
}

// ---- genotype stage ---------------------------------------------------------------------------------------------------------------

void KmerCounter::countPathKmers(bt_bloom *path_bloom, const InferenceUnit &unit, const UnitGraphs &ug) {
    std::cout << "[" << getLocalTime() << "] Counting kmers in variant cluster paths ..." << std::endl;
    PathsBatchBuilder builder;
    buildPathsBatch(&builder, unit, ug);
    unit_paths.reset(new PathsHandle());
    check(bt_paths_create(ctx, &builder.batch(), kmer_size, &unit_paths->h, nullptr), "bt_paths_create");
    check(bt_paths_count_kmers(unit_paths->h, path_bloom), "bt_paths_count_kmers");
    check(bt_sync(ctx), "bt_sync");
}

void KmerCounter::countInterclusterKmers(bt_table *table, bt_bloom *path_bloom, const std::string &intercluster_regions_prefix, const Chromosomes &chromosomes,
                                         const ChromosomePloidy &chrom_ploidy) {
    std::cout << "[" << getLocalTime() << "] Counting kmers in inter-cluster regions and decoy sequence(s) ..." << std::endl;
    // <prefix>.txt.gz: "<chromosome>\t<is decoy>\t<start>\t<end>" (0-based, inclusive; KmerCounter.cpp:349-371)
    std::map<std::string, std::vector<InterClusterRegion>> by_chrom;
    {
        std::istringstream in(readGzFile(intercluster_regions_prefix + ".txt.gz"));
        for (std::string line; std::getline(in, line);) {
            if (line.empty()) continue;
            std::istringstream ls(line);
            std::string chrom, decoy, start, end;
            if (!std::getline(ls, chrom, '\t') || !std::getline(ls, decoy, '\t') || !std::getline(ls, start, '\t') || !std::getline(ls, end, '\t'))
                throw std::runtime_error("malformed line in " + intercluster_regions_prefix + ".txt.gz: " + line);
            by_chrom[chrom].push_back(InterClusterRegion{chrom, std::stoi(decoy) != 0, (uint32_t)std::stoul(start), (uint32_t)std::stoul(end)});
        }
    }
    for (auto &entry : by_chrom) {
        const int chrom = chromosomes.find(entry.first);
        if (chrom < 0) throw std::runtime_error("chromosome " + entry.first + " of an inter-cluster region is not in the genome");
        const std::string &seq = chromosomes.sequence((size_t)chrom);
        DeviceCopy d_seq(ctx, seq.data(), seq.size());
        std::vector<uint64_t> start, len;
        std::vector<uint8_t> decoy, female, male;
        for (auto &r : entry.second) {
            if (r.end_position >= seq.size() || r.start_position > r.end_position) throw std::runtime_error("inter-cluster region outside chromosome " + entry.first);
            uint32_t f = 0, m = 0;
            if (!r.is_decoy) {
                const auto &gp = chrom_ploidy.getGenderPloidy(r.chrom_name);
                f = gp[0];
                m = gp[1];
            }
            start.push_back(r.start_position);
            len.push_back((uint64_t)r.end_position - r.start_position + 1);
            decoy.push_back(r.is_decoy ? 1 : 0);
            female.push_back((uint8_t)f);
            male.push_back((uint8_t)m);
        }
        // every region of the chromosome in one launch (a call per region is a kernel launch per region)
        check(bt_table_count_intercluster_regions(table, path_bloom, (const char *)d_seq.d, (uint32_t)start.size(), start.data(), len.data(), decoy.data(), female.data(), male.data()),
              "bt_table_count_intercluster_regions");
        check(bt_sync(ctx), "bt_sync");
    }
    checkTable(table, "countInterclusterKmers");
}

void KmerCounter::parseSampleKmers(bt_table *table, bt_bloom *path_bloom, Comm *comm) {
    const uint64_t world = comm ? (uint64_t)comm->world() : 1, rank = comm ? (uint64_t)comm->rank() : 0;
    for (size_t s = 0; s < samples.size(); s++) {
        std::cout << "[" << getLocalTime() << "] Parsing kmers from sample " << samples[s].name << " ..." << std::endl;
        KmcFile db(samples[s].file);
        if (db.kmer_length != kmer_size) throw std::runtime_error("KMC database " + samples[s].file + " holds " + std::to_string(db.kmer_length) + "-mers, not " + std::to_string(kmer_size) + "-mers");
        // rank r of w scans records [r * total / w, (r + 1) * total / w): a byte range of the .kmc_suf payload
        const uint64_t first = db.total_kmers / world * rank + std::min<uint64_t>(rank, db.total_kmers % world);
        const uint64_t count = db.total_kmers / world + (rank < db.total_kmers % world ? 1 : 0);
        const auto t_scan = std::chrono::steady_clock::now();
        uint64_t hits = bthost::parseSampleKmers(ctx, db, path_bloom, table, (uint32_t)s, 1ull << 22, first, count);
        const double scan_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scan).count();
        if (comm) comm->allreduceHist(&hits, 1);
        std::cout << "[" << getLocalTime() << "] Parsed " << db.total_kmers << " kmers (" << hits << " passed the path kmer filter)" << std::endl;
        if (getenv("BT_STAGE_TIMES")) std::cerr << "  parse sample k-mers: " << samples[s].name << " " << count << " records in " << scan_s << " s = " << (double)count / scan_s << " records/s" << std::endl;
    }
    if (!comm) return;
    // merge: a (k-mer, sample) count comes from ONE KMC record, i.e. from one rank; before the scans the replicas are identical and hold no
    // counts, so the records with a non-zero count are exactly what a rank's scans added
    checkTable(table, "parseSampleKmers");
    uint32_t row_bytes = 0;
    check(bt_table_count_row_bytes(table, &row_bytes), "bt_table_count_row_bytes");
    uint64_t mine = 0;
    check(bt_table_export_count_rows(table, nullptr, 0, &mine), "bt_table_export_count_rows");
    std::vector<uint64_t> sizes((size_t)world, 0);
    sizes[rank] = mine;
    comm->allreduceHist(sizes.data(), sizes.size());
    uint64_t total = 0;
    for (uint64_t n : sizes) total += n;
    void *d_mine = nullptr, *d_all = nullptr;
    check(bt_malloc(ctx, std::max<uint64_t>(mine, 1) * row_bytes, &d_mine), "bt_malloc");
    check(bt_malloc(ctx, std::max<uint64_t>(total, 1) * row_bytes, &d_all), "bt_malloc");
    try {
        uint64_t written = 0;
        check(bt_table_export_count_rows(table, (uint8_t *)d_mine, std::max<uint64_t>(mine, 1), &written), "bt_table_export_count_rows");
        if (written != mine) throw std::runtime_error("parseSampleKmers: the table changed between sizing and export");
        const std::vector<uint64_t> off = comm->allgatherDevice((const uint8_t *)d_mine, mine * row_bytes, (uint8_t *)d_all, std::max<uint64_t>(total, 1) * row_bytes);
        for (uint64_t r = 0; r < world; r++)
            if (r != rank && off[r + 1] > off[r]) check(bt_table_merge_count_rows(table, (const uint8_t *)d_all + off[r], (off[r + 1] - off[r]) / row_bytes), "bt_table_merge_count_rows");
        check(bt_sync(ctx), "bt_sync");
    } catch (...) {
        bt_free(ctx, d_mine);
        bt_free(ctx, d_all);
        throw;
    }
    bt_free(ctx, d_mine);
    bt_free(ctx, d_all);
    checkTable(table, "parseSampleKmers (merge)");
    std::cout << "[" << getLocalTime() << "] Merged the sample counts of " << world << " ranks (" << total << " kmers with counts)" << std::endl;
}

GibbsBatchData KmerCounter::classifyPathKmers(bt_table *table, const InferenceUnit &unit, const UnitGraphs &ug, const std::string &multigroup_kmers_bloom_prefix,
                                              const ChromosomePloidy &chrom_ploidy) {
    std::cout << "[" << getLocalTime() << "] Classifying kmers in variant cluster paths ..." << std::endl;
    if (!unit_paths) throw std::runtime_error("classifyPathKmers: countPathKmers has not run");
    const uint32_t C = (uint32_t)ug.graphs.size(), S = (uint32_t)samples.size();
    BloomHandle mg;
    check(bt_bloom_load(ctx, multigroup_kmers_bloom_prefix.c_str(), kmer_size, &mg.h), "bt_bloom_load");
    std::vector<uint32_t> num_path_kmers(C);
    std::vector<uint8_t> has_excluded(C);
    std::unique_ptr<StageScope> st(new StageScope("  classify (bt_paths_classify)"));
    check(bt_paths_classify(unit_paths->h, table, mg.h, num_path_kmers.data(), has_excluded.data()), "bt_paths_classify");
    checkTable(table, "classifyPathKmers");
    st.reset(new StageScope("  candidates (bt_paths_candidates)"));

    // ---- getHaplotypeCandidates of every cluster (VariantClusterGraph.cpp:941-1135) ----
    bt_paths_candidates_sizes sz{};
    check(bt_paths_candidates(unit_paths->h, table, &sz), "bt_paths_candidates");
    st.reset(new StageScope("  candidates: host arrays + fetch"));
    GibbsBatchData b;
    b.S = S;
    // (the arrays are sized — i.e. zero-filled, page by page — on several threads: one thread took 0.17 s for the 0.4 GB of a chr20-sized unit)
    std::vector<uint64_t> kmer_key;
    {
        const std::vector<std::function<void()>> sizing = {
            [&]() { kmer_key.resize(std::max<uint64_t>(sz.rows * 2, 1)); },
            [&]() { b.kmer_off.resize(C + 1); b.unique_off.resize(C + 1); b.multi_off.resize(C + 1); b.nestdep_off.resize(C + 1); },
            [&]() { b.hap_kmer_mult.resize(std::max<uint64_t>(sz.mult_bytes, 1)); },
            [&]() { b.kmer_has_counts.resize(std::max<uint64_t>(sz.rows, 1)); b.kmer_counts.resize(std::max<uint64_t>(sz.rows * S, 1)); },
            [&]() { b.kmer_ic_mult.resize(std::max<uint64_t>(sz.rows * 2, 1)); },
            [&]() { b.kv_off.resize(sz.rows + 1); },
            [&]() { b.kv_var.resize(std::max<uint64_t>(sz.nnz, 1)); },
            [&]() { b.kv_bits.resize(std::max<uint64_t>(sz.kv_words, 1)); },
            [&]() { b.unique_idx.resize(std::max<uint64_t>(sz.num_unique, 1)); b.multi_idx.resize(std::max<uint64_t>(sz.num_multi, 1)); },
            [&]() {
                b.hap_allele.resize(std::max<uint64_t>(sz.hap_allele, 1));
                b.hapnest_off.resize(sz.num_haplotypes + 1);
                b.hapnest_idx.resize(std::max<uint64_t>(sz.hapnest, 1));
                b.nestdep_cluster.resize(std::max<uint64_t>(sz.nestdep, 1));
                b.nestdep_var_off.resize(sz.nestdep + 1);
                b.nestdep_var.resize(std::max<uint64_t>(sz.nestdep_var, 1));
            }};
        parallelFor(sizing.size(), (unsigned)sizing.size(), [&](size_t a, size_t e, unsigned) {
            for (size_t i = a; i < e; i++) sizing[i]();
        });
    }
    bt_paths_candidates_out out{};
    out.kmer_off = b.kmer_off.data();
    out.hap_kmer_mult = b.hap_kmer_mult.data();
    out.kmer_key = kmer_key.data();
    out.kmer_has_counts = b.kmer_has_counts.data();
    out.kmer_counts = b.kmer_counts.data();
    out.kmer_ic_mult = b.kmer_ic_mult.data();
    out.kv_off = b.kv_off.data();
    out.kv_var = b.kv_var.data();
    out.kv_bits = b.kv_bits.data();
    out.unique_off = b.unique_off.data();
    out.unique_idx = b.unique_idx.data();
    out.multi_off = b.multi_off.data();
    out.multi_idx = b.multi_idx.data();
    out.hap_allele = b.hap_allele.data();
    out.hapnest_off = b.hapnest_off.data();
    out.hapnest_idx = b.hapnest_idx.data();
    out.nestdep_off = b.nestdep_off.data();
    out.nestdep_cluster = b.nestdep_cluster.data();
    out.nestdep_var_off = b.nestdep_var_off.data();
    out.nestdep_var = b.nestdep_var.data();
    check(bt_paths_candidates_fetch(unit_paths->h, &out), "bt_paths_candidates_fetch");
    unit_paths.reset();   // the enumerated paths are no longer needed
    b.hap_kmer_mult.resize(sz.mult_bytes);
    b.kmer_has_counts.resize(sz.rows);
    b.kmer_counts.resize(sz.rows * S);
    b.kmer_ic_mult.resize(sz.rows * 2);
    b.kv_var.resize(sz.nnz);
    b.kv_bits.resize(sz.kv_words);
    b.unique_idx.resize(sz.num_unique);
    b.multi_idx.resize(sz.num_multi);
    b.hap_allele.resize(sz.hap_allele);
    b.hapnest_idx.resize(sz.hapnest);
    b.nestdep_cluster.resize(sz.nestdep);
    b.nestdep_var.resize(sz.nestdep_var);

    // ---- the group structure (VariantClusterGroup.hpp:60-89) around the bundles ----
    st.reset(new StageScope("  group structure (host)"));
    const uint32_t G = (uint32_t)unit.variant_cluster_groups.size();
    b.kmer_shared.assign(sz.rows, -1);
    b.group_cluster_off.push_back(0);
    b.group_source_off.push_back(0);
    b.edge_off.push_back(0);
    for (uint32_t g = 0; g < G; g++) {
        const ClusterGroup &grp = unit.variant_cluster_groups[g];
        b.group_index.push_back(g);   // index in the unit's sorted group vector: the seeds derive from it (InferenceEngine.cpp:70,294)
        const auto &ploidy = chrom_ploidy.getSamplePloidy(grp.chrom_name);
        b.group_ploidy.insert(b.group_ploidy.end(), ploidy.begin(), ploidy.end());
        b.group_sources.insert(b.group_sources.end(), grp.source_vertices.begin(), grp.source_vertices.end());
        b.group_source_off.push_back((uint32_t)b.group_sources.size());
        // multicluster k-mers of a group share one record: number the distinct keys among the group's multicluster rows
        std::map<std::pair<uint64_t, uint64_t>, int32_t> keys;
        for (uint32_t v = 0; v < grp.clusters.size(); v++) {
            const uint32_t c = ug.group_first[g] + v;
            const uint32_t r0 = b.kmer_off[c];
            for (uint32_t i = b.multi_off[c]; i < b.multi_off[c + 1]; i++) {
                const uint64_t r = (uint64_t)r0 + b.multi_idx[i];
                auto ins = keys.emplace(std::make_pair(kmer_key[2 * r], kmer_key[2 * r + 1]), (int32_t)keys.size());
                b.kmer_shared[r] = ins.first->second;
            }
            b.cluster_idx.push_back(grp.clusters[v].cluster_idx);
            b.edges.insert(b.edges.end(), grp.out_edges[v].begin(), grp.out_edges[v].end());
            b.edge_off.push_back((uint32_t)b.edges.size());
            b.num_haplotypes.push_back((uint32_t)unit.best_paths[g][v].size());
            b.num_variants.push_back((uint32_t)ug.graphs[c].var_num_alleles.size());
            b.var_num_alleles.insert(b.var_num_alleles.end(), ug.graphs[c].var_num_alleles.begin(), ug.graphs[c].var_num_alleles.end());
            b.var_has_dependency.insert(b.var_has_dependency.end(), ug.graphs[c].var_has_dependency.begin(), ug.graphs[c].var_has_dependency.end());
        }
        b.group_num_shared.push_back((uint32_t)keys.size());
        b.group_cluster_off.push_back((uint32_t)b.cluster_idx.size());
    }
    return b;
}

}  // namespace bthost
