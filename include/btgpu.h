/*
 * btgpu.h — C ABI of libbtgpu.so: the MI355X (gfx950) implementation of BayesTyper's
 * k-mer matching + per-cluster Gibbs genotyping hot path.
 *
 * The reference (BayesTyper v1.5, C++11) has no FFI; the seams below are cut at the
 * narrowest points of its own class structure.  Every entry point cites
 * the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, opaque handles, no exceptions cross the boundary;
 *   - every call returns int: 0 = ok, non-zero = error, text via bt_last_error();
 *   - pointers named d_* are DEVICE pointers (hipMalloc / bt_malloc / a torch tensor's
 *     data_ptr); everything else is host memory owned by the caller;
 *   - batch calls are asynchronous on the context's stream; bt_sync() waits;
 *   - handles are not thread-safe, distinct handles are; one bt_ctx per GPU;
 *   - k-mers are passed 2-bit packed in two uint64 words {lo, hi}: nucleotide i sits in
 *     bits (2i, 2i+1) of the 128-bit value, A=0 C=1 G=2 T=3 — the bit layout of the
 *     reference's std::bitset<2k> (include/bayesTyper/Nucleotide.hpp:40-70).
 *     k <= 64.  Unless stated otherwise k-mers must already be canonical
 *     (include/bayesTyper/Kmer.tpp:225-255).
 */
#ifndef BTGPU_H
#define BTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_OK 0
#define BT_ERR 1

/* ------------------------------------------------------------------------------------------
 * Context, memory, timing
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_ctx bt_ctx;

/* last error message of the calling thread ("" if none) */
const char *bt_last_error(void);
/* library/ABI version, e.g. 100 = 1.0.0 */
int bt_version(void);
/* number of visible HIP devices (0 and BT_OK when there is none) */
int bt_device_count(int *count);

/* one context = one GPU + one stream (replaces the reference's `-p` thread pool for this path:
 * src/bayesTyper/main.cpp:128,214,467) */
int bt_ctx_create(int device_id, bt_ctx **out);
/* a second context on the same GPU with a stream of its own: work a helper thread enqueues (the next noise chain's sampler construction,
 * InferenceEngine.cpp:191-211 constructs a chain's genotypers while nothing else runs; here it overlaps the previous chain) does not queue behind
 * the first context's stream.  Handles created on either context live in the same device memory. */
int bt_ctx_clone(bt_ctx *ctx, bt_ctx **out);
int bt_ctx_destroy(bt_ctx *ctx);
/* run all subsequent work of this context on an externally owned hipStream_t (e.g. torch's
 * current stream); NULL restores the context's own stream */
int bt_ctx_set_stream(bt_ctx *ctx, void *hip_stream);
/* run all subsequent work on the device's default (null) stream — what torch.cuda.current_stream() is unless the
 * caller switched streams; the default stream's handle is 0, which bt_ctx_set_stream reads as "own stream" */
int bt_ctx_use_default_stream(bt_ctx *ctx);
int bt_sync(bt_ctx *ctx);
/* device properties: compute units, total/free HBM bytes, gcn arch name (buffer >= 64 bytes) */
int bt_ctx_info(bt_ctx *ctx, int *num_cu, uint64_t *hbm_total, uint64_t *hbm_free, char *arch, size_t arch_len);

int bt_malloc(bt_ctx *ctx, size_t bytes, void **d_out);
int bt_free(bt_ctx *ctx, void *d_ptr);
int bt_memset(bt_ctx *ctx, void *d_ptr, int value, size_t bytes);
int bt_memcpy_h2d(bt_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int bt_memcpy_d2h(bt_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int bt_memcpy_d2d(bt_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);

/* HIP-event timing on the context's stream (bench.py's roofline leg) */
typedef struct bt_timer bt_timer;
int bt_timer_create(bt_ctx *ctx, bt_timer **out);
int bt_timer_destroy(bt_timer *t);
int bt_timer_start(bt_timer *t);
int bt_timer_stop(bt_timer *t);
/* waits for the stop event; milliseconds between start and stop */
int bt_timer_elapsed_ms(bt_timer *t, float *ms);

/* ------------------------------------------------------------------------------------------
 * k-mer packing / canonical form
 *   mirrors KmerPair<k>::move + getLexicographicalLowestKmer
 *   (include/bayesTyper/Kmer.tpp:44-81,116-153,182-255) and Nucleotide::ntToBit
 *   (include/bayesTyper/Nucleotide.hpp:40-70)
 * ---------------------------------------------------------------------------------------- */
/* For every position i of the ASCII sequence d_seq[0..len): if the k characters ending at i are
 * all in ACGTacgt, write the canonical k-mer of seq[i-k+1..i] to d_kmers[2*i..2*i+1] and 1 to
 * d_valid[i]; else d_valid[i] = 0 (the k-mer window "resets" on any other character, exactly
 * as KmerPair::move does). */
int bt_kmers_from_sequence(bt_ctx *ctx, const char *d_seq, uint64_t len, uint32_t k,
                           uint64_t *d_kmers, uint8_t *d_valid);

/* ------------------------------------------------------------------------------------------
 * ntHash (external/ntHash/nthash.hpp:262-267 NTP64(kmer,k); :275-282 NTP64(kmer,k,seed))
 * ---------------------------------------------------------------------------------------- */
/* d_hash[i] = NTP64(kmer_i, k); if seeded != 0: NTP64(kmer_i, k, seed) */
int bt_nthash_batch(bt_ctx *ctx, const uint64_t *d_kmers, uint64_t n, uint32_t k,
                    int seeded, uint32_t seed, uint64_t *d_hash);

/* ------------------------------------------------------------------------------------------
 * Bloom filters: KmerBloom<k> and ThreadedKmerBloom<k>
 *   include/kmerBloom/KmerBloom.hpp:48-108, src/kmerBloom/KmerBloom.cpp:54-286,
 *   external/ntHash/BloomFilter.hpp:40-66,149-161,260-264
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_bloom bt_bloom;

/* KmerBloom<k>(num_kmers, fpr)               threaded == 0   (KmerBloom.cpp:54-60)
 * ThreadedKmerBloom<k>(num_kmers, fpr)       threaded != 0   (KmerBloom.cpp:204-215): 65 536
 * sub-filters each sized for ceil(num_kmers / 65536.0f) k-mers, routed by
 * NTP64(kmer,k,1029283129) % 65536 (KmerBloom.cpp:277-280). */
int bt_bloom_create(bt_ctx *ctx, uint64_t num_kmers, float fpr, uint32_t k, int threaded, bt_bloom **out);
/* KmerBloom<k>(prefix): reads <prefix>.bloomMeta / <prefix>.bloomData (KmerBloom.cpp:63-89).
 * Fails (like the reference's assert) when the stored k differs. */
int bt_bloom_load(bt_ctx *ctx, const char *prefix, uint32_t k, bt_bloom **out);
/* KmerBloom::save (KmerBloom.cpp:149-164); byte-identical files. Single filters only
 * (the reference's ThreadedKmerBloom::save is commented out, KmerBloom.hpp:100). */
int bt_bloom_save(bt_bloom *b, const char *prefix);
int bt_bloom_destroy(bt_bloom *b);
/* sizing as computed by calcOptNumBloomBits / calcOptNumHashes (KmerBloom.cpp:134-146);
 * for a threaded filter num_kmers/num_bits are per sub-filter */
int bt_bloom_info(bt_bloom *b, uint64_t *num_kmers, uint64_t *num_bits, uint32_t *num_hashes,
                  uint32_t *num_sub_filters, uint64_t *device_bytes);
/* addKmer for a batch (BloomFilter::insertF, BloomFilter.hpp:56-66) */
int bt_bloom_insert_batch(bt_bloom *b, const uint64_t *d_kmers, uint64_t n);
/* lookup for a batch (BloomFilter::containsF, BloomFilter.hpp:149-161): d_hits[i] = 0/1 */
int bt_bloom_contains_batch(bt_bloom *b, const uint64_t *d_kmers, uint64_t n, uint8_t *d_hits);
/* raw bit image of sub-filter `sub` (0 for a single filter) as the reference stores it:
 * (num_bits+7)/8 bytes, bit b at byte b/8, mask 1<<(7-b%8) */
int bt_bloom_read_bits(bt_bloom *b, uint32_t sub, uint8_t *h_out, uint64_t nbytes);
int bt_bloom_clear(bt_bloom *b);

/* ------------------------------------------------------------------------------------------
 * k-mer count table: KmerCountsHash / ObservedKmerCountsHash<N> + KmerCounts
 *   include/bayesTyper/KmerHash.hpp:73-87, include/bayesTyper/KmerCounts.hpp:35-101,
 *   src/bayesTyper/KmerCounts.cpp:40-223.  Per-key contents equal the reference's; the
 *   iteration order is free (open addressing in HBM instead of HybridHash's 16.7 M leaves).
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_table bt_table;

/* flag bits of bt_table meta byte 0 (KmerCounts.hpp:70) */
#define BT_KC_CLUSTER_OCC      0x01
#define BT_KC_MULTICLUSTER_OCC 0x02
#define BT_KC_MULTIGROUP_OCC   0x04
#define BT_KC_DECOY_OCC        0x08
#define BT_KC_MAX_MULTIPLICITY 0x10
#define BT_KC_PARAMETER        0x20

/* ObservedKmerCountsHash<N>(expected_size, threads) (src/bayesTyper/KmerHash.cpp:202-210);
 * num_samples <= 30 (src/bayesTyper/main.cpp:72).  Capacity is fixed: 2 x expected rounded
 * up to a power of two; an insert into a full table is an error reported by bt_table_status. */
int bt_table_create(bt_ctx *ctx, uint64_t expected_size, uint32_t num_samples, uint32_t k, bt_table **out);
int bt_table_destroy(bt_table *t);
/* forget every record (the capacity stays): a fresh table for the next unit (main.cpp:513 constructs one per unit) */
int bt_table_clear(bt_table *t);
/* grow to hold `expected_size` records (2 x, power of two) and move the stored records over: the reference's HybridHash
 * grows on demand (HybridHash.tpp:162); callers that know an upper bound of a stage's inserts reserve before it.
 * No-op when the table is large enough; an error when records were already dropped (overflow flag). */
int bt_table_reserve(bt_table *t, uint64_t expected_size);
/* number of stored keys, capacity, overflow flag */
int bt_table_status(bt_table *t, uint64_t *num_keys, uint64_t *capacity, int *overflowed);
/* addKmer(kmer, sorted) for a batch; optionally mark as parameter k-mer
 * (main.cpp:571-577: addKmer + isParameter(true)) */
int bt_table_insert_batch(bt_table *t, const uint64_t *d_kmers, uint64_t n, int mark_parameter);
/* findKmer for a batch: d_slots[i] = slot index or -1 */
int bt_table_find_batch(bt_table *t, const uint64_t *d_kmers, uint64_t n, int64_t *d_slots);
/* read records of slots found with bt_table_find_batch (slot -1 -> zeros):
 * h_counts[i*num_samples + s], h_meta[i*4 + {flags, max_haploid_mult, female_ic, male_ic}] */
int bt_table_read_slots(bt_table *t, const int64_t *h_slots, uint64_t n, uint8_t *h_counts, uint8_t *h_meta);
/* export every stored record (iteration order unspecified); arrays sized by bt_table_status */
int bt_table_export(bt_table *t, uint64_t *h_kmers, uint8_t *h_counts, uint8_t *h_meta, uint64_t max_records, uint64_t *num_written);
/* Merging the sample counts of tables filled by ranks that each scanned their own byte range of the samples' KMC databases
 * (KmerCounter::parseSampleKmers, KmerCounter.cpp:431-524, sharded over GPUs; a (k-mer, sample) count comes from exactly one KMC record,
 * i.e. from one rank).  A count row = 16 key bytes (lo, hi) + *row_bytes - 16 count bytes (the samples' counts, padded to 4).
 * bt_table_export_count_rows: every record with a non-zero sample count -> d_rows (device, capacity_rows rows; *h_num_rows = number of
 * such records, an error if it exceeds the capacity — call with capacity 0 and d_rows NULL to size the buffer).
 * bt_table_merge_count_rows: addKmer + saturating addSampleCount of every sample for each row (rows of OTHER ranks). */
int bt_table_count_row_bytes(bt_table *t, uint32_t *row_bytes);
int bt_table_export_count_rows(bt_table *t, uint8_t *d_rows, uint64_t capacity_rows, uint64_t *h_num_rows);
int bt_table_merge_count_rows(bt_table *t, const uint8_t *d_rows, uint64_t num_rows);

/* ObservedKmerCountsHash<N>::calculateKmerStats (src/bayesTyper/KmerHash.cpp:256-340): one pass over the table.
 * h_class_counts[7] = {total, unique, multicluster, decoy, max_multiplicity, multigroup, non_cluster} exactly as the reference
 * tallies them (:268-321).  For every PARAMETER k-mer without cluster occurrence and every sample s the observed count c enters
 * the bin (s, m = interclusterMultiplicity(gender[s])): the device accumulates the exact integer moments
 *   h_n[s*256+m] += 1, h_nonzero[s*256+m] += (c != 0), h_sum[s*256+m] += c, h_sumsq[s*256+m] += c*c
 * (order-independent, unlike the reference's running Welford update; KmerStats count / fraction / mean / variance follow from them on the host:
 * KmerStats.cpp:51-63,107-121).  gender[s]: 0 female, 1 male (Utils::Gender). */
int bt_table_kmer_stats(bt_table *t, const uint8_t *h_gender, uint64_t *h_class_counts, uint64_t *h_n, uint64_t *h_nonzero, uint64_t *h_sum,
                        uint64_t *h_sumsq);

/* KmerCounter::countInterclusterKmers for ONE region (src/bayesTyper/KmerCounter.cpp:291-338):
 * slide over d_seq[0..len), canonical k-mers that hit `path_bloom` are added to the table
 * (addKmer sorted) and get addInterclusterMultiplicity(is_decoy, {female_ploidy, male_ploidy})
 * (KmerCounts.cpp:98-118). */
int bt_table_count_intercluster(bt_table *t, bt_bloom *path_bloom, const char *d_seq, uint64_t len,
                                int is_decoy, uint32_t female_ploidy, uint32_t male_ploidy);

/* The same for ALL regions of one sequence in one launch (KmerCounter::countInterclusterKmers walks the region list of a unit,
 * src/bayesTyper/KmerCounter.cpp:349-386: one call per region would be one kernel launch per region).  Region r covers
 * d_seq[h_start[r] .. h_start[r] + h_len[r]); its k-mers get addInterclusterMultiplicity(h_is_decoy[r], {h_female_ploidy[r], h_male_ploidy[r]}).
 * The update is commutative, so the result equals the region-by-region calls in any order.  Synchronises the context's stream. */
int bt_table_count_intercluster_regions(bt_table *t, bt_bloom *path_bloom, const char *d_seq, uint32_t num_regions, const uint64_t *h_start, const uint64_t *h_len,
                                        const uint8_t *h_is_decoy, const uint8_t *h_female_ploidy, const uint8_t *h_male_ploidy);

/* KmerCounter::countInterclusterParameterKmers (src/bayesTyper/KmerCounter.cpp:161-250) for a batch of disjoint intercluster
 * regions of one device-resident sequence: region r = d_seq[h_start[r] .. h_start[r] + h_len[r]).  Every canonical k-mer of a
 * region that is NOT in the path Bloom filter is, in window order, either recorded as decoy (h_is_decoy[r]) or accepted by a
 * bernoulli_distribution(fraction) draw of mt19937(h_seed[r]) (h_seed[r] = prng_seed + intercluster_regions_idx, :176); accepted
 * k-mers are inserted with BT_KC_PARAMETER, decoy ones with BT_KC_DECOY_OCC.  The table plays KmerHash<bool>: the reference's
 * final value of a k-mer is PARAMETER && !DECOY_OCC (independent of the order regions are processed in). */
int bt_table_count_parameter_kmers(bt_table *t, bt_bloom *path_bloom, const char *d_seq, uint32_t num_regions, const uint64_t *h_start,
                                   const uint64_t *h_len, const uint8_t *h_is_decoy, const uint32_t *h_seed, float fraction);

/* the table-update half of VariantClusterGraph::classifyPathKmers for one batch of DISTINCT
 * path k-mers of distinct clusters (src/bayesTyper/VariantClusterGraph.cpp:902-938): for k-mer i
 * with max-over-paths multiplicity d_mult[i]: findKmer; absent and mult > 127 -> addKmer;
 * present -> addClusterMultiplicity(mult, multigroup_bloom.lookup(kmer)) (KmerCounts.cpp:137-159);
 * d_excluded[i] = isExcluded() after the update (0 when absent).  A k-mer may occur several
 * times in the batch (once per cluster that contains it). */
int bt_table_classify_batch(bt_table *t, bt_bloom *multigroup_bloom, const uint64_t *d_kmers,
                            const uint8_t *d_mult, uint64_t n, uint8_t *d_excluded);

/* ------------------------------------------------------------------------------------------
 * KMC count-table scan: KmerCounter::parseSampleKmers + parseSampleKmersCallBack
 *   src/bayesTyper/KmerCounter.cpp:388-524; KMC record layout external/kmc_api/kmc_file.cpp:428-494
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_kmc_scan bt_kmc_scan;

/* Describes one KMC database for the scan: k, lut_prefix_length p ((k-p)%4==0), counter_size
 * (1..4 bytes, little endian; counts must be <= 255 as asserted at KmerCounter.cpp:401),
 * total record count and the prefix LUT (4^p + 1 entries: LUT[j] = index of the first record
 * whose prefix is >= j, LUT[4^p] = total; host memory, copied). */
int bt_kmc_scan_create(bt_ctx *ctx, uint32_t k, uint32_t lut_prefix_len, uint32_t counter_size,
                       uint64_t total_records, const uint64_t *h_prefix_lut, bt_kmc_scan **out);
/* The same with an explicit LUT length for KMC2 ("0x200") databases, whose .kmc_pre holds one prefix table per signature bin,
 * concatenated (num_lut_entries = bins * 4^p + 1, last entry = total; external/kmc_api/kmc_file.cpp:186-238,428-449): the record's
 * prefix is (LUT index) mod 4^p. */
int bt_kmc_scan_create_bins(bt_ctx *ctx, uint32_t k, uint32_t lut_prefix_len, uint32_t counter_size, uint64_t total_records,
                            const uint64_t *h_prefix_lut, uint64_t num_lut_entries, bt_kmc_scan **out);
/* the database header's [min_count, max_count]: records whose counter lies outside are skipped, as CKMCFile::ReadNextKmer does
 * (external/kmc_api/kmc_file.cpp:496-511).  Default: no record is skipped. */
int bt_kmc_scan_set_count_range(bt_kmc_scan *s, uint32_t min_count, uint64_t max_count);
int bt_kmc_scan_destroy(bt_kmc_scan *s);
/* Push records [first_record, first_record + n) of the .kmc_suf payload through
 * decode -> path_bloom.lookup -> (on hit) table.addKmer(unsorted) + addSampleCount(sample, count).
 * d_records points at the first byte of record `first_record` and must be 16-byte aligned;
 * record size = (k-p)/4 + counter_size.  d_hit_count (optional, device uint64) is incremented
 * by the number of Bloom hits. */
int bt_kmc_scan_run(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx,
                    const uint8_t *d_records, uint64_t first_record, uint64_t n, uint64_t *d_hit_count);
/* The same for records in HOST memory (e.g. the memory-mapped .kmc_suf): the range is streamed through two pinned staging
 * buffers and two device buffers — host copy, H2D transfer (own stream) and scan of consecutive chunks overlap.  Blocking;
 * chunk_records = records per transfer (0: default 2^23); the staging buffers are kept with the handle; *h_hit_count = number of Bloom hits. */
int bt_kmc_scan_run_host(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const uint8_t *h_records,
                         uint64_t first_record, uint64_t n, uint64_t chunk_records, uint64_t *h_hit_count);
/* the same from the database's .kmc_suf file: the records [first_record, first_record + n) at payload_offset (4: behind the "KMCS" marker) + first_record x record size
 * are read by several threads with pread() straight into the pinned staging slots (the reference reads the file through CKMCFile's buffered reader,
 * kmc_file.cpp:428-515, one producer thread: KmerCounter.cpp:469-505) — no memory mapping, no page fault per 4 KB of a first pass */
int bt_kmc_scan_run_file(bt_kmc_scan *s, bt_bloom *path_bloom, bt_table *table, uint32_t sample_idx, const char *suf_path, uint64_t payload_offset,
                         uint64_t first_record, uint64_t n, uint64_t chunk_records, uint64_t *h_hit_count);
/* bayesTyperTools makeBloom (src/bayesTyperTools/MakeBloom.cpp:200-295): the k-mers of records [first_record, first_record + n)
 * are added to a sample's KmerBloom (created with bt_bloom_create(ctx, total_kmers, fpr, k, 0, ..), written with bt_bloom_save:
 * byte-identical .bloomMeta / .bloomData, insertion being an order-independent OR) */
int bt_kmc_scan_make_bloom(bt_kmc_scan *s, bt_bloom *sample_bloom, const uint8_t *d_records, uint64_t first_record, uint64_t n);
/* decode only (tests): d_kmers[2*i..] = packed k-mer of record i, d_counts[i] = its count */
int bt_kmc_scan_decode(bt_kmc_scan *s, const uint8_t *d_records, uint64_t first_record, uint64_t n,
                       uint64_t *d_kmers, uint32_t *d_counts);

/* ------------------------------------------------------------------------------------------
 * Path k-mer enumeration over variant-cluster graphs:
 *   VariantClusterGraph::{countPathKmers, classifyPathKmers, getHaplotypeCandidates, updateVariantPathIndices}
 *   (src/bayesTyper/VariantClusterGraph.cpp:800-1184), driven per unit by KmerCounter::{countPathKmers, classifyPathKmers}
 *   (src/bayesTyper/KmerCounter.cpp:252-289,526-555) and VariantClusterGroup::initGenotyper.
 * Input: the graphs as VariantClusterGraph holds them after findSamplePaths — vertices in topological (vertex-index) order
 * with the fields of VariantClusterGraphVertex (include/bayesTyper/VariantClusterGraphVertex.hpp:43-73) and the best-path
 * bitmaps best_paths_indices (P x |V|).  Output of the last stage: the VariantClusterHaplotypes bundle of every cluster in the
 * flattened form bt_gibbs_batch takes.
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_paths_batch {
    uint32_t num_clusters;             /* C */
    const uint32_t *vertex_off;        /* [C+1] vertices of cluster c, in vertex-index order */
    const uint32_t *num_paths;         /* [C] number of best paths (= haplotype candidates H) */
    /* ---- per vertex (NV = vertex_off[C]) ---- */
    const uint64_t *seq_off;           /* [NV+1] nucleotides of vertex v: seq[seq_off[v] .. seq_off[v+1]) */
    const uint8_t *seq;                /* one 2-bit code per byte (A 0, C 1, G 2, T 3: Nucleotide.hpp:40-70) */
    const uint16_t *vertex_variant;    /* variant_allele_idx.first, 0xFFFF = none */
    const uint16_t *vertex_allele;     /* variant_allele_idx.second */
    const uint8_t *vertex_flags;       /* bit 0 is_disconnected, bit 1 is_first_nucleotides_redundant */
    const uint32_t *vertex_nested;     /* nested_variant_cluster_index, 0xFFFFFFFF = none */
    const uint32_t *refvar_off;        /* [NV+1] -> refvar */
    const uint16_t *refvar;            /* reference_variant_indices */
    /* ---- per cluster ---- */
    const uint64_t *path_off;          /* [C+1] -> path_vertices; cluster c: num_paths[c] rows of (vertex_off[c+1]-vertex_off[c]) bytes */
    const uint8_t *path_vertices;      /* best_paths_indices, row-major (path, vertex), 0/1 */
    const uint32_t *var_off;           /* [C+1] -> per-variant arrays (variant_cluster_info) */
    const uint16_t *var_num_alleles;   /* numberOfAlleles(), incl. the missing allele of variants with has_dependency */
    const uint8_t *var_has_dependency;
    /* ---- graph edges: only needed by bt_find_paths_* (may be NULL otherwise) ---- */
    const uint32_t *in_off;            /* [NV+1] in-edges of vertex v -> in_src, in boost::in_edges order (edge insertion order) */
    const uint32_t *in_src;            /* source vertex, local to the cluster (always < the target's local index) */
} bt_paths_batch;

typedef struct bt_paths bt_paths;

/* uploads the graphs, lays every best path out as one text (k-mer windows never span a disconnected vertex or two paths) and
 * enumerates the canonical k-mer of every window once; *h_num_kmer_occurrences (optional) = number of full windows */
int bt_paths_create(bt_ctx *ctx, const bt_paths_batch *batch, uint32_t k, bt_paths **out, uint64_t *h_num_kmer_occurrences);
int bt_paths_destroy(bt_paths *p);
/* VariantClusterGraph::countPathKmers + KmerCounter::countPathKmersCallback (KmerCounter.cpp:252-289): every path k-mer is
 * added to the path Bloom filter (the reference first collects them in an unordered_set; insertion is idempotent) */
int bt_paths_count_kmers(bt_paths *p, bt_bloom *path_bloom);
/* KmerCounter::countPathMultigroupKmers (src/bayesTyper/KmerCounter.cpp:105-159) — cluster stage.  h_cluster_group[c] = index of
 * the variant-cluster group of cluster c; clusters are listed group by group, groups in index order, a group's clusters in its vertex
 * order (an error otherwise).  Every distinct path k-mer of a group is looked up in the path Bloom filter: present -> it goes into
 * the multigroup table (KmerHash<bool>), absent -> it is added to the filter; *h_num_path_kmers = sum over groups of their distinct
 * path k-mers (inference_unit->num_path_kmers).  The outcome — including the k-mers that enter the table as false positives of what
 * the filter holds at that moment — is the one of the reference run with ONE thread (with more threads the reference's outcome
 * depends on thread timing): groups in index order, a group's k-mers in the iteration order of the std::unordered_set<std::bitset<2k>>
 * (libstdc++) the reference collects them in and reuses from group to group.  Afterwards every path k-mer is in the filter. */
int bt_paths_count_multigroup(bt_paths *p, const uint32_t *h_cluster_group, bt_bloom *path_bloom, bt_table *multigroup_table, uint64_t *h_num_path_kmers);
/* VariantClusterGraph::classifyPathKmers for every cluster (VariantClusterGraph.cpp:848-939): per distinct path k-mer of a
 * cluster the maximum over its paths of the (saturating) per-path multiplicity -> table update as bt_table_classify_batch.
 * h_num_path_kmers[c] = distinct k-mers of cluster c (num_path_kmers), h_has_excluded[c] = has_excluded_kmers. */
int bt_paths_classify(bt_paths *p, bt_table *table, bt_bloom *multigroup_bloom, uint32_t *h_num_path_kmers, uint8_t *h_has_excluded);

/* host arrays receiving getHaplotypeCandidates' result for all clusters; sizes from bt_paths_candidates (R rows, NNZ incidence
 * entries, ...).  Field meaning as in bt_gibbs_batch; row ids in unique_idx / multi_idx are local to the cluster. */
typedef struct bt_paths_candidates_out {
    uint32_t *kmer_off;          /* [C+1] */
    uint8_t *hap_kmer_mult;      /* sum_c K_c * H_c */
    uint64_t *kmer_key;          /* [R*2] packed canonical k-mer of the row (identifies shared records of multicluster k-mers) */
    uint8_t *kmer_has_counts;    /* [R] the k-mer has a record in the table */
    uint8_t *kmer_counts;        /* [R*S] */
    uint8_t *kmer_ic_mult;       /* [R*2] */
    uint32_t *kv_off;            /* [R+1] */
    uint16_t *kv_var;            /* [NNZ] entries of a row sorted by variant index (the reference's order is an unordered_map's) */
    uint32_t *kv_bits;           /* ceil(H/32) words per entry */
    uint32_t *unique_off;        /* [C+1] */
    uint32_t *unique_idx;
    uint32_t *multi_off;         /* [C+1] */
    uint32_t *multi_idx;
    uint16_t *hap_allele;        /* sum_c H_c * V_c */
    uint32_t *hapnest_off;       /* [sum_c H_c + 1] */
    uint32_t *hapnest_idx;
    uint32_t *nestdep_off;       /* [C+1] */
    uint32_t *nestdep_cluster;
    uint32_t *nestdep_var_off;   /* [ND+1] */
    uint16_t *nestdep_var;
} bt_paths_candidates_out;

typedef struct bt_paths_candidates_sizes {
    uint64_t rows, mult_bytes, nnz, kv_words, num_unique, num_multi, hap_allele, num_haplotypes, hapnest, nestdep, nestdep_var;
} bt_paths_candidates_sizes;

/* VariantClusterGraph::getHaplotypeCandidates for every cluster (VariantClusterGraph.cpp:941-1135) against the classified
 * table: computes the bundle on the device + host and reports its sizes; bt_paths_candidates_fetch copies it out. */
int bt_paths_candidates(bt_paths *p, bt_table *table, bt_paths_candidates_sizes *sizes);
int bt_paths_candidates_fetch(bt_paths *p, bt_paths_candidates_out *out);

/* ------------------------------------------------------------------------------------------
 * Best-path search per sample: VariantClusterGraph::{findSamplePaths, mergePaths, isPathsRedundant, filterPaths, addPathIndices}
 *   (src/bayesTyper/VariantClusterGraph.cpp:389-798) with VariantClusterGraphPath (src/bayesTyper/VariantClusterGraphPath.cpp:38-225),
 *   driven per sample by KmerCounter::findVariantClusterPaths (src/bayesTyper/KmerCounter.cpp:59-103).
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_find_paths bt_find_paths;
/* state for the clusters of `batch` (num_paths / path_* fields ignored, in_off / in_src required): empty best_paths_indices */
int bt_find_paths_create(bt_ctx *ctx, const bt_paths_batch *batch, uint32_t k, uint32_t max_sample_haplotypes, uint32_t num_samples, bt_find_paths **out);
int bt_find_paths_destroy(bt_find_paths *f);
/* one sample: findSamplePaths of every cluster against the sample's KmerBloom with mt19937(h_seeds[c]) — the reference's seed is
 * prng_seed + (group_idx + 1) * (sample_idx + 1) + variant_cluster_idx (KmerCounter.cpp:65, VariantClusterGroup.cpp:142) — followed
 * by addPathIndices into the accumulated best paths */
int bt_find_paths_sample(bt_find_paths *f, bt_bloom *sample_bloom, const uint32_t *h_seeds);
/* best_paths_indices so far: h_num_paths[c] rows of |V_c| bytes each, clusters concatenated (bt_paths_batch::path_vertices layout) */
int bt_find_paths_sizes(bt_find_paths *f, uint32_t *h_num_paths, uint64_t *h_total_bytes);
int bt_find_paths_fetch(bt_find_paths *f, uint8_t *h_path_vertices);

/* ------------------------------------------------------------------------------------------
 * Count model LUTs: CountDistribution (src/bayesTyper/CountDistribution.cpp:215-265)
 * ---------------------------------------------------------------------------------------- */
/* layout of the two caches the sampler reads through calcCountLogProb(sample, bias=0, multiplicity, count):
 *   genomic[(s*256 + multiplicity)*256 + count]   (CountDistribution.cpp:215-234; row multiplicity 0 is never read)
 *   noise[s*256 + count]                          (CountDistribution.cpp:236-253)
 * The host computes them in fp64 exactly as the reference does; the device only gathers. */

/* ------------------------------------------------------------------------------------------
 * Gibbs genotyping of variant-cluster groups:
 *   VariantClusterGroup::{initGenotyper, shuffleBranchOrdering, estimateGenotypes, getNoiseCounts,
 *   clearGenotyperCache, resetGroup, collectGenotypes} (include/bayesTyper/VariantClusterGroup.hpp:125-134)
 *   driven as InferenceEngine does (src/bayesTyper/InferenceEngine.cpp:60-133,278-333).
 *
 * Input = the per-cluster tensor bundle VariantClusterGraph::getHaplotypeCandidates produces
 * (VariantClusterHaplotypes, include/bayesTyper/VariantClusterHaplotypes.hpp:52-109), flattened.
 * All arrays are HOST memory and are copied by bt_gibbs_create.
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_gibbs_params {
    uint32_t num_samples;                  /* S <= 30 */
    uint32_t seed;                         /* --random-seed */
    uint32_t num_chains;                   /* --number-of-gibbs-chains (20) */
    uint32_t burn_in;                      /* --gibbs-burn-in (100) */
    uint32_t num_iterations;               /* --gibbs-samples (250) */
    float kmer_subsampling_rate;           /* --kmer-subsampling-rate (0.1f) */
    uint32_t max_haplotype_variant_kmers;  /* --max-haplotype-variant-kmers (500) */
    uint32_t noise_seeding;                /* 0: genotyper seed = seed+(i+1)          (InferenceEngine.cpp:294)
                                              1: genotyper seed = seed+(i+1)*(c+1)    (InferenceEngine.cpp:70) */
    const uint8_t *gender;                 /* [S] 0 = female, 1 = male (Utils::Gender) */
} bt_gibbs_params;

typedef struct bt_gibbs_batch {
    uint32_t num_groups;     /* G */
    uint32_t num_clusters;   /* C = total number of group vertices */
    /* ---- per group ---- */
    const uint32_t *group_index;         /* [G] index i of the group in the unit's sorted group vector (seeds, Appendix B) */
    const uint32_t *group_cluster_off;   /* [G+1] vertices of group g are clusters [off[g], off[g+1]) in vertex order */
    const uint8_t *group_ploidy;         /* [G*S] chromosome ploidy per sample: 0 Null, 1 Haploid, 2 Diploid */
    const uint32_t *group_source_off;    /* [G+1] -> group_sources */
    const uint32_t *group_sources;       /* source vertices (local vertex ids), VariantClusterGroup.hpp:86 */
    const uint32_t *group_num_shared;    /* [G] number of multicluster k-mer records shared by the group's clusters */
    /* ---- per cluster (vertex) ---- */
    const uint32_t *cluster_idx;         /* [C] variant_cluster_idx (seed offset, nested-cluster identity) */
    const uint32_t *edge_off;            /* [C+1] out_edges CSR -> edges */
    const uint32_t *edges;               /* local vertex ids */
    const uint32_t *num_haplotypes;      /* [C] H */
    const uint32_t *num_variants;        /* [C] V */
    const uint32_t *kmer_off;            /* [C+1] k-mer rows of cluster c: [kmer_off[c], kmer_off[c+1]) */
    const uint8_t *hap_kmer_mult;        /* haplotype_kmer_multiplicities, row-major K x H per cluster, clusters concatenated */
    /* ---- per k-mer row (R = kmer_off[C]) ---- */
    const uint8_t *kmer_has_counts;      /* [R] KmerInfo::counts != nullptr */
    const uint8_t *kmer_counts;          /* [R*S] getSampleCount(s) */
    const uint8_t *kmer_ic_mult;         /* [R*2] getInterclusterMultiplicity(Female), (Male) */
    const int32_t *kmer_shared;          /* [R] -1, or index (< group_num_shared) of the shared record of a multicluster k-mer */
    const uint32_t *kv_off;              /* [R+1] variant_haplotype_indices CSR -> kv_var, kv_bits */
    const uint16_t *kv_var;              /* [NNZ] variant index */
    const uint32_t *kv_bits;             /* haplotype bitsets, ceil(H/32) words per entry, entries in kv order */
    /* ---- index lists ---- */
    const uint32_t *unique_off;          /* [C+1] -> unique_idx */
    const uint32_t *unique_idx;          /* unique_kmer_indices (row ids local to the cluster), first-seen order */
    const uint32_t *multi_off;           /* [C+1] -> multi_idx */
    const uint32_t *multi_idx;           /* multicluster_kmer_indices */
    /* ---- haplotypes / variants ---- */
    const uint16_t *hap_allele;          /* variant_allele_indices, row-major H x V per cluster, clusters concatenated */
    const uint32_t *hapnest_off;         /* [sum(H)+1] nested_variant_cluster_indices CSR per haplotype (sorted) */
    const uint32_t *hapnest_idx;
    const uint16_t *var_num_alleles;     /* [sum(V)] numberOfAlleles() incl. the missing allele */
    const uint8_t *var_has_dependency;   /* [sum(V)] */
    /* ---- nested_variant_cluster_dependency (VariantClusterHaplotypes.hpp:97) ---- */
    const uint32_t *nestdep_off;         /* [C+1] -> nestdep_cluster / nestdep_var_off */
    const uint32_t *nestdep_cluster;     /* child variant_cluster_idx */
    const uint32_t *nestdep_var_off;     /* [ND+1] -> nestdep_var */
    const uint16_t *nestdep_var;         /* variant indices, sorted descending (VariantClusterGraph.cpp:1130) */
} bt_gibbs_batch;

typedef struct bt_gibbs bt_gibbs;

int bt_gibbs_create(bt_ctx *ctx, const bt_gibbs_params *params, const bt_gibbs_batch *batch, bt_gibbs **out);
/* the device memory bt_gibbs_create would need for this batch (nothing is allocated): the host sizes its launches from it and
 * bt_ctx_info's free HBM (InferenceEngine.cpp:335-382 hands groups to its workers in batches; here a batch is what fits the GPU) */
int bt_gibbs_state_bytes(bt_ctx *ctx, const bt_gibbs_params *params, const bt_gibbs_batch *batch, uint64_t *bytes);
int bt_gibbs_destroy(bt_gibbs *g);
/* A batch's flat arrays held ON THE DEVICE, and samplers over any selection of its groups built from them without another pass through the host:
 * the reference keeps a unit's clusters in memory for the whole genotyping stage and constructs genotypers from them chain after chain
 * (InferenceEngine.cpp:60-75 initGenotypersCallback over the unit's groups, :172-211 the noise driver's subset per chain, :335-382 the group batches of
 * the default mode); here the unit goes to the GPU once (bt_gibbs_source_create: validated like bt_gibbs_create, uploaded, the caller's arrays are not
 * referenced afterwards) and every sampler — a noise chain's random subset, a launch-sized range of the unit — is laid out by the host from the
 * per-cluster dimensions alone and filled by the device from the resident arrays (bt_gibbs_create_from_source: group_ids = positions in the source
 * batch, in the order the sampler shall hold them; NULL = all groups; ctx = the context whose stream the sampler works on, NULL = the source's, same device).
 * bt_gibbs_create(batch) = source + sampler over all groups + source released. */
typedef struct bt_gibbs_source bt_gibbs_source;
int bt_gibbs_source_create(bt_ctx *ctx, uint32_t num_samples, const bt_gibbs_batch *batch, bt_gibbs_source **out);
int bt_gibbs_source_destroy(bt_gibbs_source *src);
int bt_gibbs_source_device_bytes(bt_gibbs_source *src, uint64_t *bytes);
int bt_gibbs_create_from_source(bt_gibbs_source *src, bt_ctx *ctx, const bt_gibbs_params *params, const uint32_t *group_ids, uint32_t num_groups, bt_gibbs **out);
int bt_gibbs_state_bytes_from_source(bt_gibbs_source *src, const bt_gibbs_params *params, const uint32_t *group_ids, uint32_t num_groups, uint64_t *bytes);
/* upload the count-model LUTs (see above); must be called before the first sweep and after every noise update */
int bt_gibbs_set_lut(bt_gibbs *g, const double *h_genomic /* [S*256*256] */, const double *h_noise /* [S*256] */);
int bt_gibbs_set_noise_lut(bt_gibbs *g, const double *h_noise /* [S*256] */);
/* initGenotyper (constructing the genotypers on first use) + shuffleBranchOrdering for every group,
 * with the seeds of chain `chain_idx` (InferenceEngine.cpp:292-295 / :60-75) */
int bt_gibbs_init_chain(bt_gibbs *g, uint32_t chain_idx);
/* estimateGenotypes(count_distribution, ploidy, collect_samples) `num_sweeps` times for every group */
int bt_gibbs_sweep(bt_gibbs *g, uint32_t num_sweeps, int collect_samples);
/* the whole default-mode schedule for every group: for each chain init_chain, burn_in sweeps without and
 * num_iterations sweeps with collection (InferenceEngine.cpp:292-306), in one launch */
int bt_gibbs_run(bt_gibbs *g);
/* getNoiseCounts of every group accumulated into a [S*256] histogram on the device, then clearGenotyperCache
 * (InferenceEngine.cpp:90-92); d_hist is zeroed first when zero_first != 0 */
int bt_gibbs_noise_counts(bt_gibbs *g, uint64_t *d_hist, int zero_first);
/* One iteration of the noise drivers with ONE host synchronisation (InferenceEngine.cpp:77-98): (the noise table of the previous
 * iteration, h_noise [S*256] or NULL, is uploaded without waiting;) one sweep of every group; the noise-count histogram of all groups
 * (+ clearGenotyperCache) lands in h_hist [S*256].  The caller draws the new rates from h_hist (after its all-reduce over the ranks) and
 * hands the rebuilt table to the next call. */
int bt_gibbs_noise_iteration(bt_gibbs *g, const double *h_noise, int collect_samples, uint64_t *h_hist);
/* A whole chain of a noise driver (InferenceEngine.cpp:135-276 estimateNoise, :384-472 estimateNoiseAndGenotypes; one iteration = :77-98) as ONE
 * resident launch: every tile keeps its workgroup — sampler state in registers / LDS — for the num_iterations iterations (collecting from iteration
 * first_collect on, 0-based), and the per-iteration exchange (noise-count histogram out, rebuilt noise table in) goes through a mailbox in pinned host
 * memory instead of a kernel launch + copies + a stream synchronisation per iteration.  The draws stay where they are exact: the caller takes the
 * histogram of iteration i from step i, reduces it over its ranks, draws the rates (CountDistribution::sampleNoiseParameters, CountDistribution.cpp:173-186:
 * libstdc++ / glibc) and hands the rebuilt table to step i + 1.
 *   begin: *resident = 1 when the chain was started this way; 0 (and BT_OK) when this batch cannot be (its workgroups do not fit the GPU together, or
 *          tiles with large dense tables want the whole-GPU refill between iterations): the caller then iterates with bt_gibbs_noise_iteration.
 *   step:  h_noise = the table for this iteration's sweep ([S*256]; NULL: unchanged; must be NULL for the first iteration, which runs with the table
 *          of bt_gibbs_set_lut / bt_gibbs_set_noise_lut); returns when this iteration's histogram is in h_hist [S*256] (caches cleared, :90-92).
 *   end:   after the last step (or earlier: the launch is told to stop at its next exchange); synchronises.  No other operation on g between begin and end.
 * Every wait on either side has a deadline (BT_NOISE_CHAIN_TIMEOUT_S, default 60 s): a stall is an error, never a hang. */
int bt_gibbs_noise_chain_begin(bt_gibbs *g, uint32_t num_iterations, uint32_t first_collect, int *resident);
int bt_gibbs_noise_chain_step(bt_gibbs *g, const double *h_noise, uint64_t *h_hist);
int bt_gibbs_noise_chain_end(bt_gibbs *g);
/* resetGroup for every group (InferenceEngine.cpp:100-113): genotypers are rebuilt by the next init_chain */
int bt_gibbs_reset_groups(bt_gibbs *g);

/* Results (collectGenotypes input): per cluster the diplotype sampling frequencies
 * (VariantClusterGenotyper.hpp:112) and the allele k-mer statistics (:104).
 * sizes: number of distinct sampled diplotypes over all clusters; number of (cluster, sample, allele) cells */
int bt_gibbs_result_sizes(bt_gibbs *g, uint64_t *num_diplotype_entries, uint64_t *num_allele_cells);
/* h_dip_off[C+1]; entry e: haplotypes (h_dip_h1[e] <= h_dip_h2[e], 0xFFFF = none), h_dip_freq[e*S + s];
 * allele cells in the order cluster, sample, variant, allele: h_stats[cell*12 + stat*4 + {count, fraction, mean, M2}]
 * for stat in {count_stats, fraction_stats, mean_stats} (KmerStats.cpp:107-121); h_cell_off[C+1] */
int bt_gibbs_result_fetch(bt_gibbs *g, uint64_t *h_dip_off, uint16_t *h_dip_h1, uint16_t *h_dip_h2, uint32_t *h_dip_freq,
                          uint64_t *h_cell_off, double *h_stats);
/* The same results as ONE string of 32-bit words in DEVICE memory, for the gather to rank 0 (bt_comm_gather_summaries) without a
 * host round trip — the reference's threads push their genotypes into one queue (InferenceEngine.cpp:335-382, 384-399); with
 * one process per GPU that queue is the gather.  Layout: [C, entries, cells, S], [entries, cells] per cluster, h1 | h2 << 16
 * per entry (order of bt_gibbs_result_fetch), counts [entry][S], one pad word if the count so far is odd, statistics
 * [cell][12] doubles.  *d_words belongs to the sampler (valid until the next call or bt_gibbs_destroy); complete on return. */
int bt_gibbs_result_words(bt_gibbs *g, const uint32_t **d_words, uint64_t *num_words);
/* compact posterior summary on the DEVICE (input of the cross-GPU gather to rank 0): for cluster c, sample s
 * d_out[(c*S+s)*2] = h1 | h2<<16 of the most frequently sampled diplotype, d_out[(c*S+s)*2+1] = its frequency */
int bt_gibbs_posterior_summary(bt_gibbs *g, uint32_t *d_out);
/* diagnostics: the diplotype drawn for (cluster, sample) in each of the first `max_sweeps` sweeps after this call:
 * h_trace[(sweep*C + c)*S + s] = h1 | h2 << 16.  Pass max_sweeps = 0 to switch tracing off. */
int bt_gibbs_trace_enable(bt_gibbs *g, uint32_t max_sweeps);
int bt_gibbs_trace_fetch(bt_gibbs *g, uint32_t *h_trace, uint64_t max_words, uint64_t *num_sweeps_recorded);
/* device bytes held by this batch (state + inputs) */
int bt_gibbs_device_bytes(bt_gibbs *g, uint64_t *bytes);

/* ------------------------------------------------------------------------------------------
 * The noise half of CountDistribution on the device (src/bayesTyper/CountDistribution.cpp:163-200 sampleNoiseParameters /
 * calcCountSuffStats, :314-352 the Poisson noise table with its tail fold) and a whole chain of a noise driver
 * (InferenceEngine.cpp:77-98 per iteration) without a host round trip per iteration.
 * bt_noise_rng = the run's generator as libstdc++ holds it: std::mt19937 (block form: 624 words + the index of the next word) and
 * the gamma distribution's normal distribution (its saved second variate).  The host exchanges it with its CountDistribution
 * before and after a chain, so host-side and device-side draws continue one stream.
 * ---------------------------------------------------------------------------------------- */
typedef struct bt_noise_rng {
    uint32_t mt[624];
    uint32_t mt_pos;             /* 0..624 (624: the next draw regenerates the block) */
    uint32_t saved_available;    /* std::normal_distribution::_M_saved_available */
    double saved;                /* std::normal_distribution::_M_saved */
} bt_noise_rng;
typedef struct bt_noise_model bt_noise_model;
/* h_prior[2*s], h_prior[2*s+1] = shape, scale of sample s's noise-rate prior (--noise-rate-prior, floats as the reference holds them) */
int bt_noise_model_create(bt_ctx *ctx, uint32_t num_samples, const float *h_prior, bt_noise_model **out);
int bt_noise_model_destroy(bt_noise_model *m);
int bt_noise_model_set_rng(bt_noise_model *m, const bt_noise_rng *h_rng);
int bt_noise_model_get_rng(bt_noise_model *m, bt_noise_rng *h_rng);
/* num_iterations iterations of a chain: { one sweep of every group of g (collecting from iteration first_collect on, 0-based);
 * noise counts of all groups + clearGenotyperCache; reduce(user, d_hist, S*256) if given — it must only ENQUEUE work on the context's
 * stream (bt_comm_allreduce_hist does); one gamma draw per sample from the model's generator (shape + sum of counts,
 * scale / (observations * scale + 1)); the rebuilt noise table becomes g's }.  Everything is enqueued at once; the call returns after
 * the last iteration with h_rates[it*S + s] = the rate drawn in iteration it.  g may be NULL (a rank without groups in this chain
 * still reduces and draws, so that all ranks' generators stay in step). */
int bt_gibbs_noise_chain(bt_gibbs *g, bt_noise_model *m, uint32_t num_iterations, uint32_t first_collect,
                         int (*reduce)(void *user, uint64_t *d_hist, uint64_t n), void *user, double *h_rates);

/* ------------------------------------------------------------------------------------------
 * Diagnostics: host-side entry points of the libstdc++-compatible primitives the sampler relies on
 * (SURVEY Appendix B.2).  They run the same __host__ __device__ code the kernels use.
 * ---------------------------------------------------------------------------------------- */
/* replay a sequence of operations on the unordered_set<uint> emulation: op[i] = 0 insert, 1 erase, 2 clear;
 * afterwards writes the iteration order to h_order (capacity universe) and its length to *n */
int bt_diag_uset_replay(uint32_t universe, const uint8_t *ops, const uint32_t *values, uint64_t num_ops, uint32_t *h_order, uint32_t *n);
/* draws from the device random-number stack seeded like std::mt19937(seed): kind 0 raw u32, 1 generate_canonical<double,53>,
 * 2 gamma(shape=a, scale=b) with one persistent distribution object, 3 uniform_int(0, a), 4 bernoulli(float a),
 * 5 std::shuffle of 0..a-1 (h_out gets the permutation as doubles, n ignored) */
int bt_diag_rng(uint32_t seed, int kind, const double *a, const double *b, uint64_t n, double *h_out);
/* Host-side run of the container replay behind bt_paths_count_multigroup: n DISTINCT k-mers (2 x u64 each) are inserted, in the given
 * order, into an emulated libstdc++ std::unordered_set<std::bitset<2k>> that starts with `initial_buckets` buckets (1 = freshly
 * constructed; a set that was clear()ed keeps its bucket count); h_rank[i] = position of k-mer i in the set's iteration order,
 * *h_final_buckets = its bucket count afterwards.  (CPU tests compare it with the real container.) */
int bt_diag_kmer_set_order(const uint64_t *h_kmers, uint32_t n, uint64_t initial_buckets, unsigned k, uint32_t *h_rank, uint64_t *h_final_buckets);

#ifdef __cplusplus
}
#endif
#endif /* BTGPU_H */
