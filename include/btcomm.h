/*
 * btcomm.h — C ABI of libbtcomm.so: the multi-GPU exchange steps of the path over RCCL (xGMI), one rank per GPU.
 *
 * The reference is a single-process, multi-threaded program; it has no counterpart of these calls.  They exist because variant-cluster
 * groups shard across GPUs:
 *   - default genotyping needs no exchange until the end: the per-(cluster, sample) posterior summaries are GATHERED to rank 0;
 *   - the noise drivers (InferenceEngine::estimateNoise / estimateNoiseAndGenotypes, src/bayesTyper/InferenceEngine.cpp:135-276,
 *     384-472) add up the noise-count histograms of all groups every iteration (CountAllocation::mergeInCountAllocations under a
 *     mutex in the reference): one ALL-REDUCE of S x 256 counters per iteration;
 *   - k-mer matching with the KMC stream sharded by byte range: the matched (k-mer, sample, count) tuples go to the rank that owns the
 *     k-mer's group: one ALL-TO-ALL (variable sizes) per sample.
 * Kept in a library of its own so that libbtgpu.so does not depend on RCCL (a process that already holds another copy of RCCL, e.g.
 * through torch.distributed, keeps using that one).  Conventions as in btgpu.h: plain C, 0 = ok, text via bt_last_error().
 */
#ifndef BTCOMM_H
#define BTCOMM_H

#include <stddef.h>
#include <stdint.h>

#include "btgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bt_comm bt_comm;

#define BT_COMM_ID_BYTES 128

/* rank 0 creates the communicator's id (ncclGetUniqueId) and hands the bytes to the other ranks by any means (file, environment, MPI) */
int bt_comm_unique_id(uint8_t id[BT_COMM_ID_BYTES]);
/* every rank: joins the communicator on its context's GPU; collectives run on the context's stream */
int bt_comm_init(bt_ctx *ctx, const uint8_t id[BT_COMM_ID_BYTES], int rank, int world_size, bt_comm **out);
int bt_comm_destroy(bt_comm *c);
int bt_comm_rank(bt_comm *c, int *rank, int *world_size);

/* in-place sum over all ranks of n unsigned 64-bit counters on the device (the S*256 noise-count histogram of bt_gibbs_noise_counts;
 * the integer moments of bt_table_kmer_stats) */
int bt_comm_allreduce_hist(bt_comm *c, uint64_t *d_hist, uint64_t n);
/* gather of variable-length word arrays to rank 0 (the posterior summaries of bt_gibbs_posterior_summary: 2 words per (cluster, sample)):
 * h_offsets[world_size + 1] (every rank) receives the word offsets of the ranks' parts; on rank 0 d_out (capacity out_capacity words)
 * receives the parts in rank order */
int bt_comm_gather_summaries(bt_comm *c, const uint32_t *d_local, uint64_t local_words, uint32_t *d_out, uint64_t out_capacity, uint64_t *h_offsets);
/* all-gather of variable-length byte arrays: every rank's d_local (local_bytes bytes) lands in every rank's d_out in rank order,
 * h_offsets[world_size + 1] = byte offsets of the parts (the count rows of bt_table_export_count_rows after a KMC scan sharded by byte range:
 * every rank merges the other ranks' rows into its replica of the count table) */
int bt_comm_allgatherv(bt_comm *c, const uint8_t *d_local, uint64_t local_bytes, uint8_t *d_out, uint64_t out_capacity, uint64_t *h_offsets);
/* variable all-to-all of byte records: h_send_bytes[r] bytes of d_send (parts in rank order, contiguous) go to rank r; d_recv receives the
 * parts of all ranks in rank order, h_recv_bytes[r] = bytes received from rank r.  Record framing (e.g. 18-byte (k-mer, sample, count)
 * tuples) is the caller's.
 * In all three variable-size calls the ranks agree on failure: capacities and null-buffer flags travel with the sizes, every rank evaluates
 * the same conditions, so either all ranks post their transfers or all return the error (no rank is left waiting in a send). */
int bt_comm_alltoallv_matches(bt_comm *c, const uint8_t *d_send, const uint64_t *h_send_bytes, uint8_t *d_recv, uint64_t recv_capacity, uint64_t *h_recv_bytes);

#ifdef __cplusplus
}
#endif
#endif
