// libbtcomm: the exchange steps of the sharded path over RCCL (include/btcomm.h).  One rank per GPU; every call is issued on the
// rank's context stream, so it orders with the kernels that produce / consume the buffers.
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "../../../include/btcomm.h"
#include "../bt_internal.hpp"

using namespace bt;

struct bt_comm {
    bt_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    uint64_t *d_sizes = nullptr;   // world * (world + 3) counters for the size exchanges
};

#define BT_NCCL(call)                                                                      \
    do {                                                                                   \
        ncclResult_t _r = (call);                                                          \
        if (_r != ncclSuccess) return fail(std::string(#call) + ": " + ncclGetErrorString(_r)); \
    } while (0)

extern "C" {

int bt_comm_unique_id(uint8_t id[BT_COMM_ID_BYTES]) {
    if (!id) return fail("bt_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) <= BT_COMM_ID_BYTES, "ncclUniqueId does not fit BT_COMM_ID_BYTES");
    ncclUniqueId u;
    BT_NCCL(ncclGetUniqueId(&u));
    std::memset(id, 0, BT_COMM_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return BT_OK;
}

int bt_comm_init(bt_ctx *ctx, const uint8_t id[BT_COMM_ID_BYTES], int rank, int world_size, bt_comm **out) {
    if (!ctx || !id || !out) return fail("bt_comm_init: null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail("bt_comm_init: rank outside [0, world_size)");
    BT_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    bt_comm *c = new bt_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world_size;
    ncclResult_t r = ncclCommInitRank(&c->comm, world_size, u, rank);
    hipError_t e = r == ncclSuccess ? hipMalloc(reinterpret_cast<void **>(&c->d_sizes), (size_t)world_size * (world_size + 3) * 8) : hipSuccess;
    if (r != ncclSuccess || e != hipSuccess) {
        if (c->comm) ncclCommDestroy(c->comm);
        delete c;
        return fail(r != ncclSuccess ? std::string("ncclCommInitRank: ") + ncclGetErrorString(r) : std::string("bt_comm_init: ") + hipGetErrorString(e));
    }
    *out = c;
    return BT_OK;
}

int bt_comm_destroy(bt_comm *c) {
    if (!c) return BT_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return BT_OK;
}

int bt_comm_rank(bt_comm *c, int *rank, int *world_size) {
    if (!c) return fail("bt_comm_rank: null handle");
    if (rank) *rank = c->rank;
    if (world_size) *world_size = c->world;
    return BT_OK;
}

int bt_comm_allreduce_hist(bt_comm *c, uint64_t *d_hist, uint64_t n) {
    if (!c || !d_hist) return fail("bt_comm_allreduce_hist: null argument");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(c->ctx->device));
    BT_NCCL(ncclAllReduce(d_hist, d_hist, n, ncclUint64, ncclSum, c->comm, c->ctx->stream));
    return BT_OK;
}

// every rank contributes `count` 64-bit words; h_all[world * count] receives all of them in rank order
static int exchange_sizes(bt_comm *c, const uint64_t *h_mine, uint64_t count, uint64_t *h_all) {
    BT_HIP(hipMemcpyAsync(c->d_sizes + (size_t)c->rank * count, h_mine, count * 8, hipMemcpyHostToDevice, c->ctx->stream));
    BT_NCCL(ncclAllGather(c->d_sizes + (size_t)c->rank * count, c->d_sizes, count, ncclUint64, c->comm, c->ctx->stream));
    BT_HIP(hipMemcpyAsync(h_all, c->d_sizes, (size_t)c->world * count * 8, hipMemcpyDeviceToHost, c->ctx->stream));
    BT_HIP(hipStreamSynchronize(c->ctx->stream));
    return BT_OK;
}

// The variable-size collectives agree on failure: every rank sends, next to its sizes, what it can receive (its capacity, whether
// its buffers are there); all ranks evaluate the same conditions on the same exchanged numbers, so either all of them post their
// sends / receives or none does — a rank that returned an error while its peers wait in ncclSend would hang the job.
// A grouped call that fails half-way is closed (ncclGroupEnd) before the error is returned.
#define BT_NCCL_IN_GROUP(call)                                                             \
    do {                                                                                   \
        ncclResult_t _r = (call);                                                          \
        if (_r != ncclSuccess) {                                                           \
            (void)ncclGroupEnd();                                                          \
            return fail(std::string(#call) + ": " + ncclGetErrorString(_r));               \
        }                                                                                  \
    } while (0)

int bt_comm_gather_summaries(bt_comm *c, const uint32_t *d_local, uint64_t local_words, uint32_t *d_out, uint64_t out_capacity, uint64_t *h_offsets) {
    if (!c || !h_offsets) return fail("bt_comm_gather_summaries: null argument");
    BT_HIP(hipSetDevice(c->ctx->device));
    // per rank: {words it sends, words it can receive (rank 0's capacity; 0 elsewhere), 1 if its pointers fit its sizes}
    const uint64_t mine[3] = {local_words, c->rank == 0 ? (d_out ? out_capacity : 0) : 0, (local_words && !d_local) ? 0u : 1u};
    std::vector<uint64_t> all((size_t)c->world * 3);
    if (exchange_sizes(c, mine, 3, all.data()) != BT_OK) return BT_ERR;
    h_offsets[0] = 0;
    bool pointers_ok = true;
    for (int r = 0; r < c->world; ++r) {
        h_offsets[r + 1] = h_offsets[r] + all[(size_t)r * 3];
        pointers_ok = pointers_ok && all[(size_t)r * 3 + 2] != 0;
    }
    if (!pointers_ok) return fail("bt_comm_gather_summaries: a rank passed a null buffer with a non-zero size");
    if (h_offsets[c->world] > all[1]) return fail("bt_comm_gather_summaries: output buffer too small (on rank 0)");   // the same verdict on every rank
    if (c->rank == 0) {
        if (local_words) BT_HIP(hipMemcpyAsync(d_out, d_local, local_words * 4, hipMemcpyDeviceToDevice, c->ctx->stream));
        BT_NCCL(ncclGroupStart());
        for (int r = 1; r < c->world; ++r)
            if (all[(size_t)r * 3]) BT_NCCL_IN_GROUP(ncclRecv(d_out + h_offsets[r], all[(size_t)r * 3], ncclUint32, r, c->comm, c->ctx->stream));
        BT_NCCL(ncclGroupEnd());
    } else if (local_words) {
        BT_NCCL(ncclSend(d_local, local_words, ncclUint32, 0, c->comm, c->ctx->stream));
    }
    return BT_OK;
}

int bt_comm_allgatherv(bt_comm *c, const uint8_t *d_local, uint64_t local_bytes, uint8_t *d_out, uint64_t out_capacity, uint64_t *h_offsets) {
    if (!c || !h_offsets) return fail("bt_comm_allgatherv: null argument");
    BT_HIP(hipSetDevice(c->ctx->device));
    const uint64_t mine[2] = {local_bytes, ((local_bytes && !d_local) || !d_out) ? 0u : out_capacity};
    std::vector<uint64_t> all((size_t)c->world * 2);
    if (exchange_sizes(c, mine, 2, all.data()) != BT_OK) return BT_ERR;
    h_offsets[0] = 0;
    for (int r = 0; r < c->world; ++r) h_offsets[r + 1] = h_offsets[r] + all[(size_t)r * 2];
    for (int r = 0; r < c->world; ++r)
        if (h_offsets[c->world] > all[(size_t)r * 2 + 1] && h_offsets[c->world]) return fail("bt_comm_allgatherv: a rank's output buffer is missing or too small");
    // every part is one broadcast from its owner into its place in every rank's output
    BT_NCCL(ncclGroupStart());
    for (int r = 0; r < c->world; ++r) {
        const uint64_t n = all[(size_t)r * 2];
        if (!n) continue;
        BT_NCCL_IN_GROUP(ncclBroadcast(r == c->rank ? (const void *)d_local : (const void *)(d_out + h_offsets[r]), d_out + h_offsets[r], n, ncclUint8, r, c->comm, c->ctx->stream));
    }
    BT_NCCL(ncclGroupEnd());
    return BT_OK;
}

int bt_comm_alltoallv_matches(bt_comm *c, const uint8_t *d_send, const uint64_t *h_send_bytes, uint8_t *d_recv, uint64_t recv_capacity, uint64_t *h_recv_bytes) {
    if (!c || !h_send_bytes || !h_recv_bytes) return fail("bt_comm_alltoallv_matches: null argument");
    BT_HIP(hipSetDevice(c->ctx->device));
    const int W = c->world;
    uint64_t send_total = 0;
    for (int r = 0; r < W; ++r) send_total += h_send_bytes[r];
    // per rank: its W send sizes, then its receive capacity (0 without a buffer), then 1 if its send buffer is there
    std::vector<uint64_t> mine((size_t)W + 2), all((size_t)W * (W + 2));
    for (int r = 0; r < W; ++r) mine[r] = h_send_bytes[r];
    mine[W] = d_recv ? recv_capacity : 0;
    mine[W + 1] = (send_total && !d_send) ? 0u : 1u;
    if (exchange_sizes(c, mine.data(), (uint64_t)W + 2, all.data()) != BT_OK) return BT_ERR;
    for (int q = 0; q < W; ++q) {   // the same checks on every rank
        uint64_t to_q = 0;
        for (int r = 0; r < W; ++r) to_q += all[(size_t)r * (W + 2) + q];
        if (to_q > all[(size_t)q * (W + 2) + W]) return fail("bt_comm_alltoallv_matches: a rank's receive buffer is missing or too small");
        if (!all[(size_t)q * (W + 2) + W + 1]) return fail("bt_comm_alltoallv_matches: a rank passed a null send buffer with non-zero sizes");
    }
    for (int r = 0; r < W; ++r) h_recv_bytes[r] = all[(size_t)r * (W + 2) + c->rank];
    uint64_t send_at = 0, recv_at = 0;
    BT_NCCL(ncclGroupStart());
    for (int r = 0; r < W; ++r) {
        if (r == c->rank) {
            if (h_send_bytes[r]) {
                const hipError_t e = hipMemcpyAsync(d_recv + recv_at, d_send + send_at, h_send_bytes[r], hipMemcpyDeviceToDevice, c->ctx->stream);
                if (e != hipSuccess) {
                    (void)ncclGroupEnd();
                    return fail(std::string("bt_comm_alltoallv_matches: ") + hipGetErrorString(e));
                }
            }
        } else {
            if (h_send_bytes[r]) BT_NCCL_IN_GROUP(ncclSend(d_send + send_at, h_send_bytes[r], ncclUint8, r, c->comm, c->ctx->stream));
            if (h_recv_bytes[r]) BT_NCCL_IN_GROUP(ncclRecv(d_recv + recv_at, h_recv_bytes[r], ncclUint8, r, c->comm, c->ctx->stream));
        }
        send_at += h_send_bytes[r];
        recv_at += h_recv_bytes[r];
    }
    BT_NCCL(ncclGroupEnd());
    return BT_OK;
}

}  // extern "C"
