// libbtgpu: k-mer packing, ntHash and Bloom-filter kernels (gfx950).
//
//   bt_kmers_from_sequence  <- KmerPair::move / getLexicographicalLowestKmer (Kmer.tpp:44-255)
//   bt_nthash_batch         <- NTP64 (nthash.hpp:262-282)
//   bt_bloom_*              <- KmerBloom / ThreadedKmerBloom (KmerBloom.cpp:54-286), BloomFilter.hpp
//
// All of these are integer, HBM/latency-bound kernels: one k-mer per lane, 256-lane workgroups,
// grid-stride loops; no MFMA.
#include "bt_internal.hpp"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>

using namespace bt;

namespace {

constexpr unsigned BLOCK = 256;

// ---------------------------------------------------------------------------------------------
// sequence -> canonical k-mers.  A workgroup stages its tile of the sequence plus the k-1
// preceding characters in LDS as 2-bit codes (0xFF = not a nucleotide), then every lane builds
// the window that ENDS at its position.
// ---------------------------------------------------------------------------------------------
constexpr unsigned SEQ_TILE = 1024;   // positions per workgroup iteration

__global__ __launch_bounds__(BLOCK) void kmers_from_sequence_kernel(const char *__restrict__ seq, uint64_t len, unsigned k,
                                                                    uint64_t *__restrict__ kmers, uint8_t *__restrict__ valid) {
    __shared__ uint8_t codes[SEQ_TILE + 64];
    const uint64_t num_tiles = (len + SEQ_TILE - 1) / SEQ_TILE;
    for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const uint64_t tile_start = tile * SEQ_TILE;
        const uint64_t halo = k - 1;
        // stage [tile_start - halo, tile_start + SEQ_TILE)
        for (unsigned j = threadIdx.x; j < SEQ_TILE + halo; j += BLOCK) {
            int64_t pos = (int64_t)tile_start - (int64_t)halo + (int64_t)j;
            uint8_t c = 0xFF;
            if (pos >= 0 && (uint64_t)pos < len) {
                int code = nt_code(seq[pos]);
                c = code < 0 ? 0xFF : (uint8_t)code;
            }
            codes[j] = c;
        }
        __syncthreads();
        for (unsigned j = threadIdx.x; j < SEQ_TILE; j += BLOCK) {
            uint64_t pos = tile_start + j;
            if (pos >= len) break;
            // window = codes[j .. j + k - 1]  (sequence positions pos-k+1 .. pos)
            Kmer fw{0, 0};
            bool ok = true;
            for (unsigned i = 0; i < k; ++i) {
                uint8_t c = codes[j + i];
                ok = ok && (c != 0xFF);
                uint64_t v = (uint64_t)(c & 3u);
                if (i < 32u) fw.lo |= v << (2u * i);
                else fw.hi |= v << (2u * (i - 32u));
            }
            if (ok) {
                Kmer can = kmer_canonical(fw, k);
                kmers[2 * pos] = can.lo;
                kmers[2 * pos + 1] = can.hi;
                valid[pos] = 1;
            } else {
                kmers[2 * pos] = 0;
                kmers[2 * pos + 1] = 0;
                valid[pos] = 0;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void nthash_kernel(const uint64_t *__restrict__ kmers, uint64_t n, unsigned k, int seeded,
                                                       uint32_t seed, uint64_t *__restrict__ out) {
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        Kmer a{kmers[2 * i], kmers[2 * i + 1]};
        uint64_t h = nthash64(a, k);
        if (seeded) h = nthash64_seeded(h, k, seed);
        out[i] = h;
    }
}

__global__ __launch_bounds__(BLOCK) void bloom_insert_kernel(BloomView b, const uint64_t *__restrict__ kmers, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        Kmer a{kmers[2 * i], kmers[2 * i + 1]};
        bloom_insert(nthash64(a, b.k), b);
    }
}

__global__ __launch_bounds__(BLOCK) void bloom_contains_kernel(BloomView b, const uint64_t *__restrict__ kmers, uint64_t n,
                                                               uint8_t *__restrict__ hits) {
    for (uint64_t i = blockIdx.x * (uint64_t)BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        Kmer a{kmers[2 * i], kmers[2 * i + 1]};
        hits[i] = bloom_contains(nthash64(a, b.k), b) ? 1 : 0;
    }
}

// KmerBloom::calcOptNumBloomBits / calcOptNumHashes (KmerBloom.cpp:134-146): float fpr, double math
uint64_t opt_num_bits(float fpr, uint64_t num_kmers) {
    double ln2 = std::log(2);
    return (uint64_t)std::ceil(-(num_kmers * std::log(fpr) / ln2 / ln2));
}
uint32_t opt_num_hashes(uint64_t num_bits, uint64_t num_kmers) {
    double frac = (double)num_bits / (double)num_kmers;
    return (uint32_t)std::ceil(frac * std::log(2));
}

int alloc_bloom(bt_ctx *ctx, bt_bloom *b) {
    b->stride = (((b->num_bits + 7) / 8) + 3) & ~3ULL;
    b->bytes = b->stride * b->num_sub;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&b->d_words), b->bytes));
    BT_HIP(hipMemsetAsync(b->d_words, 0, b->bytes, ctx->stream));
    return BT_OK;
}

}  // namespace

extern "C" {

int bt_kmers_from_sequence(bt_ctx *ctx, const char *d_seq, uint64_t len, uint32_t k, uint64_t *d_kmers, uint8_t *d_valid) {
    if (!ctx) return fail("bt_kmers_from_sequence: null ctx");
    if (k < 1 || k > 64) return fail("bt_kmers_from_sequence: k must be in 1..64");
    if (len == 0) return BT_OK;
    BT_HIP(hipSetDevice(ctx->device));
    unsigned grid = grid_for((len + SEQ_TILE - 1) / SEQ_TILE, 1, ctx->num_cu * 8);
    hipLaunchKernelGGL(kmers_from_sequence_kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, d_seq, len, k, d_kmers, d_valid);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_nthash_batch(bt_ctx *ctx, const uint64_t *d_kmers, uint64_t n, uint32_t k, int seeded, uint32_t seed, uint64_t *d_hash) {
    if (!ctx) return fail("bt_nthash_batch: null ctx");
    if (k < 1 || k > 64) return fail("bt_nthash_batch: k must be in 1..64");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(nthash_kernel, dim3(grid_for(n, BLOCK, ctx->num_cu * 16)), dim3(BLOCK), 0, ctx->stream, d_kmers, n, k, seeded,
                       seed, d_hash);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_bloom_create(bt_ctx *ctx, uint64_t num_kmers, float fpr, uint32_t k, int threaded, bt_bloom **out) {
    if (!ctx || !out) return fail("bt_bloom_create: null argument");
    if (k < 1 || k > 64) return fail("bt_bloom_create: k must be in 1..64");
    if (!(fpr > 0.0f && fpr < 1.0f)) return fail("bt_bloom_create: fpr must be in (0,1)");
    bt_bloom *b = new bt_bloom();
    b->ctx = ctx;
    b->k = k;
    if (threaded) {
        // ThreadedKmerBloom ctor (KmerBloom.cpp:204-215): ceil(num_kmers / float(65536)) per sub-filter
        b->num_sub = BT_NUM_SUB_BLOOMS;
        float per = std::ceil((float)num_kmers / (float)BT_NUM_SUB_BLOOMS);
        b->num_kmers = (uint64_t)per;
    } else {
        b->num_sub = 1;
        b->num_kmers = num_kmers;
    }
    if (b->num_kmers < 1) b->num_kmers = 1;   // KmerBloom.cpp:54: max(num_kmers_in, 1)
    b->num_bits = opt_num_bits(fpr, b->num_kmers);
    b->num_hashes = opt_num_hashes(b->num_bits, b->num_kmers);
    int rc = alloc_bloom(ctx, b);
    if (rc != BT_OK) {
        delete b;
        return rc;
    }
    *out = b;
    return BT_OK;
}

int bt_bloom_load(bt_ctx *ctx, const char *prefix, uint32_t k, bt_bloom **out) {
    if (!ctx || !prefix || !out) return fail("bt_bloom_load: null argument");
    std::string p(prefix);
    std::ifstream meta(p + ".bloomMeta");
    if (!meta.is_open()) return fail("ERROR: Unable to open file " + p + ".bloomMeta");
    std::string line;
    std::getline(meta, line);
    std::vector<std::string> tok;
    {
        std::stringstream ss(line);
        for (std::string item; std::getline(ss, item, '\t');) tok.push_back(item);
    }
    if (tok.size() != 3) return fail("bt_bloom_load: malformed " + p + ".bloomMeta");
    bt_bloom *b = new bt_bloom();
    b->ctx = ctx;
    b->k = k;
    b->num_sub = 1;
    try {
        b->num_kmers = (uint64_t)std::stol(tok[0]);
        b->num_bits = (uint64_t)std::stol(tok[1]);
        if ((uint32_t)std::stoi(tok[2]) != k) {
            delete b;
            return fail("bt_bloom_load: k-mer size in " + p + ".bloomMeta differs from k");
        }
    } catch (...) {
        delete b;
        return fail("bt_bloom_load: malformed " + p + ".bloomMeta");
    }
    if (b->num_kmers < 1 || b->num_bits < 1) {
        delete b;
        return fail("bt_bloom_load: malformed " + p + ".bloomMeta");
    }
    b->num_hashes = opt_num_hashes(b->num_bits, b->num_kmers);   // not stored; recomputed (KmerBloom.cpp:84)
    int rc = alloc_bloom(ctx, b);
    if (rc != BT_OK) {
        delete b;
        return rc;
    }
    const uint64_t nbytes = (b->num_bits + 7) / 8;
    std::vector<uint8_t> host(b->stride, 0);
    std::ifstream data(p + ".bloomData", std::ios::in | std::ios::binary);
    if (!data.is_open()) {
        bt_bloom_destroy(b);
        return fail("ERROR: Unable to open file " + p + ".bloomData");
    }
    data.read(reinterpret_cast<char *>(host.data()), (std::streamsize)nbytes);
    hipError_t e = hipMemcpyAsync(b->d_words, host.data(), b->stride, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        bt_bloom_destroy(b);
        return fail(std::string("bt_bloom_load: ") + hipGetErrorString(e));
    }
    *out = b;
    return BT_OK;
}

int bt_bloom_save(bt_bloom *b, const char *prefix) {
    if (!b || !prefix) return fail("bt_bloom_save: null argument");
    if (b->num_sub != 1) return fail("bt_bloom_save: only single filters can be saved (as in the reference)");
    std::string p(prefix);
    const uint64_t nbytes = (b->num_bits + 7) / 8;
    std::vector<uint8_t> host(b->stride);
    BT_HIP(hipSetDevice(b->ctx->device));
    BT_HIP(hipMemcpyAsync(host.data(), b->d_words, b->stride, hipMemcpyDeviceToHost, b->ctx->stream));
    BT_HIP(hipStreamSynchronize(b->ctx->stream));
    std::ofstream meta(p + ".bloomMeta");
    if (!meta.is_open()) return fail("ERROR: Unable to write file " + p + ".bloomMeta");
    meta << std::to_string(b->num_kmers) << "\t" << std::to_string(b->num_bits) << "\t" << std::to_string(b->k) << std::endl;
    meta.close();
    std::ofstream data(p + ".bloomData", std::ios::out | std::ios::binary);
    if (!data.is_open()) return fail("ERROR: Unable to write file " + p + ".bloomData");
    data.write(reinterpret_cast<const char *>(host.data()), (std::streamsize)nbytes);
    data.close();
    return BT_OK;
}

int bt_bloom_destroy(bt_bloom *b) {
    if (!b) return BT_OK;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->d_words) (void)hipFree(b->d_words);
    delete b;
    return BT_OK;
}

int bt_bloom_info(bt_bloom *b, uint64_t *num_kmers, uint64_t *num_bits, uint32_t *num_hashes, uint32_t *num_sub_filters,
                  uint64_t *device_bytes) {
    if (!b) return fail("bt_bloom_info: null bloom");
    if (num_kmers) *num_kmers = b->num_kmers;
    if (num_bits) *num_bits = b->num_bits;
    if (num_hashes) *num_hashes = b->num_hashes;
    if (num_sub_filters) *num_sub_filters = b->num_sub;
    if (device_bytes) *device_bytes = b->bytes;
    return BT_OK;
}

int bt_bloom_insert_batch(bt_bloom *b, const uint64_t *d_kmers, uint64_t n) {
    if (!b) return fail("bt_bloom_insert_batch: null bloom");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(b->ctx->device));
    hipLaunchKernelGGL(bloom_insert_kernel, dim3(grid_for(n, BLOCK, b->ctx->num_cu * 16)), dim3(BLOCK), 0, b->ctx->stream, b->view(),
                       d_kmers, n);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_bloom_contains_batch(bt_bloom *b, const uint64_t *d_kmers, uint64_t n, uint8_t *d_hits) {
    if (!b) return fail("bt_bloom_contains_batch: null bloom");
    if (n == 0) return BT_OK;
    BT_HIP(hipSetDevice(b->ctx->device));
    hipLaunchKernelGGL(bloom_contains_kernel, dim3(grid_for(n, BLOCK, b->ctx->num_cu * 16)), dim3(BLOCK), 0, b->ctx->stream, b->view(),
                       d_kmers, n, d_hits);
    BT_CHECK_LAUNCH();
    return BT_OK;
}

int bt_bloom_read_bits(bt_bloom *b, uint32_t sub, uint8_t *h_out, uint64_t nbytes) {
    if (!b || !h_out) return fail("bt_bloom_read_bits: null argument");
    if (sub >= b->num_sub) return fail("bt_bloom_read_bits: sub-filter index out of range");
    const uint64_t want = (b->num_bits + 7) / 8;
    if (nbytes < want) return fail("bt_bloom_read_bits: buffer too small");
    BT_HIP(hipSetDevice(b->ctx->device));
    BT_HIP(hipMemcpyAsync(h_out, reinterpret_cast<const uint8_t *>(b->d_words) + (uint64_t)sub * b->stride, want, hipMemcpyDeviceToHost,
                          b->ctx->stream));
    BT_HIP(hipStreamSynchronize(b->ctx->stream));
    return BT_OK;
}

int bt_bloom_clear(bt_bloom *b) {
    if (!b) return fail("bt_bloom_clear: null bloom");
    BT_HIP(hipSetDevice(b->ctx->device));
    BT_HIP(hipMemsetAsync(b->d_words, 0, b->bytes, b->ctx->stream));
    return BT_OK;
}

}  // extern "C"
