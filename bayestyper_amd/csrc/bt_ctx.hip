// libbtgpu: context, device memory, HIP-event timers, error reporting.
#include "bt_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace bt {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
int fail(const std::string &msg) {
    set_error(msg);
    return BT_ERR;
}
}  // namespace bt

namespace {
// Two single-thread kernels that need each other: each raises its flag and waits (bounded) for the other's.  Both see the other's flag only when the two
// streams they were launched on run CONCURRENTLY — HIP streams that the runtime has put on the same hardware queue run one after the other.
__global__ void stream_pair_probe_kernel(uint32_t *flags, int me, unsigned long long timeout_ticks) {
    __hip_atomic_store(&flags[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    uint32_t seen = 0;
    while (!seen && (unsigned long long)wall_clock64() - t0 < timeout_ticks) {
        seen = __hip_atomic_load(&flags[1 - me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_sleep(10);
    }
    // (a kernel that starts after the other one gave up finds the other's flag raised: only "both overlapped" counts, so each also reports whether it waited to the end)
    flags[2 + me] = seen && (unsigned long long)wall_clock64() - t0 < timeout_ticks ? 1u : 0u;
}
}  // namespace

namespace bt {
// do kernels on streams a and b overlap?  (the pair of kernels of bt_ctx_clone; d_flags: 16 bytes of device memory)
static hipError_t streams_overlap(bt_ctx *ctx, hipStream_t a, hipStream_t b, uint32_t *d_flags, unsigned long long ticks, bool *yes) {
    uint32_t h[4] = {0, 0, 0, 0};
    hipError_t e = hipStreamSynchronize(a);
    if (e == hipSuccess) e = hipStreamSynchronize(b);
    if (e == hipSuccess) e = hipMemcpy(d_flags, h, 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stream_pair_probe_kernel, dim3(1), dim3(1), 0, a, d_flags, 0, ticks);
    hipLaunchKernelGGL(stream_pair_probe_kernel, dim3(1), dim3(1), 0, b, d_flags, 1, ticks);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(a);
    if (e == hipSuccess) e = hipStreamSynchronize(b);
    if (e == hipSuccess) e = hipMemcpy(h, d_flags, 16, hipMemcpyDeviceToHost);
    *yes = e == hipSuccess && h[2] && h[3];
    return e;
}

hipError_t ctx_class_streams(bt_ctx *ctx, unsigned n, int prio, hipStream_t *out) {
    const bool probe = getenv("BT_CLASS_STREAMS_NO_PROBE") == nullptr;
    if (ctx->class_streams_probed && (ctx->class_streams_for != ctx->stream || ctx->class_streams_prio != prio)) {   // probed against another stream: start over
        for (hipStream_t st : ctx->class_streams) ctx->retired_streams.push_back(st);   // (a sampler built before may still launch on them)
        ctx->class_streams.clear();
        ctx->class_streams_probed = false;
        ctx->class_streams_concurrent = 0;
    }
    if (ctx->class_streams.size() < n) {
        int wall_khz = 0;
        if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
        const unsigned long long ticks = (unsigned long long)wall_khz * 2;   // 2 ms
        uint32_t *d_flags = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_flags), 16);
        if (e != hipSuccess) return e;
        std::vector<hipStream_t> rejected;
        // (the context's stream may be the NULL stream: kernels on it are probed all the same)
        for (int attempt = 0; attempt < 16 && ctx->class_streams.size() < n && e == hipSuccess; ++attempt) {
            hipStream_t cand = nullptr;
            e = hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio);
            if (e != hipSuccess) break;
            bool ok = true;
            if (probe) {
                e = streams_overlap(ctx, ctx->stream, cand, d_flags, ticks, &ok);
                for (size_t i = 0; ok && e == hipSuccess && i < ctx->class_streams.size(); ++i) e = streams_overlap(ctx, ctx->class_streams[i], cand, d_flags, ticks, &ok);
            }
            if (e == hipSuccess && ok) {
                // (streams accepted after a stand-in had to be taken stay uncounted: the count is of a prefix that is concurrent throughout)
                if (ctx->class_streams_concurrent == ctx->class_streams.size() && probe) ctx->class_streams_concurrent += 1;
                ctx->class_streams.push_back(cand);
            } else
                rejected.push_back(cand);
        }
        while (e == hipSuccess && ctx->class_streams.size() < n && !rejected.empty()) {   // not enough hardware queues: streams of their own all the same, in order with another one at worst
            ctx->class_streams.push_back(rejected.back());
            rejected.pop_back();
        }
        for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
        (void)hipFree(d_flags);
        if (e != hipSuccess) return e;
        if (ctx->class_streams.size() < n) return hipErrorOutOfMemory;
        ctx->class_streams_for = ctx->stream;
        ctx->class_streams_prio = prio;
        ctx->class_streams_probed = true;
        if (getenv("BT_GIBBS_DEBUG")) std::fprintf(stderr, "bt_ctx: %zu launch-class streams probed against the context's stream and each other\n", ctx->class_streams.size());
    }
    for (unsigned i = 0; i < n; ++i) out[i] = ctx->class_streams[i];
    return hipSuccess;
}
}  // namespace bt

extern "C" {

const char *bt_last_error(void) { return bt::g_last_error.c_str(); }

int bt_version(void) { return 100; }

int bt_device_count(int *count) {
    if (!count) return bt::fail("bt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return BT_OK;
}

int bt_ctx_create(int device_id, bt_ctx **out) {
    if (!out) return bt::fail("bt_ctx_create: null out");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return bt::fail("bt_ctx_create: no HIP device visible (libbtgpu has no CPU fallback)");
    }
    if (device_id < 0 || device_id >= n) return bt::fail("bt_ctx_create: device_id out of range");
    BT_HIP(hipSetDevice(device_id));
    bt_ctx *c = new bt_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    BT_HIP(hipGetDeviceProperties(&prop, device_id));
    c->num_cu = prop.multiProcessorCount;
    BT_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    *out = c;
    return BT_OK;
}

int bt_ctx_clone(bt_ctx *ctx, bt_ctx **out) {
    if (!ctx || !out) return bt::fail("bt_ctx_clone: null argument");
    BT_HIP(hipSetDevice(ctx->device));
    // The new stream must run CONCURRENTLY with the original's: the runtime deals its hardware queues (four by default) to the process's streams round robin,
    // and two streams on one hardware queue run their kernels one after the other (a sampler built on the clone while a resident noise chain occupies the
    // original would wait for the chain to end).  Candidates are probed with a pair of kernels that need each other; the first that overlaps is taken.
    // The probe launches on ctx->stream and SYNCHRONISES it (every attempt): a caller-owned stream set with bt_ctx_set_stream must have no work in flight
    // that depends on the calling thread.
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
    uint32_t *d_flags = nullptr;
    BT_HIP(hipMalloc(reinterpret_cast<void **>(&d_flags), 16));
    hipStream_t chosen = nullptr;
    std::vector<hipStream_t> rejected;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 12 && !chosen && e == hipSuccess; ++attempt) {
        hipStream_t cand = nullptr;
        e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) break;
        uint32_t h[4] = {0, 0, 0, 0};
        e = hipMemcpy(d_flags, h, 16, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        const unsigned long long ticks = (unsigned long long)wall_khz * 3;   // 3 ms
        if (e == hipSuccess) {
            hipLaunchKernelGGL(stream_pair_probe_kernel, dim3(1), dim3(1), 0, ctx->stream, d_flags, 0, ticks);
            hipLaunchKernelGGL(stream_pair_probe_kernel, dim3(1), dim3(1), 0, cand, d_flags, 1, ticks);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(cand);
        if (e == hipSuccess) e = hipMemcpy(h, d_flags, 16, hipMemcpyDeviceToHost);
        if (e == hipSuccess && h[2] && h[3]) chosen = cand;
        else rejected.push_back(cand);
    }
    if (!chosen && !rejected.empty()) {   // no candidate overlapped (a runtime with a single hardware queue): still a stream of its own, in order with the original at worst
        chosen = rejected.back();
        rejected.pop_back();
    }
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
    (void)hipFree(d_flags);
    if (e != hipSuccess || !chosen) {
        if (chosen) (void)hipStreamDestroy(chosen);   // (picked from the rejected candidates before the error was looked at)
        return bt::fail(std::string("bt_ctx_clone: ") + hipGetErrorString(e));
    }
    bt_ctx *c = new bt_ctx();
    c->device = ctx->device;
    c->num_cu = ctx->num_cu;
    c->own_stream = chosen;
    c->stream = c->own_stream;
    *out = c;
    return BT_OK;
}

int bt_ctx_destroy(bt_ctx *ctx) {
    if (!ctx) return BT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->pool_cache) (void)hipFree(ctx->pool_cache);
    for (auto &h : ctx->host_cache)
        if (h.p) (void)hipHostFree(h.p);
    for (hipStream_t st : ctx->class_streams) (void)hipStreamDestroy(st);
    for (hipStream_t st : ctx->retired_streams) (void)hipStreamDestroy(st);
    for (int b = 0; b < 2; ++b) {
        if (ctx->kmc_pin[b]) (void)hipHostFree(ctx->kmc_pin[b]);
        if (ctx->kmc_dev[b]) (void)hipFree(ctx->kmc_dev[b]);
    }
    delete ctx;
    return BT_OK;
}

int bt_ctx_set_stream(bt_ctx *ctx, void *hip_stream) {
    if (!ctx) return bt::fail("bt_ctx_set_stream: null ctx");
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return BT_OK;
}

int bt_ctx_use_default_stream(bt_ctx *ctx) {
    if (!ctx) return bt::fail("bt_ctx_use_default_stream: null ctx");
    ctx->stream = nullptr;
    return BT_OK;
}

int bt_sync(bt_ctx *ctx) {
    if (!ctx) return bt::fail("bt_sync: null ctx");
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

int bt_ctx_info(bt_ctx *ctx, int *num_cu, uint64_t *hbm_total, uint64_t *hbm_free, char *arch, size_t arch_len) {
    if (!ctx) return bt::fail("bt_ctx_info: null ctx");
    BT_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    BT_HIP(hipGetDeviceProperties(&prop, ctx->device));
    size_t f = 0, t = 0;
    BT_HIP(hipMemGetInfo(&f, &t));
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (hbm_total) *hbm_total = t;
    if (hbm_free) *hbm_free = f + ctx->pool_cache_bytes;   // (the cached pool is given back as soon as a sampler needs the room: bt_gibbs.hip)
    if (arch && arch_len) {
        std::strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return BT_OK;
}

int bt_malloc(bt_ctx *ctx, size_t bytes, void **d_out) {
    if (!ctx || !d_out) return bt::fail("bt_malloc: null argument");
    BT_HIP(hipSetDevice(ctx->device));
    void *p = nullptr;
    BT_HIP(hipMalloc(&p, bytes ? bytes : 16));
    *d_out = p;
    return BT_OK;
}

int bt_free(bt_ctx *ctx, void *d_ptr) {
    if (!ctx) return bt::fail("bt_free: null ctx");
    if (!d_ptr) return BT_OK;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    BT_HIP(hipFree(d_ptr));
    return BT_OK;
}

int bt_memset(bt_ctx *ctx, void *d_ptr, int value, size_t bytes) {
    if (!ctx) return bt::fail("bt_memset: null ctx");
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return BT_OK;
}

int bt_memcpy_h2d(bt_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return bt::fail("bt_memcpy_h2d: null ctx");
    if (!bytes) return BT_OK;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

int bt_memcpy_d2h(bt_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return bt::fail("bt_memcpy_d2h: null ctx");
    if (!bytes) return BT_OK;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

int bt_memcpy_d2d(bt_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    if (!ctx) return bt::fail("bt_memcpy_d2d: null ctx");
    if (!bytes) return BT_OK;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

int bt_timer_create(bt_ctx *ctx, bt_timer **out) {
    if (!ctx || !out) return bt::fail("bt_timer_create: null argument");
    BT_HIP(hipSetDevice(ctx->device));
    bt_timer *t = new bt_timer();
    t->ctx = ctx;
    BT_HIP(hipEventCreate(&t->start));
    BT_HIP(hipEventCreate(&t->stop));
    *out = t;
    return BT_OK;
}

int bt_timer_destroy(bt_timer *t) {
    if (!t) return BT_OK;
    if (t->start) (void)hipEventDestroy(t->start);
    if (t->stop) (void)hipEventDestroy(t->stop);
    delete t;
    return BT_OK;
}

int bt_timer_start(bt_timer *t) {
    if (!t) return bt::fail("bt_timer_start: null timer");
    BT_HIP(hipEventRecord(t->start, t->ctx->stream));
    return BT_OK;
}

int bt_timer_stop(bt_timer *t) {
    if (!t) return bt::fail("bt_timer_stop: null timer");
    BT_HIP(hipEventRecord(t->stop, t->ctx->stream));
    return BT_OK;
}

int bt_timer_elapsed_ms(bt_timer *t, float *ms) {
    if (!t || !ms) return bt::fail("bt_timer_elapsed_ms: null argument");
    BT_HIP(hipEventSynchronize(t->stop));
    BT_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return BT_OK;
}

}  // extern "C"
