// libbtgpu: context, device memory, HIP-event timers, error reporting.
#include "bt_internal.hpp"

#include <cstring>

namespace bt {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
int fail(const std::string &msg) {
    set_error(msg);
    return BT_ERR;
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
    bt_ctx *c = new bt_ctx();
    c->device = ctx->device;
    c->num_cu = ctx->num_cu;
    BT_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    *out = c;
    return BT_OK;
}

int bt_ctx_destroy(bt_ctx *ctx) {
    if (!ctx) return BT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
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
    if (hbm_free) *hbm_free = f;
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
