// Does hipExtAnyOrderLaunch let two kernels on ONE stream run concurrently on this GPU / runtime?  (hip_ext.h says the flag is "not supported on AMD GFX9xx
// boards" for the module-launch form.)  Two single-thread kernels that need each other, launched back to back on the same stream: in order they time out,
// any order they see each other.  Also: how many kernels of a stream are in flight at once (N kernels that each wait for all N flags).
// build: hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder_probe ; run on the GPU box.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void need_all(uint32_t *flags, int me, int n, unsigned long long ticks, uint32_t *ok) {
    __hip_atomic_store(&flags[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    int seen = 0;
    while (seen < n && (unsigned long long)wall_clock64() - t0 < ticks) {
        seen = 0;
        for (int i = 0; i < n; ++i) seen += __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        __builtin_amdgcn_s_sleep(10);
    }
    ok[me] = seen == n && (unsigned long long)wall_clock64() - t0 < ticks;
}

int main() {
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int n : {2, 8, 64, 512}) {
        for (int any = 0; any < 2; ++any) {
            uint32_t *flags, *ok;
            hipMalloc(&flags, n * 4);
            hipMalloc(&ok, n * 4);
            hipMemset(flags, 0, n * 4);
            hipMemset(ok, 0, n * 4);
            hipDeviceSynchronize();
            const unsigned long long ticks = (unsigned long long)khz * 20;   // 20 ms
            for (int i = 0; i < n; ++i) {
                if (any) hipExtLaunchKernelGGL(need_all, dim3(1), dim3(64), 1024 * (1 + i % 8), st, nullptr, nullptr, hipExtAnyOrderLaunch, flags, i, n, ticks, ok);
                else hipLaunchKernelGGL(need_all, dim3(1), dim3(64), 1024 * (1 + i % 8), st, flags, i, n, ticks, ok);
            }
            hipError_t e = hipStreamSynchronize(st);
            std::vector<uint32_t> h(n);
            hipMemcpy(h.data(), ok, n * 4, hipMemcpyDeviceToHost);
            int good = 0;
            for (uint32_t v : h) good += v ? 1 : 0;
            printf("%d kernels on one stream, %s: %d saw all the others (%s)\n", n, any ? "hipExtAnyOrderLaunch" : "in order", good, hipGetErrorString(e));
            hipFree(flags);
            hipFree(ok);
        }
    }
    return 0;
}
