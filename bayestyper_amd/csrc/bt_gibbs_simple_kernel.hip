// libbtgpu: gibbs_simple_kernel — the Gibbs schedule for launch classes made of tiles of two-haplotype clusters only (90 % of the
// clusters of a whole-genome batch) and the three sampling operations.  Its own translation unit: the sweep (bt_gibbs_simple.hpp) is the
// kernel body, compiled for GIBBS_SIMPLE_WAVES wavefronts per SIMD; what runs once per chain (chain start, entry / exit of the sweep loop,
// the drain of the collected runs) and the rare exact paths are out-of-line functions with their own register allocation.
#define BT_SIMPLE_TU 1
#include "bt_gibbs_kernel.hpp"

namespace {
using namespace bt;
#ifndef GIBBS_SIMPLE_WAVES
#define GIBBS_SIMPLE_WAVES 3
#endif
__global__ __launch_bounds__(LANES, GIBBS_SIMPLE_WAVES) void gibbs_simple_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg,
                                                                                  int op, uint32_t arg0, uint32_t arg1, unsigned long long *__restrict__ hist, TraceCfg tr,
                                                                                  const uint32_t *__restrict__ tile_list) {
    gibbs_body<true>(tiles, pool, Pg, op, arg0, arg1, hist, tr, tile_list);
}
}  // namespace

namespace bt {
hipError_t launch_gibbs_simple_kernel(unsigned grid, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, int op, uint32_t a0, uint32_t a1,
                                      unsigned long long *hist, TraceCfg tr, const uint32_t *tile_list) {
    hipLaunchKernelGGL(gibbs_simple_kernel, dim3(grid), dim3(LANES), lds, st, tiles, pool, P, op, a0, a1, hist, tr, tile_list);
    return hipGetLastError();
}
#ifdef BT_PROF
hipError_t simple_prof_read(unsigned long long *h_out32, int reset) {   // this unit's copy of the phase counters (tools/prof_class.py)
    hipError_t e = hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(g_bt_prof), 32 * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_bt_prof), z, 32 * 8);
    }
    return e;
}
#endif
hipError_t prepare_gibbs_simple_kernel(int max_lds) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gibbs_simple_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
}
}  // namespace bt
