// libbtgpu: gibbs_hot_kernel — the general Gibbs schedule for launch classes whose tiles all keep every vertex's hot arrays in LDS for the whole
// launch (single clusters and the nested groups whose vertices fit together: the multi-variant and nested-SV classes of a batch) and the three
// sampling operations.  Its own translation unit: with BT_HOT_ALL the accesses to those arrays are LDS accesses at compile time (bt_gibbs_tile.hpp:
// hot_core) — ds instructions — where the general kernel reaches them through generic pointers that may point to LDS or HBM (flat instructions: both
// wait counters, and a wait for any of them is a wait for all memory operations in flight).
#define BT_HOT_ALL 1
#define BT_PACKED 1
#define BT_NO_NOISE_CHAIN 1
#ifndef BT_SWEEP_OUTLINE
#define BT_SWEEP_INLINE
#endif
#include "bt_gibbs_kernel.hpp"

namespace {
using namespace bt;
__global__ __launch_bounds__(LANES * 8, GIBBS_WAVES) void gibbs_hot_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg, int op, uint32_t arg0,
                                                                           uint32_t arg1, unsigned long long *__restrict__ hist, TraceCfg tr, const uint32_t *__restrict__ tile_list) {
    if (!(op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN)) return;   // (the other operations read the arrays in HBM: gibbs_kernel; chains of a noise driver: gibbs_chain_kernel)
    gibbs_body<false>(tiles, pool, Pg, op, arg0, arg1, hist, tr, tile_list);
}
}  // namespace

namespace bt {
hipError_t launch_gibbs_hot_kernel(unsigned grid, unsigned block, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, int op, uint32_t a0, uint32_t a1,
                                   unsigned long long *hist, TraceCfg tr, const uint32_t *tile_list) {
    hipLaunchKernelGGL(gibbs_hot_kernel, dim3(grid), dim3(block), lds, st, tiles, pool, P, op, a0, a1, hist, tr, tile_list);
    return hipGetLastError();
}
#ifdef BT_PROF
hipError_t hot_prof_read(unsigned long long *h_out32, int reset) {   // this unit's copy of the phase counters (tools/prof_class.py)
    hipError_t e = hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(g_bt_prof), 32 * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_bt_prof), z, 32 * 8);
    }
    return e;
}
#endif
hipError_t occupancy_gibbs_hot_kernel(int *blocks_per_cu, int block, uint32_t lds) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(gibbs_hot_kernel), block, lds);
}
hipError_t prepare_gibbs_hot_kernel(int max_lds) { return hipFuncSetAttribute(reinterpret_cast<const void *>(gibbs_hot_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds); }
}  // namespace bt
