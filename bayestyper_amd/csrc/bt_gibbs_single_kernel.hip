// libbtgpu: gibbs_single_kernel — the general Gibbs schedule for launch classes made of tiles whose groups are ONE cluster without multicluster k-mers
// (TileDesc::logged; the multi-variant and many-candidate clusters of a batch: 8.5 % of a whole-genome batch's groups, a third of its wavefront-seconds)
// and the three sampling operations.  Its own translation unit (bt_gibbs_tile.hpp: BT_SINGLE): without the nested-group traversal, the multicluster
// sums and the immediate statistics path, and with the candidates evaluated two at a time, the sweep fits GIBBS_SINGLE_WAVES = 3 wavefronts per SIMD
// (gibbs_hot_kernel: 256 registers, two) — VariantClusterGenotyper.cpp:597-785 for nested_variant_cluster_info empty and no multicluster k-mers.
#define BT_HOT_ALL 1
#define BT_PACKED 1
#define BT_NO_NOISE_CHAIN 1
#define BT_SINGLE 1
#ifndef BT_EVAL_BLOCK
#define BT_EVAL_BLOCK 2u
#endif
#define BT_SWEEP_INLINE
#include "bt_gibbs_kernel.hpp"

namespace {
using namespace bt;
#ifndef GIBBS_SINGLE_WAVES
#define GIBBS_SINGLE_WAVES 3
#endif
__global__ __launch_bounds__(LANES * 4, GIBBS_SINGLE_WAVES) void gibbs_single_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg, int op,
                                                                                      uint32_t arg0, uint32_t arg1, unsigned long long *__restrict__ hist, TraceCfg tr,
                                                                                      const uint32_t *__restrict__ tile_list) {
    if (!(op == OP_RUN || op == OP_SWEEP || op == OP_INIT_CHAIN)) return;   // (the other operations read the arrays in HBM: gibbs_kernel; chains of a noise driver: gibbs_chain_kernel)
    gibbs_body<false>(tiles, pool, Pg, op, arg0, arg1, hist, tr, tile_list);
}
}  // namespace

namespace bt {
hipError_t launch_gibbs_single_kernel(unsigned grid, unsigned block, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, int op, uint32_t a0, uint32_t a1,
                                      unsigned long long *hist, TraceCfg tr, const uint32_t *tile_list) {
    hipLaunchKernelGGL(gibbs_single_kernel, dim3(grid), dim3(block), lds, st, tiles, pool, P, op, a0, a1, hist, tr, tile_list);
    return hipGetLastError();
}
#ifdef BT_PROF
hipError_t single_prof_read(unsigned long long *h_out32, int reset) {   // this unit's copy of the phase counters (tools/prof_class.py)
    hipError_t e = hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(g_bt_prof), 32 * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_bt_prof), z, 32 * 8);
    }
    return e;
}
#endif
hipError_t occupancy_gibbs_single_kernel(int *blocks_per_cu, int block, uint32_t lds) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(gibbs_single_kernel), block, lds);
}
hipError_t prepare_gibbs_single_kernel(int max_lds) { return hipFuncSetAttribute(reinterpret_cast<const void *>(gibbs_single_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds); }
}  // namespace bt
