// libbtgpu: gibbs_chain_kernel — a whole chain of a noise driver (InferenceEngine.cpp:77-98 per iteration; estimateNoise :135-276,
// estimateNoiseAndGenotypes :384-472) as ONE launch with every tile of the sampler resident: one wavefront per tile, whatever its kind — tiles of
// two-haplotype clusters run simple_sweeps, all others the general sweep — the iterations' exchange with the host through bt_noise_chain.hpp's mailbox.
//
// Why one kernel for all tiles (the sampling launches of the default mode use three, one per tile kind, on streams of their own): the workgroups of a chain
// wait for each other every iteration, so ALL of them must be resident at once.  Kernels launched on different HIP streams only run concurrently when the
// streams sit on different hardware queues, and the runtime deals its (by default four) hardware queues to the process's streams round robin: two launch
// classes of one chain on one hardware queue run one after the other, and the first waits for the second for ever (measured: with GPU_MAX_HW_QUEUES=2
// every chain stalled, with 4 the first chain of a process, with 8 none).  One launch has no such dependence, and with one workgroup shape the residency
// check is exact: grid <= workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor at this block size and LDS need) x CUs.
// Its own translation unit: both sweep bodies are part of this kernel.
#define BT_SWEEP_INLINE
#include "bt_gibbs_kernel.hpp"

namespace {
using namespace bt;
__global__ __launch_bounds__(LANES, 2) void gibbs_chain_kernel(const TileDesc *__restrict__ tiles, uint8_t *__restrict__ pool, const GParams *__restrict__ Pg,
                                                                const NoiseChainCtl *__restrict__ ctl, TraceCfg tr) {
    unsigned long long *arg = reinterpret_cast<unsigned long long *>(const_cast<NoiseChainCtl *>(ctl));
    if (blockIdx.x >= ctl->num_tiles) {
        // A workgroup without a tile: the batch has large tables whose per-iteration refill is shared out among the workgroups of the chain (bt_noise_help.hpp),
        // but few tiles — a 2 000-group batch at thirty samples is 70 tiles, 70 wavefronts for what ucache_prefill_kernel spreads over the whole GPU between
        // ordinary launches — so the launch is topped up with helpers: they wait for every table, take work units until there are none, and arrive.
        Env env{tiles, pool, Pg, nullptr, 0xFFFFFFFFu};
        if (!nc_begin(ctl)) return;
        for (uint32_t i = ctl->it_begin; i < ctl->n_iterations; ++i) {
            if (i > ctl->it_begin) {
                if (!nc_wait_table(ctl, i)) break;
                noise_help(env, ctl);
            }
            if (!nc_iteration_end(ctl, i)) break;
        }
        return;
    }
    if (((const TileDesc BT_CAS *)tiles)[blockIdx.x].simple) gibbs_body<true>(tiles, pool, Pg, OP_NOISE_CHAIN, 0u, 0u, arg, tr, nullptr);
    else gibbs_body<false>(tiles, pool, Pg, OP_NOISE_CHAIN, 0u, 0u, arg, tr, nullptr);
}
}  // namespace

namespace bt {
hipError_t launch_gibbs_chain_kernel(unsigned grid, uint32_t lds, hipStream_t st, const TileDesc *tiles, uint8_t *pool, const GParams *P, const NoiseChainCtl *ctl, TraceCfg tr) {
    hipLaunchKernelGGL(gibbs_chain_kernel, dim3(grid), dim3(LANES), lds, st, tiles, pool, P, ctl, tr);
    return hipGetLastError();
}
hipError_t occupancy_gibbs_chain_kernel(int *blocks_per_cu, uint32_t lds) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void *>(gibbs_chain_kernel), (int)LANES, lds);
}
hipError_t prepare_gibbs_chain_kernel(int max_lds) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gibbs_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
}
}  // namespace bt
