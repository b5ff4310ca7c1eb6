// EVM circuit: the warm instantiation (copy- / keccak- / exp-table gadgets: SHA3, *COPY, LOG, EXP), general build
#include "evm_kernel.hpp"

// With the state-sorted mapping: one step per lane over the warm lane range, step pairs staged in LDS like the hot kernel's
// (EVM_HOT_BLOCK-lane blocks; `warm_lanes` = the range's size once a collect has read it, 0 = size the grid for every pair: blocks
// past the range exit on their first load).  Without the mapping: the grid-stride form over all pairs, `grid` blocks of 256.
void zk_launch_evm_warm(hipStream_t st, u32 grid, u32 warm_lanes, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e1) {
    if (a.perm) {
        const u32 lanes = warm_lanes ? warm_lanes : a.n_pairs;
        u32 g = (lanes + EVM_HOT_BLOCK - 1) / EVM_HOT_BLOCK;
        if (!warm_lanes && g > 512u) g = 512u;  // range not known yet: a capped grid that walks the range block-stride (evm_kernel.hpp)
        if (e1)
            hipExtLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_WARM, 1, EVM_HOT_BLOCK>), dim3(g), dim3(EVM_HOT_BLOCK), 0, st, nullptr, e1, 0, a, group_start, status, tally);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_WARM, 1, EVM_HOT_BLOCK>), dim3(g), dim3(EVM_HOT_BLOCK), 0, st, a, group_start, status, tally);
        return;
    }
    if (e1)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_WARM, 1, 256>), dim3(grid), dim3(256), 0, st, nullptr, e1, 0, a, group_start, status, tally);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_WARM, 1, 256>), dim3(grid), dim3(256), 0, st, a, group_start, status, tally);
}
