// EVM circuit: the warm instantiation (copy- / keccak- / exp-table gadgets: SHA3, *COPY, LOG, EXP), general build
#include "evm_kernel.hpp"

void zk_launch_evm_warm(hipStream_t st, u32 grid, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e1) {
    if (e1)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_WARM, 1, 256>), dim3(grid), dim3(256), 0, st, nullptr, e1, 0, a, group_start, status, tally);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_WARM, 1, 256>), dim3(grid), dim3(256), 0, st, a, group_start, status, tally);
}
