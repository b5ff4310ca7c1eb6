// EVM circuit: the cold instantiation (rarely-taken execution states)
#include "evm_kernel.hpp"

void zk_launch_evm_cold(hipStream_t st, u32 grid, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e1) {
    if (e1)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_COLD, 1, 256>), dim3(grid), dim3(256), 0, st, nullptr, e1, 0, a, group_start, status, tally);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_COLD, 1, 256>), dim3(grid), dim3(256), 0, st, a, group_start, status, tally);
}
