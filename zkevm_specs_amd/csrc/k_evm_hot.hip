// EVM circuit: the hot instantiation (BASELINE config 3's opcode mix), fast build: fallback paths are deferred to the cold launch
#define EVM_FAST 1
#include "evm_kernel.hpp"

void zk_launch_evm_hot(hipStream_t st, u32 grid, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e0) {
    if (e0)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_ALL, ZK_HOT_OCC, EVM_HOT_BLOCK>), dim3(grid), dim3(EVM_HOT_BLOCK), 0, st, e0, nullptr, 0, a, group_start, status, tally);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_ALL, ZK_HOT_OCC, EVM_HOT_BLOCK>), dim3(grid), dim3(EVM_HOT_BLOCK), 0, st, a, group_start, status, tally);
}
