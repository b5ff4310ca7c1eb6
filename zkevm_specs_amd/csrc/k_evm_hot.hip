// EVM circuit: the hot instantiation (BASELINE config 3's opcode mix), fast build: fallback paths are deferred to the general build
#define EVM_FAST 1
#include "evm_kernel.hpp"

// e0 / e1: start / stop events riding on the dispatch (either may be null)
void zk_launch_evm_hot(hipStream_t st, u32 grid, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e0, hipEvent_t e1) {
    if (e0 || e1)
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_ALL, ZK_HOT_OCC, EVM_HOT_BLOCK>), dim3(grid), dim3(EVM_HOT_BLOCK), 0, st, e0, e1, 0, a, group_start, status, tally);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(evm_steps_kernel<EVM_GROUP_ALL, ZK_HOT_OCC, EVM_HOT_BLOCK>), dim3(grid), dim3(EVM_HOT_BLOCK), 0, st, a, group_start, status, tally);
}
