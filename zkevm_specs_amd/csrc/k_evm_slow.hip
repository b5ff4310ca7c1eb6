// EVM circuit: the general build of the hot gadgets, for the pairs the fast kernel deferred (see EVM_FAST in evm_circuit.hpp)
#define EVM_DEFERRED_KERNEL 1
#include "evm_kernel.hpp"

void zk_launch_evm_deferred(hipStream_t st, const EvmArgs& a, u32* status, ZkTally* tally) {
    hipLaunchKernelGGL(evm_deferred_kernel, dim3(256), dim3(256), 0, st, a, status, tally);
}
