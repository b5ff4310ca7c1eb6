// secp256k1 ECDSA verification kernel (secp256k1.hpp)
#include "kernels.hpp"

// secp256k1 ECDSA verification: one lane per signature (secp256k1.hpp); integer-ALU bound
__global__ __launch_bounds__(64) void ecdsa_verify_kernel(EcdsaArgs a, u32* status, ZkTally* tally) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 code = 0;
    if (i < a.n) {
        code = ecdsa_verify_one(a, i);
        if (status) status[i] = code;
        if (a.out) a.out[i * a.out_stride] = code;
    }
    tally_commit(tally, i, code);
}
void zk_launch_ecdsa(hipStream_t st, const EcdsaArgs& a, u32* status, ZkTally* tally) {
    // 64-lane blocks: 2^14 signatures are only 256 wavefronts, one per CU
    hipLaunchKernelGGL(ecdsa_verify_kernel, dim3((u32)((a.n + 63) / 64)), dim3(64), 0, st, a, status, tally);
}
