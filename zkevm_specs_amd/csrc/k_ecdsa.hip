// secp256k1 ECDSA verification kernel (secp256k1.hpp)
#include "kernels.hpp"

// One lane per signature (L = 1) or a lane pair per signature (L = 2: each lane runs one half of the GLV split with its own
// 128-doubling chain, the partial sums meet through one DPP-free cross-lane exchange at the end), or four (L = 4, round 6: lanes 2 / 3
// carry the halves of u1 G and add them in the ladder's addition slots, secp256k1.hpp ecdsa_partial4; 256-lane blocks).  Integer-ALU bound.
template <int L>
__global__ __launch_bounds__(L == 4 ? 256 : 64) void ecdsa_verify_kernel(EcdsaArgs a, u32* status, ZkTally* tally) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 i = a.first + gid / L;
    const int role = (int)(gid % L);
    const bool valid = i < a.n;
#if ZK_ECDSA_TAB_INTERLEAVED
    u32* tab = a.qtab + gid;  // word w of entry e at (e * 24 + w) * lanes + lane
    const u64 tab_stride = a.qtab_lanes;
#else
    // each lane's 15 x 96 B table is contiguous: the window digit is data dependent, so the lanes of a wavefront read
    // different entries, and with the lane-interleaved layout every 4-byte word came out of a different 64-byte sector
    // (FETCH_SIZE 5.2 GB per 2^17-signature launch, profiles/r02_row_kernels_profile.json)
    u32* tab = a.qtab + gid * (15u * 24u);
    const u64 tab_stride = 1;
#endif
    EcdsaPrep pr;
    u32 st = ECDSA_NOT_VERIFIED;
    SpPoint part = sp_infinity();
    if (valid) {
        st = ecdsa_prepare(a, i, pr, role == 0);
        if (st == ECDSA_PENDING) part = L == 4 ? ecdsa_partial4(pr, role, tab, tab_stride, a.gcomb)
                                               : ecdsa_partial(pr, L == 1 ? 0 : role, L == 1 ? 1 : role, tab, tab_stride, a.gcomb);
    }
    if (L >= 2) {  // every lane takes part in the exchange; lane 2i receives the partial sum of lane 2i + 1 ...
        SpPoint other;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            other.X.v[w] = (u32)__shfl_xor((int)part.X.v[w], 1);
            other.Y.v[w] = (u32)__shfl_xor((int)part.Y.v[w], 1);
            other.Z.v[w] = (u32)__shfl_xor((int)part.Z.v[w], 1);
        }
        if (valid && (role & 1) == 0 && st == ECDSA_PENDING) sp_add_ip(part, other);
    }
    if (L == 4) {  // ... and lane 4i the sum of lanes 4i + 2, 4i + 3
        SpPoint other;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            other.X.v[w] = (u32)__shfl_xor((int)part.X.v[w], 2);
            other.Y.v[w] = (u32)__shfl_xor((int)part.Y.v[w], 2);
            other.Z.v[w] = (u32)__shfl_xor((int)part.Z.v[w], 2);
        }
        if (valid && role == 0 && st == ECDSA_PENDING) sp_add_ip(part, other);
    }
    u32 code = 0;
    if (valid && role == 0) {
        code = st == ECDSA_PENDING ? ecdsa_verdict(pr, part) : st;
        if (status) status[i] = code;
        if (i < a.n0) { if (a.out) a.out[i * a.out_stride] = code; }
        else if (a.out1) a.out1[(i - a.n0) * a.out_stride1] = code;
    }
    tally_commit(tally, i, code);
}

__global__ __launch_bounds__(64) void ecdsa_comb_build_kernel(u32* table) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ECDSA_COMB_ENTRIES) ecdsa_comb_entry(e, table);
}
void zk_launch_ecdsa_comb_build(hipStream_t st, u32* table) {
    hipLaunchKernelGGL(ecdsa_comb_build_kernel, dim3((ECDSA_COMB_ENTRIES + 63) / 64), dim3(64), 0, st, table);
}
void zk_launch_ecdsa(hipStream_t st, const EcdsaArgs& a0, u32* status, ZkTally* tally) {
    // 64-lane blocks: 2^14 signatures are only 256 (512 as lane pairs) wavefronts.  The per-lane key tables cost 1,440 bytes
    // per lane (ZK_ECDSA_CHUNK_LANES of them are allocated, 189 MB): a larger batch runs as consecutive launches over the same
    // tables (in order on the stream, so a chunk's tables are free when the next one starts).
    EcdsaArgs a = a0;
    const u64 per_chunk = a.qtab_lanes / a.lanes_per_sig;  // signatures per launch
    for (a.first = 0; a.first < a.n; a.first += per_chunk) {
        const u64 m = a.n - a.first < per_chunk ? a.n - a.first : per_chunk;
        const u32 grid = (u32)((m * a.lanes_per_sig + 63) / 64);
        // (the four-lane form in 256-lane blocks: its wavefronts are few, and four of them on one CU share its instruction cache — one
        // wavefront per CU took 1.21 ms at 2^11 signatures where four per CU take 1.01)
        if (a.lanes_per_sig == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(ecdsa_verify_kernel<4>), dim3((grid + 3) / 4), dim3(256), 0, st, a, status, tally);
        else if (a.lanes_per_sig == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(ecdsa_verify_kernel<2>), dim3(grid), dim3(64), 0, st, a, status, tally);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(ecdsa_verify_kernel<1>), dim3(grid), dim3(64), 0, st, a, status, tally);
    }
}
