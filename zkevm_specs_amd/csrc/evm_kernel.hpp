// The EVM step kernel template, instantiated by k_evm_hot.hip and k_evm_cold.hip.
#pragma once
#include "kernels.hpp"

// ---------------------------------------------------------------------------------------
// EVM circuit kernels: one lane per step pair (curr, next).  Lanes are assigned through a
// permutation sorted by (kernel group, execution state) so that a 64-lane wavefront runs ONE
// gadget body instead of serialising the ~10 different execution states a window of consecutive
// steps contains; each group has its own kernel instantiation (evm_circuit.hpp).
// group_start[g] .. group_start[g+1] is the lane range of group g inside `perm`.
// ---------------------------------------------------------------------------------------
#ifndef ZK_HOT_OCC
#define ZK_HOT_OCC 2  // waves per SIMD the hot EVM kernel is compiled for
#endif
template <int G, int OCC, int BLOCK>
__global__ __launch_bounds__(BLOCK, OCC) void evm_steps_kernel(EvmArgs a, const u32* group_start, u32* status, ZkTally* tally) {
    evm_args_resolve(a);  // open-time verdicts (dense RW index, directory size, EndBlock aggregates) live in HBM
    // lane range: with the state-sorted mapping the hot instantiation owns [0, group_start[COLD]) and the
    // cold one [group_start[COLD], n); without it both walk all pairs and skip the other's states
    u32 lo = 0, hi = a.n_pairs;
    if (a.perm) {  // lane ranges of the sorted mapping (hot bins padded to whole wavefronts with EVM_NO_PAIR lanes)
        if (G == EVM_GROUP_COLD) { lo = group_start[EVM_GROUP_COLD]; hi = group_start[EVM_N_GROUPS]; }
        else hi = group_start[EVM_GROUP_COLD];
    }
    u64 t = (u64)lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ u32 s_stage[G == EVM_GROUP_ALL ? EVM_STAGE_ENTRIES * EVM_STAGE_STRIDE : 1];
    __shared__ u64 s_dir[G == EVM_GROUP_ALL ? EVM_DIR_LDS_U64 : 1];
    if (G == EVM_GROUP_ALL) {  // the grid covers every pair: one step per lane
        if ((u64)blockIdx.x * blockDim.x >= (u64)hi) return;  // the grid is sized for the largest possible padding
        if (EV_PROF_ON(a)) {  // tuning aid: entry stamps (core clock, 100 MHz wall clock)
            a.prof[EV_PROF_WAVE * 8 + 5] = __builtin_readcyclecounter();
            a.prof[EV_PROF_WAVE * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        }
        // small bytecode directories (the usual case: a handful of contracts) are mirrored in LDS by the whole block, so that
        // resolving curr.code_hash costs no dependent HBM round trips
        const bool dir_in_lds = a.codes.n != 0 && a.codes.n <= EVM_DIR_MAX_ENTRIES && a.codes.mask < EVM_DIR_MAX_SLOTS;
        if (dir_in_lds) {
            const u32 n_slots = a.codes.mask + 1u;
            for (u32 k = threadIdx.x; k < EVM_DIR_SLOT_U64; k += blockDim.x) {
                const u32 s0 = 2 * k < n_slots ? a.codes.slots[2 * k] : ZK_EMPTY_SLOT, s1 = 2 * k + 1 < n_slots ? a.codes.slots[2 * k + 1] : ZK_EMPTY_SLOT;
                s_dir[k] = (u64)s0 | ((u64)s1 << 32);
            }
            const u64* e = (const u64*)a.codes.entries;
            for (u32 k = threadIdx.x; k < a.codes.n * 12u; k += blockDim.x) s_dir[EVM_DIR_SLOT_U64 + k] = e[k];
        }
        u32 code = 0;
        u64 idx = t;
        if (t < (u64)hi && a.perm) idx = a.perm[t];
        const bool mine = t < (u64)hi && idx != (u64)EVM_NO_PAIR;
        // both steps of every pair of the wavefront go to LDS first (lane quads fetch them); the gadgets read them from there
        __attribute__((address_space(3))) u32* my = (__attribute__((address_space(3))) u32*)s_stage + threadIdx.x;
        const bool wide = evm_stage_steps_quad(a, (u32)idx, mine, my - (threadIdx.x & 63u));
        if (dir_in_lds) __syncthreads();
        if (mine) {
            code = evm_check_step<G>(a, idx, wide ? (EVM_LDS32_PTR) nullptr : (EVM_LDS32_PTR)my,
                                     dir_in_lds ? (EVM_LDS_PTR)(__attribute__((address_space(3))) u64*)s_dir : (EVM_LDS_PTR) nullptr);
            if (code == ZK_NOT_MINE) code = 0;
            else if (status) status[idx] = code;
        }
        tally_commit(tally, idx, code);
        if (EV_PROF_ON(a)) a.prof[EV_PROF_WAVE * 8 + 7] = __builtin_amdgcn_s_memrealtime();
    } else {  // small grid, grid-stride loop
        const u64 stride = (u64)gridDim.x * blockDim.x;
        for (; t < (u64)hi; t += stride) {
            const u64 idx = a.perm ? (u64)a.perm[t] : t;
            u32 code = idx == (u64)EVM_NO_PAIR ? (u32)ZK_NOT_MINE : evm_check_step<G>(a, idx);
            if (code == ZK_NOT_MINE) code = 0;
            else if (status) status[idx] = code;
            tally_commit(tally, idx, code);  // ballot over the lanes still in the loop
        }
    }
}
