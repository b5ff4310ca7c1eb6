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
    // Entry: three independent loads in flight together — the open-time verdicts (EvmDyn), the lane range of the sorted mapping
    // (group_start) and this lane's pair (perm; read before the range is known: the buffer is padded past every grid) — instead
    // of three dependent round trips of 2-3 us each under load.
    u32 perm_t = 0;
    if (G == EVM_GROUP_ALL && a.perm) perm_t = a.perm[(u64)blockIdx.x * blockDim.x + threadIdx.x];
    if (G == EVM_GROUP_ALL && a.defer_count_twin != nullptr) {  // resident sessions: this launch readies the next pass's tally and counter
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.defer_count_twin = 0u;
        tally_clear_twin(tally);
    }
    __shared__ u64 s_dir[G == EVM_GROUP_ALL ? EVM_DIR_LDS_U64 : 1];
    const bool have_dir = G == EVM_GROUP_ALL && a.dyn != nullptr && a.codes.slots != nullptr && a.codes.entries != nullptr;
    if (have_dir) {
        for (u32 k = threadIdx.x; k < EVM_DIR_SLOT_U64; k += blockDim.x) s_dir[k] = (u64)a.codes.slots[2 * k] | ((u64)a.codes.slots[2 * k + 1] << 32);
        const u64* e = (const u64*)a.codes.entries;
        for (u32 k = threadIdx.x; k < EVM_DIR_MAX_ENTRIES * 12u; k += blockDim.x) s_dir[EVM_DIR_SLOT_U64 + k] = e[k];
    }
    evm_args_resolve(a);  // open-time verdicts (dense RW index, directory size, EndBlock aggregates) live in HBM
    // lane range: with the state-sorted mapping the hot instantiation owns [0, group_start[COLD]) and the
    // cold one [group_start[COLD], n); without it both walk all pairs and skip the other's states
    u32 lo = 0, hi = a.n_pairs;
    if (a.perm) {  // lane ranges of the sorted mapping (hot bins padded to whole wavefronts with EVM_NO_PAIR lanes)
        if (G == EVM_GROUP_ALL) hi = group_start[EVM_GROUP_WARM];
        else { lo = group_start[G]; hi = group_start[G + 1]; }
    }
    u64 t = (u64)lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    // One step per lane, both steps of the pair staged in LDS: the hot instantiation, and the warm one when it is launched with
    // EVM_HOT_BLOCK-lane blocks over the sorted mapping (its gadgets read ~40 step cells each: from HBM, one dependent round trip
    // per cell, a wavefront of SHA3 / EXP steps took 150k-220k clocks).
    const bool staged_launch = G == EVM_GROUP_ALL || (G == EVM_GROUP_WARM && BLOCK == EVM_HOT_BLOCK && a.perm != nullptr);
    __shared__ u32 s_stage[(G == EVM_GROUP_ALL || (G == EVM_GROUP_WARM && BLOCK == EVM_HOT_BLOCK)) ? EVM_STAGE_ENTRIES * EVM_STAGE_STRIDE : 1];
    if (staged_launch) {  // one step per lane.  Hot: the grid covers every pair of the range.  Warm: the launch may be smaller than the
        // range (its size is not known to the host before the session's first collect: a capped grid instead of one block per
        // 128 pairs of the whole trace, most of which would only find the range empty) and walks it block-stride
      for (u32 vblock = blockIdx.x;; vblock += gridDim.x) {
        t = (u64)lo + (u64)vblock * blockDim.x + threadIdx.x;
        if ((u64)lo + (u64)vblock * blockDim.x >= (u64)hi) return;  // the grid is sized for the largest possible range / padding
        if (G != EVM_GROUP_ALL && vblock != blockIdx.x) __syncthreads();  // the stage columns of the previous iteration are done with
        if (G != EVM_GROUP_ALL) perm_t = t < (u64)hi ? a.perm[t] : 0u;
        if (G == EVM_GROUP_ALL && EV_PROF_ON(a)) {  // tuning aid: entry stamps (core clock, 100 MHz wall clock)
            a.prof[EV_PROF_WAVE * 8 + 5] = __builtin_readcyclecounter();
            a.prof[EV_PROF_WAVE * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        }
        // small bytecode directories (the usual case: a handful of contracts) are mirrored in LDS by the whole block, so that
        // resolving curr.code_hash costs no dependent HBM round trips.  The copy does not wait for the directory's size (EvmDyn):
        // it always takes the first EVM_DIR_MAX_SLOTS slots and EVM_DIR_MAX_ENTRIES entries of the session's (larger, pre-filled)
        // buffers; whether the mirror is complete — and used — is decided from the size afterwards.
        const bool dir_in_lds = have_dir && a.codes.n != 0 && a.codes.n <= EVM_DIR_MAX_ENTRIES && a.codes.mask < EVM_DIR_MAX_SLOTS;
        u32 code = 0;
        u64 idx = t;
        if (t < (u64)hi && a.perm) idx = perm_t;  // lo == 0 for the hot instantiation: t is the lane's global index
        const bool mine = t < (u64)hi && idx != (u64)EVM_NO_PAIR;
        // both steps of every pair of the wavefront go to LDS first (lane quads fetch them); the gadgets read them from there
        __attribute__((address_space(3))) u32* my = (__attribute__((address_space(3))) u32*)s_stage + threadIdx.x;
        // from the packed step records when the session has them (zk_evm_open), else from the step rows (zk_evm_verify: one pass only)
        const bool wide = a.step_recs ? evm_stage_steps_quad(a, (u32)idx, mine, my - (threadIdx.x & 63u))
                                      : evm_stage_steps_rows_quad(a, (u32)idx, mine, my - (threadIdx.x & 63u));
        if (dir_in_lds) __syncthreads();
        if (mine) {
#if EVM_FAST
            // an unstaged pair (a step cell wider than its record field) or a pair that ran into a fallback path goes to the
            // deferred list: the cold launch evaluates it with the general build of the same gadgets
            code = wide ? (u32)ZK_DEFERRED_BASE + 8u
                        : evm_check_step<G>(a, idx, (EVM_LDS32_PTR)my, dir_in_lds ? (EVM_LDS_PTR)(__attribute__((address_space(3))) u64*)s_dir : (EVM_LDS_PTR) nullptr);
            if (code >= ZK_DEFERRED_BASE && code != ZK_NOT_MINE) {
#ifdef ZK_DEFER_DEBUG  // tuning aid: per-reason counters in the last slots of the profiling buffer (ZK_EVM_PROF=1)
                if (a.prof) atomicAdd(&a.prof[4095 * 8 - 16 + (code & 15u)], 1ull);
#endif
                a.defer_list[atomicAdd(a.defer_count, 1u)] = (u32)idx;
                code = 0;
            } else
#else
#ifdef ZK_WARM_TWICE  // tuning build: the same step a second time — its stamps are those of warm instruction / data caches
            if (G == EVM_GROUP_WARM)
                code = evm_check_step<G>(a, idx, wide ? (EVM_LDS32_PTR) nullptr : (EVM_LDS32_PTR)my, (EVM_LDS_PTR) nullptr);
#endif
            code = evm_check_step<G>(a, idx, wide ? (EVM_LDS32_PTR) nullptr : (EVM_LDS32_PTR)my,
                                     dir_in_lds ? (EVM_LDS_PTR)(__attribute__((address_space(3))) u64*)s_dir : (EVM_LDS_PTR) nullptr);
#endif
            if (code == ZK_NOT_MINE) code = 0;
            else if (status) status[idx] = code;
        }
        tally_commit(tally, idx, code);
        if (G == EVM_GROUP_ALL && EV_PROF_ON(a)) a.prof[EV_PROF_WAVE * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        if (G == EVM_GROUP_ALL) return;
      }
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

#if defined(EVM_DEFERRED_KERNEL)
// The pairs the fast (hot) kernel deferred (EVM_FAST, evm_circuit.hpp): every hot gadget in its general form — generic indices,
// cell-by-cell key compares, step rows from HBM, field-arithmetic transitions.  Launched by the host only when a pass left
// such pairs behind (zk_collect / zk_read_status read the count): well-formed witnesses never pay for it.
__global__ __launch_bounds__(256, 1) void evm_deferred_kernel(EvmArgs a, u32* status, ZkTally* tally) {
    evm_args_resolve(a);
    const u32 n_def = *a.defer_count;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < (u64)n_def; k += stride) {
        const u64 idx = a.defer_list[k];
        u32 code = evm_check_step<EVM_GROUP_ALL>(a, idx);
        if (code == ZK_NOT_MINE) code = 0;
        else if (status) status[idx] = code;
        tally_commit(tally, idx, code);
    }
}
#endif
