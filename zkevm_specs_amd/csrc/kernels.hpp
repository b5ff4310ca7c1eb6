// Device helpers shared by every kernel translation unit + the host launchers the C-ABI layer (zkevm_hip.hip) calls.
// The library is built from several translation units compiled in parallel (build.sh): each k_*.hip defines its kernels
// and a plain C++ launcher; no relocatable device code is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "state_circuit.hpp"
#include "evm_circuit.hpp"
#include "row_circuits.hpp"
#include "copy_circuit.hpp"
#include "sign_circuit.hpp"
#include "keccak_table.hpp"
#include "state_assign.hpp"
#include "secp256k1.hpp"
#include "bytecode_assign.hpp"
#include "copy_assign.hpp"
#include "pi_circuit.hpp"
#include "state_rekey.hpp"

// The single-kernel row sessions keep two tallies and alternate between them: a pass accumulates into one and its first
// lane clears the other for the pass after it, so that no reset kernel sits in front of every evaluation kernel (a kernel
// boundary costs ~10 us of the 77 us State pass at 2^16 rows).  `tally` and its twin are 16 B apart in one 32 B-aligned block.
__device__ __forceinline__ void tally_clear_twin(ZkTally* tally) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ZkTally* twin = (ZkTally*)((uintptr_t)tally ^ (uintptr_t)sizeof(ZkTally));
        twin->fail_count = 0ull;
        twin->first_fail = ~0ull;
    }
}
static_assert(sizeof(ZkTally) == 16, "tally_clear_twin assumes 16-byte tallies");

// One wave-level ballot, then at most one counter atomic per wave and one atomicMin per
// failing lane (failures are rare on real witnesses; the hot path issues no atomics at all).
__device__ __forceinline__ void tally_commit(ZkTally* tally, u64 row, u32 code) {
    const unsigned long long ballot = __ballot(code != 0u);
    if (ballot != 0ull) {
        if (code != 0u) atomicMin(&tally->first_fail, (row << 32) | (unsigned long long)code);
        const int lane = threadIdx.x & 63;
        if (lane == __ffsll((long long)ballot) - 1) atomicAdd(&tally->fail_count, (unsigned long long)__popcll(ballot));
    }
}

#define ST_ROWS_PER_WAVE 63

// ---- launchers (defined in the k_*.hip units) ------------------------------------------------------------------------
// e0 / e1 (optional): the pass's start / stop events ride on the dispatch itself (no event packets between back-to-back passes) when
// zk_state_rows_events_ride(a) says the launch is one kernel of a form that takes them; otherwise the caller records them around the call
void zk_launch_state_rows(hipStream_t st, const StateArgs& a, u32* status, ZkTally* tally, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
bool zk_state_rows_events_ride(const StateArgs& a);
// hot: e0 rides on the dispatch as its start event; cold: e1 as its stop event (either may be null)
void zk_launch_evm_hot(hipStream_t st, u32 grid, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e0, hipEvent_t e1 = nullptr);
void zk_launch_evm_warm(hipStream_t st, u32 grid, u32 warm_lanes, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e1);
void zk_launch_evm_cold(hipStream_t st, u32 grid, const EvmArgs& a, const u32* group_start, u32* status, ZkTally* tally, hipEvent_t e1);
// the pairs the hot (fast) kernel deferred, evaluated by the general build (k_evm_slow.hip); launched only when there are any
void zk_launch_evm_deferred(hipStream_t st, const EvmArgs& a, u32* status, ZkTally* tally);
void zk_launch_bytecode_rows(hipStream_t st, const BytecodeArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally);
void zk_launch_copy_rows(hipStream_t st, const CopyArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally);
void zk_launch_sign_units(hipStream_t st, const SignArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally);
void zk_launch_exp_rows(hipStream_t st, const ExpArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally);
void zk_launch_fr_to_mont(hipStream_t st, const Fr& x, u64* out);
void zk_launch_sign_rpow(hipStream_t st, const Fr& r, u64* out);
void zk_launch_keccak_rpow(hipStream_t st, const Fr& r, u64* out);
void zk_launch_keccak_table(hipStream_t st, const KeccakGenArgs& g, u32* status, ZkTally* tally);
void zk_launch_state_assign(hipStream_t st, const AssignArgs& a, u32* status, ZkTally* tally);
// state_fused.hpp: the State circuit on rows computed from the ops (after zk_launch_state_assign with a.root_rank set)
void zk_launch_state_rows_fused(hipStream_t st, const StateArgs& sa, const AssignArgs& g, u32* status, ZkTally* tally, ZkTally* asg_tally);
void zk_launch_rekey_scan(hipStream_t st, const RekeyArgs& a);  // open-time class masks of the RW -> State re-keying
void zk_launch_state_rekey(hipStream_t st, const RekeyArgs& a, u32* status, ZkTally* tally);
void zk_launch_bca_rpow(hipStream_t st, const Fr& r, u64* out);
void zk_launch_bytecode_assign(hipStream_t st, const BcaArgs& a, u32* status, ZkTally* tally);
void zk_launch_pi_rows(hipStream_t st, const PiArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally);
void zk_launch_pi_copy(hipStream_t st, const PiCopyArgs& a, u32* status, ZkTally* tally);
void zk_launch_cpa_rpow(hipStream_t st, const Fr& r, u64* out);
void zk_launch_copy_assign(hipStream_t st, const CpaArgs& a, u32* status, ZkTally* tally);
void zk_launch_ecdsa(hipStream_t st, const EcdsaArgs& a, u32* status, ZkTally* tally);
void zk_launch_ecdsa_comb_build(hipStream_t st, u32* table);  // the device's 8-bit fixed-base table of G (522 KB), built once
