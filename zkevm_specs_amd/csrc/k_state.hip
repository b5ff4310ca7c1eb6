// State-circuit kernel (state_circuit.hpp)
#include "kernels.hpp"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------
// State circuit kernel.  Column-major cells make every cell load a fully coalesced 32 B/lane
// access (2 x dwordx4).  A lane loads ONLY its own row; what the checks need from the previous
// row arrives from lane - 1 through DPP moves, so every wavefront evaluates 63 rows and its
// lane 0 holds the (read-only) row in front of them.  The next row (Storage / Account last-
// access test) is re-read through L1/L2 by the few rows that need it.
// ---------------------------------------------------------------------------------------
#ifndef ZK_STATE_OCC
#define ZK_STATE_OCC 2  // waves per SIMD the State kernel is compiled for (3 was measured: 168 VGPRs + 32 B scratch, 2^20 rows 0.423 vs 0.404 ms)
#endif
__global__ __launch_bounds__(256, ZK_STATE_OCC) void state_rows_kernel(StateArgs a, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 first = a.eval_lo + wave * ST_ROWS_PER_WAVE;  // first row this wavefront evaluates
    const u64 n = a.rows.n;
    // lane l holds row first + l - 1 (lane 0: the predecessor of `first`, wrapping to n - 1)
    u64 i = lane == 0 ? (first == 0 ? n - 1 : first - 1) : first + lane - 1;
    const bool evaluate = lane != 0 && i < a.eval_hi;
    if (i >= n) i = n - 1;  // lanes past the range still take part in the DPP moves: keep their loads in bounds
    StRow C;
    u32 code = 0;
    state_load_row(a.rows, i, C, code);
    code = state_check_loaded<1>(a, i, C, C, code);
    if (!evaluate) code = 0;
    else if (status) status[i] = code;
    tally_commit(tally, i, code);
}

// The lane-group forms (state_load_row_group<L>): L = 4 lanes per row = 16 rows per wavefront, L = 2 = 32 rows; the first
// row of a wavefront is the halo row in front of the evaluated ones.
#ifndef ZK_STATE_QUAD_OCC
#define ZK_STATE_QUAD_OCC 2
#endif
#ifndef ZK_STATE_QUAD_BLOCK
#define ZK_STATE_QUAD_BLOCK 256
#endif
template <int L>
__global__ __launch_bounds__(ZK_STATE_QUAD_BLOCK, ZK_STATE_QUAD_OCC) void state_rows_group_kernel(StateArgs a, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    constexpr u32 ROWS = 64u / L - 1u;  // evaluated rows per wavefront
    const u32 lane = threadIdx.x & 63u, q = lane & (u32)(L - 1), slot = lane / (u32)L;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 first = a.eval_lo + wave * ROWS;
    const u64 n = a.rows.n;
    // slot s holds row first + s - 1 (slot 0: the predecessor of `first`, wrapping to n - 1)
    u64 i = slot == 0 ? (first == 0 ? n - 1 : first - 1) : first + slot - 1;
    const bool evaluate = slot != 0 && q == 0 && i < a.eval_hi;
    if (i >= n) i = n - 1;  // lanes past the range still take part in the cross-lane moves: keep their loads in bounds
    StRow C;
    u32 code = 0;
    state_load_row_group<L>(a.rows, i, q, C, code);
    code = state_check_loaded<L>(a, i, C, C, code);
    if (!evaluate) code = 0;
    else if (status) status[i] = code;
    tally_commit(tally, i, code);
}

#ifndef ZK_STATE_SMALL_LANES
#define ZK_STATE_SMALL_LANES 4  // lanes per row below 2^18 rows
#endif
// Measured (profiles/r02_state_lanes.txt): 2^16 rows 67.6 us (quad) vs 81.6 us (one lane per row); 2^20 rows 555 us vs 415 us —
// the quad form wins while one wavefront per SIMD is all a launch has, the one-lane form once the chip is full (a quarter of
// the cross-lane traffic and of the redundant per-quad checks).  ZK_STATE_LANES=1|4 overrides (tuning / tests).
static int state_lanes_per_row(u64 rows) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("ZK_STATE_LANES");
        forced = e ? (atoi(e) == 1 ? 1 : atoi(e) == 2 ? 2 : 4) : 0;
    }
    if (forced) return forced;
    return rows < (1ull << 18) ? ZK_STATE_SMALL_LANES : 1;
}
template <int L>
static void launch_group(hipStream_t st, const StateArgs& a, u32* status, ZkTally* tally) {
    const u64 rows_per_block = (u64)(ZK_STATE_QUAD_BLOCK / 64) * (64 / L - 1);
    const u32 grid = (u32)((a.eval_hi - a.eval_lo + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(state_rows_group_kernel<L>), dim3(grid), dim3(ZK_STATE_QUAD_BLOCK), 0, st, a, status, tally);
}
void zk_launch_state_rows(hipStream_t st, const StateArgs& a, u32* status, ZkTally* tally) {
    const int block = 256;
    const int lanes = state_lanes_per_row(a.eval_hi - a.eval_lo);
    if (lanes == 4) { launch_group<4>(st, a, status, tally); return; }
    if (lanes == 2) { launch_group<2>(st, a, status, tally); return; }
    const u64 rows_per_block = (u64)(block / 64) * ST_ROWS_PER_WAVE;  // 63 evaluated rows per wavefront
    const u32 grid = (u32)((a.eval_hi - a.eval_lo + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(state_rows_kernel, dim3(grid), dim3(block), 0, st, a, status, tally);
}
