// State circuit on rows computed from the ops in registers (state_fused.hpp): one lane per row, the previous row from the neighbouring
// lane as in state_rows_compact_kernel; two tallies — the State circuit's and the assignment's.
#include "kernels.hpp"
#include "state_fused.hpp"

template <bool RW>
__global__ __launch_bounds__(256, 2) void state_rows_fused_kernel(StateFusedArgs<RW> a, u32* status, ZkTally* tally, ZkTally* asg_tally) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 first = a.eval_lo + wave * ST_ROWS_PER_WAVE;
    const u64 n = a.rows.n;
    const u64 i_raw = lane == 0 ? (first == 0 ? n - 1 : first - 1) : first + lane - 1;
    const bool evaluate = lane != 0 && i_raw < a.eval_hi;
    const u64 i = i_raw >= n ? n - 1 : i_raw;
    StRow C;
    u32 code = 0, asg_code = 0;
    state_row_from_op<RW>(a.asg, i, C, code, asg_code);
    code = state_check_loaded<1>(a, i, C, C, code);
    if (!evaluate) code = asg_code = 0;
    else if (status) status[i] = code;
    tally_commit(tally, i, code);
    tally_commit(asg_tally, i, asg_code);
}

void zk_launch_state_rows_fused(hipStream_t st, const StateArgs& sa, const AssignArgs& g, u32* status, ZkTally* tally, ZkTally* asg_tally) {
    const int block = 256;
    const u64 rows_per_block = (u64)(block / 64) * ST_ROWS_PER_WAVE;
    const u32 grid = (u32)((sa.eval_hi - sa.eval_lo + rows_per_block - 1) / rows_per_block);
    if (g.rw) {
        StateFusedArgs<true> a;
        (StateArgs&)a = sa;
        a.asg = g;
        hipLaunchKernelGGL(state_rows_fused_kernel<true>, dim3(grid), dim3(block), 0, st, a, status, tally, asg_tally);
    } else {
        StateFusedArgs<false> a;
        (StateArgs&)a = sa;
        a.asg = g;
        hipLaunchKernelGGL(state_rows_fused_kernel<false>, dim3(grid), dim3(block), 0, st, a, status, tally, asg_tally);
    }
}
