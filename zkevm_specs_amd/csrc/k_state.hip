// State-circuit kernel (state_circuit.hpp)
#include "kernels.hpp"
#include <stdlib.h>
#include <mutex>

// ---------------------------------------------------------------------------------------
// State circuit kernels.  Column-major cells; one lane per row; what the checks need from the previous row arrives from lane - 1
// through DPP moves, so a wavefront evaluates 63 rows and its lane 0 holds the (read-only) row in front of them.
//   state_rows_dma_kernel   (default) the 14 wide cells are ordinary loads, the 42 limb / byte cells stream through a per-wavefront
//                           LDS ring with global_load_lds (state_load_row_dma)
//   state_rows_group_kernel<4>  ZK_STATE_DMA=0: a lane quad per row, everything through registers (round 2's small-batch form,
//                           kept as the comparison point; round 2's one-lane all-register kernel is gone: with the batched MPT
//                           compare it no longer fits 256 registers)
// ---------------------------------------------------------------------------------------
// The same evaluation with the 42 limb / byte cells streamed through a per-wavefront LDS ring (state_load_row_dma).
#ifndef ZK_STATE_DMA_OCC
#define ZK_STATE_DMA_OCC 2
#endif
__global__ __launch_bounds__(256, ZK_STATE_DMA_OCC) void state_rows_dma_kernel(StateArgs a, u32* status, ZkTally* tally) {
    extern __shared__ uint4 st_lds[];
    tally_clear_twin(tally);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 first = a.eval_lo + wave * ST_ROWS_PER_WAVE;
    const u64 n = a.rows.n;
    // lane j holds row rowof(j): the predecessor of `first` (wrapping to n - 1) in lane 0, then first, first + 1, ...; rows past
    // the witness are clamped to n - 1 so that every load stays in bounds (those lanes do not report)
    auto rowof = [&](u32 j) -> u64 {
        u64 r = j == 0 ? (first == 0 ? n - 1 : first - 1) : first + j - 1;
        return r >= n ? n - 1 : r;
    };
    const u64 i_raw = lane == 0 ? (first == 0 ? n - 1 : first - 1) : first + lane - 1;
    const bool evaluate = lane != 0 && i_raw < a.eval_hi;
    const u64 i = rowof(lane);
    const u64 off[2] = {rowof(lane >> 1) * 32u + (lane & 1u) * 16u, rowof(32u + (lane >> 1)) * 32u + (lane & 1u) * 16u};
    uint4* ring = st_lds + (threadIdx.x >> 6) * (ST_DMA_WAVE_BYTES / 16);
    StRow C;
    u32 code = 0;
#ifdef ZK_STATE_PROF  // tuning build (tools/state_wave_timeline.py): per-wavefront clocks + placement instead of the row status
    const u64 t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
#endif
    state_load_row_dma(a.rows, i, off, ring, lane, C, code);
#ifdef ZK_STATE_PROF
    const u64 c1 = __builtin_amdgcn_s_memtime();
#endif
    code = state_check_loaded<1>(a, i, C, C, code);
#if defined(ZK_STATE_PROF) && ZK_STATE_PROF == 2  // per-row clocks of the Storage / Account branch instead of the status
    if (status && evaluate) status[i] = code;
    code = 0;
#elif defined(ZK_STATE_PROF)
    if (!evaluate) code = 0;
    if (status && lane == 0 && first + 17 < n) {
        const u64 t1 = __builtin_amdgcn_s_memrealtime(), c2 = __builtin_amdgcn_s_memtime();
        u32* o = status + first;
        o[0] = (u32)t0; o[1] = (u32)(t0 >> 32); o[2] = (u32)t1; o[3] = (u32)(t1 >> 32);
        o[4] = (u32)(c1 - c0); o[5] = (u32)(c2 - c0);
        o[9] = (u32)(st_prof_stamp[threadIdx.x >> 6][0] - c0); o[10] = (u32)(st_prof_stamp[threadIdx.x >> 6][1] - c0);
        o[11] = (u32)(st_prof_stamp[threadIdx.x >> 6][2] - c0);
        for (int k = 3; k < 7; k++) o[9 + k] = (u32)(st_prof_stamp[threadIdx.x >> 6][k] - c0);
        o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); o[8] = 0xabcd1234u;
    }
#else
    if (!evaluate) code = 0;
    else if (status) status[i] = code;
#endif
    tally_commit(tally, i, code);
}

// ZK_OPT_STATE_COMPACT: the 15-cell rows of a device-assigned witness (st_col / state_load_row derive the limb and byte
// decompositions from the address and storage-key cells): fourteen wide cells per lane, no LDS ring.
#ifndef ZK_STATE_COMPACT_OCC
#define ZK_STATE_COMPACT_OCC 2
#endif
__global__ __launch_bounds__(256, ZK_STATE_COMPACT_OCC) void state_rows_compact_kernel(StateArgs a, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 first = a.eval_lo + wave * ST_ROWS_PER_WAVE;
    const u64 n = a.rows.n;
    const u64 i_raw = lane == 0 ? (first == 0 ? n - 1 : first - 1) : first + lane - 1;
    const bool evaluate = lane != 0 && i_raw < a.eval_hi;
    const u64 i = i_raw >= n ? n - 1 : i_raw;
    StRow C;
    u32 code = 0;
    state_load_row_compact(a.rows, i, C, code);
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

// ZK_STATE_DMA=0|1: the LDS-ring kernel (default on)
static int state_use_dma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ZK_STATE_DMA");
        v = e ? (atoi(e) != 0) : 1;
    }
    return v;
}
template <int L>
static void launch_group(hipStream_t st, const StateArgs& a, u32* status, ZkTally* tally) {
    const u64 rows_per_block = (u64)(ZK_STATE_QUAD_BLOCK / 64) * (64 / L - 1);
    const u32 grid = (u32)((a.eval_hi - a.eval_lo + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(state_rows_group_kernel<L>), dim3(grid), dim3(ZK_STATE_QUAD_BLOCK), 0, st, a, status, tally);
}
bool zk_state_rows_events_ride(const StateArgs& a) { return a.rows.skip || state_use_dma(); }
void zk_launch_state_rows(hipStream_t st, const StateArgs& a, u32* status, ZkTally* tally, hipEvent_t e0, hipEvent_t e1) {
    const int block = 256;
    if (a.rows.skip) {
        const u64 rows_per_block = (u64)(block / 64) * ST_ROWS_PER_WAVE;
        const u32 grid = (u32)((a.eval_hi - a.eval_lo + rows_per_block - 1) / rows_per_block);
        if (e0 || e1) hipExtLaunchKernelGGL(state_rows_compact_kernel, dim3(grid), dim3(block), 0, st, e0, e1, 0, a, status, tally);
        else hipLaunchKernelGGL(state_rows_compact_kernel, dim3(grid), dim3(block), 0, st, a, status, tally);
        return;
    }
    if (!state_use_dma()) {
        launch_group<4>(st, a, status, tally);
        return;
    }
    // ZK_STATE_LDS_PAD=<bytes> (experiment): extra dynamic LDS per block — 17408 makes a block 81 KB, i.e. ONE block (one wavefront per
    // SIMD, 249 VGPRs) per CU, which leaves registers and LDS for another circuit's wavefronts beside it
    static const int pad = [] { const char* e = getenv("ZK_STATE_LDS_PAD"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 65536 ? 65536 : v); }();
    const int lds = (block / 64) * ST_DMA_WAVE_BYTES + pad;
    {  // > 64 KiB of dynamic LDS has to be asked for, once per device
        static std::mutex m;
        static bool asked[64] = {false};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(m);
        if (dev >= 0 && dev < 64 && !asked[dev]) {
            (void)hipFuncSetAttribute((const void*)state_rows_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            asked[dev] = true;
        }
    }
    const u64 rows_per_block = (u64)(block / 64) * ST_ROWS_PER_WAVE;  // 63 evaluated rows per wavefront
    const u32 grid = (u32)((a.eval_hi - a.eval_lo + rows_per_block - 1) / rows_per_block);
    if (e0 || e1) hipExtLaunchKernelGGL(state_rows_dma_kernel, dim3(grid), dim3(block), lds, st, e0, e1, 0, a, status, tally);
    else hipLaunchKernelGGL(state_rows_dma_kernel, dim3(grid), dim3(block), lds, st, a, status, tally);
}
