// Bytecode / Copy / Tx+Sig / Exp row kernels
#include <stdlib.h>
#include "kernels.hpp"

// ---------------------------------------------------------------------------------------
// Bytecode / Exp circuit kernels: one lane per row, column-major witness (coalesced), next row
// re-read through L1/L2 (wraps modulo n).
// ---------------------------------------------------------------------------------------
__global__ void fr_to_mont_kernel(Fr x, u64* out) {  // one cell to Montgomery form (per-session constants)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const Fr m = fr_to_mont(x);
        for (int j = 0; j < 4; j++) out[j] = (u64)m.v[2 * j] | ((u64)m.v[2 * j + 1] << 32);
    }
}
// Bytecode: a wavefront holds 64 consecutive rows, evaluates the first 63 and takes each row's successor from lane + 1; its last
// lane is the (read-only) successor of the 63rd.  The successor of the witness's last row is row 0.
#define BC_ROWS_PER_WAVE 63
__global__ __launch_bounds__(256) void bytecode_rows_kernel(BytecodeArgs a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 n = a.rows.n;
    u64 i = lo + wave * BC_ROWS_PER_WAVE + lane;
    const bool evaluate = lane != 63u && i < hi;
    if (i >= n) i = i == n ? 0 : n - 1;  // row n is the wrap-around successor; lanes further out only keep their loads in bounds
    BcRow C;
    bytecode_load_row(a.rows, i, C);
    u32 code = bytecode_check_loaded(a, C, C);
    if (!evaluate) code = 0;
    else if (status) status[i] = code;
    tally_commit(tally, i, code);
}
// Copy: a wavefront holds 64 consecutive rows, evaluates the first 62 and takes the cells of rows i + 1 / i + 2 from lanes + 1 /
// + 2; its last two lanes are (read-only) successors.  Rows past the end wrap to the start (copy_circuit.py:92-130).
#ifndef ZK_COPY_OCC
#define ZK_COPY_OCC 1  // blocks of 256 per CU the build is sized for (tuning builds: 2 = at most 256 registers, two wavefronts per SIMD)
#endif
__global__ __launch_bounds__(256, ZK_COPY_OCC) void copy_rows_kernel(CopyArgs a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 n = a.rows.n;
    u64 i = lo + wave * CP_ROWS_PER_WAVE + lane;
    const bool evaluate = lane < (u32)CP_ROWS_PER_WAVE && i < hi;
    if (i >= n) i %= n;  // the wrap-around successors (and lanes further out, which only have to stay in bounds)
    CpRow C;
    copy_load_row(a.rows, i, C);
    u32 code = copy_check_loaded(a, C, C, C);
    if (!evaluate) code = 0;
    else if (status) status[i] = code;
    tally_commit(tally, i, code);
}
// Tx / Sig circuits: one lane per tx slot / signature row (units are independent: no halo).
__global__ void sign_rpow_kernel(Fr r, u64* out) {  // row k = r^k canonical, k < 64, one lane each (sign_fill_rpow is the host form)
    const u32 k = threadIdx.x;
    if (blockIdx.x != 0 || k >= 64u) return;
    const Fr p = fr_mont(fr_pow_small_mont(fr_to_mont(r), k), fr_from_u64(1));
    for (int j = 0; j < 4; j++) out[4 * k + j] = (u64)p.v[2 * j] | ((u64)p.v[2 * j + 1] << 32);
}
__global__ __launch_bounds__(256) void sign_units_kernel(SignArgs a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 code = 0;
    if (i < hi) {
        code = sign_check_unit(a, i);
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}
__global__ __launch_bounds__(256) void exp_rows_kernel(ExpArgs a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 code = 0;
    if (i < hi) {
        code = exp_check_row(a, i);
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}

__global__ __launch_bounds__(256) void pi_rows_kernel(PiArgs a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 code = 0;
    if (i < hi) {
        code = pi_check_row(a, i);
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}
__global__ __launch_bounds__(256) void pi_copy_kernel(PiCopyArgs a, u32* status, ZkTally* tally) {
    tally_clear_twin(tally);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 code = 0;
    if (i < a.n) {
        code = pi_copy_check(a, i);
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}
static inline u32 grid256(u64 n) { return (u32)((n + 255) / 256); }
void zk_launch_pi_copy(hipStream_t st, const PiCopyArgs& a, u32* status, ZkTally* tally) {
    hipLaunchKernelGGL(pi_copy_kernel, dim3(grid256(a.n)), dim3(256), 0, st, a, status, tally);
}
void zk_launch_bytecode_rows(hipStream_t st, const BytecodeArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    const u64 rows_per_block = 4 * BC_ROWS_PER_WAVE;  // 63 evaluated rows per wavefront
    hipLaunchKernelGGL(bytecode_rows_kernel, dim3((u32)((hi - lo + rows_per_block - 1) / rows_per_block)), dim3(256), 0, st, a, lo, hi, status, tally);
}
void zk_launch_copy_rows(hipStream_t st, const CopyArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    // small tables: one wavefront per block (2^15 rows = 532 wavefronts reach every CU; four per block only a third of them)
    if (hi <= lo) return;  // nothing to evaluate (and the kernel's wrap-around `i %= n` must never see an empty table)
    static const u32 forced = [] { const char* e = getenv("ZK_COPY_BLOCK"); const int v = e ? atoi(e) : 0; return (u32)((v == 64 || v == 128 || v == 256) ? v : 0); }();
    const u64 waves = (hi - lo + CP_ROWS_PER_WAVE - 1) / CP_ROWS_PER_WAVE;
    const u32 block = forced ? forced : (waves <= 4096 ? 64u : 256u);
    hipLaunchKernelGGL(copy_rows_kernel, dim3((u32)((waves + block / 64 - 1) / (block / 64))), dim3(block), 0, st, a, lo, hi, status, tally);
}
void zk_launch_sign_units(hipStream_t st, const SignArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    hipLaunchKernelGGL(sign_units_kernel, dim3(grid256(hi - lo)), dim3(256), 0, st, a, lo, hi, status, tally);
}
void zk_launch_exp_rows(hipStream_t st, const ExpArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    hipLaunchKernelGGL(exp_rows_kernel, dim3(grid256(hi - lo)), dim3(256), 0, st, a, lo, hi, status, tally);
}
void zk_launch_pi_rows(hipStream_t st, const PiArgs& a, u64 lo, u64 hi, u32* status, ZkTally* tally) {
    hipLaunchKernelGGL(pi_rows_kernel, dim3(grid256(hi - lo)), dim3(256), 0, st, a, lo, hi, status, tally);
}
void zk_launch_fr_to_mont(hipStream_t st, const Fr& x, u64* out) { hipLaunchKernelGGL(fr_to_mont_kernel, dim3(1), dim3(64), 0, st, x, out); }
void zk_launch_sign_rpow(hipStream_t st, const Fr& r, u64* out) { hipLaunchKernelGGL(sign_rpow_kernel, dim3(1), dim3(64), 0, st, r, out); }
