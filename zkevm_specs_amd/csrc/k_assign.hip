// Witness assignment (State, Bytecode) and keccak-table generation kernels
#include <stdlib.h>
#include "kernels.hpp"


__global__ void slots_fill_kernel_asg(u32* slots, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = ZK_EMPTY_SLOT;
}
// ---------------------------------------------------------------------------------------
// State-circuit witness assignment (state_assign.hpp): one lane per op.
//   insert -> mark (first occurrences, per-block partials) -> scan (one block) -> rank + MPT rows -> rows
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(ASG_BLOCK) void assign_insert_kernel(AssignArgs a) {
    const u64 i = (u64)blockIdx.x * ASG_BLOCK + threadIdx.x;
    if (i < a.n && asg_has_key(asg_slot(a, ASG_TAG, i))) asg_insert(a, (u32)i);
}
__global__ __launch_bounds__(ASG_BLOCK) void assign_mark_kernel(AssignArgs a) {
    __shared__ u32 s_cnt[ASG_BLOCK / 64];
    __shared__ u32 s_min[ASG_BLOCK / 64];
    const u64 i = (u64)blockIdx.x * ASG_BLOCK + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool keyed = i < a.n && asg_has_key(asg_slot(a, ASG_TAG, i));
    u32 f = ASG_NONE;
    if (keyed) f = asg_find_first(a, (u32)i);
    if (i < a.n) a.first[i] = f;
    const unsigned long long bf = __ballot(keyed && f == (u32)i), bk = __ballot(keyed);
    if (lane == 0) {
        s_cnt[w] = (u32)__popcll(bf);
        s_min[w] = bk ? (u32)i + (u32)__ffsll((long long)bk) - 1u : ASG_NONE;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 c = 0, m = ASG_NONE;
        for (int k = 0; k < ASG_BLOCK / 64; k++) { c += s_cnt[k]; m = s_min[k] < m ? s_min[k] : m; }
        a.blk_cnt[blockIdx.x] = c;
        a.blk_next[blockIdx.x] = m;
    }
}
// blk_cnt -> exclusive prefix (total in [nb]); blk_next -> min over the blocks after b.  One block.
__global__ __launch_bounds__(1024) void assign_scan_kernel(AssignArgs a) {
    __shared__ u32 s[1024];
    const u32 t = threadIdx.x, nb = a.nb;
    const u32 per = (nb + 1023u) / 1024u;
    const u32 lo = t * per < nb ? t * per : nb, hi = lo + per < nb ? lo + per : nb;
    u32 sum = 0;
    for (u32 b = lo; b < hi; b++) sum += a.blk_cnt[b];
    s[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
        const u32 v = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    u32 run = s[t] - sum;
    const u32 total = s[1023];
    for (u32 b = lo; b < hi; b++) { const u32 c = a.blk_cnt[b]; a.blk_cnt[b] = run; run += c; }
    if (t == 0) a.blk_cnt[nb] = total;
    u32 m = ASG_NONE;
    for (u32 b = lo; b < hi; b++) m = a.blk_next[b] < m ? a.blk_next[b] : m;
    __syncthreads();
    s[t] = m;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
        const u32 v = t + d < 1024 ? s[t + d] : ASG_NONE;
        __syncthreads();
        s[t] = v < s[t] ? v : s[t];
        __syncthreads();
    }
    u32 after = t + 1 < 1024 ? s[t + 1] : ASG_NONE;
    for (u32 b = hi; b > lo; b--) { const u32 c = a.blk_next[b - 1]; a.blk_next[b - 1] = after; after = c < after ? c : after; }
    if (t == 0) a.blk_next[nb] = ASG_NONE;
}
__global__ __launch_bounds__(ASG_BLOCK) void assign_rank_kernel(AssignArgs a) {
    __shared__ u32 s_cnt[ASG_BLOCK / 64];
    const u64 i = (u64)blockIdx.x * ASG_BLOCK + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool is_first = i < a.n && a.first[i] == (u32)i;
    const unsigned long long bf = __ballot(is_first);
    if (lane == 0) s_cnt[w] = (u32)__popcll(bf);
    __syncthreads();
    if (is_first) {
        u32 r = a.blk_cnt[blockIdx.x] + (u32)__popcll(bf & ((1ull << lane) - 1ull));
        for (u32 k = 0; k < w; k++) r += s_cnt[k];
        a.rank[i] = r;
        asg_write_mpt(a, i, r);
    }
}
__global__ __launch_bounds__(ASG_BLOCK) void assign_rows_kernel(AssignArgs a, u32* status, ZkTally* tally) {
    __shared__ u32 s_min[ASG_BLOCK / 64];
    const u64 i = (u64)blockIdx.x * ASG_BLOCK + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool in = i < a.n;
    const u32 f = in ? a.first[i] : ASG_NONE;
    const unsigned long long bk = __ballot(f != ASG_NONE);
    const u32 wave_base = (u32)i - lane;
    if (lane == 0) s_min[w] = bk ? wave_base + (u32)__ffsll((long long)bk) - 1u : ASG_NONE;
    __syncthreads();
    // first MPT-keyed op strictly after i: this wave, the later waves of the block, the later blocks
    const unsigned long long above = lane == 63u ? 0ull : (bk >> (lane + 1u)) << (lane + 1u);
    u32 nxt = above ? wave_base + (u32)__ffsll((long long)above) - 1u : ASG_NONE;
    for (u32 k = w + 1; k < ASG_BLOCK / 64; k++)
        if (nxt == ASG_NONE) nxt = s_min[k];
    if (nxt == ASG_NONE) nxt = a.blk_next[blockIdx.x];
    u32 code = 0;
    if (in) {
        const u64 root = 3ull + 5ull * (nxt == ASG_NONE ? a.blk_cnt[a.nb] : a.rank[a.first[nxt]]);
        code = asg_write_row(a, i, root, f == (u32)i);
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}
// Bytecode-circuit witness assignment (bytecode_assign.hpp)
__global__ void bca_rpow_kernel(Fr r, u64* out) {  // entry m = Mont(r^m), one lane each (bca_fill_rpow is the host form)
    if (blockIdx.x == 0 && threadIdx.x < BCA_RPOW_ROWS) bca_store(out + 4 * threadIdx.x, fr_pow_small_mont(fr_to_mont(r), threadIdx.x));
}
__global__ void bca_chunk_kernel(BcaArgs a) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.n_chunks) bca_chunk(a, c);
}
__global__ void bca_prefix_kernel(BcaArgs a) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < a.n_codes) bca_prefix_code(a, j);
}
__global__ __launch_bounds__(BCA_CHUNK) void bca_rlc_kernel(BcaArgs a) {  // a block per chunk, a lane per row
    if (blockIdx.x < a.n_chunks) bca_rlc_row(a, blockIdx.x, threadIdx.x);
}
__global__ __launch_bounds__(256) void bca_rows_kernel(BcaArgs a, u32* status, ZkTally* tally) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_out) {
        bca_write_row(a, i);
        if (status) status[i] = 0;  // the assignment has no failure modes of its own
    }
    tally_commit(tally, i, 0);
}
// Keccak table generation: one lane per message (keccak_table.hpp)
__global__ void keccak_rpow_kernel(Fr r, u64* out) {  // rows 0..64: r^k canonical; row 65: Mont(r^64) (kt_fill_rpow is the host form)
    const u32 k = threadIdx.x;
    if (blockIdx.x != 0 || k >= KT_RPOW_ROWS) return;
    const Fr pM = fr_pow_small_mont(fr_to_mont(r), k < 65u ? k : 64u);
    kt_store(out + 4 * k, k < 65u ? fr_mont(pM, fr_from_u64(1)) : pM);
}
__global__ __launch_bounds__(256) void keccak_table_kernel(KeccakGenArgs g, u32* status, ZkTally* tally) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 code = 0;
    if (i < g.n) {
        // long messages of the KeccakCircuit.add form go to the lane-group kernel behind this one (their status is 0: only
        // KeccakTable.add has a failing case, inputs over 64 bytes, and that verdict needs no hashing)
        if (g.long_list && g.mode == KT_MODE_CIRCUIT && g.offsets[i + 1] - g.offsets[i] >= KT_GROUP_MIN_BYTES)
            g.long_list[atomicAdd(g.long_count, 1u)] = (u32)i;
        else
            code = keccak_table_row(g, i);
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}
// two messages per wavefront; the groups walk the list of long messages grid-stride (its length is only known on the device)
__global__ __launch_bounds__(256) void keccak_table_group_kernel(KeccakGenArgs g, u32 use_lds) {
    __shared__ u64 s_state[8][64];  // per lane group: the state words and the rotated words of a round (keccak_table_row_group)
    const u32 gl = threadIdx.x & 31u;
    const int base = (int)(threadIdx.x & 32u);
    const u32 n_long = *g.long_count;
    const u32 groups = gridDim.x * (blockDim.x >> 5);
    u64* lds = use_lds ? s_state[threadIdx.x >> 5] : nullptr;
    for (u32 k = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5); k < n_long; k += groups) keccak_table_row_group(g, g.long_list[k], gl, base, lds);
}
void zk_launch_state_assign(hipStream_t st, const AssignArgs& a, u32* status, ZkTally* tally) {
    const u32 cap = a.mask + 1u;
    hipLaunchKernelGGL(slots_fill_kernel_asg, dim3((cap + 255) / 256), dim3(256), 0, st, a.slots, cap);
    hipLaunchKernelGGL(assign_insert_kernel, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
    hipLaunchKernelGGL(assign_mark_kernel, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
    hipLaunchKernelGGL(assign_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(assign_rank_kernel, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
    hipLaunchKernelGGL(assign_rows_kernel, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a, status, tally);
}
void zk_launch_bca_rpow(hipStream_t st, const Fr& r, u64* out) { hipLaunchKernelGGL(bca_rpow_kernel, dim3(1), dim3(128), 0, st, r, out); }
void zk_launch_bytecode_assign(hipStream_t st, const BcaArgs& a, u32* status, ZkTally* tally) {
    const u32 gc = (u32)((a.n_codes + 63) / 64), gk = (u32)((a.n_chunks + 63) / 64);
    if (a.n_codes) {
        hipLaunchKernelGGL(bca_chunk_kernel, dim3(gk ? gk : 1), dim3(64), 0, st, a);
        hipLaunchKernelGGL(bca_prefix_kernel, dim3(gc), dim3(64), 0, st, a);
        hipLaunchKernelGGL(bca_rlc_kernel, dim3((u32)(a.n_chunks ? a.n_chunks : 1)), dim3(BCA_CHUNK), 0, st, a);
    }
    hipLaunchKernelGGL(bca_rows_kernel, dim3((u32)((a.n_out + 255) / 256)), dim3(256), 0, st, a, status, tally);
}
// Copy-circuit witness assignment (copy_assign.hpp)
__global__ void cpa_rpow_kernel(Fr r, u64* out) {  // entry m = Mont(r^m), one lane each (cpa_fill_rpow is the host form)
    if (blockIdx.x == 0 && threadIdx.x < CPA_RPOW_ROWS) cpa_store(out + 4 * threadIdx.x, fr_pow_small_mont(fr_to_mont(r), threadIdx.x));
}
__global__ void cpa_chunk_kernel(CpaArgs a) {
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.n_chunks) cpa_chunk(a, c);
}
__global__ void cpa_prefix_kernel(CpaArgs a) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < a.n_events) cpa_prefix_event(a, j);
}
__global__ __launch_bounds__(256) void cpa_rows_kernel(CpaArgs a, u32* status, ZkTally* tally) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < a.n_rows) {
        cpa_write_row(a, j);
        if (status) status[j] = 0;  // the assignment has no failure modes of its own (domain checks happen at open)
    }
    tally_commit(tally, j, 0);
}
void zk_launch_cpa_rpow(hipStream_t st, const Fr& r, u64* out) { hipLaunchKernelGGL(cpa_rpow_kernel, dim3(1), dim3(128), 0, st, r, out); }
void zk_launch_copy_assign(hipStream_t st, const CpaArgs& a, u32* status, ZkTally* tally) {
    if (a.n_chunks) hipLaunchKernelGGL(cpa_chunk_kernel, dim3((u32)((a.n_chunks + 63) / 64)), dim3(64), 0, st, a);
    if (a.n_events) hipLaunchKernelGGL(cpa_prefix_kernel, dim3((u32)((a.n_events + 63) / 64)), dim3(64), 0, st, a);
    if (a.n_rows) hipLaunchKernelGGL(cpa_rows_kernel, dim3((u32)((a.n_rows + 255) / 256)), dim3(256), 0, st, a, status, tally);
}
void zk_launch_keccak_rpow(hipStream_t st, const Fr& r, u64* out) { hipLaunchKernelGGL(keccak_rpow_kernel, dim3(1), dim3(128), 0, st, r, out); }
void zk_launch_keccak_table(hipStream_t st, const KeccakGenArgs& g, u32* status, ZkTally* tally) {
    if (g.long_list) (void)hipMemsetAsync(g.long_count, 0, sizeof(u32), st);
    hipLaunchKernelGGL(keccak_table_kernel, dim3((u32)((g.n + 255) / 256)), dim3(256), 0, st, g, status, tally);
    if (g.long_list) {
        const u64 blocks = (g.n + 7) / 8;  // eight groups per block; at most one group per message
        static const u32 use_lds = [] { const char* e = getenv("ZK_KECCAK_LDS"); return (u32)!(e && e[0] == '0'); }();  // 0: the shuffle rounds of round 4
        hipLaunchKernelGGL(keccak_table_group_kernel, dim3((u32)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, g, use_lds);
    }
}
