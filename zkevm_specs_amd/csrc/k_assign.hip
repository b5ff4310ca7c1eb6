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
template <bool RW>
__global__ __launch_bounds__(ASG_BLOCK) void assign_insert_kernel(AssignArgs a) {
    const u64 i = (u64)blockIdx.x * ASG_BLOCK + threadIdx.x;
    if (i < a.n && asg_has_key(asg_slot<RW>(a, ASG_TAG, i))) asg_insert<RW>(a, (u32)i);
}
template <bool RW>
__global__ __launch_bounds__(ASG_BLOCK) void assign_mark_kernel(AssignArgs a) {
    __shared__ u32 s_cnt[ASG_BLOCK / 64];
    __shared__ u32 s_min[ASG_BLOCK / 64];
    const u64 i = (u64)blockIdx.x * ASG_BLOCK + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool keyed = i < a.n && asg_has_key(asg_slot<RW>(a, ASG_TAG, i));
    u32 f = ASG_NONE;
    if (keyed) f = asg_find_first<RW>(a, (u32)i);
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
template <bool RW>
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
        Fr q[ASG_MPT_NCELLS];
        asg_mpt_cells<RW>(a, i, r, q);
        u64* out = a.mpt + (u64)r * (ASG_MPT_NCELLS * 4);
#pragma unroll
        for (int c = 0; c < ASG_MPT_NCELLS; c++) asg_store(out + 4 * c, q[c]);
        if (a.mpt_slots) {  // the State circuit's MPT index, entered as the row is written (build_index + mpt_index_build_kernel otherwise)
            ZkTable t;
            t.n = (u32)a.n;
            const u64 h = state_mpt_hash_cells(q);
            const u32 v = state_mpt_slot_value(t, r, h);
            u32 s = (u32)h & a.mpt_mask;
            while (atomicCAS(&a.mpt_slots[s], ZK_EMPTY_SLOT, v) != ZK_EMPTY_SLOT) s = (s + 1) & a.mpt_mask;
        }
    }
}
// ROWS = false: only every op's root (as the rank it is made of) comes out — the rows are computed again where they are evaluated
// (state_rows_fused_kernel)
template <bool RW, bool ROWS = true>
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
        const u32 rk = nxt == ASG_NONE ? a.blk_cnt[a.nb] : a.rank[a.first[nxt]];
        if (!ROWS) {
            a.root_rank[i] = rk;
            return;
        }
        code = asg_write_row<RW>(a, i, 3ull + 5ull * rk, f == (u32)i);
        if (status) status[i] = code;
    }
    if (ROWS) tally_commit(tally, i, code);
}
// Bytecode-circuit witness assignment (bytecode_assign.hpp)
__global__ void bca_rpow_kernel(Fr r, u64* out) {  // entry m = Mont(r^m), one lane each (bca_fill_rpow is the host form)
    if (blockIdx.x == 0 && threadIdx.x < BCA_RPOW_ROWS) bca_store(out + 4 * threadIdx.x, fr_pow_small_mont(fr_to_mont(r), threadIdx.x));
}
// ---- device form (round 6): wavefront-parallel scans instead of one lane per chunk / per bytecode ------------------------------
// (the per-element functions of bytecode_assign.hpp stay the host's — CPU backend, tests/hostsim — and define the results)
__device__ __forceinline__ Fr bca_shfl_up(const Fr& x, int d) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = (u32)__shfl_up((int)x.v[k], d);
    return r;
}
__device__ __forceinline__ Fr bca_shfl(const Fr& x, int lane) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = (u32)__shfl((int)x.v[k], lane);
    return r;
}
// A wavefront per chunk, a lane per row.  (1) Horner prefix of the chunk's byte rows as an inclusive scan of the affine maps
// x -> x r + value: at distance d the left neighbour's partial is multiplied by r^d (the table's Montgomery powers) and added —
// six Montgomery products per lane; lane j ends with sum_{i <= j} value_i r^(j - i), the last byte row with the chunk's Horner
// value.  Exact for every canonical value (no wide-value fallback needed).  (2) push sizes; the counter-after-the-chunk map by
// pointer jumping over "the next position entered with 0" (six rounds of lane reads).
__global__ __launch_bounds__(64) void bca_chunk_wave_kernel(BcaArgs a) {
    const u64 c = blockIdx.x;
    if (c >= a.n_chunks) return;
    const BcaChunk ch = a.chunks[c];
    const u32 t = threadIdx.x, skip = ch.first ? 1u : 0u;
    const bool valid = t < ch.count, byte_row = valid && t >= skip;
    const u32 j = t - skip;  // index among the chunk's byte rows (meaningful when byte_row)
    const u64 g = (u64)ch.start + t;
    Fr v = fr_zero();
    if (byte_row) v = bca_in_cell(a, g, 5);
    Fr B = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Fr Bp = bca_shfl_up(B, d);
        const Fr rd = fr_load(a.rpow + 4 * (u64)d);
        if (byte_row && j >= (u32)d) B = fr_add(fr_mont(Bp, rd), B);
    }
    if (valid) {
        bca_store(a.rlc + 4 * g, B);  // the row's share of value_rlc: what the chunk's own rows up to it contribute (bca_rows_wave_kernel adds the rest)
        a.row_code[g] = ch.code;
        a.row_chunk[g] = (u32)c;
    }
    const u32 m = ch.count - skip;
    if (t == ch.count - 1u) {
        bca_store(a.chunk_acc + 4 * c, m ? B : fr_zero());
        a.chunk_m[c] = m;
    }
    const u32 sz = byte_row ? bca_push_size(v) : 0u;
    if (valid) a.track[2 * g + 1] = (uint8_t)sz;
    // zero_out[p] = counter after the chunk when position p is entered with 0: follow nxt until a terminal value
    //   nxt = p + 1 (no push) or p + sz + 1 (push data inside the chunk); terminal: the data runs past the chunk (value = what is left), or p == count (0)
    u32 nxt = t + (sz ? sz + 1u : 1u), val = 0;
    bool term = !valid;  // lanes at and past `count`: terminal 0
    if (valid && sz && t + sz + 1u > ch.count) { term = true; val = sz - (ch.count - 1u - t); }
#pragma unroll
    for (int round = 0; round < 7; round++) {  // (six doublings cover a chain of 64 links; one more for the final adoption)
        const u32 src = nxt < 64u ? nxt : 63u;
        const u32 n_term = (u32)__shfl((int)(term ? 1u : 0u), (int)src), n_val = (u32)__shfl((int)val, (int)src), n_nxt = (u32)__shfl((int)nxt, (int)src);
        if (!term) {
            if (nxt >= ch.count) { term = true; val = 0; }  // position `count`: the chunk ends with the counter at 0
            else if (n_term) { term = true; val = n_val; }
            else nxt = n_nxt;
        }
    }
    // map[left] for left = 0..32: entered with `left`, the first `left` rows are push data, then position `left` is entered with 0:
    // lane t holds zero_out[t]
    uint8_t* map = a.chunk_map + c * BCA_MAP_STRIDE;
    if (t <= 32u) map[t] = (uint8_t)(t < ch.count ? val : (t == ch.count ? 0u : t - ch.count));
}
// A block of 16 wavefronts per bytecode, a wavefront per batch of 64 chunks, 1,024 chunks per round.
//   value_rlc:     the chunks' transitions x -> x r^m + acc composed by an inclusive scan inside the batch — (A2 A1, A2 B1 + B2), A in
//                  Montgomery form, B canonical —, the batches' totals chained afterwards (a Montgomery product per earlier batch);
//   push counter:  the round's 33-entry maps staged in LDS (coalesced); lane s <= 32 of every wavefront walks ITS batch as if it were
//                  entered with counter s and records the counter entering every chunk; once the batches' results are chained (one
//                  lookup per earlier batch) every chunk picks the record of the entry state that really occurred.
// (One lane walking all chunks was 70 us for 2^17 rows — ~190 ns per dependent LDS lookup —, a scan over composed maps 192 us.)
#define BCA_ROUND_WAVES 16
#define BCA_ROUND_CHUNKS (64 * BCA_ROUND_WAVES)
__global__ __launch_bounds__(64 * BCA_ROUND_WAVES) void bca_prefix_wave_kernel(BcaArgs a) {
    __shared__ u32 s_map[BCA_ROUND_CHUNKS * (BCA_MAP_STRIDE / 4)];
    __shared__ uint8_t s_in[BCA_ROUND_CHUNKS][BCA_MAP_STRIDE];  // [chunk][s]: counter entering the chunk when its batch is entered with s
    __shared__ uint8_t s_res[BCA_ROUND_WAVES][BCA_MAP_STRIDE];  // [batch][s]: counter after the batch
    __shared__ u32 s_tot[BCA_ROUND_WAVES][16];                  // [batch]: A (8 words), B (8 words) of the whole batch
    __shared__ u32 s_carry[9];                                  // value (8 words) and counter entering the round
    const u64 code = blockIdx.x;
    if (code >= a.n_codes) return;
    const u32 c0 = a.code_chunk0[code], c1 = a.code_chunk0[code + 1];
    const u32 l = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (threadIdx.x < 9) s_carry[threadIdx.x] = 0;
    for (u32 round0 = c0; round0 < c1; round0 += BCA_ROUND_CHUNKS) {  // (uniform trip count for the whole block)
        const u32 nc = c1 - round0 < (u32)BCA_ROUND_CHUNKS ? c1 - round0 : (u32)BCA_ROUND_CHUNKS;
        {
            const u32* src = (const u32*)(a.chunk_map + (u64)round0 * BCA_MAP_STRIDE);
            for (u32 k = threadIdx.x; k < nc * (BCA_MAP_STRIDE / 4); k += 64u * BCA_ROUND_WAVES) s_map[k] = src[k];
        }
        __syncthreads();
        const u32 c = round0 + w * 64u + l;
        const bool on = c < c1;
        Fr A = frm_one(), B = fr_zero();
        if (on) {
            A = fr_load(a.rpow + 4 * (u64)a.chunk_m[c]);
            B = fr_load(a.chunk_acc + 4 * (u64)c);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const Fr Ap = bca_shfl_up(A, d), Bp = bca_shfl_up(B, d);
            if (l >= (u32)d) {
                B = fr_add(fr_mont(Bp, A), B);  // (prefix first, then this segment)
                A = fr_mont(Ap, A);
            }
        }
        if (l == 63u) {  // (lanes past the end hold identities: lane 63 has the batch's whole composition)
#pragma unroll
            for (int k = 0; k < 8; k++) { s_tot[w][k] = A.v[k]; s_tot[w][8 + k] = B.v[k]; }
        }
        if (l <= 32u) {  // the batch walked from entry state l
            const uint8_t* m = (const uint8_t*)s_map;
            u32 cur = l;
            const u32 q0 = w * 64u, q1 = q0 + 64u < nc ? q0 + 64u : nc;
            for (u32 q = q0; q < q1; q++) {
                s_in[q][l] = (uint8_t)cur;
                cur = m[q * BCA_MAP_STRIDE + cur];
            }
            s_res[w][l] = (uint8_t)cur;  // (an empty batch is the identity)
        }
        __syncthreads();
        Fr carry;
#pragma unroll
        for (int k = 0; k < 8; k++) carry.v[k] = s_carry[k];
        u32 entry = s_carry[8];
        for (u32 b = 0; b < w; b++) {  // through the batches before this one
            Fr Ab, Bb;
#pragma unroll
            for (int k = 0; k < 8; k++) { Ab.v[k] = s_tot[b][k]; Bb.v[k] = s_tot[b][8 + k]; }
            carry = fr_add(fr_mont(carry, Ab), Bb);
            entry = s_res[b][entry];
        }
        const Fr Ae = bca_shfl_up(A, 1), Be = bca_shfl_up(B, 1);
        if (on) {
            bca_store(a.chunk_in + 4 * (u64)c, l == 0 ? carry : fr_add(fr_mont(carry, Ae), Be));
            a.chunk_state[c] = s_in[w * 64u + l][entry];
        }
        __syncthreads();
        if (w == BCA_ROUND_WAVES - 1 && l == 63u) {  // the last batch's lane 63: what enters the next round
            const Fr nxt = fr_add(fr_mont(carry, A), B);
#pragma unroll
            for (int k = 0; k < 8; k++) s_carry[k] = nxt.v[k];
            s_carry[8] = s_res[w][entry];
        }
        __syncthreads();
    }
}
// A lane per OUTPUT row: value_rlc = (value entering the chunk) r^(byte rows up to this one) + the row's share from the chunk scan;
// push_data_left by walking the chunk's push sizes (one byte each) from the chunk's entry state.
__global__ __launch_bounds__(256) void bca_rows_wave_kernel(BcaArgs a, u32* status, ZkTally* tally) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_out) {
        if (i < a.n_in) {
            const u32 c = a.row_chunk[i];
            const BcaChunk ch = a.chunks[c];
            const u32 t = (u32)(i - ch.start), skip = ch.first ? 1u : 0u;
            u32 left = a.chunk_state[c];
            for (u32 q = skip; q < t; q++) {
                const u32 sz = a.track[2 * ((u64)ch.start + q) + 1];
                left = left == 0u ? sz : left - 1u;
            }
            a.track[2 * i] = (uint8_t)left;
            Fr rlc = fr_load(a.chunk_in + 4 * (u64)c);
            if (t >= skip) rlc = fr_add(fr_mont(rlc, fr_load(a.rpow + 4 * (u64)(t + 1u - skip))), fr_load(a.rlc + 4 * i));
            bca_store(a.rlc + 4 * i, rlc);
        }
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
    // (the fused form's MPT index sits behind the key slots in one buffer: one fill)
    const u32 cap = a.mask + 1u + (a.mpt_slots ? a.mpt_mask + 1u : 0u);
    hipLaunchKernelGGL(slots_fill_kernel_asg, dim3((cap + 255) / 256), dim3(256), 0, st, a.slots, cap);
    if (a.root_rank) {  // rows evaluated where they are computed: no row pass, the roots alone
        if (a.rw) {
            hipLaunchKernelGGL(assign_insert_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
            hipLaunchKernelGGL(assign_mark_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
            hipLaunchKernelGGL(assign_scan_kernel, dim3(1), dim3(1024), 0, st, a);
            hipLaunchKernelGGL(assign_rank_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(assign_rows_kernel<true, false>), dim3(a.nb), dim3(ASG_BLOCK), 0, st, a, status, tally);
        } else {
            hipLaunchKernelGGL(assign_insert_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
            hipLaunchKernelGGL(assign_mark_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
            hipLaunchKernelGGL(assign_scan_kernel, dim3(1), dim3(1024), 0, st, a);
            hipLaunchKernelGGL(assign_rank_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(assign_rows_kernel<false, false>), dim3(a.nb), dim3(ASG_BLOCK), 0, st, a, status, tally);
        }
        return;
    }
    if (a.rw) {  // ops read straight from the RW rows through the sorted order (zk_state_assign_from_rw)
        hipLaunchKernelGGL(assign_insert_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
        hipLaunchKernelGGL(assign_mark_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
        hipLaunchKernelGGL(assign_scan_kernel, dim3(1), dim3(1024), 0, st, a);
        hipLaunchKernelGGL(assign_rank_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
        hipLaunchKernelGGL(assign_rows_kernel<true>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a, status, tally);
        return;
    }
    hipLaunchKernelGGL(assign_insert_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
    hipLaunchKernelGGL(assign_mark_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
    hipLaunchKernelGGL(assign_scan_kernel, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(assign_rank_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a);
    hipLaunchKernelGGL(assign_rows_kernel<false>, dim3(a.nb), dim3(ASG_BLOCK), 0, st, a, status, tally);
}
void zk_launch_bca_rpow(hipStream_t st, const Fr& r, u64* out) { hipLaunchKernelGGL(bca_rpow_kernel, dim3(1), dim3(128), 0, st, r, out); }
void zk_launch_bytecode_assign(hipStream_t st, const BcaArgs& a, u32* status, ZkTally* tally) {
    const u32 gc = (u32)((a.n_codes + 63) / 64), gk = (u32)((a.n_chunks + 63) / 64);
    (void)gc; (void)gk;
    if (a.n_codes) {
        if (a.n_chunks) hipLaunchKernelGGL(bca_chunk_wave_kernel, dim3((u32)a.n_chunks), dim3(64), 0, st, a);
        hipLaunchKernelGGL(bca_prefix_wave_kernel, dim3((u32)a.n_codes), dim3(64 * BCA_ROUND_WAVES), 0, st, a);
    }
    hipLaunchKernelGGL(bca_rows_wave_kernel, dim3((u32)((a.n_out + 255) / 256)), dim3(256), 0, st, a, status, tally);
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
