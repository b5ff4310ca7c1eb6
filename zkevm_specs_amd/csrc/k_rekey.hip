// RW table -> State-circuit operations (state_rekey.hpp): re-keying, the lexicographic sort, the op list.
//   open:    rwk_scan_kernel      per class (state tag): OR / NOT-AND masks of the five key fields + row counts  -> host plan
//   launch:  rwk_collect / rwk_rank   (only for classes whose wide fields the plan ranks: small classes with 160 / 256-bit fields)
//            rwk_pack_kernel      compact order-preserving keys (SoA words) + per-row status + tally
//            rwk_hist / rwk_scatter   LSD radix sort of the row indices, 8 bits per pass, stable (wave match-any ranking)
//            rwk_emit_kernel      ops[12][n_ops] + flags in sorted order, StartOp in front
#include "kernels.hpp"
#include <mutex>

#define RWK_BLOCK 256
#define RWK_TILE_ITEMS 16
#define RWK_TILE (RWK_BLOCK * RWK_TILE_ITEMS)

// OR over the 64 lanes of a wave, result in every lane's copy of the return value's first lane (wave-uniform): DPP inside the rows
// of 16 (quad_perm, row_half_mirror, row_mirror), then the four row results by v_readlane.
__device__ __forceinline__ u32 rwk_wave_or(u32 v) {
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);   // quad_perm [1, 0, 3, 2]
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);   // quad_perm [2, 3, 0, 1]
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v |= (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);  // row_mirror
    return (u32)__builtin_amdgcn_readlane((int)v, 0) | (u32)__builtin_amdgcn_readlane((int)v, 16) |
           (u32)__builtin_amdgcn_readlane((int)v, 32) | (u32)__builtin_amdgcn_readlane((int)v, 48);
}

// ---- open: class masks --------------------------------------------------------------------------------------------------
// masks: u32[2][16][5][8] (OR of the field words, OR of their complements) then u32[16] row counts.
// Per wave and class present in it (one or two, rows of a class come in runs): a word that is zero in every row of the class only
// sets a bit in the class's "seen zero" bitmap (most of the 40 words: the upper words of ids, tags, counters, the storage key of
// stack rows); a word with one value in all rows costs two LDS atomics without return; only words that differ inside the wave
// are reduced across the lanes.
#define RWK_MASK_WORDS (RWK_NCLASSES * RWK_NFIELDS * 8)
#define RWK_SCAN_BLOCK 1024
__global__ __launch_bounds__(RWK_SCAN_BLOCK) void rwk_scan_kernel(RekeyArgs a) {
    __shared__ u32 s_or[RWK_MASK_WORDS], s_nand[RWK_MASK_WORDS], s_cnt[RWK_NCLASSES], s_zero[RWK_NCLASSES][2];
    for (u32 t = threadIdx.x; t < RWK_MASK_WORDS; t += RWK_SCAN_BLOCK) { s_or[t] = 0; s_nand[t] = 0; }
    if (threadIdx.x < RWK_NCLASSES) { s_cnt[threadIdx.x] = 0; s_zero[threadIdx.x][0] = 0; s_zero[threadIdx.x][1] = 0; }
    __syncthreads();
    const u32 lane = threadIdx.x & 63u;
    for (u64 base = (u64)blockIdx.x * RWK_SCAN_BLOCK; base < a.n; base += (u64)gridDim.x * RWK_SCAN_BLOCK) {
        const u64 i = base + threadIdx.x;
        const bool valid = i < a.n;
        RwkKey k;
        if (valid) k = rwk_key(a.rw + i * (RWK_RW_NCELLS * 4));
        else { k.cls = RWK_CLASS_DROPPED; k.status = 0; for (int f = 0; f < RWK_NFIELDS; f++) k.f[f] = fr_zero(); }
        unsigned long long rem = __ballot(valid);
        while (rem) {  // one round per class present in the wave
            const int first = __ffsll((long long)rem) - 1;
            const u32 c = (u32)__builtin_amdgcn_readlane((int)k.cls, first);
            const unsigned long long inc = __ballot(valid && k.cls == c);
            rem &= ~inc;
            const bool in = valid && k.cls == c;
            if (lane == 0) atomicAdd(&s_cnt[c], (u32)__popcll(inc));
            if (c == RWK_CLASS_DROPPED) continue;
            u32 zero_lo = 0, zero_hi = 0;  // (f * 8 + w): words 0..31 / 32..39
#pragma unroll
            for (int f = 0; f < RWK_NFIELDS; f++) {
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const u32 v = k.f[f].v[w];
                    if (__ballot(in && v != 0u) == 0ull) {
                        if (f * 8 + w < 32) zero_lo |= 1u << ((f * 8 + w) & 31);
                        else zero_hi |= 1u << ((f * 8 + w) & 31);
                        continue;
                    }
                    const u32 v0 = (u32)__builtin_amdgcn_readlane((int)v, first);
                    u32 orv = v0, nandv = ~v0;
                    if (__ballot(in && v != v0)) {
                        orv = rwk_wave_or(in ? v : 0u);
                        nandv = rwk_wave_or(in ? ~v : 0u);
                    }
                    if (lane == 0) {
                        const u32 slot = (c * RWK_NFIELDS + f) * 8 + w;
                        atomicOr(&s_or[slot], orv);
                        atomicOr(&s_nand[slot], nandv);
                    }
                }
            }
            if (lane == 0) {
                if (zero_lo) atomicOr(&s_zero[c][0], zero_lo);
                if (zero_hi) atomicOr(&s_zero[c][1], zero_hi);
            }
        }
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < RWK_MASK_WORDS; t += RWK_SCAN_BLOCK) {
        const u32 c = t / (RWK_NFIELDS * 8), fw = t % (RWK_NFIELDS * 8);
        const u32 o = s_or[t], nn = s_nand[t] | (((s_zero[c][fw >> 5] >> (fw & 31)) & 1u) ? 0xffffffffu : 0u);
        if (o && (__atomic_load_n(&a.masks[t], __ATOMIC_RELAXED) & o) != o) atomicOr(&a.masks[t], o);
        if (nn && (__atomic_load_n(&a.masks[RWK_MASK_WORDS + t], __ATOMIC_RELAXED) & nn) != nn) atomicOr(&a.masks[RWK_MASK_WORDS + t], nn);
    }
    if (threadIdx.x < RWK_NCLASSES && s_cnt[threadIdx.x]) atomicAdd(&a.masks[2 * RWK_MASK_WORDS + threadIdx.x], s_cnt[threadIdx.x]);
}

// ---- ranks of wide fields in small classes ------------------------------------------------------------------------------
// collect: the members of every job's class with their field value; rank: rank = number of members with a smaller value.
__global__ __launch_bounds__(RWK_BLOCK) void rwk_collect_kernel(RekeyArgs a) {
    const u64 i = (u64)blockIdx.x * RWK_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const RwkKey k = rwk_key(a.rw + i * (RWK_RW_NCELLS * 4));
    for (u32 j = 0; j < a.n_jobs; j++) {
        const RwkRankJob job = a.jobs[j];
        if (job.cls != k.cls) continue;
        const u32 pos = atomicAdd(&a.job_cursor[j], 1u);
        a.job_rows[job.base + pos] = (u32)i;
        Fr v = fr_zero();
#pragma unroll
        for (int f = 0; f < RWK_NFIELDS; f++)
            if (job.field == (u32)f) v = k.f[f];
        rwk_store(a.job_vals + (u64)(job.base + pos) * 4, v);
    }
}
__global__ __launch_bounds__(RWK_BLOCK) void rwk_rank_kernel(RekeyArgs a, u32 j) {
    __shared__ u32 s_v[RWK_BLOCK][9];  // (padded: lanes read the same entry, broadcast)
    const RwkRankJob job = a.jobs[j];
    const u32 t = blockIdx.x * RWK_BLOCK + threadIdx.x;
    const bool mine = t < job.count;
    Fr v = fr_zero();
    if (mine) v = fr_load(a.job_vals + (u64)(job.base + t) * 4);
    u32 rank = 0;
    for (u32 base = 0; base < job.count; base += RWK_BLOCK) {
        const u32 m = base + threadIdx.x;
        __syncthreads();
        if (m < job.count) {
            const Fr x = fr_load(a.job_vals + (u64)(job.base + m) * 4);
#pragma unroll
            for (int w = 0; w < 8; w++) s_v[threadIdx.x][w] = x.v[w];
        }
        __syncthreads();
        const u32 lim = job.count - base < RWK_BLOCK ? job.count - base : RWK_BLOCK;
        for (u32 q = 0; q < lim; q++) {
            bool lt = false;  // s_v[q] < v
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const u32 x = s_v[q][w];
                lt = (x < v.v[w]) || (x == v.v[w] && lt);
            }
            rank += lt ? 1u : 0u;
        }
    }
    if (mine) a.ranks[job.field][a.job_rows[job.base + t]] = rank;
}

// ---- compact keys ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RWK_BLOCK) void rwk_pack_kernel(RekeyArgs a, u32* status, ZkTally* tally) {
    __shared__ RwkPlan s_plan;
    {
        const u32* src = (const u32*)a.plan;
        u32* dst = (u32*)&s_plan;
        for (u32 t = threadIdx.x; t < sizeof(RwkPlan) / 4; t += RWK_BLOCK) dst[t] = src[t];
    }
    __syncthreads();
    const u64 i = (u64)blockIdx.x * RWK_BLOCK + threadIdx.x;
    u32 code = 0;
    if (i < a.n) {
        const RwkKey k = rwk_key(a.rw + i * (RWK_RW_NCELLS * 4));
        u32 ranks[RWK_NFIELDS];
#pragma unroll
        for (int f = 0; f < RWK_NFIELDS; f++) ranks[f] = a.ranks[f] ? a.ranks[f][i] : 0u;
        rwk_pack(s_plan, s_plan.cls[k.cls], k, ranks, a.keys + i, a.n);
        code = k.status;
        if (status) status[i] = code;
    }
    tally_commit(tally, i, code);
}

// Fast path: keys of at most 64 bits are packed into one u64 per row, and the digit histograms of all (at most 8) passes are
// taken on the way (LDS, then one global atomic per non-empty bin and block): grid-stride, 256 blocks of 1024.
__global__ __launch_bounds__(1024) void rwk_pack64_kernel(RekeyArgs a, u32* status, ZkTally* tally) {
    __shared__ RwkPlan s_plan;
    __shared__ u32 s_gh[8 * 256];
    {
        const u32* src = (const u32*)a.plan;
        u32* dst = (u32*)&s_plan;
        for (u32 t = threadIdx.x; t < sizeof(RwkPlan) / 4; t += 1024) dst[t] = src[t];
        for (u32 t = threadIdx.x; t < 8 * 256; t += 1024) s_gh[t] = 0;
    }
    __syncthreads();
    for (u64 base = (u64)blockIdx.x * 1024; base < a.n; base += (u64)gridDim.x * 1024) {
        const u64 i = base + threadIdx.x;
        u32 code = 0;
        if (i < a.n) {
            const RwkKey k = rwk_key(a.rw + i * (RWK_RW_NCELLS * 4));
            u32 ranks[RWK_NFIELDS];
#pragma unroll
            for (int f = 0; f < RWK_NFIELDS; f++) ranks[f] = a.ranks[f] ? a.ranks[f][i] : 0u;
            u32 w[2] = {0u, 0u};
            rwk_pack(s_plan, s_plan.cls[k.cls], k, ranks, w, 1);
            const u64 key = a.key_words == 2u ? (((u64)w[0] << 32) | (u64)w[1]) : (u64)w[0];
            a.key64_a[i] = key;
            for (u32 p = 0; p < a.n_passes; p++) atomicAdd(&s_gh[p * 256 + (u32)((key >> (8u * p)) & 0xffull)], 1u);
            code = k.status;
            if (status) status[i] = code;
        }
        tally_commit(tally, i, code);
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < a.n_passes * 256; t += 1024)
        if (s_gh[t]) atomicAdd(&a.sweep[t], s_gh[t]);
}
// One radix pass in one kernel: tiles take tickets in order, rank their keys (stable, as below), publish their digit counts and
// look back over the earlier tiles' descriptors for their prefix (descriptor word: count | 1 << 30 = this tile only, | 2 << 30 =
// everything up to and including this tile).  A tile only ever waits for tiles with smaller tickets, which are running already.
#define RWK_DESC_LOCAL (1u << 30)
#define RWK_DESC_INCL (2u << 30)
#define RWK_DESC_VAL 0x3fffffffu
__global__ __launch_bounds__(RWK_SW_BLOCK) void rwk_sweep_kernel(RekeyArgs a, const u64* in_key, const u32* in_idx, u64* out_key, u32* out_idx, u32 pass) {
    __shared__ u32 s_cnt[RWK_SW_BLOCK / 64][256];
    __shared__ u32 s_scan[256], s_texcl[256], s_gbase[256];
    __shared__ u32 s_tile;
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    u32* ticket = a.sweep + 8 * 256;
    u32* err = a.sweep + 8 * 256 + 8;
    u32* desc = a.sweep + RWK_SWEEP_HEAD + (u64)pass * a.ntiles_fast * 256;
    if (threadIdx.x == 0) s_tile = atomicAdd(&ticket[pass], 1u);
#pragma unroll
    for (u32 t = threadIdx.x; t < (RWK_SW_BLOCK / 64) * 256; t += RWK_SW_BLOCK) (&s_cnt[0][0])[t] = 0;
    __syncthreads();
    {   // digit bases: exclusive scan of the pass's global histogram (every thread takes part in the barriers)
        const u32 d = threadIdx.x & 255u;
        const u32 g = a.sweep[pass * 256 + d];
        if (threadIdx.x < 256) s_scan[d] = g;
        __syncthreads();
        for (u32 s = 1; s < 256; s <<= 1) {
            const u32 x = (threadIdx.x < 256 && d >= s) ? s_scan[d - s] : 0u;
            __syncthreads();
            if (threadIdx.x < 256) s_scan[d] += x;
            __syncthreads();
        }
        if (threadIdx.x < 256) s_scan[d] -= g;
        __syncthreads();
    }
    const u32 tile = s_tile;
    const u64 tile0 = (u64)tile * RWK_SW_TILE;
    const u32 shift = 8u * pass;
    u64 keys[RWK_SW_ITEMS];
    u32 rows[RWK_SW_ITEMS], meta[RWK_SW_ITEMS];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RWK_SW_ITEMS; j++) {
        const u64 p = tile0 + wv * (64 * RWK_SW_ITEMS) + j * 64 + lane;
        keys[j] = (p < a.n) ? in_key[p] : 0ull;
        rows[j] = (p < a.n) ? (in_idx ? in_idx[p] : (u32)p) : 0u;
    }
#pragma unroll
    for (int j = 0; j < RWK_SW_ITEMS; j++) {
        const u64 p = tile0 + wv * (64 * RWK_SW_ITEMS) + j * 64 + lane;
        const bool valid = p < a.n;
        const u32 dg = valid ? (u32)((keys[j] >> shift) & 0xffull) : 0u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((dg >> b) & 1u);
            same &= ((dg >> b) & 1u) ? bal : ~bal;
        }
        volatile u32* cnt = &s_cnt[wv][dg];
        const u32 before = valid ? *cnt : 0u;
        const u32 r = before + (u32)__popcll(same & below);
        if (valid && (same & below) == 0ull) *cnt = before + (u32)__popcll(same);
        meta[j] = valid ? (dg | (r << 8)) : 0xffffffffu;
    }
    __syncthreads();
    // the tile's own histogram, published at once (the later tiles' look-backs can proceed), and its exclusive scan over the digits
    u32 wave_tot[RWK_SW_BLOCK / 64], tile_tot = 0;
    u32* mine = desc + (u64)tile * 256 + (threadIdx.x & 255u);
    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < RWK_SW_BLOCK / 64; w++) { wave_tot[w] = s_cnt[w][threadIdx.x]; tile_tot += wave_tot[w]; }
        __hip_atomic_store(mine, tile_tot | (tile == 0 ? RWK_DESC_INCL : RWK_DESC_LOCAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_texcl[threadIdx.x] = tile_tot;
    }
    __syncthreads();
    {
        const u32 d = threadIdx.x & 255u;
        for (u32 s = 1; s < 256; s <<= 1) {
            const u32 x = (threadIdx.x < 256 && d >= s) ? s_texcl[d - s] : 0u;
            __syncthreads();
            if (threadIdx.x < 256) s_texcl[d] += x;
            __syncthreads();
        }
    }
    if (threadIdx.x < 256) {
        const u32 d = threadIdx.x;
        const u32 texcl = s_texcl[d] - tile_tot;
        // look-back, eight predecessors per round trip: their descriptors are independent loads; the walk ends at the first
        // inclusive one.  (One predecessor per round trip made the pass a chain of ntiles dependent agent-scope loads: 19 of its 20 us.)
        u32 excl = 0;
        int p = (int)tile - 1;
        u32 spins = 0;
        while (p >= 0) {
            u32 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = p - q >= 0 ? __hip_atomic_load(desc + (u64)(p - q) * 256 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : RWK_DESC_INCL;
            bool done = false;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (done || p < 0) continue;
                if ((v[q] >> 30) == 0u) { done = true; continue; }  // not published yet: re-read from here
                excl += v[q] & RWK_DESC_VAL;
                p = (v[q] >> 30) == 2u ? -1 : p - 1;
            }
            if (done && ++spins >= (1u << 22)) { atomicOr(err, 1u); break; }  // (never seen: a predecessor did not publish within ~seconds)
        }
        if (tile) __hip_atomic_store(mine, (excl + tile_tot) | RWK_DESC_INCL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // where digit d's run of this tile starts in the output, minus where it starts in the tile-local order
        s_gbase[d] = s_scan[d] + excl - texcl;  // (s_scan: the digit's base = exclusive scan of the pass's global histogram; wraps mod 2^32, added back below)
        u32 run = texcl;
#pragma unroll
        for (int w = 0; w < RWK_SW_BLOCK / 64; w++) { s_cnt[w][d] = run; run += wave_tot[w]; }
    }
    __syncthreads();
    // tile-local reorder through LDS: the elements of one digit become one contiguous run, so that the global stores of a wavefront
    // cover whole runs (scattered 8 + 4-byte stores of an 8-bit digit pass cost 26 us of a 2^20-key pass, the rest of the kernel 16)
    extern __shared__ u64 s_dyn[];
    u64* s_key = s_dyn;
    u32* s_row = (u32*)(s_dyn + RWK_SW_TILE);
#pragma unroll
    for (int j = 0; j < RWK_SW_ITEMS; j++)
        if (meta[j] != 0xffffffffu) {
            const u32 q = s_cnt[wv][meta[j] & 0xffu] + (meta[j] >> 8);
            s_key[q] = keys[j];
            s_row[q] = rows[j];
        }
    __syncthreads();
    const u64 left = a.n - tile0;
    const u32 n_here = left < (u64)RWK_SW_TILE ? (u32)left : (u32)RWK_SW_TILE;
#pragma unroll
    for (int j = 0; j < RWK_SW_ITEMS; j++) {
        const u32 q = (u32)j * RWK_SW_BLOCK + threadIdx.x;
        if (q < n_here) {
            const u64 k = s_key[q];
            const u32 pos = s_gbase[(u32)((k >> shift) & 0xffull)] + q;
            out_idx[pos] = s_row[q];
            if (out_key) out_key[pos] = k;
        }
    }
}

// ---- LSD radix sort of the row indices, 8 bits per pass ---------------------------------------------------------------------
// A tile is RWK_TILE consecutive positions of the current order; wave w of the block owns positions [w * 1024, (w + 1) * 1024)
// of it, in 16 rounds of 64: position order = (tile, wave, round, lane), which is what makes the pass stable.
__device__ __forceinline__ u32 rwk_digit(const RekeyArgs& a, u32 row, u32 word, u32 shift) {
    return (a.keys[(u64)word * a.n + row] >> shift) & 0xffu;
}
__global__ __launch_bounds__(RWK_BLOCK) void rwk_hist_kernel(RekeyArgs a, const u32* in, u32 word, u32 shift) {
    __shared__ u32 s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const u64 tile0 = (u64)blockIdx.x * RWK_TILE;
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int j = 0; j < RWK_TILE_ITEMS; j++) {
        const u64 p = tile0 + wv * (64 * RWK_TILE_ITEMS) + j * 64 + lane;
        if (p < a.n) {
            const u32 row = in ? in[p] : (u32)p;
            atomicAdd(&s_h[rwk_digit(a, row, word, shift)], 1u);
        }
    }
    __syncthreads();
    a.hist[(u64)blockIdx.x * 256 + threadIdx.x] = s_h[threadIdx.x];
}
__global__ __launch_bounds__(RWK_BLOCK) void rwk_scatter_kernel(RekeyArgs a, const u32* in, u32* out, u32 word, u32 shift) {
    __shared__ u32 s_cnt[RWK_BLOCK / 64][256];  // per wave: running count of every digit -> the wave's first output slot of it
    __shared__ u32 s_scan[256];
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int w = 0; w < RWK_BLOCK / 64; w++) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const u64 tile0 = (u64)blockIdx.x * RWK_TILE;
    u32 rows[RWK_TILE_ITEMS], meta[RWK_TILE_ITEMS];  // meta = digit | rank inside (wave, digit) << 8; 0xffffffff = no element
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < RWK_TILE_ITEMS; j++) {
        const u64 p = tile0 + wv * (64 * RWK_TILE_ITEMS) + j * 64 + lane;
        rows[j] = (p < a.n) ? (in ? in[p] : (u32)p) : 0u;
    }
#pragma unroll
    for (int j = 0; j < RWK_TILE_ITEMS; j++) {
        const u64 p = tile0 + wv * (64 * RWK_TILE_ITEMS) + j * 64 + lane;
        const bool valid = p < a.n;
        const u32 dg = valid ? rwk_digit(a, rows[j], word, shift) : 0u;
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((dg >> b) & 1u);
            same &= ((dg >> b) & 1u) ? bal : ~bal;
        }
        // lanes holding the same digit: the lowest of them publishes the group's size after everyone has read the running count
        volatile u32* cnt = &s_cnt[wv][dg];
        const u32 before = valid ? *cnt : 0u;
        const u32 r = before + (u32)__popcll(same & below);
        if (valid && (same & below) == 0ull) *cnt = before + (u32)__popcll(same);
        meta[j] = valid ? (dg | (r << 8)) : 0xffffffffu;
    }
    __syncthreads();
    {   // thread d: digit d's first output slot for every wave of this tile
        const u32 d = threadIdx.x;
        u32 wave_tot[RWK_BLOCK / 64], tile_tot = 0;
#pragma unroll
        for (int w = 0; w < RWK_BLOCK / 64; w++) { wave_tot[w] = s_cnt[w][d]; tile_tot += wave_tot[w]; }
        u32 before = 0, all = 0;
        for (u32 t = 0; t < a.ntiles; t++) {
            const u32 h = a.hist[(u64)t * 256 + d];
            before += t < blockIdx.x ? h : 0u;
            all += h;
        }
        // exclusive scan of `all` over the digits
        s_scan[d] = all;
        __syncthreads();
        for (u32 s = 1; s < 256; s <<= 1) {
            const u32 x = d >= s ? s_scan[d - s] : 0u;
            __syncthreads();
            s_scan[d] += x;
            __syncthreads();
        }
        u32 run = s_scan[d] - all + before;
#pragma unroll
        for (int w = 0; w < RWK_BLOCK / 64; w++) { s_cnt[w][d] = run; run += wave_tot[w]; }
        (void)tile_tot;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RWK_TILE_ITEMS; j++)
        if (meta[j] != 0xffffffffu) out[s_cnt[wv][meta[j] & 0xffu] + (meta[j] >> 8)] = rows[j];
}

// ---- the op list -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RWK_BLOCK) void rwk_emit_kernel(RekeyArgs a, const u32* order) {
    const u64 j = (u64)blockIdx.x * RWK_BLOCK + threadIdx.x;
    if (j >= a.n_ops) return;
    if (j == 0) { rwk_emit_start(a.ops, a.op_flags, a.n_ops); return; }
    const u32 row = order ? order[j - 1] : (u32)(j - 1);
    const u64* p = a.rw + (u64)row * (RWK_RW_NCELLS * 4);
    const RwkKey k = rwk_key(p);
    const RwkOp o = rwk_op(p, a.rw_flags ? a.rw_flags[row] : 0u, k);
    rwk_emit(a.ops, a.op_flags, a.n_ops, j, k, o);
}

void zk_launch_rekey_scan(hipStream_t st, const RekeyArgs& a) {
    // few, large blocks: every block ends with one global atomic per mask word it saw set (256 x ~200, not 2,700 x ~200)
    u32 grid = (u32)((a.n + RWK_SCAN_BLOCK - 1) / RWK_SCAN_BLOCK);
    if (grid > 256u) grid = 256u;
    hipLaunchKernelGGL(rwk_scan_kernel, dim3(grid), dim3(RWK_SCAN_BLOCK), 0, st, a);
}
void zk_launch_state_rekey(hipStream_t st, const RekeyArgs& a, u32* status, ZkTally* tally) {
    const u32 grid = (u32)((a.n + RWK_BLOCK - 1) / RWK_BLOCK);
    if (a.n_jobs) {
        hipMemsetAsync(a.job_cursor, 0, sizeof(u32) * a.n_jobs, st);
        hipLaunchKernelGGL(rwk_collect_kernel, dim3(grid), dim3(RWK_BLOCK), 0, st, a);
        for (u32 j = 0; j < a.n_jobs; j++)
            hipLaunchKernelGGL(rwk_rank_kernel, dim3((a.jobs[j].count + RWK_BLOCK - 1) / RWK_BLOCK), dim3(RWK_BLOCK), 0, st, a, j);
    }
    if (a.fast) {
        {   // > 64 KiB of dynamic LDS has to be asked for, once per device
            static std::mutex m;
            static bool asked[64] = {false};
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> lock(m);
            if (dev >= 0 && dev < 64 && !asked[dev]) {
                (void)hipFuncSetAttribute((const void*)rwk_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RWK_SW_TILE * 12);
                asked[dev] = true;
            }
        }
        hipMemsetAsync(a.sweep, 0, ((size_t)RWK_SWEEP_HEAD + (size_t)a.n_passes * a.ntiles_fast * 256) * 4, st);
        u32 g64 = (u32)((a.n + 1023) / 1024);
        hipLaunchKernelGGL(rwk_pack64_kernel, dim3(g64 > 256u ? 256u : g64), dim3(1024), 0, st, a, status, tally);
        const u64* kin = a.key64_a;
        const u32* iin = nullptr;
        u64* kbufs[2] = {a.key64_b, a.key64_a};
        u32* ibufs[2] = {a.idx_a, a.idx_b};
        for (u32 p = 0; p < a.n_passes; p++) {
            const bool last = p + 1 == a.n_passes;
            hipLaunchKernelGGL(rwk_sweep_kernel, dim3(a.ntiles_fast), dim3(RWK_SW_BLOCK), RWK_SW_TILE * 12, st, a, kin, iin, last ? (u64*)nullptr : kbufs[p & 1u], ibufs[p & 1u], p);
            kin = kbufs[p & 1u];
            iin = ibufs[p & 1u];
        }
        if (a.ops) hipLaunchKernelGGL(rwk_emit_kernel, dim3((u32)((a.n_ops + RWK_BLOCK - 1) / RWK_BLOCK)), dim3(RWK_BLOCK), 0, st, a, iin);
        return;
    }
    hipLaunchKernelGGL(rwk_pack_kernel, dim3(grid), dim3(RWK_BLOCK), 0, st, a, status, tally);
    const u32* in = nullptr;
    u32* bufs[2] = {a.idx_a, a.idx_b};
    for (u32 p = 0; p < a.n_passes; p++) {
        const u32 word = a.key_words - 1u - (p >> 2), shift = 8u * (p & 3u);
        u32* out = bufs[p & 1u];
        hipLaunchKernelGGL(rwk_hist_kernel, dim3(a.ntiles), dim3(RWK_BLOCK), 0, st, a, in, word, shift);
        hipLaunchKernelGGL(rwk_scatter_kernel, dim3(a.ntiles), dim3(RWK_BLOCK), 0, st, a, in, out, word, shift);
        in = out;
    }
    if (a.ops) hipLaunchKernelGGL(rwk_emit_kernel, dim3((u32)((a.n_ops + RWK_BLOCK - 1) / RWK_BLOCK)), dim3(RWK_BLOCK), 0, st, a, in);
}
