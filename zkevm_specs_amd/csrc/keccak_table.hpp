// Keccak table generation: one table row per message.
//
// Replaces, for a batch of byte strings,
//   KeccakCircuit.add   (evm_circuit/typing.py:854-865; used by assign_keccak_table, bytecode_circuit.py:182-186,
//                        and by the SHA3 / CREATE tests):  (state_tag = 2, RLC(reversed(data), r), len, Word(BE int))
//   KeccakTable.add     (util/tables.py:18-27 == tx_circuit.py:48-58; Tx / Sig circuits):
//                        (is_enabled = 1, RLC(reversed(input), r, n_bytes = 64), len, Word(digest bytes))
// RLC(reversed(data)) = sum data[len-1-i] * r^i, i.e. Horner over the message front to back
// (util/arithmetic.py:9-24,69-96).  The digest itself is third-party in the reference (pycryptodome /
// eth_utils keccak); algorithm restated in keccak.hpp.
//
// One lane per message.  Message bytes are fetched as aligned 64-bit words and funnel-shifted to
// the message's byte offset; the RLC runs in 64-byte chunks with one lazy reduction per chunk
// (64 x 8 multiply-adds + one Montgomery multiplication by r^64 instead of 64 multiplications).
#pragma once
#include "keccak.hpp"

#define KT_NCELLS 5
#define KT_RPOW_ROWS 66  // r^0 .. r^64 canonical, then r^64 in Montgomery form

enum { KT_MODE_CIRCUIT = 0, KT_MODE_TABLE = 1 };

struct KeccakGenArgs {
    const uint8_t* data;  // concatenated messages
    const u64* offsets;   // [n + 1] byte offsets into data, non-decreasing
    u64 n;
    const u64* rpow;      // [KT_RPOW_ROWS][4]
    u64* rows;            // out: [n][KT_NCELLS][4]
    u32 mode;
    // device only: messages of KT_GROUP_MIN_BYTES and more (KeccakCircuit.add mode) are left to the lane-group kernel
    u32* long_list;       // [n] their indices, in no particular order (nullptr: the one-lane form handles every message)
    u32* long_count;
};
#define KT_GROUP_MIN_BYTES 544u  // four rate blocks: below that the one-lane form's 64 messages per wavefront win

ZK_HD void kt_fill_rpow(const Fr& r, u64* out) {
    Fr p = fr_from_u64(1);
    for (int k = 0; k <= 64; k++) {
        for (int j = 0; j < 4; j++) out[4 * k + j] = (u64)p.v[2 * j] | ((u64)p.v[2 * j + 1] << 32);
        if (k < 64) p = fr_mul(p, r);
    }
    const Fr m = fr_to_mont(p);
    for (int j = 0; j < 4; j++) out[4 * 65 + j] = (u64)m.v[2 * j] | ((u64)m.v[2 * j + 1] << 32);
}

// Sequential reader of a message as little-endian 64-bit words: aligned loads, funnel-shifted to the
// message's byte offset, each aligned word fetched once.  Only aligned words that contain at least
// one message byte are touched; bytes beyond the end read as zero.
struct KtStream {
    const u64* al;  // aligned word holding the next unread byte
    u64 cur;        // its value
    u64 left;       // unread bytes
    u32 s;          // bit offset of the message inside aligned words
};
ZK_HD KtStream kt_stream(const uint8_t* p, u64 len) {
    KtStream st;
    const uintptr_t addr = (uintptr_t)p;
    st.al = (const u64*)(addr & ~(uintptr_t)7);
    st.s = (u32)(addr & 7u) * 8u;
    st.left = len;
    st.cur = len ? st.al[0] : 0;
    return st;
}
ZK_HD u64 kt_next(KtStream& st) {
    if (st.left == 0) return 0;
    u64 w = st.cur >> st.s;
    u64 nxt = 0;
    if (st.left * 8u > 64u - st.s) nxt = *++st.al;  // the following aligned word holds message bytes
    if (st.s) w |= nxt << (64u - st.s);
    st.cur = nxt;
    if (st.left < 8) {
        w &= (~0ull) >> (64u - 8u * (u32)st.left);
        st.left = 0;
    } else {
        st.left -= 8;
    }
    return w;
}

ZK_HD void keccak_f1600_regs(u64 a[25]) {
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        u64 c[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            const u64 d = c[(x + 4) % 5] ^ keccak_rol(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rol(a[x + 5 * y], ROT[x + 5 * y]);
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

// sum of byte_k * r^(top - k) over the next `cnt` <= 64 bytes of the stream, reduced to canonical form
// (the stream must end at the chunk or on an 8-byte word boundary of it: cnt == 64 or the last chunk)
ZK_HD Fr kt_chunk(KtStream& st, u32 cnt, u32 top, const u64* rpow) {
    u32 acc[9];
#pragma unroll
    for (int j = 0; j < 9; j++) acc[j] = 0;
    for (u32 wi = 0; wi * 8u < cnt; wi++) {
        const u64 w = kt_next(st);
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 idx = 8u * wi + k;
            const u32 byte = (u32)(w >> (8u * k)) & 0xffu;  // zero beyond the end of the stream
            const Fr pw = fr_load(rpow + 4 * (top >= idx ? top - idx : 0u));
            u64 c = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c += (u64)acc[j] + (u64)pw.v[j] * byte;
                acc[j] = (u32)c;
                c >>= 32;
            }
            acc[8] += (u32)c;
        }
    }
    Fr lo;
#pragma unroll
    for (int j = 0; j < 8; j++) lo.v[j] = acc[j];
    const Fr pmod = fr_modulus();
#pragma unroll
    for (int it = 0; it < 5; it++) {  // lo < 2^256 < 6p
        Fr t;
        const u32 bw = u256_sub(t, lo, pmod);
        lo = bw ? lo : t;
    }
    // acc[8] * 2^256 mod p: 2^256 mod p is the Montgomery one in canonical form
    return fr_add(lo, fr_mul(fr_from_u64(acc[8]), frm_one()));
}

ZK_HD void kt_store(u64* out, const Fr& x) {
#pragma unroll
    for (int j = 0; j < 4; j++) out[j] = (u64)x.v[2 * j] | ((u64)x.v[2 * j + 1] << 32);
}
ZK_HD u64 kt_bswap64(u64 x) {
    x = ((x & 0x00ff00ff00ff00ffull) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffull);
    x = ((x & 0x0000ffff0000ffffull) << 16) | ((x >> 16) & 0x0000ffff0000ffffull);
    return (x << 32) | (x >> 32);
}

// Row i of the table.  Returns the status code of the message: KeccakTable.add raises ValueError
// for inputs longer than 64 bytes (RLC n_bytes = 64, util/arithmetic.py:82-83); everything else is 0.
ZK_HD u32 keccak_table_row(const KeccakGenArgs& g, u64 i) {
    const u64 o0 = g.offsets[i], o1 = g.offsets[i + 1];
    const uint8_t* p = g.data + o0;
    const u64 len = o1 - o0;
    u64* out = g.rows + i * (KT_NCELLS * 4);
    if (g.mode == KT_MODE_TABLE && len > 64) {
        for (int j = 0; j < KT_NCELLS * 4; j++) out[j] = 0;
        return ((u32)ZK_VALUE_ERROR << 24) | 1u;
    }
    // digest: absorb 136-byte blocks; the last block carries the 0x01 .. 0x80 padding
    u64 a[25];
#pragma unroll
    for (int k = 0; k < 25; k++) a[k] = 0;
    const u64 nblocks = len / 136 + 1;
    KtStream st = kt_stream(p, len);
    for (u64 blk = 0; blk < nblocks; blk++) {
        const u64 off = blk * 136;
        const bool last = blk + 1 == nblocks;
        const u32 rem = last ? (u32)(len - off) : 136u;  // message bytes in this block (< 136 when last)
#pragma unroll
        for (u32 k = 0; k < 17; k++) {
            u64 w = kt_next(st);
            if (last && (rem >> 3) == k) w ^= 1ull << (8u * (rem & 7u));
            a[k] ^= w;
        }
        if (last) a[16] ^= 0x80ull << 56;
        keccak_f1600_regs(a);
    }
    // input RLC, front to back: a leading partial chunk, then whole 64-byte chunks
    const u32 m0 = (u32)(len & 63u);
    Fr acc = fr_zero();
    if (m0) {
        KtStream lead = kt_stream(p, m0);
        acc = kt_chunk(lead, m0, m0 - 1, g.rpow);
    }
    const Fr r64m = fr_load(g.rpow + 4 * 65);
    KtStream body = kt_stream(p + m0, len - m0);
    for (u64 off = m0; off < len; off += 64) acc = fr_add(fr_mulc(acc, r64m), kt_chunk(body, 64, 63, g.rpow));

    kt_store(out + 0, fr_from_u64(g.mode == KT_MODE_TABLE ? 1 : 2));
    kt_store(out + 4, acc);
    kt_store(out + 8, fr_from_u64(len));
    if (g.mode == KT_MODE_TABLE) {  // Word(bytes): lo = digest[0:16] little-endian (util/arithmetic.py:99-123)
        out[12] = a[0]; out[13] = a[1]; out[14] = 0; out[15] = 0;
        out[16] = a[2]; out[17] = a[3]; out[18] = 0; out[19] = 0;
    } else {  // Word(int.from_bytes(digest, "big"))
        out[12] = kt_bswap64(a[3]); out[13] = kt_bswap64(a[2]); out[14] = 0; out[15] = 0;
        out[16] = kt_bswap64(a[1]); out[17] = kt_bswap64(a[0]); out[18] = 0; out[19] = 0;
    }
    return 0;
}


#ifndef ZK_HOSTSIM
// ---------------------------------------------------------------------------------------
// Long messages: a GROUP of 32 lanes per message (round 4).  With one lane per message a contract's bytecode is a chain of
// len / 136 permutations of ~7,000 instructions each on ONE lane (24 KiB: 180 of them, 7.8 ms for a launch of 4,096 such
// messages on 64 wavefronts; a block's 16 contracts bound its keccak-table launch at 1.7 ms).  Here lane l = x + 5 y of the group
// holds state word A[x][y]: theta's column parities, rho-pi's permutation and chi's row neighbours are lane shuffles (nine 64-bit
// shuffles per round), the 17 rate words of a block are fetched by 17 lanes at once, and the input RLC is cut into per-lane runs
// of 64-byte chunks whose partial sums are weighted with (r^64)^k and added through a shuffle tree.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 kt_shfl64(u64 v, int lane) {
    const u32 lo = (u32)__shfl((int)(u32)v, lane), hi = (u32)__shfl((int)(u32)(v >> 32), lane);
    return (u64)lo | ((u64)hi << 32);
}
__device__ __forceinline__ u64 kt_rolv(u64 x, u32 n) { return (x << (n & 63u)) | (x >> ((64u - n) & 63u)); }
// bytes [at, at + 8) of a message as a little-endian word; bytes at or beyond `len` read as zero (never touched)
__device__ __forceinline__ u64 kt_word_at(const uint8_t* p, u64 at, u64 len) {
    u64 w = 0;
#pragma unroll
    for (u32 b = 0; b < 8; b++)
        if (at + b < len) w |= (u64)p[at + b] << (8u * b);
    return w;
}
__device__ __forceinline__ Fr kt_shfl_fr(const Fr& a, int lane) {
    Fr o;
#pragma unroll
    for (int j = 0; j < 8; j++) o.v[j] = (u32)__shfl((int)a.v[j], lane);
    return o;
}
// one message by the 32 lanes [base, base + 32) of the wavefront; every lane of the group calls this with the same i
// `lds`: 64 u64 of LDS owned by this lane group (two 32-entry arrays), or nullptr for the shuffle form of the rounds (round 4: nine
// 64-bit shuffles = 18 ds_bpermute in four dependent stages per round).  With it a round is two stages through LDS: every lane
// stores its word, reads the ten words of the columns x - 1 and x + 1 (theta without the separate parity exchange), stores its
// rotated word, and reads the three rotated words chi needs — its own pi source and the pi sources of (x + 1, y), (x + 2, y):
// 2 stores + 13 loads of 8 bytes, 11 DS instructions, two dependent round trips instead of four.
__device__ __forceinline__ void keccak_table_row_group(const KeccakGenArgs& g, u64 i, u32 gl /* lane in group */, int base, u64* lds = nullptr) {
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    const u64 o0 = g.offsets[i], o1 = g.offsets[i + 1];
    const uint8_t* p = g.data + o0;
    const u64 len = o1 - o0;
    // lane geometry (lanes 25..31 shadow lane 24's sources: their values are never read by lanes 0..24)
    const u32 l = gl < 25u ? gl : 24u;
    const u32 x = l % 5u, y = l / 5u;
    // rotation offsets r[x][y] of rho, indexed x + 5 y, six bits each
    const u64 ROT0 = (0ull) | (1ull << 6) | (62ull << 12) | (28ull << 18) | (27ull << 24) | (36ull << 30) | (44ull << 36) | (6ull << 42) | (55ull << 48) | (20ull << 54);
    const u64 ROT1 = (3ull) | (10ull << 6) | (43ull << 12) | (25ull << 18) | (39ull << 24) | (41ull << 30) | (45ull << 36) | (15ull << 42) | (21ull << 48) | (8ull << 54);
    const u64 ROT2 = (18ull) | (2ull << 6) | (61ull << 12) | (56ull << 18) | (14ull << 24);
    const u32 my_rot = (u32)(((l < 10u ? ROT0 : l < 20u ? ROT1 : ROT2) >> (6u * (l % 10u))) & 63u);
    const int col1 = base + (int)((l + 5u) % 25u), col2 = base + (int)((l + 10u) % 25u), col3 = base + (int)((l + 15u) % 25u), col4 = base + (int)((l + 20u) % 25u);
    const int xm1 = base + (int)(5u * y + (x + 4u) % 5u), xp1 = base + (int)(5u * y + (x + 1u) % 5u), xp2 = base + (int)(5u * y + (x + 2u) % 5u);
    // pi as a gather: the word that lands at (x', y') = (x, y) comes from ((x' + 3 y') mod 5, x')
    const int pi_src = base + (int)(((x + 3u * y) % 5u) + 5u * x);
    u64 a = 0;
    const u64 nblocks = len / 136 + 1;
    // the next block's rate word is fetched before this block's 24 rounds run (its eight byte loads land under them; fetched where
    // it is needed, they were ~1.5 us of exposed latency per 136-byte block)
    u64 w_next = gl < 17u ? kt_word_at(p, 8u * gl, len) : 0ull;
    for (u64 blk = 0; blk < nblocks; blk++) {
        const u64 off = blk * 136;
        u64 w = 0;
        if (gl < 17u) {
            w = w_next;
            if (blk + 1 < nblocks) w_next = kt_word_at(p, off + 136u + 8u * gl, len);
            if (blk + 1 == nblocks) {  // pad10*1 inside the last block
                const u32 rem = (u32)(len - off);
                if ((rem >> 3) == gl) w ^= 1ull << (8u * (rem & 7u));
                if (gl == 16u) w ^= 0x80ull << 56;
            }
        }
        a ^= w;
        if (lds) {
            u64* A = lds;
            u64* R = lds + 32;
            const u32 xm = (x + 4u) % 5u, xp = (x + 1u) % 5u;  // the neighbouring columns' x
            const u32 x1 = (x + 1u) % 5u, x2 = (x + 2u) % 5u;
            const u32 p0 = ((x + 3u * y) % 5u) + 5u * x, p1 = ((x1 + 3u * y) % 5u) + 5u * x1, p2 = ((x2 + 3u * y) % 5u) + 5u * x2;
            // rotate by the lane's own amount with two funnel shifts (v_alignbit_b32) on the halves, swapped first when the amount is
            // 32 or more — full-rate instructions where a 64-bit variable shift pair is not; the 24 rounds are unrolled (round constants
            // as immediates: the rolled loop fetched RC[round] with a scalar load and a branch for lane 0 every round)
            const u32 rot_swap = my_rot & 32u, rot_n = (32u - (my_rot & 31u)) & 31u;
            const bool rot_id = (my_rot & 31u) == 0u;
            const u64 iota_mask = gl == 0u ? ~0ull : 0ull;
#pragma unroll
            for (int round = 0; round < 24; round++) {
                A[gl] = a;  // (lanes 25..31 use slots 25..31: nobody reads them)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const u64 cm = A[xm] ^ A[xm + 5u] ^ A[xm + 10u] ^ A[xm + 15u] ^ A[xm + 20u];
                const u64 cp = A[xp] ^ A[xp + 5u] ^ A[xp + 10u] ^ A[xp + 15u] ^ A[xp + 20u];
                a ^= cm ^ kt_rolv(cp, 1);                                                                            // theta
                {                                                                                                    // rho
                    u32 lo = (u32)a, hi = (u32)(a >> 32);
                    if (rot_swap) { const u32 t = lo; lo = hi; hi = t; }
                    // rotl by m = my_rot & 31 on (hi:lo): lo' = (lo << m) | (hi >> (32 - m)) = alignbit(lo, hi, 32 - m), likewise hi'
                    const u32 nlo = rot_id ? lo : __builtin_amdgcn_alignbit(lo, hi, rot_n);
                    const u32 nhi = rot_id ? hi : __builtin_amdgcn_alignbit(hi, lo, rot_n);
                    R[gl] = (u64)nlo | ((u64)nhi << 32);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const u64 b0 = R[p0], b1 = R[p1], b2 = R[p2];                                                        // pi (gathers)
                a = b0 ^ (~b1 & b2);                                                                                 // chi
                a ^= RC[round] & iota_mask;                                                                          // iota (lane 0)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the next round's stores stay behind this round's loads
            }
        } else
#pragma unroll 1
        for (int round = 0; round < 24; round++) {
            const u64 c = a ^ kt_shfl64(a, col1) ^ kt_shfl64(a, col2) ^ kt_shfl64(a, col3) ^ kt_shfl64(a, col4);  // column parity C[x]
            a ^= kt_shfl64(c, xm1) ^ kt_rolv(kt_shfl64(c, xp1), 1);                                                  // theta
            const u64 b = kt_shfl64(kt_rolv(a, my_rot), pi_src);                                                     // rho + pi
            a = b ^ (~kt_shfl64(b, xp1) & kt_shfl64(b, xp2));                                                        // chi
            if (gl == 0u) a ^= RC[round];                                                                            // iota
        }
    }
    // input RLC: a leading partial chunk (lane 0), then the W whole 64-byte chunks in 32 runs of q chunks
    const u32 m0 = (u32)(len & 63u);
    const u64 W = (len - m0) / 64;
    const u64 q = (W + 31) / 32;
    const u64 c_lo = (u64)gl * q < W ? (u64)gl * q : W, c_hi = ((u64)gl + 1) * q < W ? ((u64)gl + 1) * q : W;
    Fr acc = fr_zero();
    if (gl == 0u && m0) {
        KtStream lead = kt_stream(p, m0);
        acc = kt_chunk(lead, m0, m0 - 1, g.rpow);
    }
    const Fr r64m = fr_load(g.rpow + 4 * 65);
    if (c_hi > c_lo) {
        KtStream body = kt_stream(p + m0 + 64 * c_lo, 64 * (c_hi - c_lo));
        for (u64 c = c_lo; c < c_hi; c++) acc = fr_add(fr_mulc(acc, r64m), kt_chunk(body, 64, 63, g.rpow));
    }
    {   // weight: (r^64)^(W - c_hi), square-and-multiply on the Montgomery form of r^64
        u64 e = W - c_hi;
        Fr sq = r64m;
        while (__any(e != 0)) {
            if (e & 1u) acc = fr_mulc(acc, sq);
            sq = fr_mont(sq, sq);
            e >>= 1;
        }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
        const Fr o = kt_shfl_fr(acc, base + (int)((gl + (u32)d) & 31u));
        if (gl < (u32)d) acc = fr_add(acc, o);
    }
    // digest words sit in lanes 0..3
    const u64 d0 = kt_shfl64(a, base + 0), d1 = kt_shfl64(a, base + 1), d2 = kt_shfl64(a, base + 2), d3 = kt_shfl64(a, base + 3);
    if (gl == 0u) {
        u64* out = g.rows + i * (KT_NCELLS * 4);
        kt_store(out + 0, fr_from_u64(2));
        kt_store(out + 4, acc);
        kt_store(out + 8, fr_from_u64(len));
        out[12] = kt_bswap64(d3); out[13] = kt_bswap64(d2); out[14] = 0; out[15] = 0;
        out[16] = kt_bswap64(d1); out[17] = kt_bswap64(d0); out[18] = 0; out[19] = 0;
    }
}
#endif
