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
};

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
