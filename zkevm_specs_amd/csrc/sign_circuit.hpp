// Tx and Sig circuits: the per-transaction / per-signature SignVerify checks.
//
// Reference: src/zkevm_specs/tx_circuit.py — `SignVerifyChip.verify` :205-243 and the copy constraints
// of `verify_circuit` :253-291 (one unit = one tx slot, 12 tx-table rows at fixed offsets);
// src/zkevm_specs/sig_circuit.py — `Row.verify` :64-104, `verify_circuit` :113-122.
// The secp256k1 ECDSA verification itself is a third-party call in the reference (eth_keys,
// tx_circuit.py:147-158, util/ec.py:109-117): its outcome is consumed as a pre-computed column
// `ecdsa_status` (0 = verified, 1 = returned/asserted False, other = status code of the exception).
//
// Unit layout:  bytes  u8[n][9][32]: pk_x, pk_y (chip copies), ecdsa pk_x, pk_y, msg_hash_bytes (chip),
//                                   ecdsa msg_hash_bytes, pub_key_hash, ecdsa sig_r (LE), ecdsa sig_s (LE)
//               cells  column-major [8][n]: address, msg_hash lo, hi, sig_v, sig_r lo, hi, sig_s lo, hi
//               meta   u32[n][4]: ecdsa_status, expected is_valid (Sig circuit), malformed mask (bit k: byte
//                      row k was not a 32-byte bytes object on the host: every use fails a type assert), 0
// Tx rows: row-major [n_rows][5] (tx_id, tag, index, value lo, hi) + flags bit0 value.is_word.
// Keccak table (tx_circuit.py:38-61): row-major [m][5]: is_enabled, input_rlc, input_len, output lo, hi.
#pragma once
#include "row_circuits.hpp"

enum { SG_PK_X = 0, SG_PK_Y, SG_E_PK_X, SG_E_PK_Y, SG_MSG, SG_E_MSG, SG_PK_HASH, SG_E_SIG_R, SG_E_SIG_S, SG_NBYTES_ROWS };
enum { SG_ADDRESS = 0, SG_MSG_LO, SG_MSG_HI, SG_SIG_V, SG_SIG_R_LO, SG_SIG_R_HI, SG_SIG_S_LO, SG_SIG_S_HI, SG_NCELLS };

struct SignArgs {
    const uint8_t* bytes;  // [n][9][32]
    ZkCols cells;          // [8][n]
    const u32* meta;       // [n][4]
    ZkTable keccak;
    ZkTable tx_rows;       // Tx circuit only (n = 0 for the Sig circuit)
    Fr r;
    const u64* rpow;       // [64][4]: r^0 .. r^63 (canonical), built once per session (sign_fill_rpow)
    u32 is_sig;            // 0 = Tx circuit semantics, 1 = Sig circuit semantics
};
// powers of the keccak randomness for the 64-byte public-key RLC
ZK_HD void sign_fill_rpow(const Fr& r, u64* out) {
    Fr p = fr_from_u64(1);
    for (int k = 0; k < 64; k++) {
        for (int j = 0; j < 4; j++) out[4 * k + j] = (u64)p.v[2 * j] | ((u64)p.v[2 * j + 1] << 32);
        p = fr_mul(p, r);
    }
}

// one 32-byte row of the unit's byte block, fetched with two 16-byte loads (the rows are 32-byte
// aligned: unit stride 9 x 32 B) and kept as eight little-endian words
struct B32 {
    u32 w[8];
};
ZK_HD B32 sg_bytes(const SignArgs& a, u64 i, int k) {
    const uint4* p = reinterpret_cast<const uint4*>(a.bytes + (i * SG_NBYTES_ROWS + k) * 32);
    const uint4 lo = p[0], hi = p[1];
    B32 b;
    b.w[0] = lo.x; b.w[1] = lo.y; b.w[2] = lo.z; b.w[3] = lo.w;
    b.w[4] = hi.x; b.w[5] = hi.y; b.w[6] = hi.z; b.w[7] = hi.w;
    return b;
}
ZK_HD bool sg_bytes_eq(const B32& x, const B32& y) {
    u32 d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d |= x.w[k] ^ y.w[k];
    return d == 0;
}
ZK_HD u32 sg_byte(const B32& b, int k) { return (b.w[k >> 2] >> (8 * (k & 3))) & 0xffu; }
// little-endian integer of bytes [off, off + 16)
ZK_HD Fr sg_le128(const B32& b, int off) {
    Fr r = fr_zero();
#pragma unroll
    for (int k = 0; k < 4; k++) r.v[k] = b.w[(off >> 2) + k];
    return r;
}

// RLC of 64 bytes: sum of byte_k * r^k with lazy reduction.  Each term is < 2^262, the sum < 2^268:
// nine 32-bit limbs, reduced once (64 x 8 multiply-adds instead of 64 Montgomery multiplications).
ZK_HD Fr sg_rlc64(const B32& first32, const B32& last32, const u64* rpow) {
    u32 acc[9];
#pragma unroll
    for (int j = 0; j < 9; j++) acc[j] = 0;
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const u32 byte = k < 32 ? sg_byte(first32, k) : sg_byte(last32, k - 32);
        const Fr pw = fr_load(rpow + 4 * k);  // same address in every lane
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)acc[j] + (u64)pw.v[j] * byte;
            acc[j] = (u32)c;
            c >>= 32;
        }
        acc[8] += (u32)c;
    }
    Fr lo;
#pragma unroll
    for (int j = 0; j < 8; j++) lo.v[j] = acc[j];
    const Fr p = fr_modulus();
#pragma unroll
    for (int it = 0; it < 5; it++) {  // lo < 2^256 < 6p
        Fr t;
        const u32 bw = u256_sub(t, lo, p);
        lo = bw ? lo : t;
    }
    // acc[8] * 2^256 mod p: 2^256 mod p is the Montgomery one in canonical form
    return fr_add(lo, fr_mul(fr_from_u64(acc[8]), frm_one()));
}

#define SG_FAIL(kind, site) code = (code == 0u) ? ZK_CODE(kind, site) : code
#define SG_ASSERT(cond, site) code = (code == 0u && !(cond)) ? ZK_CODE(ZK_ASSERT, site) : code

ZK_HD u32 sign_check_unit(const SignArgs& a, u64 i) {
    u32 code = 0;
    const Fr address = zk_col(a.cells, SG_ADDRESS, i);
    const Fr msg_lo = zk_col(a.cells, SG_MSG_LO, i), msg_hi = zk_col(a.cells, SG_MSG_HI, i);
    const u32 ecdsa_status = a.meta[4 * i], expect_valid = a.meta[4 * i + 1], bad = a.meta[4 * i + 2];
    const bool is_np = a.is_sig ? true : !fr_is_zero(address);  // is_not_padding (:206)

    // 0. copy constraints between the chip and the ECDSA chip
    const B32 pk_x = sg_bytes(a, i, SG_PK_X), pk_y = sg_bytes(a, i, SG_PK_Y), msg = sg_bytes(a, i, SG_MSG);
    const B32 pk_hash = sg_bytes(a, i, SG_PK_HASH);
    SG_ASSERT(!(bad & 0x5u) && sg_bytes_eq(pk_x, sg_bytes(a, i, SG_E_PK_X)), 1);
    SG_ASSERT(!(bad & 0xau) && sg_bytes_eq(pk_y, sg_bytes(a, i, SG_E_PK_Y)), 2);
    SG_ASSERT(!(bad & 0x30u) && sg_bytes_eq(msg, sg_bytes(a, i, SG_E_MSG)), 3);
    if (a.is_sig) {
        // sig_r/sig_s.int_value() == int.from_bytes(ecdsa sig bytes, "little")  (sig_circuit.py:70-71)
#pragma unroll
        for (int which = 0; which < 2; which++) {
            const Fr lo = zk_col(a.cells, SG_SIG_R_LO + 2 * which, i), hi = zk_col(a.cells, SG_SIG_R_HI + 2 * which, i);
            const B32 b = sg_bytes(a, i, SG_E_SIG_R + which);
            // lo + (hi << 128) as a 384-bit integer vs the 256-bit byte value
            u32 sum[12];
            u64 c = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) {  // (unrolled: a run-time index would put lo / hi / sum on the stack)
                c += (u64)(k < 8 ? lo.v[k] : 0u) + (u64)((k >= 4) ? hi.v[k - 4] : 0u);
                sum[k] = (u32)c;
                c >>= 32;
            }
            bool eq = c == 0;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                u32 want = 0;
                if (k < 8) want = b.w[k];
                eq = eq && sum[k] == want;
            }
            SG_ASSERT(eq, 12 + which);
        }
        SG_ASSERT(fr_le_u64(zk_col(a.cells, SG_SIG_V, i), 1), 14);
    }
    // 1. keccak(pub_key_bytes) == pub_key_hash through the keccak table.  The RLC input is
    //    pk_y bytes then pk_x bytes in little-endian positions (see the oracle for the derivation).
    {
        const Fr acc = sg_rlc64(pk_y, pk_x, a.rpow);
        const B32& h = pk_hash;
        Fr q[KECCAK_NCELLS];
        q[0] = fr_from_u64(is_np ? 1 : 0);
        q[1] = is_np ? acc : fr_zero();
        q[2] = fr_from_u64(is_np ? 64 : 0);
        q[3] = is_np ? sg_le128(h, 0) : fr_zero();        // Word(bytes): lo = bytes[0:16] little-endian
        q[4] = is_np ? sg_le128(h, 16) : fr_zero();
        if (code == 0u) SG_ASSERT(!(bad & 0x40u) && keccak_contains(a.keccak, q), 4);
    }
    // 2. low 20 bytes of the hash (big-endian) == address
    {
        Fr addr = fr_zero();
#pragma unroll
        for (int k = 0; k < 20; k++) addr.v[k >> 2] |= sg_byte(pk_hash, 31 - k) << (8 * (k & 3));
        SG_ASSERT(fr_eq(addr, address), 5);
    }
    // 3. Word(msg_hash_bytes) (select(is_not_padding) for Tx) == msg_hash
    {
        const Fr lo = is_np ? sg_le128(msg, 0) : fr_zero(), hi = is_np ? sg_le128(msg, 16) : fr_zero();
        SG_ASSERT(fr_eq(lo, msg_lo) && fr_eq(hi, msg_hi), 6);
    }
    // 4. ECDSA outcome (pre-computed column)
    if (a.is_sig) {
        if (ecdsa_status >= 2) SG_FAIL(ecdsa_status >> 24, 7);
        else SG_ASSERT((ecdsa_status == 0 ? 1u : 0u) == (expect_valid ? 1u : 0u), 15);
        return code;
    }
    if (ecdsa_status == 1) SG_FAIL(ZK_ASSERT, 7);
    else if (ecdsa_status >= 2) SG_FAIL(ecdsa_status >> 24, 7);
    // copy constraints to the tx-table rows at fixed offsets (tx_circuit.py:270-289)
    {
        const u64 caller = i * 12 + 3, sign = i * 12 + 11;
        if (caller >= a.tx_rows.n) { SG_FAIL(ZK_INDEX_ERROR, 8); return code; }
        SG_ASSERT(!(a.tx_rows.flags ? (a.tx_rows.flags[caller] & 1u) : true), 8);  // value.value()
        SG_ASSERT(fr_eq(zk_table_cell(a.tx_rows, (u32)caller, 3), address), 9);
        if (sign >= a.tx_rows.n) { SG_FAIL(ZK_INDEX_ERROR, 10); return code; }
        SG_ASSERT(fr_eq(zk_table_cell(a.tx_rows, (u32)sign, 3), msg_lo), 10);
        SG_ASSERT(fr_eq(zk_table_cell(a.tx_rows, (u32)sign, 4), msg_hi), 11);
    }
    return code;
}
