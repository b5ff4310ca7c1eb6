// Bytecode and Exp circuits: per-row constraint evaluation (row-stencil siblings of the State
// circuit: one lane per row, column-major witness, neighbour rows wrap modulo n).
//
// Bytecode circuit — reference src/zkevm_specs/bytecode_circuit.py: `check_bytecode_row` :37-67 and
//   the transition helpers :71-100; driver loop in reference tests/test_bytecode_circuit.py:26-47.
//   Witness (12 cells): q_first, q_last, hash lo, hi, tag, index, value, is_code, push_data_left,
//   value_rlc, length, push_data_size (bytecode_circuit.py:15-26).
//   Keccak table row (5 cells): state_tag, input_rlc, input_len, output lo, hi (table.py:511-515).
//   The push table (byte -> push size, :174-179) is evaluated in closed form.
// Exp circuit — reference src/zkevm_specs/exp_circuit.py: `verify_step` :14-85 under
//   ConstraintSystem conditions (util/constraint_system.py:27-74), loop :88-97.
//   Witness (21 cells): q_usable, is_step, identifier, is_last, base lo,hi, exponent lo,hi,
//   exponentiation lo,hi, a lo,hi, b lo,hi, c lo,hi, d lo,hi, q lo,hi, r (table.py:519-535).
// Status = (kind << 24) | site, sites numbered in the reference's evaluation order.
#pragma once
#include "common.hpp"

// ----------------------------------------------------------------------------------------------
// Bytecode circuit
// ----------------------------------------------------------------------------------------------
enum { BC_Q_FIRST = 0, BC_Q_LAST, BC_HASH_LO, BC_HASH_HI, BC_TAG, BC_INDEX, BC_VALUE, BC_IS_CODE, BC_PUSH_LEFT,
       BC_VALUE_RLC, BC_LENGTH, BC_PUSH_SIZE, BC_NCELLS };

struct BytecodeArgs {
    ZkCols rows;
    ZkTable keccak;
    Fr r;  // keccak randomness (canonical)
    const u64* r_mont;  // optional: the same in Montgomery form, computed once per session (one product per row instead of two)
};


// `row in keccak_table` with all five cells given (set membership, bytecode_circuit.py:100).  The query stays in registers: the
// candidate row comes in one batch of loads and is compared with compile-time cell indices (zk_row_matches indexes the query at
// run time, which put it on the stack: 176 B per lane in the Bytecode kernel, 208 in the Tx / Sig one).
ZK_HD bool keccak_contains(const ZkTable& t, const Fr (&q)[KECCAK_NCELLS]) {
    u32 kind;
    (void)table_probe_inline<KECCAK_NCELLS, 0x1fu>(t, keccak_key_hash_cells(q[1], q[2]), q, kind);
    return kind != (u32)ZK_LOOKUP_UNSAT;  // one matching row or several: contained either way
}

#define RC_ASSERT(cond, site) code = (code == 0u && !(cond)) ? ZK_CODE(ZK_ASSERT, site) : code

// One row's twelve cells.  A lane loads only its own row; the nine cells the transition checks need from row i + 1 come from the
// lane that holds it (wave_shl:1 DPP moves on the device — the State kernel's neighbour exchange, mirrored; round 2 re-read them
// through L2: 1.44 x the algorithmic traffic), or from a second load on the host.
struct BcRow {
    Fr c[BC_NCELLS];
};
ZK_HD void bytecode_load_row(const ZkCols& w, u64 i, BcRow& R) {
#pragma unroll
    for (int k = 0; k < BC_NCELLS; k++) R.c[k] = zk_col(w, k, i);
}
#ifndef ZK_HOSTSIM
ZK_HD Fr bc_next_lane(const Fr& x) {  // the same cell in lane + 1 (every lane of the wavefront takes part)
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.v[k], 0x130, 0xf, 0xf, false);  // wave_shl:1
    return r;
}
#define BC_NEXT(cell) bc_next_lane(C.c[cell])
#else
#define BC_NEXT(cell) (N.c[cell])
#endif
// Row C against its successor N (host: loaded; device: the neighbour lane's registers, so every lane must call this).
ZK_HD u32 bytecode_check_loaded(const BytecodeArgs& a, const BcRow& C, const BcRow& N) {
    (void)N;
    u32 code = 0;
    const Fr& q_first = C.c[BC_Q_FIRST]; const Fr& q_last = C.c[BC_Q_LAST];
    const Fr& tag = C.c[BC_TAG];
    const Fr& hash_lo = C.c[BC_HASH_LO]; const Fr& hash_hi = C.c[BC_HASH_HI];
    const Fr& index = C.c[BC_INDEX]; const Fr& value = C.c[BC_VALUE]; const Fr& length = C.c[BC_LENGTH];
    const Fr& value_rlc = C.c[BC_VALUE_RLC];
    // the successor's cells, moved unconditionally with all lanes active
    const Fr ntag = BC_NEXT(BC_TAG), n_length = BC_NEXT(BC_LENGTH), n_index = BC_NEXT(BC_INDEX), n_is_code = BC_NEXT(BC_IS_CODE);
    const Fr n_hash_lo = BC_NEXT(BC_HASH_LO), n_hash_hi = BC_NEXT(BC_HASH_HI), n_value_rlc = BC_NEXT(BC_VALUE_RLC);
    const Fr n_value = BC_NEXT(BC_VALUE), n_left = BC_NEXT(BC_PUSH_LEFT);
    const bool is_header = fr_eq_u64(tag, 1), is_byte = fr_eq_u64(tag, 2);
    const bool next_header = fr_eq_u64(ntag, 1), next_byte = fr_eq_u64(ntag, 2);
    // EMPTY_HASH = keccak256("") as a Word (util/hash.py:13)
    const Fr empty_lo = fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull);
    const Fr empty_hi = fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull);
    const bool hdr_to_hdr_ok_len = fr_is_zero(length);
    const bool hdr_to_hdr_ok_hash = fr_eq(hash_lo, empty_lo) && fr_eq(hash_hi, empty_hi);

    if (fr_eq_u64(q_first, 1)) RC_ASSERT(is_header, 1);
    if (fr_is_zero(q_last)) {
        if (is_header) {
            RC_ASSERT(fr_eq(value, length), 2);
            RC_ASSERT(fr_is_zero(index), 3);
            if (next_byte) {  // check_bytecode_row_header_to_byte :71-76
                RC_ASSERT(fr_eq(n_length, length), 4);
                RC_ASSERT(fr_is_zero(n_index), 5);
                RC_ASSERT(fr_eq_u64(n_is_code, 1), 6);
                RC_ASSERT(fr_eq(n_hash_lo, hash_lo) && fr_eq(n_hash_hi, hash_hi), 7);
                RC_ASSERT(fr_eq(n_value_rlc, n_value), 8);
            }
            if (next_header) {  // check_bytecode_row_header_to_header :80-82
                RC_ASSERT(hdr_to_hdr_ok_len, 9);
                RC_ASSERT(hdr_to_hdr_ok_hash, 10);
            }
        }
        if (is_byte) {
            const Fr& push_size = C.c[BC_PUSH_SIZE]; const Fr& push_left = C.c[BC_PUSH_LEFT];
            const Fr& is_code = C.c[BC_IS_CODE];
            // (value, push_data_size) in push_table: value a byte, size = get_push_size(value) (opcode.py:432)
            const u32 v = value.v[0] & 0xffu;
            const u32 want = (v >= 0x60u && v <= 0x7fu) ? v - 0x5fu : 0u;
            RC_ASSERT(fr_le_u64(value, 255) && fr_eq_u64(push_size, want), 11);
            RC_ASSERT(fr_eq_u64(is_code, fr_is_zero(push_left) ? 1 : 0), 12);
            if (next_byte) {  // check_bytecode_row_byte_to_byte :86-94
                RC_ASSERT(fr_eq(n_length, length), 13);
                RC_ASSERT(fr_eq(n_index, fr_add_u64(index, 1)), 14);
                RC_ASSERT(fr_eq(n_hash_lo, hash_lo) && fr_eq(n_hash_hi, hash_hi), 15);
                const Fr want_rlc = fr_add(a.r_mont ? fr_mulc(value_rlc, fr_load(a.r_mont)) : fr_mul(value_rlc, a.r), n_value);
                RC_ASSERT(fr_eq(n_value_rlc, want_rlc), 16);
                if (fr_eq_u64(is_code, 1)) RC_ASSERT(fr_eq(n_left, push_size), 17);
                else RC_ASSERT(fr_eq(n_left, fr_sub_u64(push_left, 1)), 18);
            }
            if (next_header) {  // check_bytecode_row_byte_to_header :98-100
                RC_ASSERT(fr_eq(fr_add_u64(index, 1), length), 19);
                Fr q[KECCAK_NCELLS];
                q[0] = fr_from_u64(2);
                q[1] = value_rlc;
                q[2] = length;
                q[3] = hash_lo;
                q[4] = hash_hi;
                if (code == 0u) RC_ASSERT(keccak_contains(a.keccak, q), 20);
            }
        }
    }
    if (fr_eq_u64(q_last, 1)) {
        RC_ASSERT(is_header, 21);
        RC_ASSERT(hdr_to_hdr_ok_len, 22);
        RC_ASSERT(hdr_to_hdr_ok_hash, 23);
    }
    return code;
}
#ifdef ZK_HOSTSIM
ZK_HD u32 bytecode_check_row(const BytecodeArgs& a, u64 i) {  // host build: both rows loaded
    BcRow C, N;
    bytecode_load_row(a.rows, i, C);
    bytecode_load_row(a.rows, i + 1 == a.rows.n ? 0 : i + 1, N);
    return bytecode_check_loaded(a, C, N);
}
#endif

// ----------------------------------------------------------------------------------------------
// Exp circuit
// ----------------------------------------------------------------------------------------------
enum { EX_Q_USABLE = 0, EX_IS_STEP, EX_ID, EX_IS_LAST, EX_BASE, EX_EXPONENT = 6, EX_EXPONENTIATION = 8, EX_A = 10,
       EX_B = 12, EX_C = 14, EX_D = 16, EX_Q = 18, EX_R = 20, EX_NCELLS = 21 };

struct ExpArgs {
    ZkCols rows;
};

struct ExWord {
    Fr lo, hi;
};
ZK_HD ExWord ex_word(const ZkCols& w, int c, u64 i) {
    ExWord x;
    x.lo = zk_col(w, c, i);
    x.hi = zk_col(w, c + 1, i);
    return x;
}
ZK_HD bool ex_word_eq(const ExWord& a, const ExWord& b) { return fr_eq(a.lo, b.lo) && fr_eq(a.hi, b.hi); }
ZK_HD bool ex_fits(const ExWord& a) { return fr_fits128(a.lo) && fr_fits128(a.hi); }

// 64x64 -> 128 accumulate into a 256-bit integer at a 64-bit limb offset
ZK_HD void ex_acc_mul64(U256& acc, u64 a, u64 b, int limb_off) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0, p01 = (u64)a0 * b1, p10 = (u64)a1 * b0, p11 = (u64)a1 * b1;
    u32 wv[4];
    u64 c = p00;
    wv[0] = (u32)c;
    c = (c >> 32) + (u32)p01 + (u32)p10;
    wv[1] = (u32)c;
    c = (c >> 32) + (p01 >> 32) + (p10 >> 32) + (u32)p11;
    wv[2] = (u32)c;
    c = (c >> 32) + (p11 >> 32);
    wv[3] = (u32)c;
    u64 carry = 0;
    for (int k = 0; k < 8 - 2 * limb_off; k++) {
        carry += (u64)acc.v[2 * limb_off + k] + (k < 4 ? wv[k] : 0u);
        acc.v[2 * limb_off + k] = (u32)carry;
        carry >>= 32;
    }
}
// carries of util.mul_add_words (util/arithmetic.py:245-276) for well-formed a, b (cells < 2^128)
ZK_HD void ex_mul_add_carries(const ExWord& a, const ExWord& b, const ExWord& c, const ExWord& d, Fr& carry_lo, Fr& carry_hi) {
    const U256 av = u256_from_lo_hi(a.lo, a.hi), bv = u256_from_lo_hi(b.lo, b.hi);
    u64 a64[4], b64[4];
    for (int k = 0; k < 4; k++) { a64[k] = u256_limb64(av, k); b64[k] = u256_limb64(bv, k); }
    U256 lo = fr_zero(), mid = fr_zero();
    ex_acc_mul64(lo, a64[0], b64[0], 0);
    ex_acc_mul64(lo, a64[0], b64[1], 1);
    ex_acc_mul64(lo, a64[1], b64[0], 1);
    ex_acc_mul64(mid, a64[0], b64[2], 0);
    ex_acc_mul64(mid, a64[1], b64[1], 0);
    ex_acc_mul64(mid, a64[2], b64[0], 0);
    ex_acc_mul64(mid, a64[0], b64[3], 1);
    ex_acc_mul64(mid, a64[1], b64[2], 1);
    ex_acc_mul64(mid, a64[2], b64[1], 1);
    ex_acc_mul64(mid, a64[3], b64[0], 1);
    // both carries only go into nine-byte range checks: integer shift instead of a Montgomery product (common.hpp)
    carry_lo = div_2p128_for_range9(fr_add(lo, c.lo), d.lo);
    carry_hi = div_2p128_for_range9(fr_add(fr_add(mid, c.hi), carry_lo), d.hi);
}

#define EX_FAIL(kind, site) code = (code == 0u) ? ZK_CODE(kind, site) : code
// cond * x == 0 in a prime field  <=>  cond == 0 or x == 0   (ConstraintSystem._eval, :27-30)
#define EX_ZERO(cond_zero, x_zero, site) RC_ASSERT((cond_zero) || (x_zero), site)

ZK_HD u32 exp_check_row(const ExpArgs& a, u64 i) {
    const ZkCols& w = a.rows;
    const u64 in = i + 1 == w.n ? 0 : i + 1;
    u32 code = 0;
    const Fr is_step = zk_col(w, EX_IS_STEP, i), is_last = zk_col(w, EX_IS_LAST, i), r = zk_col(w, EX_R, i);
    const ExWord base = ex_word(w, EX_BASE, i), exponent = ex_word(w, EX_EXPONENT, i);
    const ExWord exn = ex_word(w, EX_EXPONENTIATION, i);
    const ExWord A = ex_word(w, EX_A, i), B = ex_word(w, EX_B, i), C = ex_word(w, EX_C, i), D = ex_word(w, EX_D, i);
    const ExWord Q = ex_word(w, EX_Q, i);
    const Fr one_m_last = fr_sub(fr_from_u64(1), is_last);
    const Fr one_m_r = fr_sub(fr_from_u64(1), r);
    // conditions are products of field elements: zero iff a factor is zero
    const bool c1z = fr_is_zero(is_step) || fr_is_zero(one_m_last);
    const bool c2z = fr_is_zero(is_step);
    const bool c3z = c1z || fr_is_zero(r);
    const bool c4z = c1z || fr_is_zero(one_m_r);
    const bool c5z = fr_is_zero(is_last);

    // every step except the last (:16-24)
    {
        const ExWord nbase = ex_word(w, EX_BASE, in), nd = ex_word(w, EX_D, in);
        EX_ZERO(c1z, ex_word_eq(base, nbase), 1);
        EX_ZERO(c1z, ex_word_eq(A, nd), 2);
        EX_ZERO(c1z, fr_eq(zk_col(w, EX_ID, i), zk_col(w, EX_ID, in)), 3);
    }
    // every step (:27-52); constrain_bool: cond * value in {0, 1}
    {
        // cond * value in {0, 1}: trivially so when both factors are 0 / 1 (the usual case), else the field product decides
        const bool step01 = fr_le_u64(is_step, 1);
        RC_ASSERT((step01 && fr_le_u64(is_last, 1)) || fr_le_u64(fr_mul(is_step, is_last), 1), 4);
        RC_ASSERT((step01 && fr_le_u64(r, 1)) || fr_le_u64(fr_mul(is_step, r), 1), 5);
        if (!fr_fits128(A.lo) || !fr_fits128(A.hi)) EX_FAIL(ZK_OVERFLOW_ERROR, 6);  // a.to_64s()
        if (!fr_fits128(B.lo) || !fr_fits128(B.hi)) EX_FAIL(ZK_OVERFLOW_ERROR, 7);  // b.to_64s()
        Fr clo, chi;
        ex_mul_add_carries(A, B, C, D, clo, chi);
        if (fr_byte_len(clo) > 9) EX_FAIL(ZK_CONSTRAINT, 8);   // cs.range_check is unconditional (:64-69)
        if (fr_byte_len(chi) > 9) EX_FAIL(ZK_CONSTRAINT, 9);
        // sites 10, 11: constraints that hold identically (carry is defined from them)
        EX_ZERO(c2z, ex_word_eq(exn, D), 12);
        EX_ZERO(c2z, fr_is_zero(C.lo) && fr_is_zero(C.hi), 13);
        // mul_add_words(Word(2), q, Word.from_lo(r), exponent) (:44-52)
        RC_ASSERT(fr_fits128(r), 15);  // Word.from_lo(r) sanity check
        if (!fr_fits128(Q.lo) || !fr_fits128(Q.hi)) EX_FAIL(ZK_OVERFLOW_ERROR, 17);  // q.to_64s()
        ExWord two, rw;
        two.lo = fr_from_u64(2); two.hi = fr_zero();
        rw.lo = r; rw.hi = fr_zero();
        ex_mul_add_carries(two, Q, rw, exponent, clo, chi);
        if (fr_byte_len(clo) > 9) EX_FAIL(ZK_CONSTRAINT, 18);
        if (fr_byte_len(chi) > 9) EX_FAIL(ZK_CONSTRAINT, 19);
    }
    const ExWord nexp = ex_word(w, EX_EXPONENT, in);
    // odd exponent (:55-64)
    EX_ZERO(c3z, fr_eq(nexp.lo, fr_sub_u64(exponent.lo, 1)), 22);
    EX_ZERO(c3z, fr_eq(nexp.hi, exponent.hi), 23);
    EX_ZERO(c3z, ex_word_eq(base, B), 24);
    // even exponent (:67-77)
    EX_ZERO(c4z, fr_eq(nexp.lo, Q.lo), 25);
    EX_ZERO(c4z, fr_eq(nexp.hi, Q.hi), 26);
    EX_ZERO(c4z, ex_word_eq(A, B), 27);
    // last step (:80-85)
    EX_ZERO(c5z, fr_eq_u64(exponent.lo, 2), 28);
    EX_ZERO(c5z, fr_is_zero(exponent.hi), 29);
    EX_ZERO(c5z, ex_word_eq(base, A), 30);
    EX_ZERO(c5z, ex_word_eq(base, B), 31);
    return code;
}
