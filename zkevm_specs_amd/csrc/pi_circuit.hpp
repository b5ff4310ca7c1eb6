// Public-inputs (PI) circuit: per-row gate and lookup evaluation.
//
// Reference: `check_row(row, row_next, calldata_gas_cost_table, fixed_u16_table, keccak_table, circuit_len)`
// src/zkevm_specs/pi_circuit.py:150-322, called for every row with the next row wrapping modulo n by `verify_circuit`
// (:447-459).  Row = 24 cells, column-major (order: oracle/pi_oracle.py / flatten.flatten_pi_rows): the running keccak-RLC
// and per-value linear combination of the raw public-input bytes (gates 1-4), the keccak-table membership of the digest on
// the first row (5), the calldata rows' tx_id / index / gas-cost accumulation with their is-zero inverses (6-22, incl. the
// fixed u16 range lookup), the tx-table rows' CallDataLength -> gas-cost-table lookup (23-26) and the withdrawal ids (27, 28).
// Site numbers follow the reference's evaluation order; every `assert` is an AssertionError, the two `lookup()` calls raise
// LookupUnsatFailure (table.py:864-884; all their fields are given, so a match is unique).
// Products that are only compared with zero are evaluated with the zero-product rule of the prime field; the is-zero /
// is-nonzero indicators (x * x_inv) are real field products.
#pragma once
#include "common.hpp"

enum { PI_Q_BYTES_LAST = 0, PI_Q_TX_TABLE, PI_Q_TX_CALLDATA, PI_Q_TX_CALLDATA_START, PI_Q_KECCAK, PI_Q_VALUE_START, PI_TX_ID_INV, PI_TX_LO_INV,
       PI_TX_DIFF_INV, PI_GAS_COST, PI_IS_FINAL, PI_Q_WD, PI_RPI_BYTES, PI_RPI_RLC, PI_RPI_LC, PI_DIGEST_LO, PI_DIGEST_HI, PI_Q_BYTE_EN,
       PI_TX_ID, PI_TX_TAG, PI_TX_INDEX, PI_TX_LO, PI_WD_ID, PI_WD_AMOUNT, PI_NCELLS };
enum { PI_GAS_NCELLS = 3 };

struct PiArgs {
    ZkCols rows;
    ZkTable keccak;      // (is_enabled, input_rlc, input_len, output lo, hi), keyed on (rlc, len)
    ZkTable gas;         // TxCallDataGasCostAccRow (tx_id, is_final, gas_cost_acc), keyed on all three
    Fr circuit_len;
    Fr keccak_rand_m;    // Montgomery form
    Fr byte_pow_base_m;  // Montgomery form
};

ZK_HD u64 pi_gas_key_hash_cells(const Fr& a, const Fr& b, const Fr& c) { return zk_hash_cell(zk_hash_cell(zk_hash_cell(0x9a5c057u, a), b), c); }
ZK_HD u64 pi_gas_key_hash(const ZkTable& t, u32 r) { return pi_gas_key_hash_cells(zk_table_cell(t, r, 0), zk_table_cell(t, r, 1), zk_table_cell(t, r, 2)); }
ZK_HD bool pi_gas_contains(const ZkTable& t, const Fr& a, const Fr& b, const Fr& c) {
    if (t.n == 0) return false;
    u32 slot = (u32)pi_gas_key_hash_cells(a, b, c) & t.mask;
    for (u32 probes = 0; probes <= t.mask; probes++) {
        const u32 r = t.slots[slot];
        if (r == ZK_EMPTY_SLOT) return false;
        if (fr_eq(zk_table_cell(t, r, 0), a) && fr_eq(zk_table_cell(t, r, 1), b) && fr_eq(zk_table_cell(t, r, 2), c)) return true;
        slot = (slot + 1) & t.mask;
    }
    return false;
}
ZK_HD bool pi_keccak_contains(const ZkTable& t, const Fr q[5]) {
    if (t.n == 0) return false;
    u32 slot = (u32)keccak_key_hash_cells(q[1], q[2]) & t.mask;
    for (u32 probes = 0; probes <= t.mask; probes++) {
        const u32 r = t.slots[slot];
        if (r == ZK_EMPTY_SLOT) return false;
        bool m = true;
        for (int c = 0; c < 5; c++) m = m && fr_eq(zk_table_cell(t, r, c), q[c]);
        if (m) return true;
        slot = (slot + 1) & t.mask;
    }
    return false;
}

#define PI_ASSERT(cond, site) code = (code == 0u && !(cond)) ? ZK_CODE(ZK_ASSERT, site) : code
#define PI_FAIL(kind, site) code = (code == 0u) ? ZK_CODE(kind, site) : code

ZK_HD u32 pi_check_row(const PiArgs& a, u64 i) {
    const ZkCols& w = a.rows;
    const u64 in = i + 1 == w.n ? 0 : i + 1;
    u32 code = 0;
    const Fr one = fr_from_u64(1);
    const Fr en = zk_col(w, PI_Q_BYTE_EN, i), last = zk_col(w, PI_Q_BYTES_LAST, i), vstart = zk_col(w, PI_Q_VALUE_START, i);
    const Fr bytes = zk_col(w, PI_RPI_BYTES, i), rlc = zk_col(w, PI_RPI_RLC, i), lc = zk_col(w, PI_RPI_LC, i);
    const bool en_z = fr_is_zero(en);
    // 1: rpi_bytes_keccakrlc[last] = rpi_bytes[last]
    PI_ASSERT(en_z || fr_is_zero(last) || fr_eq(rlc, bytes), 1);
    // 2: rpi_bytes_keccakrlc[i] = keccak_rand * rpi_bytes_keccakrlc[i + 1] + rpi_bytes[i]
    PI_ASSERT(en_z || fr_eq(last, one) || fr_eq(rlc, fr_add(fr_mulc(zk_col(w, PI_RPI_RLC, in), a.keccak_rand_m), bytes)), 2);
    // 3: rpi_value_lc[i] = rpi_value_lc[i + 1] * byte_pow_base + rpi_bytes[i]
    PI_ASSERT(en_z || fr_eq(vstart, one) || fr_eq(lc, fr_add(fr_mulc(zk_col(w, PI_RPI_LC, in), a.byte_pow_base_m), bytes)), 3);
    // 4: rpi_value_lc[i] = rpi_bytes[i]
    PI_ASSERT(en_z || fr_is_zero(vstart) || fr_eq(lc, bytes), 4);
    // 5: (q, q * rlc, q * circuit_len, rpi_digest_word.select(q)) in keccak_table; select() builds a checked Word
    {
        const Fr q = zk_col(w, PI_Q_KECCAK, i);
        Fr t[5];
        t[0] = q;
        t[1] = fr_mul(q, rlc);
        t[2] = fr_mul(q, a.circuit_len);
        t[3] = fr_mul(zk_col(w, PI_DIGEST_LO, i), q);
        t[4] = fr_mul(zk_col(w, PI_DIGEST_HI, i), q);
        PI_ASSERT(fr_fits128(t[3]) && fr_fits128(t[4]) && pi_keccak_contains(a.keccak, t), 5);
    }
    const Fr tx_id = zk_col(w, PI_TX_ID, i), tx_lo = zk_col(w, PI_TX_LO, i), id_inv = zk_col(w, PI_TX_ID_INV, i), lo_inv = zk_col(w, PI_TX_LO_INV, i);
    const Fr n_tx_id = zk_col(w, PI_TX_ID, in), n_tx_lo = zk_col(w, PI_TX_LO, in);
    if (!fr_is_zero(zk_col(w, PI_Q_TX_CALLDATA, i))) {
        const Fr diff_inv = zk_col(w, PI_TX_DIFF_INV, i), gas_cost = zk_col(w, PI_GAS_COST, i), is_final = zk_col(w, PI_IS_FINAL, i);
        const Fr n_gas_cost = zk_col(w, PI_GAS_COST, in), tx_index = zk_col(w, PI_TX_INDEX, i), n_tx_index = zk_col(w, PI_TX_INDEX, in);
        const Fr d = fr_sub(n_tx_id, tx_id);
        const Fr id_nz = fr_mul(tx_id, id_inv);                               // is_tx_id_nonzero
        const Fr id_next_nz = fr_mul(n_tx_id, zk_col(w, PI_TX_ID_INV, in));   // is_tx_id_next_nonzero
        const Fr neq_next = fr_mul(d, diff_inv);                              // tx_id_not_equal_to_next
        const Fr byte_nz = fr_mul(tx_lo, lo_inv), byte_next_nz = fr_mul(n_tx_lo, zk_col(w, PI_TX_LO_INV, in));
        PI_ASSERT(fr_is_zero(tx_id) || fr_eq(id_nz, one), 6);
        PI_ASSERT(fr_is_zero(tx_lo) || fr_eq(byte_nz, one), 7);
        PI_ASSERT(fr_is_zero(d) || fr_eq(neq_next, one), 8);
        const bool id_z_z = fr_eq(id_nz, one);  // is_tx_id_zero == 0
        PI_ASSERT(id_z_z || fr_is_zero(tx_id), 9);
        PI_ASSERT(id_z_z || fr_is_zero(n_tx_id), 10);
        PI_ASSERT(id_z_z || fr_is_zero(is_final), 11);
        PI_ASSERT(id_z_z || fr_is_zero(gas_cost), 12);
        // gas cost of a byte: 16 * nonzero + 4 * (1 - nonzero) = 4 + 12 * nonzero
        const Fr gas = fr_add_u64(fr_mul_u64(byte_nz, 12), 4), gas_next = fr_add_u64(fr_mul_u64(byte_next_nz, 12), 4);
        {   // fixed u16 lookup of tx_id_not_equal_to_next * is_tx_id_next_nonzero * (tx_id_next - tx_id - 1)
            const Fr v = fr_mul(fr_mul(neq_next, id_next_nz), fr_sub(d, one));
            if (!fr_le_u64(v, 65535)) PI_FAIL(ZK_LOOKUP_UNSAT, 13);
        }
        const bool id_nz_z = fr_is_zero(id_nz);
        const bool eq_next_z = fr_eq(neq_next, one);  // tx_id_equal_to_next == 0
        PI_ASSERT(id_nz_z || eq_next_z || fr_eq(n_tx_index, fr_add(tx_index, one)), 14);
        PI_ASSERT(id_nz_z || fr_is_zero(d) || fr_is_zero(n_tx_index), 15);
        PI_ASSERT(id_nz_z || eq_next_z || fr_eq(n_gas_cost, fr_add(gas_cost, gas_next)), 16);
        PI_ASSERT(id_nz_z || fr_is_zero(id_next_nz) || fr_is_zero(d) || fr_eq(n_gas_cost, gas_next), 17);
        PI_ASSERT(id_nz_z || fr_eq(id_next_nz, one) || fr_is_zero(n_gas_cost), 18);
        PI_ASSERT(id_nz_z || eq_next_z || fr_is_zero(is_final), 19);
        PI_ASSERT(id_nz_z || fr_is_zero(d) || fr_eq(is_final, one), 20);
        const bool start_z = fr_is_zero(zk_col(w, PI_Q_TX_CALLDATA_START, i));
        PI_ASSERT(start_z || id_nz_z || fr_is_zero(tx_index), 21);
        PI_ASSERT(start_z || id_nz_z || fr_eq(gas_cost, gas), 22);
    }
    if (!fr_is_zero(zk_col(w, PI_Q_TX_TABLE, i))) {
        const Fr is_cdl = fr_sub_u64(zk_col(w, PI_TX_TAG, i), 8);  // tag - TxTag.CallDataLength
        const Fr p1 = fr_mul(is_cdl, id_inv);                      // row_is_cdl * tx_id_inv
        const Fr len_nz = fr_mul(tx_lo, lo_inv);
        PI_ASSERT(fr_is_zero(is_cdl) || fr_eq(p1, one), 23);
        PI_ASSERT(fr_is_zero(tx_lo) || fr_eq(len_nz, one), 24);
        const Fr cdl_row = fr_sub(one, p1);                        // is_calldata_length_row
        PI_ASSERT(fr_is_zero(cdl_row) || fr_eq(len_nz, one) || fr_is_zero(n_tx_lo), 25);
        const Fr cond = fr_mul(cdl_row, len_nz);
        if (code == 0u && !pi_gas_contains(a.gas, fr_mul(tx_id, cond), cond, fr_mul(n_tx_lo, cond))) PI_FAIL(ZK_LOOKUP_UNSAT, 26);
    }
    if (!fr_is_zero(zk_col(w, PI_Q_WD, i))) {
        if (!fr_is_zero(zk_col(w, PI_Q_WD, in))) PI_ASSERT(fr_eq(zk_col(w, PI_WD_ID, in), fr_add(zk_col(w, PI_WD_ID, i), one)), 27);
        PI_ASSERT(!fr_is_zero(zk_col(w, PI_WD_AMOUNT, i)), 28);
    }
    return code;
}

// ---- PI circuit copy constraints (pi_circuit.py:355-445) ----------------------------------------------------------------------------
// The reference walks `witness.copy_constrains` (byte strings cut out of the raw public inputs, big-endian) in a fixed order
// and asserts, per entry, `cell == bytes_to_fq(entry[::-1])` against a cell of the block / tx / withdrawal table or of the
// public inputs (bytes_to_fq asserts len <= MAX_N_BYTES = 31 first, util/arithmetic.py:227-229); the very first statement
// compares two words cell by cell (`rows[0].rpi_digest_word == public_inputs.pi_keccak`, :358).  One lane per constraint; the
// tally's first failing index is the reference's first failing statement.  Sites: 1 = bytes_to_fq's length assert, 2 = the
// equality.  Wire: cells[n][4], bytes[n][32] (the entry as popped, left-aligned), lens[n] (PI_COPY_CELL: the 32 bytes are a
// canonical cell, little-endian, compared as is).
#define PI_COPY_CELL 0xFFFFFFFFu
struct PiCopyArgs {
    const u64* cells;
    const uint8_t* bytes;
    const u32* lens;
    u64 n;
};
ZK_HD u32 pi_copy_check(const PiCopyArgs& a, u64 i) {
    const Fr cell = fr_load(a.cells + i * 4);
    const u32 len = a.lens[i];
    const uint8_t* b = a.bytes + i * 32;
    Fr v = fr_zero();
    if (len == PI_COPY_CELL) {
        for (int k = 0; k < 32; k++) v.v[k >> 2] |= (u32)b[k] << (8 * (k & 3));
    } else {
        if (len > 31u) return ZK_CODE(ZK_ASSERT, 1);
        // int.from_bytes(entry[::-1], "little") == the entry read big-endian; at most 31 bytes: below the modulus, no reduction
        for (u32 k = 0; k < len; k++) {
            const u32 pos = len - 1u - k;  // byte k of the entry has weight 256^(len - 1 - k)
            v.v[pos >> 2] |= (u32)b[k] << (8 * (pos & 3u));
        }
    }
    return fr_eq(cell, v) ? 0u : ZK_CODE(ZK_ASSERT, 2);
}

