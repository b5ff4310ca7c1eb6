// Copy circuit: per-row constraint evaluation on a 3-row window + lookups into the RW, bytecode and
// tx tables.
//
// Reference: src/zkevm_specs/copy_circuit.py — `verify_row` :23-59, `verify_step` :62-89 (both under
// ConstraintSystem conditions, util/constraint_system.py:27-74), `lt` :16-20, and the loop with the
// table lookups `verify_copy_table` :92-130 (rows (i+1)%n and (i+2)%n wrap).
// Witness (column-major, 20 cells; CopyCircuitRow, evm_circuit/table.py:472-491): q_step, is_first,
//   is_last, id lo, id hi, tag, addr, src_addr_end, bytes_left, value, rlc_acc, is_code, is_pad,
//   rw_counter, rwc_inc_left, is_memory, is_bytecode, is_tx_calldata, is_tx_log, is_rlc_acc;
//   flags bit0 = id.is_word.  Tables use the EVM circuit's layouts (evm_circuit.hpp).
// Status = (kind << 24) | site, sites in the reference's evaluation order.
#pragma once
#include "evm_circuit.hpp"

enum { CP_Q_STEP = 0, CP_IS_FIRST, CP_IS_LAST, CP_ID_LO, CP_ID_HI, CP_TAG, CP_ADDR, CP_SRC_END, CP_BYTES_LEFT, CP_VALUE,
       CP_RLC_ACC, CP_IS_CODE, CP_IS_PAD, CP_RWC, CP_RWC_INC_LEFT, CP_IS_MEMORY, CP_IS_BYTECODE, CP_IS_TX_CALLDATA,
       CP_IS_TX_LOG, CP_IS_RLC_ACC, CP_NCELLS };

struct CopyArgs {
    ZkCols rows;
    ZkTable rw, bytecode, tx;
    const ZkRwMeta* rw_meta;
    Fr r;  // keccak randomness
};

#define CP_FAIL(kind, site) code = (code == 0u) ? ZK_CODE(kind, site) : code
#define CP_ASSERT(cond, site) code = (code == 0u && !(cond)) ? ZK_CODE(ZK_ASSERT, site) : code
// cond * x == 0 over a prime field  <=>  cond == 0 or x == 0
#define CP_ZERO(cond_zero, x_zero, site) CP_ASSERT((cond_zero) || (x_zero), site)

// "exactly one distinct matching row" lookup with the given query cells (generic index)
template <int NCELLS>
ZK_HD u32 copy_table_lookup(const ZkTable& t, u64 h, const Fr (&q)[NCELLS], u32 mask, u32& row) {
    Fr tmp[NCELLS];
    for (int c = 0; c < NCELLS; c++) tmp[c] = ((mask >> c) & 1u) ? q[c] : fr_zero();
    const u64 res = table_probe_generic(t, h, tmp, mask);
    row = (u32)res;
    return (u32)(res >> 32);  // 0 / ZK_LOOKUP_UNSAT / ZK_LOOKUP_AMBIGUOUS
}
ZK_HD u32 copy_rw_lookup(const CopyArgs& a, const Fr& rwc, const Fr& rw, u32 tag, const Fr& id, const Fr& addr, u32& row) {
    Fr q[RW_NCELLS];
    for (int c = 0; c < RW_NCELLS; c++) q[c] = fr_zero();
    q[R_RWC] = rwc; q[R_RW] = rw; q[R_TAG] = fr_from_u64(tag); q[R_ID] = id; q[R_ADDR] = addr;
    const u32 mask = 0x1fu;
    const ZkRwMeta* m = a.rw_meta;
    if (m && m->dense) {
        const u64 off = fr_lo64(rwc) - m->base;
        bool ok = fr_fits64(rwc) && fr_lo64(rwc) >= m->base && off < (u64)a.rw.n;
        row = ok ? (u32)off : 0u;
        for (int c = 1; c < 5; c++) ok = ok & fr_eq(zk_table_cell(a.rw, row, c), q[c]);
        return ok ? 0u : (u32)ZK_LOOKUP_UNSAT;
    }
    return copy_table_lookup<RW_NCELLS>(a.rw, rw_key_hash_cell(rwc), q, mask, row);
}

ZK_HD u32 copy_check_row(const CopyArgs& a, u64 i) {
    const ZkCols& w = a.rows;
    const u64 n = w.n, i1 = i + 1 >= n ? i + 1 - n : i + 1, i2 = i + 2 >= n ? (i + 2 - n) % n : i + 2;
    u32 code = 0;
    const Fr one = fr_from_u64(1);
    const Fr q_step = zk_col(w, CP_Q_STEP, i), is_first = zk_col(w, CP_IS_FIRST, i), is_last = zk_col(w, CP_IS_LAST, i);
    const Fr tag = zk_col(w, CP_TAG, i), addr = zk_col(w, CP_ADDR, i), src_end = zk_col(w, CP_SRC_END, i);
    const Fr value = zk_col(w, CP_VALUE, i), rlc_acc = zk_col(w, CP_RLC_ACC, i), is_pad = zk_col(w, CP_IS_PAD, i);
    const Fr rwc = zk_col(w, CP_RWC, i), inc_left = zk_col(w, CP_RWC_INC_LEFT, i);
    const Fr is_memory = zk_col(w, CP_IS_MEMORY, i), is_bytecode = zk_col(w, CP_IS_BYTECODE, i);
    const Fr is_tx_calldata = zk_col(w, CP_IS_TX_CALLDATA, i), is_tx_log = zk_col(w, CP_IS_TX_LOG, i);
    const Fr is_rlc_acc = zk_col(w, CP_IS_RLC_ACC, i);
    const Fr id_lo = zk_col(w, CP_ID_LO, i), id_hi = zk_col(w, CP_ID_HI, i);
    const bool id_is_word = w.flags ? (w.flags[i] & 1u) : false;
    const Fr n_is_last = zk_col(w, CP_IS_LAST, i1);

    // ---- verify_row (:23-59) ------------------------------------------------------------------
    CP_ASSERT(fr_le_u64(is_first, 1), 1);
    CP_ASSERT(fr_le_u64(is_last, 1), 2);
    CP_ZERO(fr_eq(q_step, one), fr_is_zero(is_first), 3);   // (1 - q_step) * is_first
    CP_ZERO(fr_is_zero(q_step), fr_is_zero(is_last), 4);    // q_step * is_last
    CP_ASSERT(fr_eq_u64(is_memory, fr_eq_u64(tag, 2) ? 1 : 0), 5);
    CP_ASSERT(fr_eq_u64(is_bytecode, fr_eq_u64(tag, 1) ? 1 : 0), 6);
    CP_ASSERT(fr_eq_u64(is_tx_calldata, fr_eq_u64(tag, 3) ? 1 : 0), 7);
    CP_ASSERT(fr_eq_u64(is_tx_log, fr_eq_u64(tag, 4) ? 1 : 0), 8);
    CP_ASSERT(fr_eq_u64(is_rlc_acc, fr_eq_u64(tag, 5) ? 1 : 0), 9);
    {
        const bool cz = fr_eq(fr_add(is_last, n_is_last), one);  // 1 - (is_last + next.is_last) == 0
        CP_ZERO(cz, fr_eq(id_lo, zk_col(w, CP_ID_LO, i2)) && fr_eq(id_hi, zk_col(w, CP_ID_HI, i2)), 10);
        CP_ZERO(cz, fr_eq(tag, zk_col(w, CP_TAG, i2)), 11);
        CP_ZERO(cz, fr_eq(fr_add_u64(addr, 1), zk_col(w, CP_ADDR, i2)), 12);
        CP_ZERO(cz, fr_eq(src_end, zk_col(w, CP_SRC_END, i2)), 13);
    }
    // (1 - is_pad) * (is_memory + is_tx_log): a select when is_pad is 0 / 1 (the usual case)
    const Fr rw_diff = fr_is_zero(is_pad) ? fr_add(is_memory, is_tx_log)
                                          : (fr_eq(is_pad, one) ? fr_zero() : fr_mul(fr_sub(one, is_pad), fr_add(is_memory, is_tx_log)));
    {
        const bool cz = fr_eq(is_last, one);  // 1 - is_last == 0
        CP_ZERO(cz, fr_eq(fr_add(rwc, rw_diff), zk_col(w, CP_RWC, i1)), 14);
        CP_ZERO(cz, fr_eq(fr_sub(inc_left, rw_diff), zk_col(w, CP_RWC_INC_LEFT, i1)), 15);
        CP_ZERO(cz, fr_eq(rlc_acc, zk_col(w, CP_RLC_ACC, i1)), 16);
    }
    CP_ZERO(fr_is_zero(is_last), fr_eq(inc_left, rw_diff), 17);
    CP_ZERO(fr_is_zero(is_last) || fr_is_zero(is_rlc_acc), fr_eq(rlc_acc, value), 18);

    // ---- verify_step (:62-89) -----------------------------------------------------------------
    {
        const bool qz = fr_is_zero(q_step);
        const Fr bytes_left = zk_col(w, CP_BYTES_LEFT, i);
        CP_ZERO(qz, fr_is_zero(n_is_last) || fr_eq(bytes_left, one), 19);
        CP_ZERO(qz, fr_eq(n_is_last, one) || fr_eq(fr_sub_u64(fr_sub(bytes_left, zk_col(w, CP_BYTES_LEFT, i2)), 1), fr_zero()), 20);
        CP_ZERO(qz, fr_is_zero(is_pad) || fr_is_zero(value), 21);
        if (fr_is_zero(is_tx_log)) {
            // lt(addr, src_addr_end, 5) asserts both operands fit 5 bytes — unconditionally (:16-20)
            CP_ASSERT(fr_byte_len(addr) <= 5 && fr_byte_len(src_end) <= 5, 22);
            const u32 lt = fr_lt(addr, src_end) ? 1u : 0u;
            CP_ZERO(qz, fr_eq_u64(is_pad, 1 - lt), 23);
        }
        CP_ZERO(qz, fr_is_zero(zk_col(w, CP_IS_PAD, i1)), 24);
        const Fr nvalue = zk_col(w, CP_VALUE, i1);
        CP_ZERO(qz || fr_eq(zk_col(w, CP_IS_RLC_ACC, i1), one), fr_eq(value, nvalue), 25);
        CP_ZERO(qz || fr_is_zero(is_first), fr_eq(value, nvalue), 26);
        const bool c27z = fr_eq(q_step, one) || fr_eq(is_last, one) || fr_is_zero(is_rlc_acc);
        if (!c27z) CP_ASSERT(fr_eq(zk_col(w, CP_VALUE, i2), fr_add(fr_mul(value, a.r), nvalue)), 27);
    }
    if (code) return code;

    // ---- table lookups (:107-130) -------------------------------------------------------------
    const bool not_pad = fr_is_zero(is_pad);
    if (fr_eq(is_memory, one) && not_pad) {
        CP_ASSERT(!id_is_word, 28);  // row.id.value()
        if (code) return code;
        u32 row;
        const u32 k = copy_rw_lookup(a, rwc, fr_sub(one, q_step), TG_Memory, id_lo, addr, row);
        if (k) { CP_FAIL(k, 29); return code; }
        CP_ASSERT(!(a.rw.flags ? (a.rw.flags[row] & 1u) : true), 30);  // .value.value()
        CP_ASSERT(fr_eq(zk_table_cell(a.rw, row, R_VAL_LO), value), 31);
    }
    if (fr_eq(is_bytecode, one) && not_pad) {
        Fr q[BYTECODE_NCELLS];
        q[B_HASH_LO] = id_lo; q[B_HASH_HI] = id_hi; q[B_TAG] = fr_from_u64(2); q[B_INDEX] = addr;
        q[B_IS_CODE] = zk_col(w, CP_IS_CODE, i); q[B_VALUE] = fr_zero();
        u32 row;
        const u32 k = copy_table_lookup<BYTECODE_NCELLS>(a.bytecode, bc_key_hash_cells(q[0], q[1], q[2], q[3]), q, 0x1fu, row);
        if (code == 0u && k) { CP_FAIL(k, 32); return code; }
        CP_ASSERT(fr_eq(zk_table_cell(a.bytecode, row, B_VALUE), value), 34);
    }
    if (fr_eq(is_tx_calldata, one) && not_pad) {
        CP_ASSERT(!id_is_word, 35);
        if (code) return code;
        Fr q[TX_NCELLS];
        q[0] = id_lo; q[1] = fr_from_u64(TXC_CallData); q[2] = addr; q[3] = fr_zero(); q[4] = fr_zero();
        u32 row;
        const u32 k = copy_table_lookup<TX_NCELLS>(a.tx, tx_key_hash_cells(q[0], q[1], q[2]), q, 0x7u, row);
        if (k) { CP_FAIL(k, 36); return code; }
        CP_ASSERT(!(a.tx.flags ? (a.tx.flags[row] & 1u) : true), 37);
        CP_ASSERT(fr_eq(zk_table_cell(a.tx, row, 3), value), 38);
    }
    if (fr_eq(is_tx_log, one)) {
        CP_ASSERT(!id_is_word, 39);
        if (code) return code;
        u32 row;
        const u32 k = copy_rw_lookup(a, rwc, one, TG_TxLog, id_lo, addr, row);
        if (k) { CP_FAIL(k, 40); return code; }
        CP_ASSERT(!(a.rw.flags ? (a.rw.flags[row] & 1u) : true), 41);
        CP_ASSERT(fr_eq(zk_table_cell(a.rw, row, R_VAL_LO), value), 42);
    }
    return code;
}
