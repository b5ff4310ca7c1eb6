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

// RW lookup by (rw_counter, rw, tag, id, address): the dense index when the rows carry consecutive rw_counters (the row's five key
// cells compared in registers, its value cell and type bits from the same batch of loads), else the inline open-addressing probe.
ZK_HD u32 copy_rw_lookup(const CopyArgs& a, bool dense, u64 base, const Fr& rwc, const Fr& rw, u32 tag, const Fr& id, const Fr& addr, u32& row, Fr& val_lo) {
    if (dense) {
        const u64 off = fr_lo64(rwc) - base;
        bool ok = fr_fits64(rwc) && fr_lo64(rwc) >= base && off < (u64)a.rw.n;
        row = ok ? (u32)off : 0u;
        const Fr k_rw = zk_table_cell(a.rw, row, R_RW), k_tag = zk_table_cell(a.rw, row, R_TAG), k_id = zk_table_cell(a.rw, row, R_ID);
        const Fr k_addr = zk_table_cell(a.rw, row, R_ADDR);
        val_lo = zk_table_cell(a.rw, row, R_VAL_LO);
        ok = ok & fr_eq(k_rw, rw) & fr_eq_u64(k_tag, tag) & fr_eq(k_id, id) & fr_eq(k_addr, addr);
        return ok ? 0u : (u32)ZK_LOOKUP_UNSAT;
    }
    Fr q[RW_NCELLS];
#pragma unroll
    for (int c = 0; c < RW_NCELLS; c++) q[c] = fr_zero();
    q[R_RWC] = rwc; q[R_RW] = rw; q[R_TAG] = fr_from_u64(tag); q[R_ID] = id; q[R_ADDR] = addr;
    u32 kind;
    row = table_probe_inline<RW_NCELLS, 0x1fu>(a.rw, rw_key_hash_cell(rwc), q, kind, &val_lo, R_VAL_LO);
    return kind;
}

// One row's twenty cells + its type bit.  A lane loads only its own row; what the gates need from rows i + 1 and i + 2 comes from
// the lanes that hold them (wave_shl DPP moves on the device, once and twice), or from two more loads on the host.  Round 4 read
// the neighbours' cells where a gate asked for them — behind `||` short circuits, i.e. as a chain of dependent, conditional loads:
// 32,976 rows took 43 us, the time of ~15 serial round trips, not of 21 MB.
struct CpRow {
    Fr c[CP_NCELLS];
    u32 flags;
};
ZK_HD void copy_load_row(const ZkCols& w, u64 i, CpRow& R) {
#pragma unroll
    for (int k = 0; k < CP_NCELLS; k++) R.c[k] = zk_col(w, k, i);
    R.flags = w.flags ? w.flags[i] : 0u;
}
#ifndef ZK_HOSTSIM
ZK_HD Fr cp_lane_plus1(const Fr& x) {  // the same cell in lane + 1 (every lane of the wavefront takes part)
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.v[k], 0x130, 0xf, 0xf, false);  // wave_shl:1
    return r;
}
#define CP_N1(cell) cp_lane_plus1(C.c[cell])
#define CP_N2(cell) cp_lane_plus1(cp_lane_plus1(C.c[cell]))
#else
#define CP_N1(cell) (N1.c[cell])
#define CP_N2(cell) (N2.c[cell])
#endif
#define CP_ROWS_PER_WAVE 62  // device: lanes 62 and 63 of a wavefront only hold the successors of rows 60 / 61

// Row C against its two successors (host: loaded; device: the neighbour lanes' registers, so every lane must call this).
ZK_HD u32 copy_check_loaded(const CopyArgs& a, const CpRow& C, const CpRow& N1, const CpRow& N2) {
    (void)N1; (void)N2;
    u32 code = 0;
    const Fr one = fr_from_u64(1);
    // the successors' cells, moved unconditionally with all lanes active
    const Fr n_is_last = CP_N1(CP_IS_LAST), n_rwc = CP_N1(CP_RWC), n_inc_left = CP_N1(CP_RWC_INC_LEFT), n_rlc_acc = CP_N1(CP_RLC_ACC);
    const Fr n_is_pad = CP_N1(CP_IS_PAD), nvalue = CP_N1(CP_VALUE), n_is_rlc_acc = CP_N1(CP_IS_RLC_ACC);
    const Fr nn_id_lo = CP_N2(CP_ID_LO), nn_id_hi = CP_N2(CP_ID_HI), nn_tag = CP_N2(CP_TAG), nn_addr = CP_N2(CP_ADDR);
    const Fr nn_src_end = CP_N2(CP_SRC_END), nn_bytes_left = CP_N2(CP_BYTES_LEFT), nn_value = CP_N2(CP_VALUE);
    // the dense-index verdict of the RW table: one independent load, issued with the rest
    const ZkRwMeta* m = a.rw_meta;
    const bool dense = m && m->dense;
    const u64 rw_base = m ? m->base : 0ull;
    const Fr& q_step = C.c[CP_Q_STEP]; const Fr& is_first = C.c[CP_IS_FIRST]; const Fr& is_last = C.c[CP_IS_LAST];
    const Fr& tag = C.c[CP_TAG]; const Fr& addr = C.c[CP_ADDR]; const Fr& src_end = C.c[CP_SRC_END];
    const Fr& value = C.c[CP_VALUE]; const Fr& rlc_acc = C.c[CP_RLC_ACC]; const Fr& is_pad = C.c[CP_IS_PAD];
    const Fr& rwc = C.c[CP_RWC]; const Fr& inc_left = C.c[CP_RWC_INC_LEFT];
    const Fr& is_memory = C.c[CP_IS_MEMORY]; const Fr& is_bytecode = C.c[CP_IS_BYTECODE];
    const Fr& is_tx_calldata = C.c[CP_IS_TX_CALLDATA]; const Fr& is_tx_log = C.c[CP_IS_TX_LOG];
    const Fr& is_rlc_acc = C.c[CP_IS_RLC_ACC];
    const Fr& id_lo = C.c[CP_ID_LO]; const Fr& id_hi = C.c[CP_ID_HI];
    const bool id_is_word = (C.flags & 1u) != 0u;

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
        CP_ZERO(cz, fr_eq(id_lo, nn_id_lo) && fr_eq(id_hi, nn_id_hi), 10);
        CP_ZERO(cz, fr_eq(tag, nn_tag), 11);
        CP_ZERO(cz, fr_eq(fr_add_u64(addr, 1), nn_addr), 12);
        CP_ZERO(cz, fr_eq(src_end, nn_src_end), 13);
    }
    // (1 - is_pad) * (is_memory + is_tx_log): a select when is_pad is 0 / 1 (the usual case)
    const Fr rw_diff = fr_is_zero(is_pad) ? fr_add(is_memory, is_tx_log)
                                          : (fr_eq(is_pad, one) ? fr_zero() : fr_mul(fr_sub(one, is_pad), fr_add(is_memory, is_tx_log)));
    {
        const bool cz = fr_eq(is_last, one);  // 1 - is_last == 0
        CP_ZERO(cz, fr_eq(fr_add(rwc, rw_diff), n_rwc), 14);
        CP_ZERO(cz, fr_eq(fr_sub(inc_left, rw_diff), n_inc_left), 15);
        CP_ZERO(cz, fr_eq(rlc_acc, n_rlc_acc), 16);
    }
    CP_ZERO(fr_is_zero(is_last), fr_eq(inc_left, rw_diff), 17);
    CP_ZERO(fr_is_zero(is_last) || fr_is_zero(is_rlc_acc), fr_eq(rlc_acc, value), 18);

    // ---- verify_step (:62-89) -----------------------------------------------------------------
    {
        const bool qz = fr_is_zero(q_step);
        const Fr& bytes_left = C.c[CP_BYTES_LEFT];
        CP_ZERO(qz, fr_is_zero(n_is_last) || fr_eq(bytes_left, one), 19);
        CP_ZERO(qz, fr_eq(n_is_last, one) || fr_eq(fr_sub_u64(fr_sub(bytes_left, nn_bytes_left), 1), fr_zero()), 20);
        CP_ZERO(qz, fr_is_zero(is_pad) || fr_is_zero(value), 21);
        if (fr_is_zero(is_tx_log)) {
            // lt(addr, src_addr_end, 5) asserts both operands fit 5 bytes — unconditionally (:16-20)
            CP_ASSERT(fr_byte_len(addr) <= 5 && fr_byte_len(src_end) <= 5, 22);
            const u32 lt = fr_lt(addr, src_end) ? 1u : 0u;
            CP_ZERO(qz, fr_eq_u64(is_pad, 1 - lt), 23);
        }
        CP_ZERO(qz, fr_is_zero(n_is_pad), 24);
        CP_ZERO(qz || fr_eq(n_is_rlc_acc, one), fr_eq(value, nvalue), 25);
        CP_ZERO(qz || fr_is_zero(is_first), fr_eq(value, nvalue), 26);
        const bool c27z = fr_eq(q_step, one) || fr_eq(is_last, one) || fr_is_zero(is_rlc_acc);
        if (!c27z) CP_ASSERT(fr_eq(nn_value, fr_add(fr_mul(value, a.r), nvalue)), 27);
    }
    if (code) return code;

    // ---- table lookups (:107-130) -------------------------------------------------------------
    const bool not_pad = fr_is_zero(is_pad);
    if (fr_eq(is_memory, one) && not_pad) {
        CP_ASSERT(!id_is_word, 28);  // row.id.value()
        if (code) return code;
        u32 row;
        Fr val_lo;
        const u32 k = copy_rw_lookup(a, dense, rw_base, rwc, fr_sub(one, q_step), TG_Memory, id_lo, addr, row, val_lo);
        if (k) { CP_FAIL(k, 29); return code; }
        CP_ASSERT(!(a.rw.flags ? (a.rw.flags[row] & 1u) : true), 30);  // .value.value()
        CP_ASSERT(fr_eq(val_lo, value), 31);
    }
    if (fr_eq(is_bytecode, one) && not_pad) {
        Fr q[BYTECODE_NCELLS];
        q[B_HASH_LO] = id_lo; q[B_HASH_HI] = id_hi; q[B_TAG] = fr_from_u64(2); q[B_INDEX] = addr;
        q[B_IS_CODE] = C.c[CP_IS_CODE]; q[B_VALUE] = fr_zero();
        u32 k;
        Fr bval;
        (void)table_probe_inline<BYTECODE_NCELLS, 0x1fu>(a.bytecode, bc_key_hash_cells(q[0], q[1], q[2], q[3]), q, k, &bval, B_VALUE);
        if (code == 0u && k) { CP_FAIL(k, 32); return code; }
        CP_ASSERT(fr_eq(bval, value), 34);
    }
    if (fr_eq(is_tx_calldata, one) && not_pad) {
        CP_ASSERT(!id_is_word, 35);
        if (code) return code;
        Fr q[TX_NCELLS];
        q[0] = id_lo; q[1] = fr_from_u64(TXC_CallData); q[2] = addr; q[3] = fr_zero(); q[4] = fr_zero();
        u32 k;
        Fr tval;
        const u32 row = table_probe_inline<TX_NCELLS, 0x7u>(a.tx, tx_key_hash_cells(q[0], q[1], q[2]), q, k, &tval, 3);
        if (k) { CP_FAIL(k, 36); return code; }
        CP_ASSERT(!(a.tx.flags ? (a.tx.flags[row] & 1u) : true), 37);
        CP_ASSERT(fr_eq(tval, value), 38);
    }
    if (fr_eq(is_tx_log, one)) {
        CP_ASSERT(!id_is_word, 39);
        if (code) return code;
        u32 row;
        Fr val_lo;
        const u32 k = copy_rw_lookup(a, dense, rw_base, rwc, one, TG_TxLog, id_lo, addr, row, val_lo);
        if (k) { CP_FAIL(k, 40); return code; }
        CP_ASSERT(!(a.rw.flags ? (a.rw.flags[row] & 1u) : true), 41);
        CP_ASSERT(fr_eq(val_lo, value), 42);
    }
    return code;
}
#ifdef ZK_HOSTSIM
ZK_HD u32 copy_check_row(const CopyArgs& a, u64 i) {  // host build: the three rows loaded
    const u64 n = a.rows.n;
    CpRow C, N1, N2;
    copy_load_row(a.rows, i, C);
    copy_load_row(a.rows, (i + 1) % n, N1);
    copy_load_row(a.rows, (i + 2) % n, N2);
    return copy_check_loaded(a, C, N1, N2);
}
#endif
