// Copy-circuit witness assignment on the device (the remaining piece).
//
// Replaces the reference's `CopyCircuit.copy(r, rw_dict, src_id, src_tag, dst_id, dst_tag, src_addr, src_addr_end, dst_addr,
// copy_length, src_data, log_id)` (src/zkevm_specs/evm_circuit/typing.py:1010-1091; `_append_row` :1093-1151): one copy
// event becomes 2 x copy_length circuit rows (a read row and a write row per byte), the memory reads / writes and tx-log
// writes it implies are appended to the RW table (:1122-1131 -> RWDictionary.memory_read / memory_write / tx_log_write,
// :482-492, :532-556), and `Tables._convert_copy_circuit_to_table` (evm_circuit/table.py:627-651) derives one copy-TABLE row
// per event from its first two rows.  Outputs: the 20-cell circuit rows column-major + type bits (exactly what zk_copy_open
// takes), the 14-cell copy-table rows (what zk_evm_tables.copy takes) and the 14-cell RW rows in rw_counter order.
//
// The reference walks an event byte by byte, threading two pieces of state: the RWDictionary's rw_counter and, for RlcAcc
// destinations (SHA3 inputs, RETURN data of a CREATE), the running `rlc_acc = rlc_acc * r + value`.  Here every output row
// is computed on its own lane:
//   * rw_counter of a row is closed-form: reads so far (only memory sources, only below src_addr_end) + writes so far (only
//     memory / tx-log destinations), so `rwc_inc_left` needs no back-patching pass;
//   * the Horner recurrence is cut into 64-byte chunks: cpa_chunk (the chunk's Horner value from 0, as ONE lazy dot product with the
//     Montgomery powers of r: cpa_dot) -> cpa_prefix (one lane per event: incoming value per chunk, event total); the per-byte
//     running value of a row is incoming * r^(t + 1) + the dot product of the chunk's first t + 1 bytes, computed by the row's own lane;
//   * cpa_write_row: one lane per OUTPUT row, coalesced 32 B/lane stores of the 20 cells; the lane of an event's first row
//     also writes its copy-table row, the lanes of rows that touch the RW table write their RW row.
#pragma once
#include "common.hpp"

enum { CPA_EV_NCELLS = 12, CPA_ROW_NCELLS = 20, CPA_TABLE_NCELLS = 14, CPA_RW_NCELLS = 14, CPA_CHUNK = 64, CPA_RPOW_ROWS = CPA_CHUNK + 1 };
static_assert(CPA_CHUNK <= 64, "the per-chunk integer accumulators (nine 32-bit limbs + carries) are sized for at most 64 byte terms");
enum { CPA_BYTECODE = 1, CPA_MEMORY = 2, CPA_TX_CALLDATA = 3, CPA_TX_LOG = 4, CPA_RLC_ACC = 5 };  // CopyDataTypeTag (table.py:308-315)
enum { CPA_TARGET_MEMORY = 9, CPA_TARGET_TX_LOG = 10, CPA_TX_LOG_DATA = 3 };                     // Target, TxLogFieldTag.Data
#define CPA_NONE 0xffffffffu

// Per-event plumbing, computed on the host from the event cells (integers only; the 256-bit ids stay in the event cells)
struct CpaEvent {
    u64 src_addr, src_end, dst_addr, length, log_id, rwc;
    u64 row0;    // first output row of the event
    u64 rw0;     // first RW row of the event
    u64 data0;   // first source byte of the event in `data`
    u64 rlc0;    // first per-byte running value of the event in `rlc` (RlcAcc destinations)
    u32 table_idx;  // its copy-table row, CPA_NONE when the event copies nothing
    u32 src_tag, dst_tag, flags;  // flags: bit0 src_id is a Word, bit1 dst_id is a Word
    u32 chunk0, n_chunks;         // its Horner chunks (RlcAcc destinations)
};
struct CpaChunk {
    u32 event;
    u32 start;  // first byte index inside the event
    u32 count;
    u32 pad;
};
struct CpaArgs {
    const u64* events;      // [n_events][12][4]
    const CpaEvent* ev;     // [n_events]
    const u64* row0;        // [n_events + 1]: first output row per event (binary-searched by the row lanes)
    u64 n_events, n_rows;
    const uint16_t* data;   // value | is_code << 8
    const u64* rpow;        // [65][4] r^m in Montgomery form
    const CpaChunk* chunks;
    u64 n_chunks;
    u64* chunk_acc;         // [n_chunks][4]
    u64* chunk_in;          // [n_chunks][4]
    u64* ev_rlc;            // [n_events][4] final rlc_acc of the event (0 for other destinations)
    u64* rlc;               // (unused since round 5: the row lanes compute the running values themselves)
    u64* rows;              // out [20][n_rows][4]
    u32* row_flags;         // out [n_rows]
    u64* table;             // out [n_table][14][4]
    u64* rw;                // out [n_rw][14][4]
    u32* rw_flags;          // out [n_rw]
};

ZK_HD void cpa_store(u64* out, const Fr& x) {
    for (int j = 0; j < 4; j++) out[j] = (u64)x.v[2 * j] | ((u64)x.v[2 * j + 1] << 32);
}
ZK_HD void cpa_store_u64(u64* out, u64 v) { out[0] = v; out[1] = 0; out[2] = 0; out[3] = 0; }
ZK_HD void cpa_fill_rpow(const Fr& r, u64* out) {  // single lane, once per session
    const Fr rM = fr_to_mont(r);
    Fr acc = frm_one();
    for (int m = 0; m < CPA_RPOW_ROWS; m++) {
        cpa_store(out + 4 * m, acc);
        acc = fr_mont(acc, rM);
    }
}
// bytes actually read: i < length with src_addr + i < src_addr_end
ZK_HD u64 cpa_n_real(const CpaEvent& e) {
    if (e.src_end <= e.src_addr) return 0;
    const u64 room = e.src_end - e.src_addr;
    return room < e.length ? room : e.length;
}
ZK_HD u32 cpa_value(const CpaArgs& a, const CpaEvent& e, u64 i, u64 n_real) { return i < n_real ? (u32)(a.data[e.data0 + i] & 0xffu) : 0u; }

// sum over s < upto of value(start + s) * r^(upto - 1 - s), canonical: the Horner value of `upto` bytes WITHOUT the dependent chain of
// `upto` Montgomery products — the bytes times the Montgomery powers of r accumulate as plain integers (each term < 2^262, 64 of
// them < 2^268: nine 32-bit limbs, 8 multiply-adds per byte, independent loads), reduced once and taken out of Montgomery form
// with one product.  Round 4's per-chunk loops did `acc = acc * r + byte` with a dependent byte load per step: 64 load + product
// round trips = 49 us for a chunk launch whatever its size.
ZK_HD Fr cpa_dot(const CpaArgs& a, const CpaEvent& e, u64 start, u32 upto, u64 n_real) {
    u32 acc[9];
#pragma unroll
    for (int j = 0; j < 9; j++) acc[j] = 0;
#pragma unroll 8
    for (u32 s = 0; s < upto; s++) {
        const u32 v = cpa_value(a, e, start + s, n_real);
        const Fr pw = fr_load(a.rpow + 4 * (u64)(upto - 1u - s));
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (u64)acc[j] + (u64)pw.v[j] * v;
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
    for (int it = 0; it < 5; it++) {  // lo < 2^256 < 6 p
        Fr t;
        const u32 bw = u256_sub(t, lo, p);
        lo = bw ? lo : t;
    }
    // + acc[8] * 2^256 (mod p): 2^256 mod p is the Montgomery one; the sum is the Montgomery form of the Horner value
    const Fr xM = fr_add(lo, fr_mul(fr_from_u64(acc[8]), frm_one()));
    return fr_mont(xM, fr_from_u64(1));
}
ZK_HD void cpa_chunk(const CpaArgs& a, u64 c) {
    const CpaChunk ch = a.chunks[c];
    const CpaEvent& e = a.ev[ch.event];
    cpa_store(a.chunk_acc + 4 * c, cpa_dot(a, e, (u64)ch.start, ch.count, cpa_n_real(e)));
}
ZK_HD void cpa_prefix_event(const CpaArgs& a, u64 j) {
    const CpaEvent& e = a.ev[j];
    Fr running = fr_zero();
    for (u32 c = e.chunk0; c < e.chunk0 + e.n_chunks; c++) {
        cpa_store(a.chunk_in + 4 * (u64)c, running);
        running = fr_add(fr_mont(running, fr_load(a.rpow + 4 * a.chunks[c].count)), fr_load(a.chunk_acc + 4 * (u64)c));
    }
    cpa_store(a.ev_rlc + 4 * j, running);
}
// running value after byte i of an RlcAcc event: (value entering the byte's chunk) * r^(t + 1) + the Horner value of bytes 0..t of
// the chunk — independent of every other output, so the row lanes compute it where they need it (no per-byte array, no launch
// of its own: round 5 first had one lane per running value in a kernel between the prefix and the rows)
ZK_HD Fr cpa_rlc_value(const CpaArgs& a, const CpaEvent& e, u64 i, u64 n_real) {
    const u64 c = (u64)e.chunk0 + i / CPA_CHUNK;
    const u32 t = (u32)(i % CPA_CHUNK);
    const Fr head = fr_mont(fr_load(a.chunk_in + 4 * c), fr_load(a.rpow + 4 * (u64)(t + 1u)));  // canonical x Montgomery power -> canonical
    return fr_add(head, cpa_dot(a, e, i - t, t + 1u, n_real));
}
// Output row j (CopyCircuitRow, table.py:472-491: q_step, is_first, is_last, id lo, hi, tag, addr, src_addr_end, bytes_left,
// value, rlc_acc, is_code, is_pad, rw_counter, rwc_inc_left, is_memory, is_bytecode, is_tx_calldata, is_tx_log, is_rlc_acc)
ZK_HD void cpa_write_row(const CpaArgs& a, u64 j) {
    // event of row j: the last one whose first row is <= j
    u64 lo = 0, hi = a.n_events;
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (a.row0[mid] <= j) lo = mid; else hi = mid;
    }
    const u64 ei = lo;
    const CpaEvent e = a.ev[ei];
    const u64 local = j - e.row0, i = local >> 1;
    const bool is_write = local & 1u;
    const u64 n_real = cpa_n_real(e);
    const bool src_mem = e.src_tag == CPA_MEMORY, dst_mem = e.dst_tag == CPA_MEMORY, dst_log = e.dst_tag == CPA_TX_LOG;
    const bool dst_rw = dst_mem || dst_log, dst_rlc = e.dst_tag == CPA_RLC_ACC;
    const bool is_pad = i >= n_real;
    const u32 d = is_pad ? 0u : (u32)a.data[e.data0 + i];
    const u32 value = d & 0xffu;
    const u32 is_code = (e.src_tag == CPA_BYTECODE || e.dst_tag == CPA_BYTECODE) ? ((d >> 8) & 1u) : 0u;
    const u64 reads_before = src_mem ? (i < n_real ? i : n_real) : 0;
    const u64 reads_incl = src_mem ? (i + 1 < n_real ? i + 1 : n_real) : 0;
    const u64 writes_before = dst_rw ? i : 0;
    const u64 total = (src_mem ? n_real : 0) + (dst_rw ? e.length : 0);
    const u64 off = (is_write ? reads_incl : reads_before) + writes_before;
    const u64 rwc = e.rwc + off;
    const u64* ev_cells = a.events + ei * CPA_EV_NCELLS * 4;
    const Fr id_lo = fr_load(ev_cells + (is_write ? 3 : 0) * 4), id_hi = fr_load(ev_cells + (is_write ? 4 : 1) * 4);
    const u32 tag = is_write ? e.dst_tag : e.src_tag;
    const Fr rlc_acc = dst_rlc ? fr_load(a.ev_rlc + 4 * ei) : fr_zero();
    u64 addr = is_write ? e.dst_addr + i : e.src_addr + i;
    if (is_write && dst_log) addr += ((u64)CPA_TX_LOG_DATA << 32) + (e.log_id << 48);
    const Fr wvalue = (is_write && dst_rlc) ? cpa_rlc_value(a, e, i, n_real) : fr_from_u64(value);
    const u64 n = a.n_rows;
#define CPA_OUT(c) (a.rows + ((u64)(c) * n + j) * 4)
    cpa_store_u64(CPA_OUT(0), is_write ? 0 : 1);
    cpa_store_u64(CPA_OUT(1), (!is_write && i == 0) ? 1 : 0);
    cpa_store_u64(CPA_OUT(2), (is_write && i == e.length - 1) ? 1 : 0);
    cpa_store(CPA_OUT(3), id_lo);
    cpa_store(CPA_OUT(4), id_hi);
    cpa_store_u64(CPA_OUT(5), tag);
    cpa_store_u64(CPA_OUT(6), addr);
    cpa_store_u64(CPA_OUT(7), is_write ? 0 : e.src_end);
    cpa_store_u64(CPA_OUT(8), is_write ? 0 : e.length - i);
    cpa_store(CPA_OUT(9), wvalue);
    cpa_store(CPA_OUT(10), rlc_acc);
    cpa_store_u64(CPA_OUT(11), is_code);
    cpa_store_u64(CPA_OUT(12), (!is_write && is_pad) ? 1 : 0);
    cpa_store_u64(CPA_OUT(13), rwc);
    cpa_store_u64(CPA_OUT(14), total - off);
    cpa_store_u64(CPA_OUT(15), tag == CPA_MEMORY);
    cpa_store_u64(CPA_OUT(16), tag == CPA_BYTECODE);
    cpa_store_u64(CPA_OUT(17), tag == CPA_TX_CALLDATA);
    cpa_store_u64(CPA_OUT(18), tag == CPA_TX_LOG);
    cpa_store_u64(CPA_OUT(19), tag == CPA_RLC_ACC);
#undef CPA_OUT
    a.row_flags[j] = is_write ? ((e.flags >> 1) & 1u) : (e.flags & 1u);
    // RW row of this copy row (memory_read below src_addr_end / memory_write / tx_log_write): value FQ, value_prev Word(0)
    if ((!is_write && src_mem && !is_pad) || (is_write && dst_rw)) {
        u64* w = a.rw + (e.rw0 + off) * CPA_RW_NCELLS * 4;
        cpa_store_u64(w + 0, rwc);
        cpa_store_u64(w + 4, is_write ? 1 : 0);
        cpa_store_u64(w + 8, (is_write && dst_log) ? CPA_TARGET_TX_LOG : CPA_TARGET_MEMORY);
        cpa_store(w + 12, id_lo);
        cpa_store_u64(w + 16, addr);
        for (int c = 5; c < 8; c++) cpa_store_u64(w + 4 * c, 0);
        cpa_store(w + 32, wvalue);
        for (int c = 9; c < CPA_RW_NCELLS; c++) cpa_store_u64(w + 4 * c, 0);
        a.rw_flags[e.rw0 + off] = 2u;
    }
    // copy-table row of the event (_convert_copy_circuit_to_table: the is_first row and the row after it)
    if (!is_write && i == 0) {
        u64* t = a.table + (u64)e.table_idx * CPA_TABLE_NCELLS * 4;
        u64 dst_addr0 = e.dst_addr;
        if (dst_log) dst_addr0 += ((u64)CPA_TX_LOG_DATA << 32) + (e.log_id << 48);
        cpa_store_u64(t + 0, 1);
        cpa_store(t + 4, id_lo);
        cpa_store(t + 8, id_hi);
        cpa_store_u64(t + 12, e.src_tag);
        cpa_store(t + 16, fr_load(ev_cells + 3 * 4));
        cpa_store(t + 20, fr_load(ev_cells + 4 * 4));
        cpa_store_u64(t + 24, e.dst_tag);
        cpa_store_u64(t + 28, e.src_addr);
        cpa_store_u64(t + 32, e.src_end);
        cpa_store_u64(t + 36, dst_addr0);
        cpa_store_u64(t + 40, e.length);
        cpa_store(t + 44, rlc_acc);
        cpa_store_u64(t + 48, e.rwc);
        cpa_store_u64(t + 52, total);
    }
}
