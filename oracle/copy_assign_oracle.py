"""Copy-circuit witness assignment — CPU restatement (TEST INFRASTRUCTURE, like everything under oracle/).

`CopyCircuit.copy(r, rw_dict, src_id, src_tag, dst_id, dst_tag, src_addr, src_addr_end, dst_addr, copy_length, src_data,
log_id)` of the reference (src/zkevm_specs/evm_circuit/typing.py:1010-1091, `_append_row` :1093-1151) expands one copy
event into 2 x copy_length circuit rows (a read row and a write row per byte), appends the memory reads / writes and tx-log
writes it implies to the RWDictionary (:1122-1131, :482-492, :532-556), and `Tables._convert_copy_circuit_to_table`
(evm_circuit/table.py:627-651) derives one copy-TABLE row per event from its first two rows.  This module restates all
three on Python ints over the event wire format of `zk_copy_assign` (include/zkevm_hip.h):

    events  [n][12]  src_id lo, hi, src_tag, dst_id lo, hi, dst_tag, src_addr, src_addr_end, dst_addr, length, log_id,
                     rw_counter (rw_dict.rw_counter when copy() is called)
    flags   [n]      bit0 src_id is a Word, bit1 dst_id is a Word
    data             per event the bytes read below src_addr_end: value | is_code << 8, events back to back (offsets[n + 1])

Pinned by tests/golden/copy_assign_cases.npz: every `CopyCircuit.copy` call of the reference's own tests with the rows,
RW rows and copy-table rows the unmodified reference produced (oracle/gen_golden_copy_assign.py).
"""
from .wire import P

BYTECODE, MEMORY, TX_CALLDATA, TX_LOG, RLC_ACC = 1, 2, 3, 4, 5  # CopyDataTypeTag (table.py:308-315)
TARGET_MEMORY, TARGET_TX_LOG = 9, 10                            # Target (table.py:184-216)
TX_LOG_DATA = 3                                                  # TxLogFieldTag.Data (table.py:275-287)
(E_SRC_LO, E_SRC_HI, E_SRC_TAG, E_DST_LO, E_DST_HI, E_DST_TAG, E_SRC_ADDR, E_SRC_END, E_DST_ADDR, E_LEN, E_LOG_ID, E_RWC) = range(12)


def n_real(ev):
    """bytes actually read: i < length with src_addr + i < src_addr_end"""
    return max(0, min(ev[E_LEN], ev[E_SRC_END] - ev[E_SRC_ADDR]))


def assign(events, flags, data, offsets, r):
    """-> (rows [N][20], row_flags [N], copy_table [m][14], rw_rows [k][14], rw_flags [k]) as lists of ints"""
    rows, row_flags, table, rw_rows, rw_flags = [], [], [], [], []
    for e, (ev, fl) in enumerate(zip(events, flags)):
        src_tag, dst_tag, length = ev[E_SRC_TAG], ev[E_DST_TAG], ev[E_LEN]
        rw_counter = ev[E_RWC]
        first = len(rows)
        rlc = 0
        base = int(offsets[e])
        for i in range(length):
            if ev[E_SRC_ADDR] + i < ev[E_SRC_END]:
                is_pad = 0
                d = int(data[base + i])
                value, is_code = d & 0xFF, (d >> 8) & 1
                if not (src_tag == BYTECODE or dst_tag == BYTECODE):
                    is_code = 0
            else:
                is_pad, value, is_code = 1, 0, 0
            # read row (_append_row with is_write = False)
            addr = ev[E_SRC_ADDR] + i
            rwc = rw_counter
            if src_tag == MEMORY and not is_pad:
                rw_rows.append([rw_counter, 0, TARGET_MEMORY, ev[E_SRC_LO], addr, 0, 0, 0, value, 0, 0, 0, 0, 0])
                rw_flags.append(2)  # value: FQ (not a word); value_prev: Word(0)
                rw_counter += 1
            assert src_tag != TX_LOG, "TxLog is write-only (typing.py:1128)"
            rows.append([1, 1 if i == 0 else 0, 0, ev[E_SRC_LO], ev[E_SRC_HI], src_tag, addr % P, ev[E_SRC_END] % P, (length - i) % P, value, 0,
                         is_code, is_pad, rwc % P, 0, int(src_tag == MEMORY), int(src_tag == BYTECODE), int(src_tag == TX_CALLDATA),
                         int(src_tag == TX_LOG), int(src_tag == RLC_ACC)])
            row_flags.append(fl & 1)
            # write row
            if dst_tag == RLC_ACC:
                rlc = (rlc * r + value) % P
            addr = ev[E_DST_ADDR] + i
            wvalue = rlc if dst_tag == RLC_ACC else value
            rwc = rw_counter
            if dst_tag == MEMORY:
                rw_rows.append([rw_counter, 1, TARGET_MEMORY, ev[E_DST_LO], addr, 0, 0, 0, wvalue, 0, 0, 0, 0, 0])
                rw_flags.append(2)
                rw_counter += 1
            elif dst_tag == TX_LOG:
                addr += (TX_LOG_DATA << 32) + (ev[E_LOG_ID] << 48)
                rw_rows.append([rw_counter, 1, TARGET_TX_LOG, ev[E_DST_LO], addr, 0, 0, 0, wvalue, 0, 0, 0, 0, 0])
                rw_flags.append(2)
                rw_counter += 1
            rows.append([0, 0, 1 if i == length - 1 else 0, ev[E_DST_LO], ev[E_DST_HI], dst_tag, addr % P, 0, 0, wvalue, 0, is_code, 0,
                         rwc % P, 0, int(dst_tag == MEMORY), int(dst_tag == BYTECODE), int(dst_tag == TX_CALLDATA),
                         int(dst_tag == TX_LOG), int(dst_tag == RLC_ACC)])
            row_flags.append((fl >> 1) & 1)
        # rwc_inc_left (and rlc_acc for RlcAcc destinations) once the event's final rw_counter is known (typing.py:1082-1089)
        for row in rows[first:]:
            row[14] = (rw_counter - row[13]) % P
            if dst_tag == RLC_ACC:
                row[10] = rlc
        if length > 0:  # _convert_copy_circuit_to_table: the is_first row and the row after it
            a, b = rows[first], rows[first + 1]
            table.append([a[1], a[3], a[4], a[5], b[3], b[4], b[5], a[6], a[7], b[6], a[8], a[10], a[13], a[14]])
    return rows, row_flags, table, rw_rows, rw_flags
