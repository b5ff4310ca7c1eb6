"""Host-side mirror of `zkevm_specs.copy_circuit.verify_copy_table(copy_circuit, tables, r)`
(copy_circuit.py:92-130), evaluated on the MI355X (`zk_copy_verify`).  Like the reference, the first failing row's
exception propagates (the loop has no try/except)."""
from . import oneshot
from .errors import raise_for_code
from .flatten import _n, flatten_bytecode_table, flatten_copy_rows, flatten_rw_table, flatten_tx_table


def verify_copy_table(copy_circuit, tables, r):
    rows = list(copy_circuit.table())
    if not rows:
        return None
    cols, flags = flatten_copy_rows(rows)
    rw, rw_flags = flatten_rw_table(tables.rw_table)
    tx, tx_flags = flatten_tx_table(tables.tx_table)
    res, _ = oneshot.copy_verify(cols, flags, _n(r), rw, rw_flags, flatten_bytecode_table(tables.bytecode_table), tx, tx_flags)
    raise_for_code(res.first_fail_code, f"Copy circuit row {res.first_fail_row}")
    return res


class CopyEvents:
    """Builder for the input of the device-side `CopyCircuit.copy` (zk_copy_assign): `copy()` takes the reference's arguments
    (evm_circuit/typing.py:1010-1023) with the RWDictionary replaced by its current rw_counter, records one event and
    returns the rw_counter after it; `assign()` expands all recorded events on the GPU into the circuit rows, their type
    bits, the copy-table rows and the RW rows (wire arrays, ready for verify / engine.open_copy / the EVM circuit's tables)."""

    def __init__(self, r):
        self.r = _n(r)
        self.events, self.flags, self.data, self.offsets = [], [], [], [0]

    def copy(self, rw_counter, src_id, src_tag, dst_id, dst_tag, src_addr, src_addr_end, dst_addr, copy_length, src_data, log_id=0):
        def ident(x):
            if hasattr(x, "lo"):
                return _n(x.lo), _n(x.hi), 1
            return _n(x), 0, 0

        s_lo, s_hi, s_w = ident(src_id)
        d_lo, d_hi, d_w = ident(dst_id)
        src_tag, dst_tag, length = int(src_tag), int(dst_tag), int(copy_length)
        n_real = 0
        for i in range(length):
            if int(src_addr) + i < int(src_addr_end):
                assert int(src_addr) + i in src_data, f"Cannot find data at the offset {int(src_addr) + i}"  # typing.py:1030
                v = src_data[int(src_addr) + i]
                v, c = (v if (src_tag == 1 or dst_tag == 1) else (v, 0))
                v, c = _n(v), _n(c)
                if not (0 <= v < 256 and c in (0, 1)):
                    raise ValueError("copy source data outside the wire's domain (bytes, is_code in {0, 1})")
                self.data.append(v | (c << 8))
                n_real += 1
        self.offsets.append(len(self.data))
        self.events.append([s_lo, s_hi, src_tag, d_lo, d_hi, dst_tag, int(src_addr), int(src_addr_end), int(dst_addr), length, int(log_id),
                            int(rw_counter)])
        self.flags.append(s_w | (d_w << 1))
        return int(rw_counter) + (n_real if src_tag == 2 else 0) + (length if dst_tag in (2, 4) else 0)

    def wire(self):
        import numpy as np

        from .wire import rows_to_rowmajor

        return (rows_to_rowmajor(self.events, 12), np.array(self.flags, dtype=np.uint32), np.array(self.data, dtype=np.uint16),
                np.array(self.offsets, dtype=np.uint64))

    def assign(self):
        """-> (rows uint64[20, n, 4], row_flags uint32[n], copy_table uint64[m, 14, 4], rw uint64[k, 14, 4], rw_flags uint32[k])"""
        events, flags, data, offsets = self.wire()
        res, rows, rf, table, rw, rwf = oneshot.copy_assign(events, flags, data, offsets, self.r)  # zk_copy_assign
        return rows, rf, table, rw, rwf
