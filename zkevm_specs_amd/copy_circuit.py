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
