"""Marshalling of the reference's witness objects into the C-ABI wire format.

Duck-typed: works on the reference's own objects (`zkevm_specs.state_circuit.Row`,
`MPTTableRow`, `StepState`, `RWTableRow`, ...) or on this package's mirrors — only attribute
names are used, nothing is imported from the reference.
"""
from .wire import rows_to_colmajor, rows_to_rowmajor
import numpy as np


def _n(x):
    """canonical integer of an FQ / Expression / int / bool / IntEnum"""
    if hasattr(x, "expr"):
        return x.expr().n
    if hasattr(x, "n"):
        return x.n
    return int(x)


def _is_word(x):
    return bool(getattr(x, "is_word", True))


STATE_NCELLS = 57
MPT_NCELLS = 12


def state_row_cells(row):
    """reference state_circuit.Row (:63-96) -> 57 ints + flags"""
    keys = row.keys
    cells = [_n(row.rw_counter), _n(row.is_write), _n(keys[0]), _n(keys[1]), _n(keys[2]),
             _n(keys[3]), _n(keys[4].lo), _n(keys[4].hi)]
    cells += [_n(x) for x in row.key2_limbs]
    cells += [_n(x) for x in row.key45_bytes]
    cells += [_n(row.value.lo), _n(row.value.hi), _n(row.initial_value.lo),
              _n(row.initial_value.hi), _n(row.root.lo), _n(row.root.hi),
              _n(row.lexicographic_ordering_selector)]
    flags = (1 if _is_word(row.value) else 0) | (2 if _is_word(row.initial_value) else 0)
    return cells, flags


def flatten_state_rows(rows):
    cf = [state_row_cells(r) for r in rows]
    cols = rows_to_colmajor([c for c, _ in cf], STATE_NCELLS)
    flags = np.array([f for _, f in cf], dtype=np.uint32)
    return cols, flags


def mpt_row_cells(m):
    """MPTTableRow (evm_circuit/table.py:461-468) -> 12 ints"""
    return [_n(m.address), _n(m.proof_type), _n(m.storage_key.lo), _n(m.storage_key.hi),
            _n(m.root.lo), _n(m.root.hi), _n(m.root_prev.lo), _n(m.root_prev.hi),
            _n(m.value.lo), _n(m.value.hi), _n(m.value_prev.lo), _n(m.value_prev.hi)]


def flatten_mpt_table(mpt_table):
    rows = sorted(set(tuple(mpt_row_cells(m)) for m in mpt_table))
    return rows_to_rowmajor(rows, MPT_NCELLS)
