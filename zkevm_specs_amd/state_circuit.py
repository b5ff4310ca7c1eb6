"""Host-side mirror of the reference's State-circuit interface, evaluated on the MI355X.

Reference seam (src/zkevm_specs/state_circuit.py): `check_state_row(row, row_prev, row_next,
tables)` :492 is called once per row by the driver loop in tests/test_state_circuit.py:26-30
(neighbours wrap modulo n).  `verify_state_rows` is that loop as one device pass;
`check_state_row` keeps the per-row signature for callers that compose it by hand.

Error behaviour matches the reference: a failing `assert` raises AssertionError, a missing
MPT row raises LookupUnsatFailure, an invalid tag/field tag raises ValueError — the class is
decoded from the device status code (csrc/common.hpp ZkKind).
"""
from . import engine
from .errors import raise_for_code
from .flatten import flatten_mpt_table, flatten_state_rows


def verify_state_rows(rows, tables, success=True, return_status=False):
    """Evaluate every row of the State circuit on the GPU.

    rows: sequence of reference-style `Row`s; tables: object with `.mpt_table` (set of
    MPTTableRow).  Like the reference's test driver, stops at the first failing row: raises
    the exception the reference would raise there when `success` is True; when `success` is
    False it asserts that an AssertionError occurred (tests/test_state_circuit.py:17-38).
    """
    cols, flags = flatten_state_rows(rows)
    mpt = flatten_mpt_table(tables.mpt_table)
    with engine.open_state(cols, flags, mpt) as s:
        res = s.run()
        status = s.read_status() if return_status else None
    if return_status:
        return res, status
    if res.ok:
        assert success, "State circuit unexpectedly satisfied"
        return res
    if success or res.first_fail_kind != 1:
        raise_for_code(res.first_fail_code, f"State circuit row {res.first_fail_row}")
    return res


def check_state_row(row, row_prev, row_next, tables):
    """Single-row form with the reference's signature (three-row window on the device)."""
    cols, flags = flatten_state_rows([row_prev, row, row_next])
    mpt = flatten_mpt_table(tables.mpt_table)
    with engine.open_state(cols, flags, mpt) as s:
        s.run()
        status = s.read_status()
    raise_for_code(int(status[1]), "State circuit row")
