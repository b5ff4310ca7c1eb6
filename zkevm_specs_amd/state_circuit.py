"""Host-side mirror of the reference's State-circuit interface, evaluated on the MI355X.

Reference seam (src/zkevm_specs/state_circuit.py): `check_state_row(row, row_prev, row_next,
tables)` :492 is called once per row by the driver loop in tests/test_state_circuit.py:26-30
(neighbours wrap modulo n).  `verify_state_rows` is that loop as one device pass;
`check_state_row` keeps the per-row signature for callers that compose it by hand.

Error behaviour matches the reference: a failing `assert` raises AssertionError, a missing
MPT row raises LookupUnsatFailure, an invalid tag/field tag raises ValueError — the class is
decoded from the device status code (csrc/common.hpp ZkKind).
"""
from . import oneshot
from .errors import raise_for_code
from .flatten import flatten_mpt_table, flatten_state_ops, flatten_state_rows


class StateWitness(tuple):
    """(cols uint64[57, n, 4], flags uint32[n]): the State-circuit rows in wire form, as `assign_state_circuit`
    returns them and `verify_state_rows` accepts them."""

    def __new__(cls, cols, flags):
        return super().__new__(cls, (cols, flags))


_MOCK_MPT_SITES = (1, 2, 4)  # csrc/state_assign.hpp: failures inside _mock_mpt_updates


def _assign(ops, mpt_only):
    from .errors import raise_for_code

    wire_ops, wire_flags = ops if isinstance(ops, tuple) else flatten_state_ops(ops)
    res, status, rows, flags, mpt = oneshot.state_assign(wire_ops, wire_flags)  # zk_state_assign
    if not res.ok:
        # `_mock_mpt_updates` runs over every op before the first `op2row` (state_circuit.py:856, :880-883)
        for sites in ((_MOCK_MPT_SITES,) if mpt_only else (_MOCK_MPT_SITES, (3,))):
            for i in status.nonzero()[0]:
                if (int(status[i]) & 0xFFFFFF) in sites:
                    raise_for_code(int(status[i]), f"state op {i}")
    return rows, flags, mpt


def assign_state_circuit(ops):
    """`assign_state_circuit(ops)` of the reference (state_circuit.py:855-884) on the GPU: ops = list of reference-style
    `Operation`s (or the (ops, flags) wire arrays) -> StateWitness.  Raises what the reference raises (AssertionError
    from `Word(...)` inside the mock MPT updates, OverflowError from `address.to_bytes(20)`)."""
    rows, flags, _ = _assign(ops, mpt_only=False)
    return StateWitness(rows, flags)


def mpt_table_from_ops(ops):
    """`mpt_table_from_ops(ops)` (state_circuit.py:887-888) on the GPU -> MPT rows uint64[m, 12, 4], one per distinct
    (address, field_tag, storage_key) of the Account / Storage ops, in first-occurrence order."""
    return _assign(ops, mpt_only=True)[2]


def verify_state_rows(rows, tables, success=True, return_status=False):
    """Evaluate every row of the State circuit on the GPU.

    rows: sequence of reference-style `Row`s; tables: object with `.mpt_table` (set of
    MPTTableRow).  Like the reference's test driver, stops at the first failing row: raises
    the exception the reference would raise there when `success` is True; when `success` is
    False it asserts that an AssertionError occurred (tests/test_state_circuit.py:17-38).
    """
    cols, flags = rows if isinstance(rows, StateWitness) else flatten_state_rows(rows)
    mpt = tables.mpt_table if hasattr(tables.mpt_table, "shape") else flatten_mpt_table(tables.mpt_table)
    res, status = oneshot.state_verify(cols, flags, mpt)  # zk_state_verify
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
    _, status = oneshot.state_verify(cols, flags, mpt)
    raise_for_code(int(status[1]), "State circuit row")
