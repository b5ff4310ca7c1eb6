"""Host-side mirror of `zkevm_specs.exp_circuit.verify_exp_circuit` (exp_circuit.py:88-97), on the MI355X (`zk_exp_verify`)."""
from . import oneshot
from .errors import raise_for_code
from .flatten import flatten_exp_rows


def verify_exp_circuit(exp_circuit):
    """exp_circuit: object with `.table()` returning ExpCircuitRow-like rows (typing.py:868-880).
    The reference propagates the first failing row's exception (no try/except in the loop)."""
    rows = list(exp_circuit.table())
    if not rows:
        return None
    res, _ = oneshot.exp_verify(flatten_exp_rows(rows))
    raise_for_code(res.first_fail_code, f"Exp circuit row {res.first_fail_row}")
    return res
