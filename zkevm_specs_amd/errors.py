"""Exception classes of the reference's boundary and the status-code -> exception mapping.

Reference: ConstraintUnsatFailure (evm_circuit/instruction.py:53, util/constraint_system.py:7),
LookupUnsatFailure / LookupAmbiguousFailure / WrongQueryKey (evm_circuit/table.py:363-378).
"""


class ConstraintUnsatFailure(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class LookupUnsatFailure(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class LookupAmbiguousFailure(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class WrongQueryKey(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class UnsupportedOnDevice(Exception):
    """The engine has no device implementation for this execution state / gadget."""


KIND_OK, KIND_ASSERT, KIND_CONSTRAINT, KIND_LOOKUP_UNSAT, KIND_LOOKUP_AMBIGUOUS = 0, 1, 2, 3, 4
KIND_WRONG_QUERY_KEY, KIND_NOT_IMPLEMENTED, KIND_TYPE_ERROR, KIND_OVERFLOW_ERROR = 5, 6, 7, 8
KIND_VALUE_ERROR, KIND_ZERO_DIVISION, KIND_NAME_ERROR, KIND_INDEX_ERROR, KIND_UNSUPPORTED = 9, 10, 11, 12, 15

KIND_NAMES = {
    0: "ok", 1: "AssertionError", 2: "ConstraintUnsatFailure", 3: "LookupUnsatFailure",
    4: "LookupAmbiguousFailure", 5: "WrongQueryKey", 6: "NotImplementedError", 7: "TypeError",
    8: "OverflowError", 9: "ValueError", 10: "ZeroDivisionError", 11: "UnboundLocalError", 12: "IndexError", 13: "AttributeError", 15: "UnsupportedOnDevice",
}


def _boundary_exception(name, msg, where):
    """One of the four exception classes of the reference's boundary.  When the caller has the reference loaded in this
    process (a drop-in user has: its tests build `Tables` / `StepState` with it) the mirror raises the reference's OWN class
    objects, so `except LookupUnsatFailure` written against `zkevm_specs` keeps catching; this module never imports the
    reference itself.  ConstraintUnsatFailure exists twice there: evm_circuit/instruction.py:53 (EVM circuit) and
    util/constraint_system.py:7 (the ConstraintSystem circuits: Exp, Copy)."""
    import sys

    tab = sys.modules.get("zkevm_specs.evm_circuit.table")
    if name == "ConstraintUnsatFailure":
        mod = sys.modules.get("zkevm_specs.evm_circuit.instruction" if where.startswith("EVM circuit") else "zkevm_specs.util.constraint_system")
        return getattr(mod, name, ConstraintUnsatFailure)(msg)
    cls = getattr(tab, name, None)
    if cls is None:
        return globals()[name](msg)
    if name == "WrongQueryKey":  # table.py:363-378: the reference's constructors take the table name and the query
        e = cls("device", set())
    elif name == "LookupUnsatFailure":
        e = cls("device", msg)
    else:
        e = cls("device", msg, [])
    e.message = msg
    return e


def exception_for_code(code, where=""):
    kind, site = code >> 24, code & 0xFFFFFF
    msg = f"{where}: constraint site {site} unsatisfied" if where else f"constraint site {site} unsatisfied"
    if kind == KIND_ASSERT:
        return AssertionError(_boundary_exception("ConstraintUnsatFailure", msg, where))
    if kind == KIND_CONSTRAINT:
        return _boundary_exception("ConstraintUnsatFailure", msg, where)
    if kind == KIND_LOOKUP_UNSAT:
        return _boundary_exception("LookupUnsatFailure", msg, where)
    if kind == KIND_LOOKUP_AMBIGUOUS:
        return _boundary_exception("LookupAmbiguousFailure", msg, where)
    if kind == KIND_WRONG_QUERY_KEY:
        return _boundary_exception("WrongQueryKey", msg, where)
    if kind == KIND_NOT_IMPLEMENTED:
        return NotImplementedError(msg)
    if kind == KIND_TYPE_ERROR:
        return TypeError(msg)
    if kind == KIND_OVERFLOW_ERROR:
        return OverflowError(msg)
    if kind == KIND_VALUE_ERROR:
        return ValueError(msg)
    if kind == KIND_ZERO_DIVISION:
        return ZeroDivisionError(msg)
    if kind == KIND_NAME_ERROR:
        return UnboundLocalError(msg)
    if kind == KIND_INDEX_ERROR:
        return IndexError(msg)
    if kind == 13:
        return AttributeError(msg)
    if kind == KIND_UNSUPPORTED:
        return UnsupportedOnDevice(msg)
    return RuntimeError(f"{msg} (unknown kind {kind})")


def raise_for_code(code, where=""):
    if code != 0:
        raise exception_for_code(code, where)


def kind_for_exception(e):
    """Status-code kind of a Python exception instance (inverse of exception_for_code)."""
    names = {v: k for k, v in KIND_NAMES.items()}
    names["NameError"] = KIND_NAME_ERROR
    for cls in type(e).__mro__:
        if cls.__name__ in names:
            return names[cls.__name__]
    return KIND_UNSUPPORTED
