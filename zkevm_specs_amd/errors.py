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


def exception_for_code(code, where=""):
    kind, site = code >> 24, code & 0xFFFFFF
    msg = f"{where}: constraint site {site} unsatisfied" if where else f"constraint site {site} unsatisfied"
    if kind == KIND_ASSERT:
        return AssertionError(ConstraintUnsatFailure(msg))
    if kind == KIND_CONSTRAINT:
        return ConstraintUnsatFailure(msg)
    if kind == KIND_LOOKUP_UNSAT:
        return LookupUnsatFailure(msg)
    if kind == KIND_LOOKUP_AMBIGUOUS:
        return LookupAmbiguousFailure(msg)
    if kind == KIND_WRONG_QUERY_KEY:
        return WrongQueryKey(msg)
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
