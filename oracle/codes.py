"""Status-code vocabulary shared by the oracle and the tests (mirrors csrc/common.hpp ZkKind)."""
OK = 0
ASSERT = 1
CONSTRAINT = 2
LOOKUP_UNSAT = 3
LOOKUP_AMBIGUOUS = 4
WRONG_QUERY_KEY = 5
NOT_IMPLEMENTED = 6
TYPE_ERROR = 7
OVERFLOW_ERROR = 8
VALUE_ERROR = 9
ZERO_DIVISION = 10
NAME_ERROR = 11
INDEX_ERROR = 12
ATTRIBUTE_ERROR = 13
UNSUPPORTED = 15

KIND_NAMES = {
    OK: "ok", ASSERT: "AssertionError", CONSTRAINT: "ConstraintUnsatFailure",
    LOOKUP_UNSAT: "LookupUnsatFailure", LOOKUP_AMBIGUOUS: "LookupAmbiguousFailure",
    WRONG_QUERY_KEY: "WrongQueryKey", NOT_IMPLEMENTED: "NotImplementedError",
    TYPE_ERROR: "TypeError", OVERFLOW_ERROR: "OverflowError", VALUE_ERROR: "ValueError",
    ZERO_DIVISION: "ZeroDivisionError", NAME_ERROR: "NameError", INDEX_ERROR: "IndexError", ATTRIBUTE_ERROR: "AttributeError",
    UNSUPPORTED: "Unsupported",
}
NAME_TO_KIND = {v: k for k, v in KIND_NAMES.items()}


def code(kind, site):
    return (kind << 24) | (site & 0xFFFFFF)


def kind_of(c):
    return c >> 24


def site_of(c):
    return c & 0xFFFFFF


class Fail(Exception):
    """Raised by oracle checks: carries the status code of the failing site."""

    def __init__(self, kind, site):
        super().__init__(f"{KIND_NAMES.get(kind, kind)} at site {site}")
        self.code = code(kind, site)
