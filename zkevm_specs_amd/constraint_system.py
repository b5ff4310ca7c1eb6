"""`ConstraintSystem` (reference util/constraint_system.py:12-74) without the reference installed, and its batch form.

A gate is `cond * (lhs - rhs) == 0` with an optional multiplicative selector `cond`; a violated gate is an
`AssertionError` carrying a `ConstraintUnsatFailure`, and `range_check` RAISES `ConstraintUnsatFailure` (the two exception routes
the boundary keeps apart: kind 1 vs kind 2 of `zk_result.first_fail_code`).

`BatchConstraintSystem` evaluates the same gates over `FrArray` columns — n rows per call through `zk_fr_op` — and reports the
rows that violate them instead of raising on the first one (`violations`), or raises like the scalar form (`check()`).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .arithmetic import FQ, MAX_N_BYTES, FrArray, Word
from .errors import ConstraintUnsatFailure


class ConstraintSystem:
    def __init__(self, cond=None):
        self.cond = cond

    # `with cs.condition(sel): ...` scopes a selector; conditions do not nest
    def condition(self, cond):
        assert self.cond is None, "Don't support recursive conditions"
        self.cond = cond
        return self

    def __enter__(self):
        return self

    def __exit__(self, e_type, e_value, traceback):
        if e_type is not None:
            raise e_value
        self.cond = None
        return self

    def _gate(self, expr) -> FQ:
        v = FQ(expr.expr())
        return v if self.cond is None else FQ(self.cond.expr()) * v

    def _require(self, ok: bool, message: str) -> None:
        assert ok, ConstraintUnsatFailure(message)

    def constrain_equal(self, lhs, rhs) -> None:
        self._require(self._gate(FQ(lhs.expr()) - rhs.expr()) == 0, f"Expected values to be equal, but got {lhs} and {rhs}")

    def constrain_equal_word(self, lhs: Word, rhs: Word) -> None:
        lo, hi = FQ(lhs.lo.expr()) - rhs.lo.expr(), FQ(lhs.hi.expr()) - rhs.hi.expr()
        self._require(self._gate(lo) == 0 and self._gate(hi) == 0, f"Expected words to be equal, but got {lhs} and {rhs}")

    def constrain_zero(self, value) -> None:
        self._require(self._gate(value) == 0, f"Expected value to be 0, but got {value}")

    def constrain_zero_word(self, value: Word) -> None:
        self._require(self._gate(value.lo) == 0 and self._gate(value.hi) == 0, f"Expected word to be 0, but got {value}")

    def constrain_bool(self, value) -> None:
        self._require(self._gate(value).n in (0, 1), f"Expected value to be a bool, but got {value}")

    def is_zero(self, value) -> FQ:
        return FQ(int(FQ(value.expr()).n == 0))

    def is_equal(self, lhs, rhs) -> FQ:
        return self.is_zero(FQ(lhs.expr()) - rhs.expr())

    def range_check(self, value, n_bytes: int) -> bytes:
        assert n_bytes <= MAX_N_BYTES, "Too many bytes to composite an integer in field"
        v = FQ(value.expr()).n
        if v >> (8 * n_bytes):
            raise ConstraintUnsatFailure(f"Value {value} has too many bytes to fit {n_bytes} bytes")
        return v.to_bytes(n_bytes, "little")


class BatchConstraintSystem:
    """The same gates over columns: every argument is an `FrArray` (or something `FrArray(...)` accepts) of n rows."""

    def __init__(self, n_rows: int, cond: Optional[FrArray] = None):
        self.n = int(n_rows)
        self.cond = cond
        self.violations = np.zeros(self.n, dtype=bool)  # rows with at least one violated gate so far
        self.first_site = np.zeros(self.n, dtype=np.int32)  # 1-based ordinal of the first violated gate of each row
        self._site = 0

    def condition(self, cond):
        assert self.cond is None, "Don't support recursive conditions"
        self.cond = FrArray(cond)
        return self

    def __enter__(self):
        return self

    def __exit__(self, e_type, e_value, traceback):
        if e_type is not None:
            raise e_value
        self.cond = None
        return self

    def _gate(self, expr: FrArray) -> FrArray:
        return expr if self.cond is None else self.cond * expr

    def _record(self, bad: np.ndarray) -> np.ndarray:
        self._site += 1
        fresh = bad & ~self.violations
        self.first_site[fresh] = self._site
        self.violations |= bad
        return bad

    def constrain_equal(self, lhs, rhs) -> np.ndarray:
        return self._record(~self._gate(FrArray(lhs) - FrArray(rhs)).is_zero())

    def constrain_zero(self, value) -> np.ndarray:
        return self._record(~self._gate(FrArray(value)).is_zero())

    def constrain_bool(self, value) -> np.ndarray:
        g = self._gate(FrArray(value))
        return self._record(~(g.is_zero() | (g == 1)))

    def is_zero(self, value) -> FrArray:
        z = np.zeros((self.n, 4), dtype=np.uint64)
        z[:, 0] = FrArray(value).is_zero()
        return FrArray(z)

    def is_equal(self, lhs, rhs) -> FrArray:
        return self.is_zero(FrArray(lhs) - FrArray(rhs))

    def range_check(self, value, n_bytes: int) -> np.ndarray:
        """rows whose value does NOT fit n_bytes (the scalar form raises for such a row)"""
        assert n_bytes <= MAX_N_BYTES, "Too many bytes to composite an integer in field"
        c = FrArray(value).cells
        full, rem = divmod(8 * n_bytes, 64)
        bad = c[:, full + (1 if rem else 0):].any(axis=1)
        if rem:
            bad |= (c[:, full] >> np.uint64(rem)) != 0
        return self._record(bad)

    def check(self) -> None:
        """raise like the scalar form would on the first violating row"""
        if self.violations.any():
            row = int(np.argmax(self.violations))
            raise AssertionError(ConstraintUnsatFailure(f"gate {int(self.first_site[row])} violated in row {row} ({int(self.violations.sum())} rows fail)"))
