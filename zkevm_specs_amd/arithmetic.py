"""Hand-composition surface of the boundary ("plus FQ, Word, RLC, ConstraintSystem ... for code that composes
gadgets by hand") — usable WITHOUT the reference installed beside the package.

Two layers:

* scalar objects with the reference's names and behaviour — `FQ` (BN254 scalar field: `+ - * / ** neg == inv`, `.n`, `.expr()`,
  `zero() / one()`; reference util/arithmetic.py:41-63 over py_ecc.bn128.FQ), `RLC` (:69-96), `Word` / `WordOrValue`
  (:99-195), `linear_combine_bytes` (:9-24), `bytes_to_fq` (:227), `byte_size` (:220), `add_words` (:236-242),
  `mul_add_words` (:245-276), `Expression` / `cast_expr` (:201-217).  A scalar lives on the host as a Python int — one field
  operation per call cannot amortise a kernel launch;
* `FrArray` — the batch form: n field elements as `uint64[n, 4]` canonical little-endian cells (the wire format of
  include/zkevm_hip.h), every operator ONE call of the C entry `zk_fr_op` over the whole array (HIP: `fr_op_kernel`, u32-limb
  Montgomery; `ZK_BACKEND=cpu`: the same functions on the host).  `linear_combine_bytes_batch` is the Horner recombination of
  `linear_combine_bytes` over n byte strings at once.

The reference's division is multiplication by the inverse with `inv(0) == 0` (py_ecc `prime_field_inv`); both layers keep that.
"""
from __future__ import annotations

from typing import Protocol, Sequence, Tuple, Union, runtime_checkable

import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # bn128.curve_order (util/arithmetic.py:42-45)
MAX_N_BYTES = 31  # util/param.py: the most bytes that always fit one field element


def _n(x) -> int:
    """int of anything FQ-like (int, FQ, objects.FQ, something with .expr())"""
    if isinstance(x, int):
        return x
    if hasattr(x, "n"):
        return int(x.n)
    if hasattr(x, "expr"):
        return int(x.expr().n)
    raise TypeError(f"expected an int or a field element, got {type(x).__name__}")


class FQ:
    """Element of the BN254 scalar field; canonical value in `.n`."""

    __slots__ = ("n",)
    field_modulus = P

    def __init__(self, value=0):
        if isinstance(value, bool):
            value = int(value)
        self.n = _n(value) % P

    # --- construction helpers --------------------------------------------------------------------------------------
    @classmethod
    def zero(cls) -> "FQ":
        return cls(0)

    @classmethod
    def one(cls) -> "FQ":
        return cls(1)

    def expr(self) -> "FQ":
        return FQ(self.n)

    # --- ring operations -------------------------------------------------------------------------------------------
    def __add__(self, o):
        return FQ(self.n + _n(o))

    __radd__ = __add__

    def __sub__(self, o):
        return FQ(self.n - _n(o))

    def __rsub__(self, o):
        return FQ(_n(o) - self.n)

    def __mul__(self, o):
        return FQ(self.n * _n(o))

    __rmul__ = __mul__

    def __neg__(self):
        return FQ(-self.n)

    def __pow__(self, e: int):
        if e < 0:
            return self.inv() ** (-e)
        return FQ(pow(self.n, e, P))

    def inv(self) -> "FQ":
        """multiplicative inverse; inv(0) == 0 (prime_field_inv's convention)"""
        return FQ(pow(self.n, P - 2, P)) if self.n else FQ(0)

    def __truediv__(self, o):
        return self * FQ(o).inv()

    def __rtruediv__(self, o):
        return FQ(o) * self.inv()

    # --- comparison / hashing --------------------------------------------------------------------------------------
    def __eq__(self, o) -> bool:
        if isinstance(o, int):
            return self.n == o % P
        if hasattr(o, "n"):
            return self.n == int(o.n) % P
        raise TypeError(f"cannot compare FQ with {type(o).__name__}")  # py_ecc's FQ.__eq__ raises on foreign types too

    def __ne__(self, o) -> bool:
        return not self == o

    def __hash__(self) -> int:
        return hash(self.n)

    def __int__(self) -> int:
        return self.n

    # (no __bool__: like py_ecc's FQ an element is always truthy — `if cond:` in the reference's ConstraintSystem._eval relies on it)

    def __repr__(self) -> str:
        return hex(self.n)


IntOrFQ = Union[int, FQ]


@runtime_checkable
class Expression(Protocol):
    def expr(self) -> FQ: ...


def cast_expr(expression, ty):
    if not isinstance(expression, ty):
        raise TypeError(f"Casting Expression to {ty}, but got {type(expression)}")
    return expression


def linear_combine_bytes(seq: Sequence[IntOrFQ], base: IntOrFQ, range_check: bool = True) -> FQ:
    """seq[0] + seq[1] base + seq[2] base^2 + ... (little-endian Horner);
    >>> linear_combine_bytes([1, 2, 3], 10).n
    321
    """
    acc, b = 0, _n(base)
    for limb in reversed(list(seq)):
        v = _n(limb)
        if range_check:
            assert 0 <= v < 256, "Each byte should fit in 8-bit"
        acc = (acc * b + v) % P
    return FQ(acc)


def bytes_to_fq(value: bytes) -> FQ:
    assert len(value) <= MAX_N_BYTES
    return FQ(int.from_bytes(value, "little"))


class RLC:
    """A little-endian byte string of n_bytes with its random linear combination."""

    def __init__(self, value: Union[int, bytes], randomness: IntOrFQ = 0, n_bytes: int = 32) -> None:
        raw = value.to_bytes(n_bytes, "little") if isinstance(value, int) else bytes(value)
        if len(raw) > n_bytes:
            raise ValueError(f"RLC expects to have {n_bytes} bytes, but got {len(raw)} bytes")
        self.le_bytes = raw + b"\x00" * (n_bytes - len(raw))
        self.int_value = int.from_bytes(self.le_bytes, "little")
        self.rlc_value = linear_combine_bytes(self.le_bytes, randomness)

    def expr(self) -> FQ:
        return FQ(self.rlc_value)

    def __hash__(self) -> int:
        return hash(self.rlc_value)

    def __repr__(self) -> str:
        return f"RLC({self.int_value})"


def byte_size(value: Union[int, RLC]) -> int:
    if isinstance(value, RLC):
        return len(value.le_bytes.rstrip(b"\x00"))
    return (value.bit_length() + 7) // 8


class Word:
    """256-bit word as (lo, hi) 128-bit halves."""

    def __init__(self, value, check: bool = True) -> None:
        if isinstance(value, tuple):
            self.lo, self.hi = value
            assert not check or (_n(self.lo) < 1 << 128 and _n(self.hi) < 1 << 128)
            return
        if isinstance(value, int):
            assert not check or value < 1 << 256
            value = value.to_bytes(32, "little")  # a negative int raises OverflowError here, as in the reference
        assert isinstance(value, bytes)
        assert len(value) == 32, f"Word expects to receive 32 bytes, but got {len(value)} bytes"
        self.lo, self.hi = bytes_to_fq(value[:16]), bytes_to_fq(value[16:])

    @classmethod
    def from_lo(cls, lo):
        return cls((lo, FQ(0)))

    def int_value(self) -> int:
        return _n(self.lo) + (_n(self.hi) << 128)

    def to_lo_hi(self) -> Tuple[FQ, FQ]:
        return FQ(self.lo), FQ(self.hi)

    def to_64s(self) -> Tuple[FQ, ...]:
        lo, hi = _n(self.lo).to_bytes(16, "little"), _n(self.hi).to_bytes(16, "little")
        return tuple(bytes_to_fq(h[k:k + 8]) for h in (lo, hi) for k in (0, 8))

    def to_le_bytes(self) -> Tuple[FQ, ...]:
        return tuple(FQ(b) for b in _n(self.lo).to_bytes(16, "little") + _n(self.hi).to_bytes(16, "little"))

    def select(self, selector: IntOrFQ) -> "Word":
        s = FQ(selector)
        return Word((s * self.lo, s * self.hi))

    def __add__(self, other: "Word") -> "Word":
        """cell-wise sum (for selects), NOT a 256-bit addition"""
        return Word((FQ(self.lo) + other.lo, FQ(self.hi) + other.hi))

    def __eq__(self, other) -> bool:
        assert isinstance(other, Word)
        return _n(self.lo) % P == _n(other.lo) % P and _n(self.hi) % P == _n(other.hi) % P

    def __hash__(self) -> int:
        return hash((_n(self.lo), _n(self.hi)))

    def __repr__(self) -> str:
        return f"Word({hex(self.int_value())})"


class WordOrValue(Word):
    """a Word, or a single value that fits the field (then `hi` is 0 and `value()` returns it)"""

    def __init__(self, value) -> None:
        self.is_word = isinstance(value, Word)
        self.lo, self.hi = (value.lo, value.hi) if self.is_word else (value, FQ(0))

    def value(self):
        assert not self.is_word
        return self.lo

    def __repr__(self) -> str:
        return super().__repr__() if self.is_word else f"Value({hex(_n(self.lo))})"


def add_words(addends: Sequence[Word]) -> Tuple[Word, FQ]:
    """256-bit sum of the words and the carry out of bit 256"""
    lo = sum(_n(w.lo) for w in addends) % P
    carry_lo, sum_lo = divmod(lo, 1 << 128)
    hi = (sum(_n(w.hi) for w in addends) + carry_lo) % P
    carry_hi, sum_hi = divmod(hi, 1 << 128)
    return Word((FQ(sum_lo), FQ(sum_hi))), FQ(carry_hi)


def mul_add_words(a: Word, b: Word, c: Word, d: Word):
    """a * b + c == d over 64-bit limbs: returns (overflow, (carry_lo, carry_hi), [(lhs, rhs), (lhs, rhs)]); the carries are
    FIELD quotients by 2^128 (the caller range-checks them to 9 bytes, instruction.py:613-627)"""
    a0, a1, a2, a3 = a.to_64s()
    b0, b1, b2, b3 = b.to_64s()
    c_lo, c_hi = c.to_lo_hi()
    d_lo, d_hi = d.to_lo_hi()
    t0 = a0 * b0
    t1 = a0 * b1 + a1 * b0
    t2 = a0 * b2 + a1 * b1 + a2 * b0
    t3 = a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0
    two64, two128 = FQ(1 << 64), FQ(1 << 128)
    low, high = t0 + t1 * two64 + c_lo, t2 + t3 * two64 + c_hi
    carry_lo = (low - d_lo) / two128
    carry_hi = (high + carry_lo - d_hi) / two128
    overflow = carry_hi + a1 * b3 + a2 * b2 + a3 * b1 + a2 * b3 + a3 * b2 + a3 * b3
    return overflow, (carry_lo, carry_hi), [(low, d_lo + carry_lo * two128), (high + carry_lo, d_hi + carry_hi * two128)]


# ---- batch layer ------------------------------------------------------------------------------------------------------
_OP_ADD, _OP_SUB, _OP_MUL, _OP_NEG, _OP_INV, _OP_DIV = 0, 1, 2, 4, 5, 6


def _cells(values) -> np.ndarray:
    """ints / FQs / an (n, 4) uint64 array -> canonical uint64[n, 4]"""
    if isinstance(values, FrArray):
        return values.cells
    if isinstance(values, np.ndarray) and values.dtype == np.uint64 and values.ndim == 2 and values.shape[1] == 4:
        return np.ascontiguousarray(values)
    ints = [_n(v) % P for v in values]
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in ints), dtype="<u8").reshape(len(ints), 4).copy()


class FrArray:
    """n field elements in the wire format (uint64[n, 4], canonical, little-endian limbs); operators run through `zk_fr_op`."""

    __slots__ = ("cells",)

    def __init__(self, values):
        self.cells = _cells(values)

    def __len__(self) -> int:
        return int(self.cells.shape[0])

    def _other(self, o) -> np.ndarray:
        if isinstance(o, (int, FQ)):
            return np.broadcast_to(_cells([o]), self.cells.shape)
        c = _cells(o)
        if c.shape != self.cells.shape:
            raise ValueError(f"FrArray length mismatch: {self.cells.shape[0]} vs {c.shape[0]}")
        return c

    def _op(self, op: int, o=None) -> "FrArray":
        from . import engine  # the library is bound on first use: importing this module needs no GPU

        b = self.cells if o is None else self._other(o)
        out = FrArray.__new__(FrArray)
        out.cells = engine.fr_op(op, self.cells, b) if len(self) else self.cells.copy()
        return out

    def __add__(self, o):
        return self._op(_OP_ADD, o)

    __radd__ = __add__

    def __sub__(self, o):
        return self._op(_OP_SUB, o)

    def __rsub__(self, o):
        return FrArray(self._other(o))._op(_OP_SUB, self)

    def __mul__(self, o):
        return self._op(_OP_MUL, o)

    __rmul__ = __mul__

    def __neg__(self):
        return self._op(_OP_NEG)

    def inv(self):
        """element-wise inverse, inv(0) == 0"""
        return self._op(_OP_INV)

    def __truediv__(self, o):
        return self._op(_OP_DIV, o)

    def __eq__(self, o):  # element-wise, like numpy
        return (self.cells == self._other(o)).all(axis=1)

    def is_zero(self) -> np.ndarray:
        return ~self.cells.any(axis=1)

    def to_ints(self):
        raw = np.ascontiguousarray(self.cells).astype("<u8").tobytes()
        return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(self))]

    def __getitem__(self, i):
        if isinstance(i, slice):
            out = FrArray.__new__(FrArray)
            out.cells = np.ascontiguousarray(self.cells[i])
            return out
        return FQ(int.from_bytes(np.ascontiguousarray(self.cells[i]).astype("<u8").tobytes(), "little"))

    def __repr__(self) -> str:
        return f"FrArray(n={len(self)})"


def linear_combine_bytes_batch(byte_rows: np.ndarray, base: IntOrFQ, range_check: bool = True) -> FrArray:
    """`linear_combine_bytes` of every row of a uint8[n, k] array at once: k multiply-add passes over n elements"""
    byte_rows = np.asarray(byte_rows)
    if range_check:
        assert byte_rows.size == 0 or (byte_rows.min() >= 0 and byte_rows.max() < 256), "Each byte should fit in 8-bit"
    n, k = byte_rows.shape
    acc = FrArray(np.zeros((n, 4), dtype=np.uint64))
    for j in range(k - 1, -1, -1):
        col = np.zeros((n, 4), dtype=np.uint64)
        col[:, 0] = byte_rows[:, j]
        acc = acc * base + FrArray(col)
    return acc
