"""The hand-composition surface (zkevm_specs_amd/arithmetic.py, constraint_system.py): reference names and
behaviour without the reference installed.  Scalars against Python big-int arithmetic (authoritative: (a * b) % p) and the
reference's own pinned facts (the `linear_combine_bytes` doctest, util/arithmetic.py:15-17; `FQ(8).inv()` making
is_mul / is_div / is_mod exactly 0 / 1, execution/mul_div_mod.py:14-16; carries of `mul_add_words` landing on <= 9 bytes,
instruction.py:613-627); `FrArray` through the C entry `zk_fr_op` — here on the CPU backend (child process: the backend is
chosen at first import), on the MI355X in the gpu-marked test."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from zkevm_specs_amd import arithmetic as A
from zkevm_specs_amd.constraint_system import ConstraintSystem
from zkevm_specs_amd.errors import ConstraintUnsatFailure

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = A.P
NASTY = [0, 1, 2, P - 1, P - 2, (1 << 128) - 1, 1 << 128, (1 << 253), (1 << 64) - 1, 1 << 64, P // 2, P // 2 + 1, 0xDEADBEEF]


def test_fq_against_bigint():
    rng = random.Random(1)
    vals = NASTY + [rng.randrange(P) for _ in range(200)]
    for a in vals:
        for b in vals[::7]:
            x, y = A.FQ(a), A.FQ(b)
            assert (x + y).n == (a + b) % P and (x - y).n == (a - b) % P and (x * y).n == a * b % P
            assert (-x).n == -a % P and (x + b).n == (a + b) % P and (b - x).n == (b - a) % P
            if b % P:
                assert ((x / y) * y).n == a % P
            else:
                assert (x / y).n == 0  # inv(0) == 0
    assert A.FQ(0).inv().n == 0 and A.FQ(P) == 0 and A.FQ(-1).n == P - 1 and A.FQ(A.FQ(5)).n == 5
    assert A.FQ(3) ** 5 == 243 and A.FQ(3) ** -1 == A.FQ(3).inv() and hash(A.FQ(7)) == hash(7) and repr(A.FQ(255)) == "0xff"
    assert A.FQ.zero() == 0 and A.FQ.one() == 1 and A.FQ(9).expr() == 9 and isinstance(A.FQ(1), A.Expression)
    with pytest.raises(TypeError):
        A.FQ(1) == "1"
    # execution/mul_div_mod.py:14-16: the three selectors come out exactly 0 / 1
    inv8 = A.FQ(8).inv()
    for op, want in ((0x02, (1, 0, 0)), (0x04, (0, 1, 0)), (0x06, (0, 0, 1))):
        o = A.FQ(op)
        got = ((o - 4) * (o - 6) * inv8, -(o - 2) * (o - 6) * A.FQ(4).inv(), (o - 2) * (o - 4) * inv8)
        assert tuple(g.n for g in got) == want


def test_linear_combine_rlc_word():
    assert A.linear_combine_bytes([1, 2, 3], 10) == 1 + 2 * 10 + 3 * 100  # the reference's doctest
    with pytest.raises(AssertionError):
        A.linear_combine_bytes([256], 10)
    assert A.linear_combine_bytes([256], 10, range_check=False) == 256
    r = A.FQ(0x1234567890ABCDEF1234567890ABCDEF)
    v = 0x0102030405060708090A0B0C0D0E0F101112131415161718191A1B1C1D1E1F20
    rlc = A.RLC(v, r)
    assert rlc.int_value == v and rlc.le_bytes == v.to_bytes(32, "little")
    assert rlc.expr().n == sum(b * pow(r.n, i, P) for i, b in enumerate(v.to_bytes(32, "little"))) % P
    assert A.RLC(b"\x01\x02", r, 4).le_bytes == b"\x01\x02\x00\x00" and A.byte_size(A.RLC(b"\x01\x02", r, 4)) == 2 and A.byte_size(0x1FF) == 2
    with pytest.raises(ValueError):
        A.RLC(b"\x00" * 5, r, 4)
    w = A.Word(v)
    assert w.int_value() == v and w.lo.n == v & ((1 << 128) - 1) and w.hi.n == v >> 128
    assert [x.n for x in w.to_64s()] == [(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
    assert bytes(x.n for x in w.to_le_bytes()) == v.to_bytes(32, "little")
    assert w == A.Word((A.FQ(w.lo), A.FQ(w.hi))) and (w + w).lo.n == 2 * w.lo.n % P and w.select(0).int_value() == 0 and w.select(1) == w
    assert A.Word.from_lo(A.FQ(5)).int_value() == 5 and repr(A.Word(255)) == "Word(0xff)"
    with pytest.raises(AssertionError):
        A.Word(1 << 256)
    with pytest.raises(OverflowError):
        A.Word(-1)
    with pytest.raises(AssertionError):
        w == 5
    wv = A.WordOrValue(A.FQ(7))
    assert not wv.is_word and wv.value() == 7 and repr(wv) == "Value(0x7)"
    with pytest.raises(AssertionError):
        A.WordOrValue(w).value()
    assert A.cast_expr(A.FQ(1), A.FQ) == 1
    with pytest.raises(TypeError):
        A.cast_expr(A.FQ(1), A.RLC)


def test_add_and_mul_add_words():
    rng = random.Random(2)
    M = 1 << 256
    for _ in range(300):
        a, b, c = (rng.choice([0, 1, M - 1, 1 << 128, (1 << 128) - 1, rng.randrange(M), rng.randrange(1 << 64)]) for _ in range(3))
        s, carry = A.add_words([A.Word(a), A.Word(b), A.Word(c)])
        assert s.int_value() == (a + b + c) % M and carry.n == (a + b + c) >> 256
        d = (a * b + c) % M
        overflow, (clo, chi), cons = A.mul_add_words(A.Word(a), A.Word(b), A.Word(c), A.Word(d))
        assert all(l == r for l, r in cons)
        assert clo.n < 1 << 72 and chi.n < 1 << 72          # what range_check(carry, 9) asserts (instruction.py:613-627)
        assert (overflow.n == 0) == (a * b + c < M) or overflow.n != 0
        # a wrong d: the field quotient by 2^128 is not a small integer any more
        bad = (d + 1) % M
        _, (blo, bhi), _ = A.mul_add_words(A.Word(a), A.Word(b), A.Word(c), A.Word(bad))
        assert blo.n >= 1 << 72 or bhi.n >= 1 << 72


def test_constraint_system():
    cs = ConstraintSystem()
    cs.constrain_equal(A.FQ(5), A.FQ(5))
    cs.constrain_zero(A.FQ(P))
    cs.constrain_bool(A.FQ(1))
    cs.constrain_equal_word(A.Word(9), A.Word(9))
    cs.constrain_zero_word(A.Word(0))
    with pytest.raises(AssertionError) as e:
        cs.constrain_equal(A.FQ(5), A.FQ(6))
    assert isinstance(e.value.args[0], ConstraintUnsatFailure) and "equal" in e.value.args[0].message
    with pytest.raises(AssertionError):
        cs.constrain_bool(A.FQ(2))
    with pytest.raises(AssertionError):
        cs.constrain_zero_word(A.Word(1 << 128))
    assert cs.is_zero(A.FQ(0)) == 1 and cs.is_zero(A.FQ(3)) == 0 and cs.is_equal(A.FQ(4), A.FQ(4)) == 1
    assert cs.range_check(A.FQ(0x1234), 2) == b"\x34\x12"
    with pytest.raises(ConstraintUnsatFailure):  # raised, not asserted (constraint_system.py:69)
        cs.range_check(A.FQ(1 << 16), 2)
    with pytest.raises(AssertionError):
        cs.range_check(A.FQ(1), 32)
    # a selector switches gates off; conditions do not nest; the scope ends with the block
    with cs.condition(A.FQ(0)) as c0:
        c0.constrain_equal(A.FQ(1), A.FQ(2))
        c0.constrain_bool(A.FQ(7))
        with pytest.raises(AssertionError):
            cs.condition(A.FQ(1))
    assert cs.cond is None
    with pytest.raises(AssertionError):
        with cs.condition(A.FQ(3)) as c1:
            c1.constrain_zero(A.FQ(2))


FRARRAY_CHILD = r'''
import os, random, sys
import numpy as np
sys.path.insert(0, os.environ["ZK_ROOT"])
from zkevm_specs_amd import arithmetic as A
from zkevm_specs_amd.constraint_system import BatchConstraintSystem
P = A.P
rng = random.Random(3)
nasty = [0, 1, 2, P - 1, P - 2, (1 << 128) - 1, 1 << 128, 1 << 253, (1 << 64) - 1, 1 << 64, P // 2]
a = nasty + [rng.randrange(P) for _ in range(2000)]
b = [rng.choice(nasty + [rng.randrange(P)]) for _ in a]
X, Y = A.FrArray(a), A.FrArray(b)
assert X.to_ints() == a and len(X) == len(a) and X[3].n == a[3] and X[1:4].to_ints() == a[1:4]
assert (X + Y).to_ints() == [(x + y) % P for x, y in zip(a, b)]
assert (X - Y).to_ints() == [(x - y) % P for x, y in zip(a, b)]
assert (X * Y).to_ints() == [x * y % P for x, y in zip(a, b)]
assert (-X).to_ints() == [-x % P for x in a]
inv = X.inv().to_ints()
assert all((x * i) % P == (1 if x else 0) and (x or i == 0) for x, i in zip(a, inv))
assert (X / Y).to_ints() == [x * pow(y, P - 2, P) % P for x, y in zip(a, b)]
assert (X * 3 + 5).to_ints() == [(3 * x + 5) % P for x in a] and (7 - X).to_ints() == [(7 - x) % P for x in a]
assert ((X == Y) == np.array([x == y for x, y in zip(a, b)])).all() and X.is_zero().tolist() == [x == 0 for x in a]
rows = np.array([[rng.randrange(256) for _ in range(32)] for _ in range(300)], dtype=np.uint8)
r = A.FQ(rng.randrange(P))
got = A.linear_combine_bytes_batch(rows, r).to_ints()
assert got == [A.linear_combine_bytes(list(map(int, row)), r).n for row in rows]
# the gates over columns: rows that violate, first violated gate per row, selector
n = 64
v = [i % 3 for i in range(n)]
cs = BatchConstraintSystem(n)
assert cs.constrain_bool(v).tolist() == [x == 2 for x in v]
assert cs.constrain_equal(v, [0] * n).tolist() == [x != 0 for x in v]
assert cs.first_site.tolist() == [0 if x == 0 else (2 if x == 1 else 1) for x in v]
assert cs.range_check([1 << (8 * (i % 33)) for i in range(n)], 9).tolist() == [(i % 33) >= 9 for i in range(n)]
assert cs.is_equal(v, [1] * n).to_ints() == [int(x == 1) for x in v]
with cs.condition([int(x != 2) for x in v]) as c:
    assert not c.constrain_zero([int(x == 2) for x in v]).any()
try:
    cs.check(); raise SystemExit("check() must raise")
except AssertionError as e:
    assert "row 1" in e.args[0].message
print("frarray ok")
'''


def _run_child(backend_env):
    p = subprocess.run([sys.executable, "-c", FRARRAY_CHILD], env=dict(os.environ, ZK_ROOT=ROOT, **backend_env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600)
    assert p.returncode == 0 and b"frarray ok" in p.stdout, p.stdout.decode()[-3000:]


def test_frarray_through_zk_fr_op_cpu_backend():
    _run_child({"ZK_BACKEND": "cpu"})


@pytest.mark.gpu
def test_frarray_through_zk_fr_op_hip():
    _run_child({"ZK_BACKEND": "hip"})
