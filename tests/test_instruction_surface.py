"""`zkevm_specs_amd.instruction.Instruction`: the table-free half of the reference's Instruction
(src/zkevm_specs/evm_circuit/instruction.py:145-665) with its names and failure behaviour.  Everywhere: properties against Python
big-int arithmetic and the exception classes the reference documents.  In the build container (where the
reference can be imported through oracle/refshim): a differential run against the reference's own class on the same inputs."""
import os
import random
import subprocess
import sys

import pytest

from zkevm_specs_amd.arithmetic import FQ, P, Word
from zkevm_specs_amd.errors import ConstraintUnsatFailure
from zkevm_specs_amd.instruction import Instruction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M256 = (1 << 256) - 1
NASTY = [0, 1, 2, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, (1 << 255) - 1, 1 << 255, (1 << 255) + 1, M256, M256 - 1, 0xDEADBEEF << 100]


def _words(rng, k):
    return NASTY + [rng.getrandbits(rng.choice([8, 64, 128, 200, 256])) for _ in range(k)]


def test_asserts_and_selectors():
    I = Instruction()
    I.constrain_zero(FQ(0)); I.constrain_not_zero(FQ(3)); I.constrain_equal(FQ(P + 2), FQ(2)); I.constrain_bool(FQ(1)); I.constrain_in(FQ(5), [FQ(4), FQ(5)])
    I.constrain_zero_word(Word(0)); I.constrain_not_zero_word(Word(1 << 200)); I.constrain_equal_word(Word(77), Word(77)); I.constrain_in_word(Word(3), [Word(1), Word(3)])
    for bad in (lambda: I.constrain_zero(FQ(1)), lambda: I.constrain_not_zero(FQ(P)), lambda: I.constrain_equal(FQ(1), FQ(2)), lambda: I.constrain_bool(FQ(2)),
                lambda: I.constrain_in(FQ(6), [FQ(4)]), lambda: I.constrain_zero_word(Word(1 << 128)), lambda: I.constrain_not_zero_word(Word(0)),
                lambda: I.constrain_equal_word(Word(1), Word(2))):
        with pytest.raises(AssertionError) as e:
            bad()
        assert isinstance(e.value.args[0], ConstraintUnsatFailure)  # constrain_*: caught by verify_steps (main.py:36)
    assert I.is_zero(FQ(0)) == 1 and I.is_zero(FQ(9)) == 0 and I.is_equal(FQ(4), FQ(P + 4)) == 1 and I.is_zero_word(Word(0)) == 1 and I.is_zero_word(Word(1 << 128)) == 0
    assert I.is_equal_word(Word(5), Word(5)) == 1 and I.is_equal_word(Word(5), Word(5 + (1 << 128))) == 0
    assert [x.n for x in I.continuous_selectors(FQ(3), 5)] == [1, 1, 1, 0, 0] and I.pair_select(FQ(7), FQ(7), FQ(8)) == (1, 0)
    assert I.select(FQ(1), "a", "b") == "a" and I.select(FQ(0), "a", "b") == "b"
    with pytest.raises(AssertionError):
        I.select(FQ(2), 1, 2)
    assert I.compare(FQ(3), FQ(5), 4) == (1, 0) and I.compare(FQ(5), FQ(5), 4) == (0, 1) and I.min(FQ(3), FQ(5), 4) == 3 and I.max(FQ(3), FQ(5), 4) == 5
    with pytest.raises(AssertionError):
        I.compare(FQ(1 << 40), FQ(1), 4)
    assert I.constant_divmod(FQ(100), FQ(7), 1) == (14, 2) and I.is_u64_overflow(FQ(1 << 64)) == 1 and I.is_memory_overflow(FQ(0x1FFFFFFFE0)) == 0
    hit = []
    I.condition(FQ(1), lambda: hit.append(1)); I.condition(FQ(0), lambda: hit.append(2))
    assert hit == [1] and I.sum([FQ(1), 2, FQ(P - 1)]) == 2


def test_ranges_and_words():
    I = Instruction()
    assert I.range_check(FQ(0x1234), 2) == b"\x34\x12"
    with pytest.raises(ConstraintUnsatFailure):  # RAISED, not asserted (instruction.py:534): verify_steps does not swallow it
        I.range_check(FQ(1 << 16), 2)
    with pytest.raises(AssertionError):
        I.range_check(FQ(1), 32)
    w = Word(0xAABBCCDDEEFF00112233445566778899AABBCCDD)
    assert I.word_to_address(w) == w.int_value() and I.address_to_word(FQ(w.int_value())) == w and I.word_to_u64(Word(5)) == 5
    with pytest.raises(ConstraintUnsatFailure):
        I.word_to_u64(Word(1 << 64))
    with pytest.raises(AssertionError):
        I.address_to_word(FQ(1 << 160))
    assert I.byte_size(Word(0)) == 0 and I.byte_size(Word(0x1FF)) == 2 and I.byte_size(Word(M256)) == 32
    assert I.is_neg_word(Word(1 << 255)) == 1 and I.is_neg_word(Word((1 << 255) - 1)) == 0
    assert I.bytes_to_fq(b"\x01\x02") == 0x0201 and I.bytes_to_fq([FQ(1), FQ(2)]) == 0x0201


def test_word_arithmetic_against_bigints():
    I = Instruction()
    rng = random.Random(11)
    ws = _words(rng, 40)
    for a in ws:
        x_abs, neg = I.abs_word(Word(a))
        signed = a - (1 << 256) if a >> 255 else a
        assert neg == (1 if signed < 0 else 0) and x_abs.int_value() == abs(signed) % (1 << 256)
        for b in ws[::5]:
            d, borrow = I.sub_word(Word(a), Word(b))
            assert d.int_value() == (a - b) % (1 << 256) and borrow == (1 if a < b else 0)
            s, carry = I.add_words([Word(a), Word(b)])
            assert s.int_value() == (a + b) % (1 << 256) and carry == (a + b) >> 256
            lt, eq = I.compare_word(Word(a), Word(b))
            assert lt == (1 if a < b else 0) and eq == (1 if a == b else 0)
            for c in ws[::13]:
                full = a * b + c
                overflow = I.mul_add_words(Word(a), Word(b), Word(c), Word(full & M256))
                assert (overflow.n == 0) == (full >> 256 == 0)
                I.mul_add_words_512(Word(a), Word(b), Word(c), Word(full >> 256), Word(full & M256))
                if full & M256 != (full + 1) & M256:
                    # a wrong product: the field quotient by 2^128 is no nine-byte carry -> range_check RAISES
                    with pytest.raises((ConstraintUnsatFailure, AssertionError)) as e:
                        I.mul_add_words(Word(a), Word(b), Word(c), Word((full + 1) & M256))
                    assert e.type is ConstraintUnsatFailure
    k = (1 << 64) - 3
    assert I.mul_word_by_u64(Word(1 << 190), FQ(k)).int_value() == (k << 190)
    with pytest.raises(AssertionError):
        I.mul_word_by_u64(Word(M256), FQ(2))


REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference lives in the build container only")
def test_differential_against_the_reference_class():
    """same inputs through the reference's Instruction (child process: PYTHONPATH = oracle/refshim + the reference) and through the
    mirror; results and exception classes must agree"""
    script = r'''
import json, random, sys
sys.path.insert(0, sys.argv[1])
from zkevm_specs.evm_circuit.instruction import Instruction as R, ConstraintUnsatFailure as RC
from zkevm_specs.util.arithmetic import FQ as RFQ, Word as RW
from zkevm_specs_amd.instruction import Instruction as M
from zkevm_specs_amd.errors import ConstraintUnsatFailure as MC
from zkevm_specs_amd.arithmetic import FQ as MFQ, Word as MW
r, m = R(None, None, None, False, False), M()
rng = random.Random(5)
M256 = (1 << 256) - 1
nasty = [0, 1, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, (1 << 255) - 1, 1 << 255, M256]
def outcome(f):
    try:
        v = f()
    except AssertionError as e:
        return ("assert", type(e.args[0]).__name__ if e.args else "")
    except Exception as e:
        return ("raise", type(e).__name__)
    def norm(x):
        if isinstance(x, tuple): return [norm(y) for y in x]
        if hasattr(x, "int_value"): return ["W", x.int_value()]
        if hasattr(x, "n"): return x.n
        return x
    return ("ok", norm(v))
bad = 0
n = 0
for _ in range(400):
    a, b, c = (rng.choice(nasty + [rng.getrandbits(rng.choice([64, 128, 256]))]) for _ in range(3))
    d = (a * b + c + rng.choice([0, 0, 0, 1, 1 << 128])) & M256
    cases = [
        (lambda I, F, W: I.mul_add_words(W(a), W(b), W(c), W(d))),
        (lambda I, F, W: I.mul_add_words_512(W(a), W(b), W(c), W(((a * b + c) >> 256) & M256), W(d))),
        (lambda I, F, W: I.abs_word(W(a))), (lambda I, F, W: I.sub_word(W(a), W(b))), (lambda I, F, W: I.compare_word(W(a), W(b))),
        (lambda I, F, W: I.word_to_fq(W(a), rng_n)), (lambda I, F, W: I.range_check(F(a % (1 << 250)), rng_n)),
        (lambda I, F, W: I.mul_word_by_u64(W(a), F(b & ((1 << 64) - 1)))), (lambda I, F, W: I.address_to_word(F(a % (1 << 170)))),
        (lambda I, F, W: I.constant_divmod(F(a % (1 << 200)), F((b % 1000) + 1), rng_n)), (lambda I, F, W: I.is_neg_word(W(a))),
        (lambda I, F, W: I.byte_size(W(a))), (lambda I, F, W: I.is_equal_word(W(a), W(b))), (lambda I, F, W: I.compare(F(a % (1 << 100)), F(b % (1 << 130)), 14)),
    ]
    for f in cases:
        rng_n = rng.choice([1, 8, 9, 16, 20, 31])
        x, y = outcome(lambda: f(r, RFQ, RW)), outcome(lambda: f(m, MFQ, MW))
        n += 1
        if x != y:
            bad += 1
            if bad < 5: print("MISMATCH", x, y)
print(json.dumps({"cases": n, "bad": bad}))
'''
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "oracle", "refshim"), REF, ROOT]), PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run([sys.executable, "-c", script, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    import json

    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["bad"] == 0 and res["cases"] >= 5000, p.stdout[-2000:]
