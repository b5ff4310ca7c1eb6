"""Bytecode- and Exp-circuit oracles (TEST INFRASTRUCTURE — see oracle/__init__.py).

Python-integer restatements of `check_bytecode_row` (reference src/zkevm_specs/bytecode_circuit.py:
37-100) and `exp_circuit.verify_step` (src/zkevm_specs/exp_circuit.py:14-85, evaluated through
`ConstraintSystem`, util/constraint_system.py:12-74) over the flattened wire rows documented in
csrc/row_circuits.hpp; site numbers are the kernels'.  Pinned to the reference by
oracle/gen_golden_rows.py (tests/golden/bytecode_cases.npz, exp_cases.npz).
"""
from .codes import ASSERT, CONSTRAINT, OK, OVERFLOW_ERROR, Fail
from .wire import P

EMPTY_HASH_LO = 0xE500B653CA82273B7BFAD8045D85A470
EMPTY_HASH_HI = 0xC5D2460186F7233C927E7DB2DCC703C0
M128 = (1 << 128) - 1
INV_2P128 = pow(1 << 128, -1, P)


def _a(cond, site):
    if not cond:
        raise Fail(ASSERT, site)


# ---- Bytecode circuit --------------------------------------------------------------------------
(Q_FIRST, Q_LAST, HASH_LO, HASH_HI, TAG, INDEX, VALUE, IS_CODE, PUSH_LEFT, VALUE_RLC, LENGTH, PUSH_SIZE) = range(12)


def _push_size(v):
    return v - 0x5F if 0x60 <= v <= 0x7F else 0


def bytecode_check_row(rows, i, keccak_set, r):
    cur, nxt = rows[i], rows[(i + 1) % len(rows)]
    try:
        hdr_hdr_len = cur[LENGTH] == 0
        hdr_hdr_hash = cur[HASH_LO] == EMPTY_HASH_LO and cur[HASH_HI] == EMPTY_HASH_HI
        if cur[Q_FIRST] == 1:
            _a(cur[TAG] == 1, 1)
        if cur[Q_LAST] == 0:
            if cur[TAG] == 1:
                _a(cur[VALUE] == cur[LENGTH], 2)
                _a(cur[INDEX] == 0, 3)
                if nxt[TAG] == 2:
                    _a(nxt[LENGTH] == cur[LENGTH], 4)
                    _a(nxt[INDEX] == 0, 5)
                    _a(nxt[IS_CODE] == 1, 6)
                    _a(nxt[HASH_LO] == cur[HASH_LO] and nxt[HASH_HI] == cur[HASH_HI], 7)
                    _a(nxt[VALUE_RLC] == nxt[VALUE], 8)
                if nxt[TAG] == 1:
                    _a(hdr_hdr_len, 9)
                    _a(hdr_hdr_hash, 10)
            if cur[TAG] == 2:
                _a(cur[VALUE] <= 255 and cur[PUSH_SIZE] == _push_size(cur[VALUE]), 11)
                _a(cur[IS_CODE] == int(cur[PUSH_LEFT] == 0), 12)
                if nxt[TAG] == 2:
                    _a(nxt[LENGTH] == cur[LENGTH], 13)
                    _a(nxt[INDEX] == (cur[INDEX] + 1) % P, 14)
                    _a(nxt[HASH_LO] == cur[HASH_LO] and nxt[HASH_HI] == cur[HASH_HI], 15)
                    _a(nxt[VALUE_RLC] == (cur[VALUE_RLC] * r + nxt[VALUE]) % P, 16)
                    if cur[IS_CODE] == 1:
                        _a(nxt[PUSH_LEFT] == cur[PUSH_SIZE], 17)
                    else:
                        _a(nxt[PUSH_LEFT] == (cur[PUSH_LEFT] - 1) % P, 18)
                if nxt[TAG] == 1:
                    _a((cur[INDEX] + 1) % P == cur[LENGTH], 19)
                    _a((2, cur[VALUE_RLC], cur[LENGTH], cur[HASH_LO], cur[HASH_HI]) in keccak_set, 20)
        if cur[Q_LAST] == 1:
            _a(cur[TAG] == 1, 21)
            _a(hdr_hdr_len, 22)
            _a(hdr_hdr_hash, 23)
    except Fail as f:
        return f.code
    return OK


def bytecode_verify_rows(rows, keccak_rows, r):
    ks = set(tuple(k) for k in keccak_rows)
    return [bytecode_check_row(rows, i, ks, r) for i in range(len(rows))]


# ---- Exp circuit ---------------------------------------------------------------------------------
(X_Q_USABLE, X_IS_STEP, X_ID, X_IS_LAST, X_BASE, _b1, X_EXPONENT, _e1, X_EXPN, _x1, X_A, _a1, X_B, _b2, X_C, _c1, X_D, _d1,
 X_Q, _q1, X_R) = range(21)


def _carries(a, b, c, d):
    av, bv = a[0] | (a[1] << 128), b[0] | (b[1] << 128)
    a64 = [(av >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    b64 = [(bv >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    t0 = a64[0] * b64[0]
    t1 = a64[0] * b64[1] + a64[1] * b64[0]
    t2 = a64[0] * b64[2] + a64[1] * b64[1] + a64[2] * b64[0]
    t3 = a64[0] * b64[3] + a64[1] * b64[2] + a64[2] * b64[1] + a64[3] * b64[0]
    clo = (t0 + (t1 << 64) + c[0] - d[0]) * INV_2P128 % P
    chi = (t2 + (t3 << 64) + c[1] + clo - d[1]) * INV_2P128 % P
    return clo, chi


def exp_check_row(rows, i):
    r0, r1 = rows[i], rows[(i + 1) % len(rows)]
    W = lambda r, c: (r[c], r[c + 1])  # noqa: E731
    try:
        is_step, is_last, rr = r0[X_IS_STEP], r0[X_IS_LAST], r0[X_R]
        c1 = is_step * (1 - is_last) % P
        c2 = is_step
        c3 = c1 * rr % P
        c4 = c1 * (1 - rr) % P
        c5 = is_last

        def zero(cond, ok, site):
            _a(cond == 0 or ok, site)

        zero(c1, W(r0, X_BASE) == W(r1, X_BASE), 1)
        zero(c1, W(r0, X_A) == W(r1, X_D), 2)
        zero(c1, r0[X_ID] == r1[X_ID], 3)
        _a(c2 * is_last % P in (0, 1), 4)
        _a(c2 * rr % P in (0, 1), 5)
        a, b, c, d, q = W(r0, X_A), W(r0, X_B), W(r0, X_C), W(r0, X_D), W(r0, X_Q)
        if a[0] > M128 or a[1] > M128:
            raise Fail(OVERFLOW_ERROR, 6)
        if b[0] > M128 or b[1] > M128:
            raise Fail(OVERFLOW_ERROR, 7)
        clo, chi = _carries(a, b, c, d)
        if clo >= 256**9:
            raise Fail(CONSTRAINT, 8)
        if chi >= 256**9:
            raise Fail(CONSTRAINT, 9)
        zero(c2, W(r0, X_EXPN) == d, 12)
        zero(c2, c == (0, 0), 13)
        _a(rr <= M128, 15)
        if q[0] > M128 or q[1] > M128:
            raise Fail(OVERFLOW_ERROR, 17)
        clo, chi = _carries((2, 0), q, (rr, 0), W(r0, X_EXPONENT))
        if clo >= 256**9:
            raise Fail(CONSTRAINT, 18)
        if chi >= 256**9:
            raise Fail(CONSTRAINT, 19)
        e0, e1 = W(r0, X_EXPONENT), W(r1, X_EXPONENT)
        zero(c3, e1[0] == (e0[0] - 1) % P, 22)
        zero(c3, e1[1] == e0[1], 23)
        zero(c3, W(r0, X_BASE) == b, 24)
        zero(c4, e1[0] == q[0], 25)
        zero(c4, e1[1] == q[1], 26)
        zero(c4, a == b, 27)
        zero(c5, e0[0] == 2, 28)
        zero(c5, e0[1] == 0, 29)
        zero(c5, W(r0, X_BASE) == a, 30)
        zero(c5, W(r0, X_BASE) == b, 31)
    except Fail as f:
        return f.code
    return OK


def exp_verify_rows(rows):
    return [exp_check_row(rows, i) for i in range(len(rows))]
