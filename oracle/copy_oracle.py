"""Copy-circuit oracle (TEST INFRASTRUCTURE — see oracle/__init__.py).

Python-integer restatement of `verify_row`, `verify_step` and the lookups of `verify_copy_table`
(reference src/zkevm_specs/copy_circuit.py:16-130) over the flattened wire rows (layout and site
numbers: csrc/copy_circuit.hpp).  Pinned to the reference by oracle/gen_golden_copy.py.
"""
from .codes import ASSERT, LOOKUP_AMBIGUOUS, LOOKUP_UNSAT, OK, Fail
from .wire import P

(Q_STEP, IS_FIRST, IS_LAST, ID_LO, ID_HI, TAG, ADDR, SRC_END, BYTES_LEFT, VALUE, RLC_ACC, IS_CODE, IS_PAD, RWC, INC_LEFT,
 IS_MEMORY, IS_BYTECODE, IS_TX_CALLDATA, IS_TX_LOG, IS_RLC_ACC) = range(20)


def _a(cond, site):
    if not cond:
        raise Fail(ASSERT, site)


def _zero(cond, ok, site):
    _a(cond % P == 0 or ok, site)


class CopyTables:
    def __init__(self, rw, rw_flags, bytecode, tx, tx_flags):
        self.rw = [tuple(r) for r in rw]
        self.rw_flags = list(rw_flags)
        self.bytecode = [tuple(r) for r in bytecode]
        self.tx = [tuple(r) for r in tx]
        self.tx_flags = list(tx_flags)
        self.rw_idx, self.bc_idx, self.tx_idx = {}, {}, {}
        for i, r in enumerate(self.rw):
            self.rw_idx.setdefault(r[0], []).append(i)
        for i, r in enumerate(self.bytecode):
            self.bc_idx.setdefault(r[:4], []).append(i)
        for i, r in enumerate(self.tx):
            self.tx_idx.setdefault(r[:3], []).append(i)


def _lookup(rows, cands, query, site):
    first = None
    for i in cands:
        r = rows[i]
        if all(r[c] == v for c, v in query):
            if first is None:
                first = i
            elif rows[first] != r:
                raise Fail(LOOKUP_AMBIGUOUS, site)
    if first is None:
        raise Fail(LOOKUP_UNSAT, site)
    return first


def check_row(rows, flags, i, T, r):
    n = len(rows)
    r0, r1, r2 = rows[i], rows[(i + 1) % n], rows[(i + 2) % n]
    try:
        _a(r0[IS_FIRST] in (0, 1), 1)
        _a(r0[IS_LAST] in (0, 1), 2)
        _zero(1 - r0[Q_STEP], r0[IS_FIRST] == 0, 3)
        _zero(r0[Q_STEP], r0[IS_LAST] == 0, 4)
        _a(r0[IS_MEMORY] == int(r0[TAG] == 2), 5)
        _a(r0[IS_BYTECODE] == int(r0[TAG] == 1), 6)
        _a(r0[IS_TX_CALLDATA] == int(r0[TAG] == 3), 7)
        _a(r0[IS_TX_LOG] == int(r0[TAG] == 4), 8)
        _a(r0[IS_RLC_ACC] == int(r0[TAG] == 5), 9)
        c = 1 - (r0[IS_LAST] + r1[IS_LAST])
        _zero(c, r0[ID_LO] == r2[ID_LO] and r0[ID_HI] == r2[ID_HI], 10)
        _zero(c, r0[TAG] == r2[TAG], 11)
        _zero(c, (r0[ADDR] + 1) % P == r2[ADDR], 12)
        _zero(c, r0[SRC_END] == r2[SRC_END], 13)
        rw_diff = (1 - r0[IS_PAD]) * (r0[IS_MEMORY] + r0[IS_TX_LOG]) % P
        c = 1 - r0[IS_LAST]
        _zero(c, (r0[RWC] + rw_diff) % P == r1[RWC], 14)
        _zero(c, (r0[INC_LEFT] - rw_diff) % P == r1[INC_LEFT], 15)
        _zero(c, r0[RLC_ACC] == r1[RLC_ACC], 16)
        _zero(r0[IS_LAST], r0[INC_LEFT] == rw_diff, 17)
        _zero(r0[IS_LAST] * r0[IS_RLC_ACC], r0[RLC_ACC] == r0[VALUE], 18)
        q = r0[Q_STEP]
        _zero(q, (r1[IS_LAST] * (1 - r0[BYTES_LEFT])) % P == 0, 19)
        _zero(q, ((1 - r1[IS_LAST]) * (r0[BYTES_LEFT] - r2[BYTES_LEFT] - 1)) % P == 0, 20)
        _zero(q, (r0[IS_PAD] * r0[VALUE]) % P == 0, 21)
        if r0[IS_TX_LOG] == 0:
            _a(r0[ADDR] < 256**5 and r0[SRC_END] < 256**5, 22)
            lt = int(r0[ADDR] < r0[SRC_END])
            _zero(q, (1 - lt) % P == r0[IS_PAD], 23)
        _zero(q, r1[IS_PAD] == 0, 24)
        _zero(q * (1 - r1[IS_RLC_ACC]), r0[VALUE] == r1[VALUE], 25)
        _zero(q * r0[IS_FIRST], r0[VALUE] == r1[VALUE], 26)
        _zero((1 - q) * (1 - r0[IS_LAST]) * r0[IS_RLC_ACC], r2[VALUE] == (r0[VALUE] * r + r1[VALUE]) % P, 27)
        id_is_word = bool(flags[i] & 1)
        if r0[IS_MEMORY] == 1 and r0[IS_PAD] == 0:
            _a(not id_is_word, 28)
            k = _lookup(T.rw, T.rw_idx.get(r0[RWC], ()), [(0, r0[RWC]), (1, (1 - q) % P), (2, 9), (3, r0[ID_LO]), (4, r0[ADDR])], 29)
            _a(not (T.rw_flags[k] & 1), 30)
            _a(T.rw[k][8] == r0[VALUE], 31)
        if r0[IS_BYTECODE] == 1 and r0[IS_PAD] == 0:
            key = (r0[ID_LO], r0[ID_HI], 2, r0[ADDR])
            k = _lookup(T.bytecode, T.bc_idx.get(key, ()), [(0, key[0]), (1, key[1]), (2, 2), (3, key[3]), (4, r0[IS_CODE])], 32)
            _a(T.bytecode[k][5] == r0[VALUE], 34)
        if r0[IS_TX_CALLDATA] == 1 and r0[IS_PAD] == 0:
            _a(not id_is_word, 35)
            key = (r0[ID_LO], 13, r0[ADDR])
            k = _lookup(T.tx, T.tx_idx.get(key, ()), [(0, key[0]), (1, 13), (2, key[2])], 36)
            _a(not (T.tx_flags[k] & 1), 37)
            _a(T.tx[k][3] == r0[VALUE], 38)
        if r0[IS_TX_LOG] == 1:
            _a(not id_is_word, 39)
            k = _lookup(T.rw, T.rw_idx.get(r0[RWC], ()), [(0, r0[RWC]), (1, 1), (2, 10), (3, r0[ID_LO]), (4, r0[ADDR])], 40)
            _a(not (T.rw_flags[k] & 1), 41)
            _a(T.rw[k][8] == r0[VALUE], 42)
    except Fail as f:
        return f.code
    return OK


def verify_rows(rows, flags, T, r):
    return [check_row(rows, flags, i, T, r) for i in range(len(rows))]
