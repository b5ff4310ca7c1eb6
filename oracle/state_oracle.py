"""State-circuit oracle (TEST INFRASTRUCTURE — see oracle/__init__.py).

Python-integer restatement of `check_state_row` and its per-tag helpers
(reference src/zkevm_specs/state_circuit.py:188-613) over the flattened wire rows
(57 integer cells + type bits, layout documented in csrc/state_circuit.hpp).  Each check
carries the same site number as the HIP kernel so tests can compare full status codes.
Pinned against the reference itself by oracle/gen_golden.py (tests/golden/state_*.npz).
"""
from .codes import ASSERT, LOOKUP_UNSAT, VALUE_ERROR, Fail, OK
from .wire import P

RWC, IS_WRITE, TAG, ID, ADDR, FIELD_TAG, KEY_LO, KEY_HI = range(8)
LIMB0, BYTE0 = 8, 18
VAL_LO, VAL_HI, INIT_LO, INIT_HI, ROOT_LO, ROOT_HI, LEX = 50, 51, 52, 53, 54, 55, 56
NCELLS = 57


def _a(cond, site):
    if not cond:
        raise Fail(ASSERT, site)


def _pack(r):
    """keys_rwc_to_limbs_in_order (state_circuit.py:552-565): big-int packing, low 31 limbs."""
    key_bytes = r[BYTE0 : BYTE0 + 32]
    if any(b > 255 for b in key_bytes):  # bytes(...) would raise ValueError
        return None
    v = r[TAG]
    v = v * 2**28 + r[ID]
    v = v * 2**160 + r[ADDR]
    v = v * 2**16 + r[FIELD_TAG]
    v = v * 2**32 + int.from_bytes(bytes(key_bytes), "little")
    v = v * 2**32 + r[RWC]
    return v & ((1 << 496) - 1)


def _keys_eq(a, b):
    return a[TAG : KEY_HI + 1] == b[TAG : KEY_HI + 1]


def check_row(rows, flags, i, mpt_set):
    """Status code of row i (prev/next wrap modulo n: tests/test_state_circuit.py:27-28)."""
    try:
        _check(rows, flags, i, mpt_set)
    except Fail as f:
        return f.code
    return OK


def _check(rows, flags, i, mpt_set):
    n = len(rows)
    r, rp, rn = rows[i], rows[(i - 1) % n], rows[(i + 1) % n]
    val_word, init_word = bool(flags[i] & 1), bool(flags[i] & 2)
    tag = r[TAG]
    # 0.0 (:498-502)
    _a(1 <= tag <= 12, 1)
    _a(r[ID] <= 2**28 - 1, 2)
    _a(r[FIELD_TAG] <= 24, 3)
    # 0.1 (:505-509)
    lc = 0
    for k in range(10):
        _a(r[LIMB0 + k] <= 65535, 4)
        lc += r[LIMB0 + k] << (16 * k)
    _a(r[ADDR] == lc % P, 5)
    # 0.2 (:512-517)
    for b in range(32):
        _a(r[BYTE0 + b] <= 255, 6)
    lo = int.from_bytes(bytes(r[BYTE0 : BYTE0 + 16]), "little")
    hi = int.from_bytes(bytes(r[BYTE0 + 16 : BYTE0 + 32]), "little")
    _a(r[KEY_LO] == lo and r[KEY_HI] == hi, 7)
    # 0.3 (:520)
    _a(r[IS_WRITE] in (0, 1), 8)
    is_read = r[IS_WRITE] == 0
    # 0.4 (:552-570)
    kp = _pack(rp)
    if kp is None:
        raise Fail(VALUE_ERROR, 9)
    kc = _pack(r)
    if tag != 1:
        _a(kp < kc, 10)
    keq = _keys_eq(r, rp)
    # 0.5 (:577-581)
    if is_read and keq:
        _a(r[VAL_LO] == rp[VAL_LO] and r[VAL_HI] == rp[VAL_HI], 11)
    if keq:
        _a(r[INIT_LO] == rp[INIT_LO] and r[INIT_HI] == rp[INIT_HI], 12)
    # 8 (:584-585)
    if tag != 1:
        _a(r[RWC] != 0, 13)

    key_zero = r[KEY_LO] == 0 and r[KEY_HI] == 0
    root_same = r[ROOT_LO] == rp[ROOT_LO] and r[ROOT_HI] == rp[ROOT_HI]
    val_zero = r[VAL_LO] == 0 and r[VAL_HI] == 0
    init_zero = r[INIT_LO] == 0 and r[INIT_HI] == 0

    if tag == 1:  # Start (:216-236)
        _a(r[FIELD_TAG] == 0, 20)
        _a(r[ADDR] == 0, 21)
        _a(r[ID] == 0, 22)
        _a(key_zero, 23)
        _a(r[VAL_HI] == 0, 24)
        _a(r[INIT_HI] == 0, 25)
        _a(r[LEX] * ((r[RWC] - rp[RWC] - 1) % P) % P == 0, 26)
        _a(not val_word, 27)
        _a(r[VAL_LO] == 0, 28)
        _a(not init_word, 29)
        _a(r[INIT_LO] == 0, 30)
        if r[LEX] != 0:
            _a(root_same, 31)
    elif tag == 2:  # Memory (:240-266)
        _a(r[FIELD_TAG] == 0, 40)
        _a(key_zero, 41)
        _a(r[VAL_HI] == 0, 42)
        _a(r[INIT_HI] == 0, 43)
        if not keq and is_read:
            _a(not val_word, 44)
            _a(r[VAL_LO] == 0, 45)
        _a(r[ADDR] <= 2**32 - 1, 46)
        _a(not val_word, 47)
        _a(r[VAL_LO] <= 255, 48)
        _a(not init_word, 49)
        _a(r[INIT_LO] == 0, 50)
        _a(root_same, 51)
    elif tag == 3:  # Stack (:270-301)
        _a(r[FIELD_TAG] == 0, 60)
        _a(key_zero, 61)
        if not keq:
            _a(r[IS_WRITE] == 1, 62)
        _a(r[ADDR] <= 1023, 63)
        if tag == rp[TAG] and r[ID] == rp[ID]:
            _a((r[ADDR] - rp[ADDR]) % P <= 1, 64)
        _a(init_zero, 65)
        _a(root_same, 66)
    elif tag in (4, 6):  # Storage (:305-324) / Account (:349-380)
        if tag == 4:
            _a(r[FIELD_TAG] == 0, 70)
            proof = 4 if (val_zero and init_zero) else 6
        else:
            if not 1 <= r[FIELD_TAG] <= 4:
                raise Fail(VALUE_ERROR, 90)
            _a(r[ID] == 0, 91)
            _a(key_zero, 92)
            if r[FIELD_TAG] == 1:
                _a(r[VAL_HI] == 0, 93)
                _a(r[INIT_HI] == 0, 94)
            proof = 4 if (val_zero and init_zero and r[FIELD_TAG] == 3) else r[FIELD_TAG]
        if not _keys_eq(r, rn):
            q = (r[ADDR], proof, r[KEY_LO], r[KEY_HI], r[ROOT_LO], r[ROOT_HI], rp[ROOT_LO],
                 rp[ROOT_HI], r[VAL_LO], r[VAL_HI], r[INIT_LO], r[INIT_HI])
            if q not in mpt_set:
                raise Fail(LOOKUP_UNSAT, 71 if tag == 4 else 95)
        else:
            _a(root_same, 73 if tag == 4 else 97)
    elif tag == 5:  # CallContext (:328-345)
        _a(r[ADDR] == 0, 80)
        _a(key_zero, 81)
        _a(r[FIELD_TAG] <= 24, 82)
        if not keq and is_read:
            _a(not val_word, 83)
            _a(r[VAL_LO] == 0, 84)
        _a(init_zero, 85)
        _a(root_same, 86)
    elif tag == 7:  # TxRefund (:387-402)
        _a(r[ADDR] == 0, 100)
        _a(r[FIELD_TAG] == 0, 101)
        _a(key_zero, 102)
        _a(root_same, 103)
        _a(init_zero, 104)
        if not keq and is_read:
            _a(val_zero, 105)
    elif tag == 8:  # TxAccessListAccount (:406-419)
        _a(r[FIELD_TAG] == 0, 110)
        _a(key_zero, 111)
        _a(r[VAL_HI] == 0, 112)
        _a(r[INIT_HI] == 0, 113)
        _a(root_same, 114)
        if not keq and is_read:
            _a(not val_word, 115)
            _a(r[VAL_LO] == 0, 116)
    elif tag == 9:  # TxAccessListAccountStorage (:423-435)
        _a(r[FIELD_TAG] == 0, 120)
        _a(r[VAL_HI] == 0, 121)
        _a(r[INIT_HI] == 0, 122)
        _a(root_same, 123)
        if not keq and is_read:
            _a(not val_word, 124)
            _a(r[VAL_LO] == 0, 125)
    elif tag == 10:  # TxLog (:439-453)
        if r[FIELD_TAG] != 2:
            _a(r[VAL_HI] == 0, 130)
            _a(r[INIT_HI] == 0, 131)
        _a(r[IS_WRITE] == 1, 132)
        _a(root_same, 133)
    elif tag == 11:  # TxReceipt (:460-488)
        _a(r[ADDR] == 0, 140)
        _a(key_zero, 141)
        _a(r[VAL_HI] == 0, 142)
        _a(r[INIT_HI] == 0, 143)
        if r[FIELD_TAG] == 1:
            _a(not val_word, 144)
            _a(r[VAL_LO] in (0, 1), 145)
        same_tag = tag == rp[TAG]
        if r[ID] != rp[ID] and same_tag:
            _a(r[ID] == (rp[ID] + 1) % P, 146)
            if r[FIELD_TAG] == 2:
                _a(not val_word, 147)
                _a(not (flags[(i - 1) % n] & 1), 148)
                _a(r[VAL_LO] > rp[VAL_LO], 149)
        if not same_tag:
            _a(r[ID] == 1, 150)
        _a(1 <= r[ID] <= 2**11, 151)
        _a(root_same, 152)
    else:  # tag == 12 (:613)
        raise Fail(VALUE_ERROR, 160)


def verify_rows(rows, flags, mpt_rows):
    """Evaluate every row independently; returns list of status codes."""
    mpt_set = set(tuple(m) for m in mpt_rows)
    return [check_row(rows, flags, i, mpt_set) for i in range(len(rows))]
