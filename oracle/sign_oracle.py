"""Tx / Sig circuit oracle (TEST INFRASTRUCTURE — see oracle/__init__.py).

Python-integer restatement of `SignVerifyChip.verify` + the copy constraints of
`tx_circuit.verify_circuit` (reference src/zkevm_specs/tx_circuit.py:205-291) and of
`sig_circuit.Row.verify` (src/zkevm_specs/sig_circuit.py:64-104) over the flattened unit records
(layout and site numbers: csrc/sign_circuit.hpp).  Pinned by oracle/gen_golden_sign.py.
"""
from .codes import ASSERT, INDEX_ERROR, OK, Fail
from .wire import P


def _a(cond, site):
    if not cond:
        raise Fail(ASSERT, site)


def check_unit(u, keccak_set, r, is_sig, tx_rows=None, tx_flags=None, index=0):
    """u: dict with bytes fields (pk_x, pk_y, e_pk_x, e_pk_y, msg, e_msg, pk_hash, e_sig_r, e_sig_s), int cells
    (address, msg_lo, msg_hi, sig_v, sig_r_lo, sig_r_hi, sig_s_lo, sig_s_hi) and ecdsa_status / expect_valid."""
    try:
        is_np = 1 if is_sig else int(u["address"] != 0)
        bad = u.get("malformed", 0)
        _a(not (bad & 0x5) and u["pk_x"] == u["e_pk_x"], 1)
        _a(not (bad & 0xA) and u["pk_y"] == u["e_pk_y"], 2)
        _a(not (bad & 0x30) and u["msg"] == u["e_msg"], 3)
        if is_sig:
            _a(u["sig_r_lo"] + (u["sig_r_hi"] << 128) == int.from_bytes(u["e_sig_r"], "little"), 12)
            _a(u["sig_s_lo"] + (u["sig_s_hi"] << 128) == int.from_bytes(u["e_sig_s"], "little"), 13)
            _a(u["sig_v"] in (0, 1), 14)
        # RLC(reversed(pk_bytes), r, 64) with pk_bytes = reversed(pk_x) + reversed(pk_y): its little-endian
        # byte string is pk_y followed by pk_x (tx_circuit.py:216-223, util/arithmetic.py:77-87)
        le = bytes(u["pk_y"]) + bytes(u["pk_x"])
        acc = 0
        for b in reversed(le):
            acc = (acc * r + b) % P
        h = bytes(u["pk_hash"])
        h_lo, h_hi = int.from_bytes(h[:16], "little"), int.from_bytes(h[16:], "little")
        _a(not (bad & 0x40) and (is_np, is_np * acc % P, is_np * 64, is_np * h_lo, is_np * h_hi) in keccak_set, 4)
        _a(int.from_bytes(h[-20:], "big") == u["address"], 5)
        m = bytes(u["msg"])
        _a((is_np * int.from_bytes(m[:16], "little"), is_np * int.from_bytes(m[16:], "little")) == (u["msg_lo"], u["msg_hi"]), 6)
        st = u["ecdsa_status"]
        if is_sig:
            if st >= 2:
                raise Fail(st >> 24, 7)
            _a((st == 0) == bool(u["expect_valid"]), 15)
            return OK
        if st == 1:
            raise Fail(ASSERT, 7)
        if st >= 2:
            raise Fail(st >> 24, 7)
        caller, sign = index * 12 + 3, index * 12 + 11
        if caller >= len(tx_rows):
            raise Fail(INDEX_ERROR, 8)
        _a(not (tx_flags[caller] & 1), 8)
        _a(tx_rows[caller][3] == u["address"], 9)
        if sign >= len(tx_rows):
            raise Fail(INDEX_ERROR, 10)
        _a(tx_rows[sign][3] == u["msg_lo"], 10)
        _a(tx_rows[sign][4] == u["msg_hi"], 11)
    except Fail as f:
        return f.code
    return OK


FIELDS_BYTES = ("pk_x", "pk_y", "e_pk_x", "e_pk_y", "msg", "e_msg", "pk_hash", "e_sig_r", "e_sig_s")
FIELDS_CELLS = ("address", "msg_lo", "msg_hi", "sig_v", "sig_r_lo", "sig_r_hi", "sig_s_lo", "sig_s_hi")


def units_from_wire(bytes_arr, cells, meta):
    from .wire import colmajor_to_rows

    rows = colmajor_to_rows(cells)
    out = []
    for i in range(bytes_arr.shape[0]):
        u = {k: bytes(bytes_arr[i, j].tolist()) for j, k in enumerate(FIELDS_BYTES)}
        u.update({k: rows[i][j] for j, k in enumerate(FIELDS_CELLS)})
        u["ecdsa_status"], u["expect_valid"], u["malformed"] = int(meta[i, 0]), int(meta[i, 1]), int(meta[i, 2])
        out.append(u)
    return out


def verify_units(bytes_arr, cells, meta, keccak_rows, r, is_sig, tx_rows=None, tx_flags=None):
    ks = set(tuple(k) for k in keccak_rows)
    units = units_from_wire(bytes_arr, cells, meta)
    return [check_unit(u, ks, r, is_sig, tx_rows, tx_flags, i) for i, u in enumerate(units)]
