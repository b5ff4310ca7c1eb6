"""The one-shot C entries (`zk_*_verify`, `zk_keccak_table`, `zk_state_assign`, `zk_bytecode_assign`, `zk_ecdsa_verify`):
host buffers in, tally + per-row status out — exactly the calls INTEGRATION.md's reference-side stub makes.  The host
mirrors (`evm_circuit.verify_steps`, `state_circuit.verify_state_rows`, ...) go through these; `engine.py` is the session
form (upload once, many passes) for benchmarks and device-resident pipelines.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import ZkResult, check
from .engine import Result, _expect, _randomness_cells


def _p(x, n=1):
    return _lib.ptr(x) if (x is not None and n) else None


def _c(x, dtype=None):
    return None if x is None else np.ascontiguousarray(x, dtype=dtype)


def state_verify(rows, flags, mpt, device=None):
    """zk_state_verify -> (Result, status uint32[n])"""
    lib = _lib.init(device)
    rows, flags, mpt = _c(rows), _c(flags), _c(mpt)
    _expect(rows, "state rows", 8, (57, None, 4))
    _expect(flags, "state flags", 4, (rows.shape[1],))
    _expect(mpt, "mpt", 8, (None, 12, 4))
    n, m = int(rows.shape[1]), 0 if mpt is None else int(mpt.shape[0])
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_state_verify(_p(rows), _p(flags), n, _p(mpt, m), m, 0, _p(status), ctypes.byref(r)), "zk_state_verify")
    return Result(r), status


_EVM_CELLS = {"steps": 13, "rw": 14, "bytecode": 6, "tx": 5, "block": 4, "copy": 14, "keccak": 5, "exp": 11, "withdrawals": 4,
              "sig": 9, "ecc": 13, "aux": None}


def evm_verify(wire, begin_with_first_step=False, end_with_last_step=False, opts=0, device=None):
    """zk_evm_verify over a wire dict (flatten.flatten_evm) -> (Result, status uint32[n_steps - 1])"""
    lib = _lib.init(device)
    a = {k: _c(wire.get(k)) for k in list(_EVM_CELLS) + ["rw_flags", "tx_flags", "block_flags", "aux_kind"]}
    for k, nc in _EVM_CELLS.items():
        _expect(a[k], k, 8, (None, nc, 4))
    for k, of in (("rw_flags", "rw"), ("tx_flags", "tx"), ("block_flags", "block"), ("aux_kind", "steps")):
        if a[k] is not None and a[of] is not None:
            _expect(a[k], k, 4, (a[of].shape[0],))

    def rows(k):
        return 0 if a[k] is None else int(a[k].shape[0])

    def p(k, n=1):
        v = _p(a[k], n)
        return v.value if v is not None else None

    n_steps = rows("steps")
    t = _lib.ZkEvmTables(
        p("steps"), n_steps, p("rw", rows("rw")), p("rw_flags", rows("rw")), rows("rw"), p("bytecode", rows("bytecode")), rows("bytecode"),
        p("tx", rows("tx")), p("tx_flags", rows("tx")), rows("tx"), p("block", rows("block")), p("block_flags", rows("block")), rows("block"),
        int(bool(begin_with_first_step)), int(bool(end_with_last_step)), p("copy", rows("copy")), rows("copy"),
        p("keccak", rows("keccak")), rows("keccak"), p("exp", rows("exp")), rows("exp"),
        p("aux", rows("aux")), p("aux_kind", rows("aux")), p("withdrawals", rows("withdrawals")), rows("withdrawals"),
        p("sig", rows("sig")), rows("sig"), p("ecc", rows("ecc")), rows("ecc"), int(a["aux"].shape[1]) if rows("aux") else 0, 0)
    status, r = np.zeros(max(n_steps - 1, 1), dtype=np.uint32), ZkResult()
    check(lib.zk_evm_verify(ctypes.byref(t), int(opts), _p(status), ctypes.byref(r)), "zk_evm_verify")
    return Result(r), status[: n_steps - 1]


def bytecode_verify(rows, keccak, randomness, device=None):
    lib = _lib.init(device)
    rows, keccak = _c(rows), _c(keccak)
    rc = _c(_randomness_cells(randomness, None))
    _expect(rows, "bytecode rows", 8, (12, None, 4))
    _expect(keccak, "keccak", 8, (None, 5, 4))
    n, m = int(rows.shape[1]), 0 if keccak is None else int(keccak.shape[0])
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_bytecode_verify(_p(rows), n, _p(keccak, m), m, _p(rc), 0, _p(status), ctypes.byref(r)), "zk_bytecode_verify")
    return Result(r), status


def exp_verify(rows, device=None):
    lib = _lib.init(device)
    rows = _c(rows)
    _expect(rows, "exp rows", 8, (21, None, 4))
    n = int(rows.shape[1])
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_exp_verify(_p(rows), n, 0, _p(status), ctypes.byref(r)), "zk_exp_verify")
    return Result(r), status


def copy_verify(rows, row_flags, randomness, rw, rw_flags, bytecode, tx, tx_flags, opts=0, device=None):
    lib = _lib.init(device)
    rows, row_flags, rw, rw_flags, bytecode, tx, tx_flags = (_c(x) for x in (rows, row_flags, rw, rw_flags, bytecode, tx, tx_flags))
    rc = _c(_randomness_cells(randomness, None))
    _expect(rows, "copy rows", 8, (20, None, 4))
    _expect(row_flags, "copy row_flags", 4, (rows.shape[1],))
    _expect(rw, "rw", 8, (None, 14, 4))
    _expect(bytecode, "bytecode", 8, (None, 6, 4))
    _expect(tx, "tx", 8, (None, 5, 4))

    def nr(x):
        return 0 if x is None else int(x.shape[0])

    def p(x, n=1):
        v = _p(x, n)
        return v.value if v is not None else None

    n = int(rows.shape[1])
    t = _lib.ZkCopyTables(p(rows), p(row_flags), n, p(rc), p(rw, nr(rw)), p(rw_flags, nr(rw)), nr(rw), p(bytecode, nr(bytecode)),
                          nr(bytecode), p(tx, nr(tx)), p(tx_flags, nr(tx)), nr(tx))
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_copy_verify(ctypes.byref(t), int(opts), _p(status), ctypes.byref(r)), "zk_copy_verify")
    return Result(r), status


def sign_verify(wire, randomness, is_sig, device=None):
    lib = _lib.init(device)
    a = {k: _c(wire.get(k)) for k in ("bytes", "cells", "meta", "keccak", "tx_rows", "tx_flags")}
    rc = _c(_randomness_cells(randomness, None))
    n = int(a["bytes"].shape[0])
    _expect(a["bytes"], "sign bytes", 1, (None, 9, 32))
    _expect(a["cells"], "sign cells", 8, (8, n, 4))
    _expect(a["meta"], "sign meta", 4, (n, 4))
    _expect(a["keccak"], "keccak", 8, (None, 5, 4))
    _expect(a["tx_rows"], "tx_rows", 8, (None, 5, 4))

    def nr(x):
        return 0 if x is None else int(x.shape[0])

    def p(x, m=1):
        v = _p(x, m)
        return v.value if v is not None else None

    t = _lib.ZkSignUnits(p(a["bytes"]), p(a["cells"]), p(a["meta"]), n, p(rc), p(a["keccak"], nr(a["keccak"])), nr(a["keccak"]),
                         p(a["tx_rows"], nr(a["tx_rows"])), p(a["tx_flags"], nr(a["tx_rows"])), nr(a["tx_rows"]), int(bool(is_sig)))
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_sign_verify(ctypes.byref(t), 0, _p(status), ctypes.byref(r)), "zk_sign_verify")
    return Result(r), status


def keccak_table(data, offsets, randomness, mode=0, device=None):
    """zk_keccak_table -> (Result, status, rows uint64[n, 5, 4])"""
    lib = _lib.init(device)
    data, offsets = _c(data, np.uint8), _c(offsets, np.uint64)
    rc = _c(_randomness_cells(randomness, None))
    n, nb = int(offsets.shape[0]) - 1, int(data.shape[0])
    rows, status, r = np.zeros((n, 5, 4), dtype=np.uint64), np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_keccak_table(_p(data, nb), nb, _p(offsets), n, _p(rc), int(mode), _p(rows), 0, _p(status), ctypes.byref(r)),
          "zk_keccak_table")
    return Result(r), status, rows


def state_assign(ops, op_flags, device=None):
    """zk_state_assign -> (Result, status, rows uint64[57, n, 4], row_flags uint32[n], mpt uint64[m, 12, 4])"""
    lib = _lib.init(device)
    ops, op_flags = _c(ops), _c(op_flags)
    _expect(ops, "state ops", 8, (12, None, 4))
    n = int(ops.shape[1])
    _expect(op_flags, "op_flags", 4, (n,))
    rows, rflags = np.zeros((57, n, 4), dtype=np.uint64), np.zeros(n, dtype=np.uint32)
    mpt, n_mpt = np.zeros((n, 12, 4), dtype=np.uint64), ctypes.c_uint64()
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_state_assign(_p(ops), _p(op_flags), n, _p(rows), _p(rflags), _p(mpt), ctypes.byref(n_mpt), 0, _p(status),
                              ctypes.byref(r)), "zk_state_assign")
    return Result(r), status, rows, rflags, mpt[: int(n_mpt.value)]


def state_ops_from_rw(rw, rw_flags, device=None):
    """zk_state_ops_from_rw -> (Result, status uint32[n] per RW row, ops uint64[12, n_ops, 4], op_flags uint32[n_ops])"""
    lib = _lib.init(device)
    rw, rw_flags = _c(rw), _c(rw_flags, np.uint32)
    _expect(rw, "rw table", 8, (None, 14, 4))
    n = int(rw.shape[0])
    _expect(rw_flags, "rw_flags", 4, (n,))
    ops, flags = np.zeros(12 * (n + 1) * 4, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint32)
    status, r, n_ops = np.zeros(n, dtype=np.uint32), ZkResult(), ctypes.c_uint64()
    check(lib.zk_state_ops_from_rw(_p(rw), _p(rw_flags), n, _p(ops), _p(flags), ctypes.byref(n_ops), 0, _p(status), ctypes.byref(r)),
          "zk_state_ops_from_rw")
    m = int(n_ops.value)
    return Result(r), status, ops[: 48 * m].reshape(12, m, 4), flags[:m]


def state_verify_from_rw(rw, rw_flags, device=None):
    """zk_state_verify_from_rw -> (Result of the State circuit, status uint32[n_ops] per State row)"""
    lib = _lib.init(device)
    rw, rw_flags = _c(rw), _c(rw_flags, np.uint32)
    _expect(rw, "rw table", 8, (None, 14, 4))
    n = int(rw.shape[0])
    _expect(rw_flags, "rw_flags", 4, (n,))
    status, r, n_ops = np.zeros(n + 1, dtype=np.uint32), ZkResult(), ctypes.c_uint64()
    check(lib.zk_state_verify_from_rw(_p(rw), _p(rw_flags), n, 0, _p(status), ctypes.byref(n_ops), ctypes.byref(r)), "zk_state_verify_from_rw", lib)
    return Result(r), status[: int(n_ops.value)]


def bytecode_assign(in_rows, offsets, lengths, k, randomness, device=None):
    """zk_bytecode_assign -> (Result, rows uint64[12, 2^k, 4])"""
    lib = _lib.init(device)
    in_rows, offsets, lengths = _c(in_rows), _c(offsets, np.uint64), _c(lengths, np.uint64)
    rc = _c(_randomness_cells(randomness, None))
    _expect(in_rows, "unrolled bytecode rows", 8, (None, 6, 4))
    n_rows, n_codes = int(in_rows.shape[0]), int(lengths.shape[0])
    rows, r = np.zeros((12, 1 << int(k), 4), dtype=np.uint64), ZkResult()
    check(lib.zk_bytecode_assign(_p(in_rows, n_rows), n_rows, _p(offsets), _p(lengths, n_codes), n_codes, int(k), _p(rc), _p(rows), 0,
                                 ctypes.byref(r)), "zk_bytecode_assign")
    return Result(r), rows


def ecdsa_verify(sig_bytes, v=None, layout=0, v_stride=1, device=None):
    """zk_ecdsa_verify -> (Result, status uint32[n])"""
    lib = _lib.init(device)
    sig_bytes, v = _c(sig_bytes, np.uint8), _c(v, np.uint32)
    _expect(sig_bytes, "signature bytes", 1, (None, 5 if layout == 0 else 9, 32))
    n = int(sig_bytes.shape[0])
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_ecdsa_verify(_p(sig_bytes), int(layout), _p(v), int(v_stride), n, 0, _p(status), ctypes.byref(r)), "zk_ecdsa_verify")
    return Result(r), status


def copy_assign(events, flags, data, offsets, randomness, device=None):
    """zk_copy_assign -> (Result, rows uint64[20, n, 4], row_flags uint32[n], table uint64[m, 14, 4], rw uint64[k, 14, 4], rw_flags)"""
    from .engine import _copy_events_struct, copy_assign_sizes

    lib = _lib.init(device)
    events, flags, data, offsets = _c(events), _c(flags, np.uint32), _c(data, np.uint16), _c(offsets, np.uint64)
    rc = _c(_randomness_cells(randomness, None))
    _expect(events, "copy events", 8, (None, 12, 4))
    n_rows, n_table, n_rw = copy_assign_sizes(events, flags, data, offsets, device)
    rows, rf = np.zeros((20, n_rows, 4), dtype=np.uint64), np.zeros(n_rows, dtype=np.uint32)
    table = np.zeros((n_table, 14, 4), dtype=np.uint64)
    rw, rwf = np.zeros((n_rw, 14, 4), dtype=np.uint64), np.zeros(n_rw, dtype=np.uint32)
    t = _copy_events_struct(events, flags, data, offsets, rc)
    r = ZkResult()
    check(lib.zk_copy_assign(ctypes.byref(t), _p(rows), _p(rf), _p(table, n_table), _p(rw, n_rw), _p(rwf, n_rw), 0, ctypes.byref(r)), "zk_copy_assign")
    return Result(r), rows, rf, table, rw, rwf


def pi_verify(rows, keccak, gas, circuit_len, keccak_rand=255, byte_pow_base=255, device=None):
    """zk_pi_verify -> (Result, status uint32[n])"""
    lib = _lib.init(device)
    rows, keccak, gas = _c(rows), _c(keccak), _c(gas)
    kr, bp = _c(_randomness_cells(int(keccak_rand), None)), _c(_randomness_cells(int(byte_pow_base), None))
    _expect(rows, "pi rows", 8, (24, None, 4))
    _expect(keccak, "keccak", 8, (None, 5, 4))
    _expect(gas, "gas-cost table", 8, (None, 3, 4))
    n, m, k = int(rows.shape[1]), 0 if keccak is None else int(keccak.shape[0]), 0 if gas is None else int(gas.shape[0])
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_pi_verify(_p(rows), n, _p(keccak, m), m, _p(gas, k), k, int(circuit_len), _p(kr), _p(bp), 0, _p(status), ctypes.byref(r)),
          "zk_pi_verify")
    return Result(r), status


def pi_copy_verify(cells, data, lens, device=None):
    """zk_pi_copy_verify over cells uint64[n, 4], data uint8[n, 32], lens uint32[n] -> (Result, status uint32[n])"""
    lib = _lib.init(device)
    cells, data, lens = _c(cells), _c(data), _c(lens)
    _expect(cells, "pi copy cells", 8, (None, 4))
    _expect(data, "pi copy bytes", 1, (cells.shape[0], 32))
    _expect(lens, "pi copy lens", 4, (cells.shape[0],))
    n = int(cells.shape[0])
    status, r = np.zeros(n, dtype=np.uint32), ZkResult()
    check(lib.zk_pi_copy_verify(_p(cells), _p(data), _p(lens), n, 0, _p(status), ctypes.byref(r)), "zk_pi_copy_verify")
    return Result(r), status

