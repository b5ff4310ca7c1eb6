"""Session objects over the C ABI: upload once, launch many passes, collect the tally."""
import ctypes

import numpy as np

from . import _lib
from ._lib import OPT_DEVICE_PTRS, ZkResult, check


class Result:
    """Outcome of one evaluation pass over all rows (mirrors zk_result)."""

    def __init__(self, r: ZkResult):
        self.fail_count = int(r.fail_count)
        self.first_fail_row = None if r.first_fail_row == 0xFFFFFFFFFFFFFFFF else int(r.first_fail_row)
        self.first_fail_code = int(r.first_fail_code)
        self.first_fail_kind = self.first_fail_code >> 24
        self.first_fail_site = self.first_fail_code & 0xFFFFFF
        self.launches = int(r.launches)
        self.rows_evaluated = int(r.rows_evaluated)
        self.kernel_ms = float(r.kernel_ms)

    @property
    def ok(self):
        return self.fail_count == 0

    def __repr__(self):
        return (f"Result(ok={self.ok}, fail_count={self.fail_count}, first_fail_row={self.first_fail_row}, "
                f"kind={self.first_fail_kind}, site={self.first_fail_site}, kernel_ms={self.kernel_ms:.4f})")


def _is_device(x):
    return hasattr(x, "is_cuda") and x.is_cuda


class Session:
    """RAII wrapper of zk_session*.  Inputs may be numpy arrays (staged to HBM by the library)
    or torch CUDA tensors (used in place; the caller keeps them alive)."""

    def __init__(self, handle, n, keepalive, lib=None):
        self._h = handle
        self.n = n
        self._keep = keepalive
        self._lib = lib if lib is not None else _lib.load()  # the library that opened the session (HIP, or the CPU backend's)

    def launch(self, status_dev=None):
        if status_dev is not None:
            _expect(status_dev, "status_dev", 4, (self.n,))
            if not _is_contiguous(status_dev):
                raise ValueError("status_dev must be contiguous")
        check(self._lib.zk_launch(self._h, _lib.ptr(status_dev)), "zk_launch", self._lib)

    def collect(self):
        r = ZkResult()
        check(self._lib.zk_collect(self._h, ctypes.byref(r)), "zk_collect", self._lib)
        return Result(r)

    def run(self):
        self.launch()
        return self.collect()

    def set_stream(self, stream):
        """Bind the session to another HIP stream of its device (a torch.cuda.Stream, a raw handle, or None = the
        engine's own stream); passes already enqueued are waited for first."""
        h = getattr(stream, "cuda_stream", stream)
        check(self._lib.zk_session_set_stream(self._h, ctypes.c_void_p(h) if h else None), "zk_session_set_stream", self._lib)

    def set_range(self, row_lo, row_hi):
        """Row-circuit sessions: evaluate rows [row_lo, row_hi) only (the rest is a read-only halo)."""
        check(self._lib.zk_set_range(self._h, int(row_lo), int(row_hi)), "zk_set_range", self._lib)

    def timing(self):
        """EVM sessions, after a collect: (open_ms, span_ms) — device spans of the open's kernels and of open + first pass"""
        a, b = ctypes.c_double(), ctypes.c_double()
        check(self._lib.zk_session_timing(self._h, ctypes.byref(a), ctypes.byref(b)), "zk_session_timing", self._lib)
        return a.value, b.value

    def read_status(self):
        out = np.empty(self.n, dtype=np.uint32)
        check(self._lib.zk_read_status(self._h, _lib.ptr(out)), "zk_read_status", self._lib)
        return out

    def close(self):
        if self._h:
            self._lib.zk_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _is_contiguous(a):
    return a.is_contiguous() if hasattr(a, "is_contiguous") else a.flags["C_CONTIGUOUS"]


def _itemsize(a):
    return a.element_size() if hasattr(a, "element_size") else a.dtype.itemsize


def _expect(a, name, itemsize, shape):
    """Wire-format check of one argument: element size (uint64 cells travel as int64 tensors on the device, so the size
    is what is checked, plus 'integer') and shape (None = any extent).  The kernels index these buffers blindly."""
    if a is None:
        return
    kind_ok = (not a.dtype.is_floating_point and not a.dtype.is_complex) if hasattr(a, "is_cuda") else a.dtype.kind in "iu"
    if not kind_ok or _itemsize(a) != itemsize:
        raise TypeError(f"{name}: expected {itemsize}-byte integers, got {a.dtype}")
    if len(a.shape) != len(shape) or any(e is not None and int(d) != e for d, e in zip(a.shape, shape)):
        raise ValueError(f"{name}: expected shape {tuple('n' if e is None else e for e in shape)}, got {tuple(a.shape)}")


def _prep(arrs, outputs=()):
    """Host arrays are made C-contiguous (a copy is fine: they are staged anyway); device tensors are used in place.
    `outputs`: indices of buffers the device WRITES — those must already be contiguous (a silent copy would swallow
    the results)."""
    dev = [_is_device(a) for a in arrs if a is not None]
    if any(dev) and not all(dev):
        raise ValueError("mix of host and device buffers")
    is_dev = bool(dev) and all(dev)
    out = []
    for k, a in enumerate(arrs):
        if a is None:
            out.append(None)
        elif k in outputs:
            if not _is_contiguous(a):
                raise ValueError("output buffer must be contiguous")
            out.append(a)
        elif is_dev:
            out.append(a.contiguous())
        else:
            out.append(np.ascontiguousarray(a))
    return out, (OPT_DEVICE_PTRS if is_dev else 0)


def _randomness_cells(randomness, like):
    """int -> one canonical cell (uint64[4]), on the device when `like` is a CUDA tensor"""
    if not isinstance(randomness, int):
        return randomness
    cells = np.frombuffer(int(randomness).to_bytes(32, "little"), dtype="<u8").copy()
    if _is_device(like):
        import torch

        return torch.from_numpy(cells.view(np.int64)).to(like.device)
    return cells


def open_state(rows, flags, mpt, device=None, compact=False):
    """rows uint64[57, n, 4], flags uint32[n], mpt uint64[m, 12, 4] -> Session.  compact: rows uint64[15, n, 4], a device-assigned
    witness without the limb / byte columns (ZK_OPT_STATE_COMPACT, include/zkevm_hip.h)"""
    lib = _lib.init(device)
    _expect(rows, "state rows", 8, (15 if compact else 57, None, 4))
    _expect(flags, "state flags", 4, (rows.shape[1],))
    _expect(mpt, "mpt", 8, (None, 12, 4))
    (rows, flags, mpt), opts = _prep([rows, flags, mpt])
    if compact:
        opts |= _lib.OPT_STATE_COMPACT
    n = rows.shape[1]
    m = mpt.shape[0] if mpt is not None else 0
    h = ctypes.c_void_p()
    check(lib.zk_state_open(_lib.ptr(rows), _lib.ptr(flags), n, _lib.ptr(mpt) if m else None, m, opts,
                            ctypes.byref(h)), "zk_state_open")
    return Session(h, n, (rows, flags, mpt), lib=lib)


def _evm_tables(wire, begin_with_first_step, end_with_last_step):
    """wire dict -> (ZkEvmTables, opts, kept arrays); host arrays are made contiguous, device tensors are used in place"""
    names = ["steps", "rw", "rw_flags", "bytecode", "tx", "tx_flags", "block", "block_flags", "copy", "keccak", "exp", "aux",
             "aux_kind", "withdrawals", "sig", "ecc"]
    cells = {"steps": 13, "rw": 14, "bytecode": 6, "tx": 5, "block": 4, "copy": 14, "keccak": 5, "exp": 11, "withdrawals": 4,
             "sig": 9, "ecc": 13, "aux": None}
    for k, nc in cells.items():
        _expect(wire.get(k), k, 8, (None, nc, 4))
    for k, of in (("rw_flags", "rw"), ("tx_flags", "tx"), ("block_flags", "block"), ("aux_kind", "steps")):
        if wire.get(k) is not None and wire.get(of) is not None:
            _expect(wire[k], k, 4, (wire[of].shape[0],))
    arrs, opts = _prep([wire.get(k) for k in names])
    a = dict(zip(names, arrs))

    def rows(x):
        return 0 if x is None else int(x.shape[0])

    def p(x):
        v = _lib.ptr(x)
        return v.value if v is not None else None

    t = _lib.ZkEvmTables(
        p(a["steps"]), int(a["steps"].shape[0]),
        p(a["rw"]) if rows(a["rw"]) else None, p(a["rw_flags"]) if rows(a["rw"]) else None, rows(a["rw"]),
        p(a["bytecode"]) if rows(a["bytecode"]) else None, rows(a["bytecode"]),
        p(a["tx"]) if rows(a["tx"]) else None, p(a["tx_flags"]) if rows(a["tx"]) else None, rows(a["tx"]),
        p(a["block"]) if rows(a["block"]) else None, p(a["block_flags"]) if rows(a["block"]) else None, rows(a["block"]),
        int(bool(begin_with_first_step)), int(bool(end_with_last_step)),
        p(a["copy"]) if rows(a["copy"]) else None, rows(a["copy"]),
        p(a["keccak"]) if rows(a["keccak"]) else None, rows(a["keccak"]),
        p(a["exp"]) if rows(a["exp"]) else None, rows(a["exp"]),
        p(a["aux"]) if rows(a["aux"]) else None, p(a["aux_kind"]) if rows(a["aux"]) else None,
        p(a["withdrawals"]) if rows(a["withdrawals"]) else None, rows(a["withdrawals"]),
        p(a["sig"]) if rows(a["sig"]) else None, rows(a["sig"]),
        p(a["ecc"]) if rows(a["ecc"]) else None, rows(a["ecc"]),
        int(a["aux"].shape[1]) if rows(a["aux"]) else 0, 0)
    return t, opts, arrs, int(a["steps"].shape[0]) - 1


def open_evm(wire, begin_with_first_step=False, end_with_last_step=False, device=None, state_sort=True,
             generic_index=False, side_stream=False, single_pass=False):
    """wire: dict with steps uint64[n, 13, 4] (row-major), rw/rw_flags, bytecode, tx/tx_flags, block/block_flags
    and optionally copy uint64[m, 14, 4], keccak uint64[m, 5, 4], exp uint64[m, 11, 4], sig uint64[m, 9, 4], ecc uint64[m, 13, 4],
    aux uint64[n, 2 or 12, 4] + aux_kind, withdrawals uint64[m, 4, 4]
    (numpy arrays or torch CUDA tensors) -> Session over the n-1 step pairs."""
    lib = _lib.init(device)
    t, opts, arrs, n_pairs = _evm_tables(wire, begin_with_first_step, end_with_last_step)
    if not state_sort:
        opts |= _lib.OPT_NO_STATE_SORT
    if generic_index:
        opts |= _lib.OPT_GENERIC_INDEX
    if single_pass:  # the session will evaluate ONE pass: skip the packed step records (ZK_OPT_SINGLE_PASS, include/zkevm_hip.h)
        opts |= _lib.OPT_SINGLE_PASS
    if side_stream:  # warm / cold launches beside the hot one: pays when other sessions' passes share the device (include/zkevm_hip.h)
        opts |= _lib.OPT_SIDE_STREAM
    h = ctypes.c_void_p()
    check(lib.zk_evm_open(ctypes.byref(t), opts, ctypes.byref(h)), "zk_evm_open")
    return Session(h, n_pairs, arrs, lib=lib)


def evm_verify(wire, begin_with_first_step=False, end_with_last_step=False, status_dev=None, device=None):
    """The one-shot C entry zk_evm_verify (open + one pass + collect + close) over a wire dict of host arrays or of torch
    CUDA tensors (then status_dev, an optional CUDA uint32[n - 1] tensor, receives the per-pair status) -> Result."""
    lib = _lib.init(device)
    t, opts, arrs, n_pairs = _evm_tables(wire, begin_with_first_step, end_with_last_step)
    if status_dev is not None:
        _expect(status_dev, "status_dev", 4, (n_pairs,))
    r = ZkResult()
    check(lib.zk_evm_verify(ctypes.byref(t), opts, _lib.ptr(status_dev), ctypes.byref(r)), "zk_evm_verify")
    return Result(r)


class EvmOneShot:
    """A prepared call of the one-shot C entry `zk_evm_verify` over a resident wire dict: the argument block is marshalled
    once, `__call__` is the C call alone (what a foreign caller of the ABI pays — bench.py's timed region)."""

    def __init__(self, wire, begin_with_first_step=False, end_with_last_step=False, status_dev=None, device=None):
        self._lib = _lib.init(device)
        self._t, self._opts, self._keep, self.n_pairs = _evm_tables(wire, begin_with_first_step, end_with_last_step)
        if status_dev is not None:
            _expect(status_dev, "status_dev", 4, (self.n_pairs,))
        self._status = status_dev
        self._status_ptr = _lib.ptr(status_dev)
        self._r = ZkResult()
        self._tref, self._rref = ctypes.byref(self._t), ctypes.byref(self._r)

    def __call__(self):
        rc = self._lib.zk_evm_verify(self._tref, self._opts, self._status_ptr, self._rref)
        if rc:
            check(rc, "zk_evm_verify", self._lib)
        return self._r

    def result(self):
        return Result(self._r)


class EvmBatch:
    """A prepared call of `zk_evm_verify_batch` over resident wire dicts (the witness list may name the same dict several times):
    n independent verifications, software-pipelined two deep by the library."""

    def __init__(self, wires, order, device=None, flags=None):
        """`flags`: per wire, (begin_with_first_step, end_with_last_step) — carried in each witness's own argument block"""
        self._lib = _lib.init(device)
        flags = flags if flags is not None else [(False, False)] * len(wires)
        prepared = [_evm_tables(w, bool(f[0]), bool(f[1])) for w, f in zip(wires, flags)]
        self._keep = prepared
        self._opts = prepared[0][1]
        # one `opts` word goes to the library for the whole batch: host and device wires (ZK_OPT_DEVICE_PTRS) cannot be mixed
        if any(p[1] != self._opts for p in prepared):
            raise ValueError("EvmBatch: every witness must be prepared with the same options (all host arrays or all device tensors)")
        self.n = len(order)
        self.n_pairs = prepared[0][3]                    # of the first witness (kept for callers of the equal-size case)
        self.n_pairs_each = [prepared[k][3] for k in order]  # per witness, in call order
        arr_t = ctypes.POINTER(_lib.ZkEvmTables) * self.n
        self._ptrs = arr_t(*[ctypes.pointer(prepared[k][0]) for k in order])
        self._results = (ZkResult * self.n)()

    def __call__(self):
        rc = self._lib.zk_evm_verify_batch(self._ptrs, self.n, self._opts, self._results)
        if rc:
            check(rc, "zk_evm_verify_batch", self._lib)
        return self._results

    def results(self):
        return [Result(r) for r in self._results]


def last_timing(lib=None):
    """(open_ms, pass_ms, span_ms) of the calling thread's last one-shot zk_evm_verify (device spans by HIP events; -1 = not measured)"""
    lib = lib if lib is not None else _lib.load()
    a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.zk_last_timing(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return a.value, b.value, c.value


def open_bytecode(rows, keccak, randomness, device=None):
    """rows uint64[12, n, 4], keccak uint64[m, 5, 4], randomness uint64[4] (or an int) -> Session"""
    lib = _lib.init(device)
    randomness = _randomness_cells(randomness, rows)
    _expect(rows, "bytecode rows", 8, (12, None, 4))
    _expect(keccak, "keccak", 8, (None, 5, 4))
    _expect(randomness, "randomness", 8, (4,))
    (rows, keccak, randomness), opts = _prep([rows, keccak, randomness])
    n = rows.shape[1]
    m = keccak.shape[0] if keccak is not None else 0
    h = ctypes.c_void_p()
    check(lib.zk_bytecode_open(_lib.ptr(rows), n, _lib.ptr(keccak) if m else None, m, _lib.ptr(randomness), opts,
                               ctypes.byref(h)), "zk_bytecode_open")
    return Session(h, n, (rows, keccak, randomness), lib=lib)


def open_exp(rows, device=None):
    """rows uint64[21, n, 4] -> Session"""
    lib = _lib.init(device)
    _expect(rows, "exp rows", 8, (21, None, 4))
    (rows,), opts = _prep([rows])
    h = ctypes.c_void_p()
    check(lib.zk_exp_open(_lib.ptr(rows), rows.shape[1], opts, ctypes.byref(h)), "zk_exp_open")
    return Session(h, rows.shape[1], (rows,), lib=lib)


def open_copy(rows, row_flags, randomness, rw, rw_flags, bytecode, tx, tx_flags, device=None, generic_index=False):
    """Copy circuit session: rows uint64[20, n, 4] + flags, randomness (int or uint64[4]), EVM-format tables."""
    lib = _lib.init(device)
    randomness = _randomness_cells(randomness, rows)
    _expect(rows, "copy rows", 8, (20, None, 4))
    _expect(row_flags, "copy row_flags", 4, (rows.shape[1],))
    _expect(rw, "rw", 8, (None, 14, 4))
    _expect(bytecode, "bytecode", 8, (None, 6, 4))
    _expect(tx, "tx", 8, (None, 5, 4))
    arrs, opts = _prep([rows, row_flags, randomness, rw, rw_flags, bytecode, tx, tx_flags])
    rows, row_flags, randomness, rw, rw_flags, bytecode, tx, tx_flags = arrs

    def nrows(x):
        return 0 if x is None else int(x.shape[0])

    def p(x, n=1):
        v = _lib.ptr(x) if n else None
        return v.value if v is not None else None

    t = _lib.ZkCopyTables(p(rows), p(row_flags), int(rows.shape[1]), p(randomness),
                          p(rw, nrows(rw)), p(rw_flags, nrows(rw)), nrows(rw), p(bytecode, nrows(bytecode)), nrows(bytecode),
                          p(tx, nrows(tx)), p(tx_flags, nrows(tx)), nrows(tx))
    if generic_index:
        opts |= _lib.OPT_GENERIC_INDEX
    h = ctypes.c_void_p()
    check(lib.zk_copy_open(ctypes.byref(t), opts, ctypes.byref(h)), "zk_copy_open")
    return Session(h, int(rows.shape[1]), arrs, lib=lib)


def open_sign(wire, randomness, is_sig, device=None):
    """Tx / Sig circuit session over `wire` = dict(bytes, cells, meta, keccak, tx_rows, tx_flags)."""
    lib = _lib.init(device)
    randomness = _randomness_cells(randomness, wire.get("bytes"))
    names = ["bytes", "cells", "meta", "keccak", "tx_rows", "tx_flags"]
    _expect(wire.get("bytes"), "sign bytes", 1, (None, 9, 32))
    _expect(wire.get("cells"), "sign cells", 8, (8, wire["bytes"].shape[0], 4))
    _expect(wire.get("meta"), "sign meta", 4, (wire["bytes"].shape[0], 4))
    _expect(wire.get("keccak"), "keccak", 8, (None, 5, 4))
    _expect(wire.get("tx_rows"), "tx_rows", 8, (None, 5, 4))
    arrs, opts = _prep([wire.get(k) for k in names] + [randomness])
    a = dict(zip(names + ["r"], arrs))

    def nrows(x):
        return 0 if x is None else int(x.shape[0])

    def p(x, n=1):
        v = _lib.ptr(x) if n else None
        return v.value if v is not None else None

    n = int(a["bytes"].shape[0])
    t = _lib.ZkSignUnits(p(a["bytes"]), p(a["cells"]), p(a["meta"]), n, p(a["r"]), p(a["keccak"], nrows(a["keccak"])),
                         nrows(a["keccak"]), p(a["tx_rows"], nrows(a["tx_rows"])), p(a["tx_flags"], nrows(a["tx_rows"])),
                         nrows(a["tx_rows"]), int(bool(is_sig)))
    h = ctypes.c_void_p()
    check(lib.zk_sign_open(ctypes.byref(t), opts, ctypes.byref(h)), "zk_sign_open")
    return Session(h, n, arrs, lib=lib)


KECCAK_MODE_CIRCUIT = 0  # KeccakCircuit.add rows (EVM / bytecode circuits)
KECCAK_MODE_TABLE = 1    # KeccakTable.add rows (Tx / Sig circuits)


class KeccakSession(Session):
    """Keccak-table generation session: launch()/collect() like the circuits, rows() for the table."""

    def rows(self):
        out = np.empty((self.n, 5, 4), dtype=np.uint64)
        check(self._lib.zk_keccak_read_rows(self._h, _lib.ptr(out)), "zk_keccak_read_rows", self._lib)
        return out


def pack_messages(messages):
    """list of bytes -> (data uint8[total], offsets uint64[n + 1])"""
    offsets = np.zeros(len(messages) + 1, dtype=np.uint64)
    if len(messages):
        offsets[1:] = np.cumsum([len(m) for m in messages], dtype=np.uint64)
    data = np.frombuffer(b"".join(bytes(m) for m in messages), dtype=np.uint8).copy()
    return data, offsets


def open_keccak(data, offsets, randomness, mode=KECCAK_MODE_CIRCUIT, rows_dev=None, device=None):
    """data uint8[total], offsets uint64[n + 1], randomness (int or uint64[4]) -> KeccakSession.
    numpy arrays are staged to HBM; torch CUDA tensors are used in place (rows_dev: optional
    CUDA uint64[n, 5, 4] tensor receiving the rows)."""
    lib = _lib.init(device)
    if isinstance(randomness, int):
        randomness = np.frombuffer(int(randomness).to_bytes(32, "little"), dtype="<u8").copy()
        if _is_device(data):
            import torch
            randomness = torch.from_numpy(randomness.view(np.int64)).to(data.device)
    _expect(data, "keccak data", 1, (None,))
    _expect(offsets, "keccak offsets", 8, (None,))
    _expect(rows_dev, "keccak rows_dev", 8, (int(offsets.shape[0]) - 1, 5, 4))
    (data, offsets, randomness, rows_dev), opts = _prep([data, offsets, randomness, rows_dev], outputs=(3,))
    n = int(offsets.shape[0]) - 1
    n_bytes = int(data.shape[0]) if data is not None else 0
    h = ctypes.c_void_p()
    check(lib.zk_keccak_open(_lib.ptr(data) if n_bytes else None, n_bytes, _lib.ptr(offsets), n, _lib.ptr(randomness),
                             int(mode), _lib.ptr(rows_dev), opts, ctypes.byref(h)), "zk_keccak_open")
    return KeccakSession(h, n, (data, offsets, randomness, rows_dev), lib=lib)


def keccak_table(messages, randomness, mode=KECCAK_MODE_CIRCUIT, device=None):
    """Keccak table rows uint64[n, 5, 4] of a list of byte strings, computed on the GPU
    (replaces KeccakCircuit.add / KeccakTable.add loops; see include/zkevm_hip.h).  Raises the
    reference's ValueError for a mode-1 input longer than 64 bytes."""
    from .errors import exception_for_code
    if len(messages) == 0:
        return np.zeros((0, 5, 4), dtype=np.uint64)
    data, offsets = pack_messages(messages)
    with open_keccak(data, offsets, randomness, mode, device=device) as s:
        res = s.run()
        if not res.ok:
            raise exception_for_code(res.first_fail_code, f"keccak table: message {res.first_fail_row}")
        return s.rows()


class AssignSession(Session):
    """State-witness assignment session: launch()/collect() like the circuits; n_mpt()/read() for the outputs."""

    def n_mpt(self):
        m = ctypes.c_uint64()
        check(self._lib.zk_state_assign_read(self._h, None, None, None, 0, ctypes.byref(m)), "zk_state_assign_read", self._lib)
        return int(m.value)

    compact = False

    def read(self):
        """-> (rows uint64[57, n, 4] (15 with compact), flags uint32[n], mpt uint64[m, 12, 4]) on the host"""
        m = self.n_mpt()
        rows = np.empty((15 if self.compact else 57, self.n, 4), dtype=np.uint64)
        flags = np.empty(self.n, dtype=np.uint32)
        mpt = np.empty((m, 12, 4), dtype=np.uint64)
        got = ctypes.c_uint64()
        check(self._lib.zk_state_assign_read(self._h, _lib.ptr(rows), _lib.ptr(flags), _lib.ptr(mpt) if m else None, m,
                                               ctypes.byref(got)), "zk_state_assign_read")
        return rows, flags, mpt


def open_state_assign(ops, op_flags, rows_dev=None, row_flags_dev=None, mpt_dev=None, device=None, compact=False):
    """ops uint64[12, n, 4] (column-major Operation slots, include/zkevm_hip.h), op_flags uint32[n] -> AssignSession.
    numpy inputs are staged to HBM; torch CUDA tensors are used in place, and rows_dev uint64[57, n, 4] /
    row_flags_dev uint32[n] / mpt_dev uint64[n, 12, 4] (optional CUDA tensors) then receive the outputs, ready
    to be handed to open_state()."""
    lib = _lib.init(device)
    _expect(ops, "state ops", 8, (12, None, 4))
    n = int(ops.shape[1])
    _expect(op_flags, "op_flags", 4, (n,))
    _expect(rows_dev, "rows_dev", 8, (15 if compact else 57, n, 4))
    _expect(row_flags_dev, "row_flags_dev", 4, (n,))
    _expect(mpt_dev, "mpt_dev", 8, (n, 12, 4))
    (ops, op_flags, rows_dev, row_flags_dev, mpt_dev), opts = _prep([ops, op_flags, rows_dev, row_flags_dev, mpt_dev],
                                                                    outputs=(2, 3, 4))
    if compact:
        opts |= _lib.OPT_STATE_COMPACT
    h = ctypes.c_void_p()
    check(lib.zk_state_assign_open(_lib.ptr(ops), _lib.ptr(op_flags), n, _lib.ptr(rows_dev), _lib.ptr(row_flags_dev),
                                   _lib.ptr(mpt_dev), opts, ctypes.byref(h)), "zk_state_assign_open")
    s = AssignSession(h, n, (ops, op_flags, rows_dev, row_flags_dev, mpt_dev), lib=lib)
    s.compact = bool(compact)
    return s


def open_state_assign_from_rw(rw, rw_flags, rows_dev=None, row_flags_dev=None, mpt_dev=None, device=None, compact=False):
    """rw uint64[n, 14, 4] + rw_flags uint32[n] (the EVM circuit's RW table) -> AssignSession over the n_ops = 1 + kept rows State
    rows (session.n): re-keying + sort + witness assignment in one session, the op list never materialised
    (zk_state_assign_from_rw_open).  With CUDA tensors, rows_dev / row_flags_dev / mpt_dev are FLAT buffers of capacity
    57 * (n + 1) * 4 / n + 1 / (n + 1) * 12 * 4 receiving the outputs packed for n_ops: rows_dev[: 57 * n_ops * 4].view(57, n_ops, 4)
    etc. are what open_state takes."""
    lib = _lib.init(device)
    _expect(rw, "rw table", 8, (None, 14, 4))
    n = int(rw.shape[0])
    _expect(rw_flags, "rw_flags", 4, (n,))
    for buf, name, size, cap in ((rows_dev, "rows_dev", 8, (15 if compact else 57) * 4 * (n + 1)), (row_flags_dev, "row_flags_dev", 4, n + 1), (mpt_dev, "mpt_dev", 8, 48 * (n + 1))):
        _expect(buf, name, size, (None,))
        if buf is not None and int(buf.shape[0]) < cap:
            raise ValueError(f"{name} holds fewer than {cap} entries")
    (rw, rw_flags, rows_dev, row_flags_dev, mpt_dev), opts = _prep([rw, rw_flags, rows_dev, row_flags_dev, mpt_dev], outputs=(2, 3, 4))
    if compact:
        opts |= _lib.OPT_STATE_COMPACT
    h, n_ops = ctypes.c_void_p(), ctypes.c_uint64()
    check(lib.zk_state_assign_from_rw_open(_lib.ptr(rw), _lib.ptr(rw_flags), n, _lib.ptr(rows_dev), _lib.ptr(row_flags_dev), _lib.ptr(mpt_dev),
                                           opts, ctypes.byref(n_ops), ctypes.byref(h)), "zk_state_assign_from_rw_open")
    s = AssignSession(h, int(n_ops.value), (rw, rw_flags, rows_dev, row_flags_dev, mpt_dev), lib=lib)
    s.compact = bool(compact)
    return s


def open_state_verify_from_rw(rw, rw_flags, device=None):
    """rw uint64[n, 14, 4] + rw_flags uint32[n] (the EVM circuit's RW table) -> Session over the n_ops = 1 + kept rows State rows
    (session.n), which are evaluated where they are computed and never stored (zk_state_verify_from_rw_open): collect() is the State
    circuit's Result; an RW row the re-keying rejects or an op the assignment raises on makes collect() raise EngineError."""
    lib = _lib.init(device)
    _expect(rw, "rw table", 8, (None, 14, 4))
    n = int(rw.shape[0])
    _expect(rw_flags, "rw_flags", 4, (n,))
    (rw, rw_flags), opts = _prep([rw, rw_flags])
    h, n_ops = ctypes.c_void_p(), ctypes.c_uint64()
    check(lib.zk_state_verify_from_rw_open(_lib.ptr(rw), _lib.ptr(rw_flags), n, opts, ctypes.byref(n_ops), ctypes.byref(h)), "zk_state_verify_from_rw_open", lib)
    return Session(h, int(n_ops.value), (rw, rw_flags), lib=lib)


class RekeySession(Session):
    """RW table -> State-circuit operations (zk_state_ops_from_rw_*): launch()/collect() like the circuits (status = one code per
    RW row); n_ops = StartOp + the rows kept; read() -> (ops uint64[12, n_ops, 4], op_flags uint32[n_ops]) on the host."""

    n_ops = 0

    def read(self):
        ops = np.empty((12, self.n_ops, 4), dtype=np.uint64)
        flags = np.empty(self.n_ops, dtype=np.uint32)
        check(self._lib.zk_state_ops_from_rw_read(self._h, _lib.ptr(ops), _lib.ptr(flags), None), "zk_state_ops_from_rw_read", self._lib)
        return ops, flags


def open_state_ops_from_rw(rw, rw_flags, ops_dev=None, op_flags_dev=None, device=None):
    """rw uint64[n, 14, 4] + rw_flags uint32[n] (the EVM circuit's RW table) -> RekeySession.  numpy inputs are staged to HBM;
    torch CUDA tensors are used in place, and ops_dev (a flat CUDA buffer of at least 12 * (n + 1) * 4 uint64) / op_flags_dev
    (n + 1 uint32) then receive the op list packed for session.n_ops ops: ops_dev[: 12 * n_ops * 4].view(12, n_ops, 4) is what
    open_state_assign takes."""
    lib = _lib.init(device)
    _expect(rw, "rw table", 8, (None, 14, 4))
    n = int(rw.shape[0])
    _expect(rw_flags, "rw_flags", 4, (n,))
    _expect(ops_dev, "ops_dev", 8, (None,))
    _expect(op_flags_dev, "op_flags_dev", 4, (None,))
    if ops_dev is not None and int(ops_dev.shape[0]) < 48 * (n + 1):
        raise ValueError("ops_dev holds fewer than 12 * (n + 1) cells")
    if op_flags_dev is not None and int(op_flags_dev.shape[0]) < n + 1:
        raise ValueError("op_flags_dev holds fewer than n + 1 entries")
    (rw, rw_flags, ops_dev, op_flags_dev), opts = _prep([rw, rw_flags, ops_dev, op_flags_dev], outputs=(2, 3))
    h, n_ops = ctypes.c_void_p(), ctypes.c_uint64()
    check(lib.zk_state_ops_from_rw_open(_lib.ptr(rw), _lib.ptr(rw_flags), n, _lib.ptr(ops_dev), _lib.ptr(op_flags_dev), opts,
                                        ctypes.byref(n_ops), ctypes.byref(h)), "zk_state_ops_from_rw_open")
    s = RekeySession(h, n, (rw, rw_flags, ops_dev, op_flags_dev), lib=lib)
    s.n_ops = int(n_ops.value)
    return s


class BytecodeAssignSession(Session):
    """Bytecode-witness assignment session: launch()/collect() like the circuits; rows() for the 2^k circuit rows."""

    def rows(self):
        out = np.empty((12, self.n, 4), dtype=np.uint64)
        check(self._lib.zk_bytecode_assign_read(self._h, _lib.ptr(out)), "zk_bytecode_assign_read", self._lib)
        return out


def open_bytecode_assign(in_rows, offsets, lengths, k, randomness, rows_dev=None, device=None):
    """in_rows uint64[n, 6, 4] (unrolled BytecodeTableRows, input order), offsets uint64[m + 1], lengths uint64[m], k,
    randomness (int or uint64[4]) -> BytecodeAssignSession over the 2^k circuit rows; rows_dev: optional CUDA tensor
    uint64[12, 2^k, 4] receiving them in place (ready for open_bytecode)."""
    lib = _lib.init(device)
    randomness = _randomness_cells(randomness, in_rows)
    _expect(in_rows, "unrolled bytecode rows", 8, (None, 6, 4))
    _expect(offsets, "offsets", 8, (int(lengths.shape[0]) + 1,))
    _expect(lengths, "lengths", 8, (None,))
    _expect(rows_dev, "rows_dev", 8, (12, 1 << int(k), 4))
    (in_rows, offsets, lengths, randomness, rows_dev), opts = _prep([in_rows, offsets, lengths, randomness, rows_dev],
                                                                    outputs=(4,))
    n_rows, n_codes = int(in_rows.shape[0]), int(lengths.shape[0])
    h = ctypes.c_void_p()
    check(lib.zk_bytecode_assign_open(_lib.ptr(in_rows) if n_rows else None, n_rows, _lib.ptr(offsets), _lib.ptr(lengths) if n_codes else None,
                                      n_codes, int(k), _lib.ptr(randomness), _lib.ptr(rows_dev), opts, ctypes.byref(h)),
          "zk_bytecode_assign_open")
    return BytecodeAssignSession(h, 1 << int(k), (in_rows, offsets, lengths, randomness, rows_dev), lib=lib)


def open_pi(rows, keccak, gas, circuit_len, keccak_rand=255, byte_pow_base=255, device=None):
    """Public-inputs circuit session: rows uint64[24, n, 4], keccak uint64[m, 5, 4], gas uint64[k, 3, 4] (include/zkevm_hip.h)"""
    lib = _lib.init(device)
    kr, bp = _randomness_cells(int(keccak_rand), rows), _randomness_cells(int(byte_pow_base), rows)
    _expect(rows, "pi rows", 8, (24, None, 4))
    _expect(keccak, "keccak", 8, (None, 5, 4))
    _expect(gas, "gas-cost table", 8, (None, 3, 4))
    (rows, keccak, gas, kr, bp), opts = _prep([rows, keccak, gas, kr, bp])
    n, m, k = int(rows.shape[1]), int(keccak.shape[0]) if keccak is not None else 0, int(gas.shape[0]) if gas is not None else 0
    h = ctypes.c_void_p()
    check(lib.zk_pi_open(_lib.ptr(rows), n, _lib.ptr(keccak) if m else None, m, _lib.ptr(gas) if k else None, k, int(circuit_len), _lib.ptr(kr),
                         _lib.ptr(bp), opts, ctypes.byref(h)), "zk_pi_open")
    return Session(h, n, (rows, keccak, gas, kr, bp), lib=lib)


def _copy_events_struct(events, flags, data, offsets, randomness):
    def p(x):
        v = _lib.ptr(x)
        return v.value if v is not None else None

    return _lib.ZkCopyEvents(p(events), p(flags), int(events.shape[0]), p(data) if data is not None and int(data.shape[0]) else None,
                             p(offsets), p(randomness))


def copy_assign_sizes(events, flags, data, offsets, device=None):
    """(n_rows, n_table, n_rw) a list of copy events expands to (host arithmetic over the events)"""
    lib = _lib.init(device)
    _expect(events, "copy events", 8, (None, 12, 4))
    (events, flags, data, offsets), opts = _prep([events, flags, data, offsets])
    t = _copy_events_struct(events, flags, data, offsets, None)
    a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    check(lib.zk_copy_assign_sizes(ctypes.byref(t), opts, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "zk_copy_assign_sizes")
    return int(a.value), int(b.value), int(c.value)


class CopyAssignSession(Session):
    """Copy-circuit witness assignment session: launch()/collect() like the circuits; read() for the outputs."""

    def read(self):
        """-> (rows uint64[20, n, 4], row_flags uint32[n], table uint64[m, 14, 4], rw uint64[k, 14, 4], rw_flags uint32[k]) on the host"""
        rows, rf = np.empty((20, self.n, 4), dtype=np.uint64), np.empty(self.n, dtype=np.uint32)
        table = np.empty((self.n_table, 14, 4), dtype=np.uint64)
        rw, rwf = np.empty((self.n_rw, 14, 4), dtype=np.uint64), np.empty(self.n_rw, dtype=np.uint32)
        check(self._lib.zk_copy_assign_read(self._h, _lib.ptr(rows), _lib.ptr(rf), _lib.ptr(table) if self.n_table else None,
                                              _lib.ptr(rw) if self.n_rw else None, _lib.ptr(rwf) if self.n_rw else None), "zk_copy_assign_read")
        return rows, rf, table, rw, rwf


def open_copy_assign(events, flags, data, offsets, randomness, rows_dev=None, row_flags_dev=None, table_dev=None, rw_dev=None,
                     rw_flags_dev=None, device=None):
    """events uint64[n, 12, 4] (row-major copy events, include/zkevm_hip.h), flags uint32[n], data uint16[total], offsets
    uint64[n + 1], randomness (int or uint64[4]) -> CopyAssignSession.  numpy inputs are staged to HBM; torch CUDA tensors are
    used in place and the optional *_dev tensors (sized with copy_assign_sizes) receive the outputs, ready for open_copy /
    open_evm."""
    lib = _lib.init(device)
    randomness = _randomness_cells(randomness, events)
    _expect(events, "copy events", 8, (None, 12, 4))
    _expect(flags, "copy event flags", 4, (events.shape[0],))
    _expect(data, "copy source data", 2, (None,))
    _expect(offsets, "copy data offsets", 8, (int(events.shape[0]) + 1,))
    n_rows, n_table, n_rw = copy_assign_sizes(events, flags, data, offsets, device)
    _expect(rows_dev, "rows_dev", 8, (20, n_rows, 4))
    _expect(row_flags_dev, "row_flags_dev", 4, (n_rows,))
    _expect(table_dev, "table_dev", 8, (n_table, 14, 4))
    _expect(rw_dev, "rw_dev", 8, (n_rw, 14, 4))
    _expect(rw_flags_dev, "rw_flags_dev", 4, (n_rw,))
    arrs, opts = _prep([events, flags, data, offsets, randomness, rows_dev, row_flags_dev, table_dev, rw_dev, rw_flags_dev],
                       outputs=(5, 6, 7, 8, 9))
    events, flags, data, offsets, randomness, rows_dev, row_flags_dev, table_dev, rw_dev, rw_flags_dev = arrs
    t = _copy_events_struct(events, flags, data, offsets, randomness)
    h = ctypes.c_void_p()
    check(lib.zk_copy_assign_open(ctypes.byref(t), _lib.ptr(rows_dev), _lib.ptr(row_flags_dev), _lib.ptr(table_dev), _lib.ptr(rw_dev),
                                  _lib.ptr(rw_flags_dev), opts, ctypes.byref(h)), "zk_copy_assign_open")
    s = CopyAssignSession(h, n_rows, arrs, lib=lib)
    s.n_table, s.n_rw = n_table, n_rw
    return s


ECDSA_LAYOUT_PACKED = 0  # uint8[n, 5, 32]: pk_x LE, pk_y LE, msg_hash BE, sig_r LE, sig_s LE
ECDSA_LAYOUT_TX_UNITS = 1   # uint8[n, 9, 32]: the Tx units' byte rows (open_sign's wire["bytes"]; msg_hash little-endian)
ECDSA_LAYOUT_SIG_UNITS = 2  # the Sig units' byte rows (msg_hash big-endian; v = meta[:, 3])


def open_ecdsa(sig_bytes, v=None, layout=ECDSA_LAYOUT_PACKED, out_dev=None, out_stride=1, device=None, v_stride=1):
    """secp256k1 ECDSA verification session: status per signature = the `ecdsa_status` column of the Tx / Sig units
    (0 verified, 1 not verified, else the exception's code; include/zkevm_hip.h).  out_dev: optional CUDA uint32
    tensor receiving status i at element i * out_stride (e.g. the units' meta tensor with stride 4)."""
    lib = _lib.init(device)
    _expect(sig_bytes, "signature bytes", 1, (None, 5 if layout == ECDSA_LAYOUT_PACKED else 9, 32))
    (sig_bytes, v, out_dev), opts = _prep([sig_bytes, v, out_dev], outputs=(2,))
    n = int(sig_bytes.shape[0])
    h = ctypes.c_void_p()
    check(lib.zk_ecdsa_open(_lib.ptr(sig_bytes), int(layout), _lib.ptr(v), int(v_stride), n, _lib.ptr(out_dev),
                            int(out_stride), opts, ctypes.byref(h)), "zk_ecdsa_open")
    return Session(h, n, (sig_bytes, v, out_dev), lib=lib)


def open_ecdsa_batches(batches, device=None):
    """One launch over one or two signature arrays (zk_ecdsa_open_batches): `batches` = list of dicts with the arguments of
    open_ecdsa (sig_bytes, v, layout, out_dev, out_stride, v_stride).  Statuses: batch 0's signatures first."""
    lib = _lib.init(device)
    keep, arr, opts_all, n_total = [], (_lib.ZkEcdsaBatch * len(batches))(), None, 0
    for k, b in enumerate(batches):
        layout = b.get("layout", ECDSA_LAYOUT_PACKED)
        _expect(b["sig_bytes"], "signature bytes", 1, (None, 5 if layout == ECDSA_LAYOUT_PACKED else 9, 32))
        (sb, v, od), opts = _prep([b["sig_bytes"], b.get("v"), b.get("out_dev")], outputs=(2,))
        assert opts_all is None or opts == opts_all, "mix of host and device batches"
        opts_all = opts
        keep += [sb, v, od]
        n = int(sb.shape[0])
        n_total += n
        pv = lambda x: None if x is None else _lib.ptr(x).value  # noqa: E731
        arr[k] = _lib.ZkEcdsaBatch(pv(sb), int(layout), pv(v), int(b.get("v_stride", 1)), n, pv(od), int(b.get("out_stride", 1)))
    h = ctypes.c_void_p()
    check(lib.zk_ecdsa_open_batches(arr, len(batches), opts_all, ctypes.byref(h)), "zk_ecdsa_open_batches")
    return Session(h, n_total, tuple(keep), lib=lib)


def ecdsa_status(sig_bytes, v=None, layout=ECDSA_LAYOUT_PACKED, device=None, v_stride=1):
    """-> uint32[n] ecdsa_status computed on the GPU"""
    with open_ecdsa(sig_bytes, v, layout, device=device, v_stride=v_stride) as s:
        s.run()
        return s.read_status()


def fr_op(op, a, b):
    """Vector Fr op on the device (host numpy in/out): a, b uint64[n, 4]."""
    lib = _lib.init()
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    check(lib.zk_fr_op(int(op), _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), a.shape[0], 0), "zk_fr_op")
    return out
