"""Marshalling of the reference's witness objects into the C-ABI wire format.

Duck-typed: works on the reference's own objects (`zkevm_specs.state_circuit.Row`,
`MPTTableRow`, `StepState`, `RWTableRow`, ...) or on this package's mirrors — only attribute
names are used, nothing is imported from the reference.
"""
import os

from .wire import FR_MODULUS, rows_to_colmajor, rows_to_rowmajor
import numpy as np

# The per-cell loops below define the results.  Where _flatten_ext.so is built (csrc/flatten_ext.c, by csrc/build.sh) the big
# tables and the steps take the same walk in C — ~0.09 us per cell against ~1.2 us — and the de-duplication / ordering runs on the
# packed rows in numpy; tests/test_flatten_ext.py holds the two paths equal, array for array.  ZK_FLATTEN_PY=1 keeps the loops.
try:
    from . import _flatten_ext as _ext
except ImportError:  # not built (a source checkout before build()): the Python loops
    _ext = None
USE_EXT = _ext is not None and os.environ.get("ZK_FLATTEN_PY") != "1"
_N, _LO, _HI, _INT, _BOOL = 0, 1, 2, 3, 4  # cell modes of _flatten_ext.pack


def _spec(*cells):
    """'a.b' -> FQ(int) of row.a.b; ('lo', 'a') / ('hi', 'a') -> the halves of a Word / WordOrValue / bare value; ('int', 'a'); ('bool', 'a')"""
    out = []
    for c in cells:
        mode, path = (_N, c) if isinstance(c, str) else ({"lo": _LO, "hi": _HI, "int": _INT, "bool": _BOOL}[c[0]], c[1])
        out.append((mode, tuple(int(t) if t.isdigit() else t for t in path.split("."))))
    return tuple(out)


def _pack(objs, cells, flags=()):
    """-> (uint64[n, ncells, 4] row-major, uint32[n] flags) through _flatten_ext"""
    objs = objs if isinstance(objs, (list, tuple)) else list(objs)
    cb, fb = _ext.pack(objs, cells, flags)
    return np.frombuffer(cb, dtype="<u8").reshape(len(objs), len(cells), 4), np.frombuffer(fb, dtype="<u4")


def _dedup_rows(rows, flags=None):
    """_dedup on packed rows: set semantics on the cells, the first occurrence's flags, ascending order of the cells as integers (the
    order of sorted() over tuples of ints).  Rows are ranked by refining groups one 64-bit limb at a time, most significant first,
    skipping limbs that are the same in every row, until every row stands alone (an RW table: after its rw_counter limb) or the limbs
    run out (what still shares a group then is one row several times)."""
    n, nc = rows.shape[:2]
    if flags is None:
        flags = np.zeros(n, dtype=np.uint32)
    if n == 0:
        return np.zeros((0, nc, 4), dtype=np.uint64), np.zeros(0, dtype=np.uint32)
    words = np.ascontiguousarray(rows).reshape(n, nc * 4)
    varies = words.min(axis=0) != words.max(axis=0)
    group, n_groups = np.zeros(n, dtype=np.int64), 1
    for col in (4 * c + limb for c in range(nc) for limb in (3, 2, 1, 0)):
        if n_groups == n:
            break
        if not varies[col]:
            continue
        v = np.ascontiguousarray(words[:, col])
        idx = np.lexsort((v, group))  # by group, then by this limb
        gs, vs = group[idx], v[idx]
        starts = np.empty(n, dtype=bool)
        starts[0] = True
        np.not_equal(gs[1:], gs[:-1], out=starts[1:])
        starts[1:] |= vs[1:] != vs[:-1]
        ranks = np.cumsum(starts) - 1
        group = np.empty(n, dtype=np.int64)
        group[idx] = ranks
        n_groups = int(ranks[-1]) + 1
    idx = np.argsort(group, kind="stable")  # ascending rank; inside a group (duplicates) the original order: its first row is kept
    gs = group[idx]
    keep = np.empty(n, dtype=bool)
    keep[0] = True
    np.not_equal(gs[1:], gs[:-1], out=keep[1:])
    first = idx[keep]
    return rows[first].view(np.uint64), flags[first].astype(np.uint32, copy=False)  # (fancy indexing: fresh, contiguous arrays)


def _n(x):
    """canonical integer of an FQ / Expression / int / bool / IntEnum"""
    if hasattr(x, "expr"):
        return x.expr().n
    if hasattr(x, "n"):
        return x.n
    return int(x) % FR_MODULUS  # plain ints behave like FQ(int) (reduction mod p, e.g. rw_counter=-1)


def _is_word(x):
    return bool(getattr(x, "is_word", True))


STATE_NCELLS = 57
MPT_NCELLS = 12


def state_row_cells(row):
    """reference state_circuit.Row (:63-96) -> 57 ints + flags"""
    keys = row.keys
    cells = [_n(row.rw_counter), _n(row.is_write), _n(keys[0]), _n(keys[1]), _n(keys[2]),
             _n(keys[3]), _n(keys[4].lo), _n(keys[4].hi)]
    cells += [_n(x) for x in row.key2_limbs]
    cells += [_n(x) for x in row.key45_bytes]
    cells += [_n(row.value.lo), _n(row.value.hi), _n(row.initial_value.lo),
              _n(row.initial_value.hi), _n(row.root.lo), _n(row.root.hi),
              _n(row.lexicographic_ordering_selector)]
    flags = (1 if _is_word(row.value) else 0) | (2 if _is_word(row.initial_value) else 0)
    return cells, flags


_STATE_SPEC = _spec("rw_counter", "is_write", "keys.0", "keys.1", "keys.2", "keys.3", "keys.4.lo", "keys.4.hi", *[f"key2_limbs.{k}" for k in range(10)],
                    *[f"key45_bytes.{k}" for k in range(32)], "value.lo", "value.hi", "initial_value.lo", "initial_value.hi", "root.lo", "root.hi",
                    "lexicographic_ordering_selector")
_STATE_FLAGS = ((1, 0, ("value",)), (2, 0, ("initial_value",)))


def flatten_state_rows(rows):
    if USE_EXT:
        cells, flags = _pack(rows, _STATE_SPEC, _STATE_FLAGS)
        return np.ascontiguousarray(cells.transpose(1, 0, 2)), flags.astype(np.uint32)
    cf = [state_row_cells(r) for r in rows]
    cols = rows_to_colmajor([c for c, _ in cf], STATE_NCELLS)
    flags = np.array([f for _, f in cf], dtype=np.uint32)
    return cols, flags


def mpt_row_cells(m):
    """MPTTableRow (evm_circuit/table.py:461-468) -> 12 ints"""
    return [_n(m.address), _n(m.proof_type), _n(m.storage_key.lo), _n(m.storage_key.hi),
            _n(m.root.lo), _n(m.root.hi), _n(m.root_prev.lo), _n(m.root_prev.hi),
            _n(m.value.lo), _n(m.value.hi), _n(m.value_prev.lo), _n(m.value_prev.hi)]


def flatten_mpt_table(mpt_table):
    rows = sorted(set(tuple(mpt_row_cells(m)) for m in mpt_table))
    return rows_to_rowmajor(rows, MPT_NCELLS)


# ---- EVM circuit ------------------------------------------------------------------------
STEP_NCELLS = 13
RW_NCELLS = 14
BYTECODE_NCELLS = 6
TX_NCELLS = 5
BLOCK_NCELLS = 4


def _word_cells(x):
    """(lo, hi, is_word) of a Word / WordOrValue / bare FQ field"""
    if hasattr(x, "lo"):
        return _n(x.lo), _n(x.hi), _is_word(x)
    return _n(x), 0, False


def step_cells(s):
    """StepState (evm_circuit/step.py:6-75) -> 13 ints"""
    return [int(s.execution_state), _n(s.rw_counter), _n(s.call_id), int(bool(s.is_root)), int(bool(s.is_create)),
            _n(s.code_hash.lo), _n(s.code_hash.hi), _n(s.program_counter), _n(s.stack_pointer), _n(s.gas_left),
            _n(s.memory_word_size), _n(s.reversible_write_counter), _n(s.log_id)]


_STEP_SPEC = _spec(("int", "execution_state"), "rw_counter", "call_id", ("bool", "is_root"), ("bool", "is_create"), "code_hash.lo", "code_hash.hi",
                   "program_counter", "stack_pointer", "gas_left", "memory_word_size", "reversible_write_counter", "log_id")


def flatten_steps(steps):
    """row-major uint64[n_steps, 13, 4] (see include/zkevm_hip.h: the EVM kernel gathers steps)"""
    if USE_EXT:
        return _pack(steps, _STEP_SPEC)[0].copy()
    return rows_to_rowmajor([step_cells(s) for s in steps], STEP_NCELLS)


def rw_row_cells(r):
    """RWTableRow (evm_circuit/table.py:447-457) -> 14 ints + flags"""
    vlo, vhi, vw = _word_cells(r.value)
    plo, phi, pw = _word_cells(r.value_prev)
    cells = [_n(r.rw_counter), _n(r.rw), _n(r.key0), _n(r.id), _n(r.address), _n(r.field_tag),
             _n(r.storage_key.lo), _n(r.storage_key.hi), vlo, vhi, plo, phi, _n(r.aux0.lo), _n(r.aux0.hi)]
    return cells, (1 if vw else 0) | (2 if pw else 0)


def _dedup(pairs):
    """Tables are sets keyed on the cells (type bits do not take part in hashing/equality,
    util/arithmetic.py:133-141): keep the first occurrence, sorted for determinism."""
    seen = {}
    for cells, flag in pairs:
        seen.setdefault(tuple(cells), flag)
    keys = sorted(seen)
    return [list(k) for k in keys], [seen[k] for k in keys]


def _iter_table(t):
    """A table argument may be a set/list of rows or a witness object with table_assignments()."""
    if t is None:
        return []
    if hasattr(t, "table_assignments"):
        return list(t.table_assignments())
    return list(t)


_RW_SPEC = _spec("rw_counter", "rw", "key0", "id", "address", "field_tag", "storage_key.lo", "storage_key.hi", ("lo", "value"), ("hi", "value"),
                 ("lo", "value_prev"), ("hi", "value_prev"), "aux0.lo", "aux0.hi")
_RW_FLAGS = ((1, 1, ("value",)), (2, 1, ("value_prev",)))


def flatten_rw_table(rw_table):
    if USE_EXT:
        return _dedup_rows(*_pack(_iter_table(rw_table), _RW_SPEC, _RW_FLAGS))
    rows, flags = _dedup([rw_row_cells(r) for r in _iter_table(rw_table)])
    return rows_to_rowmajor(rows, RW_NCELLS), np.array(flags, dtype=np.uint32)


_BYTECODE_SPEC = _spec("bytecode_hash.lo", "bytecode_hash.hi", "field_tag", "index", "is_code", "value")


def flatten_bytecode_table(bytecode_table):
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(bytecode_table), _BYTECODE_SPEC)[0])[0]
    rows, _ = _dedup([([_n(r.bytecode_hash.lo), _n(r.bytecode_hash.hi), _n(r.field_tag), _n(r.index),
                        _n(r.is_code), _n(r.value)], 0) for r in _iter_table(bytecode_table)])
    return rows_to_rowmajor(rows, BYTECODE_NCELLS)


_TX_SPEC = _spec("tx_id", "field_tag", "call_data_index_or_zero", ("lo", "value"), ("hi", "value"))
_BLOCK_SPEC = _spec("field_tag", "block_number_or_zero", ("lo", "value"), ("hi", "value"))
_VALUE_FLAG = ((1, 1, ("value",)),)


def flatten_tx_table(tx_table):
    if USE_EXT:
        return _dedup_rows(*_pack(_iter_table(tx_table), _TX_SPEC, _VALUE_FLAG))
    pairs = []
    for r in _iter_table(tx_table):
        lo, hi, w = _word_cells(r.value)
        pairs.append(([_n(r.tx_id), _n(r.field_tag), _n(r.call_data_index_or_zero), lo, hi], 1 if w else 0))
    rows, flags = _dedup(pairs)
    return rows_to_rowmajor(rows, TX_NCELLS), np.array(flags, dtype=np.uint32)


def flatten_block_table(block_table):
    if USE_EXT:
        return _dedup_rows(*_pack(_iter_table(block_table), _BLOCK_SPEC, _VALUE_FLAG))
    pairs = []
    for r in _iter_table(block_table):
        lo, hi, w = _word_cells(r.value)
        pairs.append(([_n(r.field_tag), _n(r.block_number_or_zero), lo, hi], 1 if w else 0))
    rows, flags = _dedup(pairs)
    return rows_to_rowmajor(rows, BLOCK_NCELLS), np.array(flags, dtype=np.uint32)


COPY_T_NCELLS, KECCAK_T_NCELLS, EXP_T_NCELLS = 14, 5, 11


_COPY_T_SPEC = _spec("is_first", "src_id.lo", "src_id.hi", "src_tag", "dst_id.lo", "dst_id.hi", "dst_tag", "src_addr", "src_addr_end", "dst_addr",
                     "length", "rlc_acc", "rw_counter", "rwc_inc")
_EXP_T_SPEC = _spec("is_step", "identifier", "is_last", "base_limb0", "base_limb1", "base_limb2", "base_limb3", "exponent.lo", "exponent.hi",
                    "exponentiation.lo", "exponentiation.hi")


def flatten_copy_table(copy_table):
    """set of CopyTableRow (evm_circuit/table.py:494-507) -> uint64[m, 14, 4]; ids match on their
    lo/hi cells only (TableRow.match, table.py:389-401), so no type bits travel"""
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(copy_table), _COPY_T_SPEC)[0])[0]
    rows, _ = _dedup([([_n(r.is_first), _n(r.src_id.lo), _n(r.src_id.hi), _n(r.src_tag), _n(r.dst_id.lo), _n(r.dst_id.hi),
                        _n(r.dst_tag), _n(r.src_addr), _n(r.src_addr_end), _n(r.dst_addr), _n(r.length), _n(r.rlc_acc),
                        _n(r.rw_counter), _n(r.rwc_inc)], 0) for r in _iter_table(copy_table)])
    return rows_to_rowmajor(rows, COPY_T_NCELLS)


def flatten_exp_table(exp_table):
    """set of ExpTableRow (evm_circuit/table.py:538-548) -> uint64[m, 11, 4]"""
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(exp_table), _EXP_T_SPEC)[0])[0]
    rows, _ = _dedup([([_n(r.is_step), _n(r.identifier), _n(r.is_last), _n(r.base_limb0), _n(r.base_limb1),
                        _n(r.base_limb2), _n(r.base_limb3), _n(r.exponent.lo), _n(r.exponent.hi),
                        _n(r.exponentiation.lo), _n(r.exponentiation.hi)], 0) for r in _iter_table(exp_table)])
    return rows_to_rowmajor(rows, EXP_T_NCELLS)


AUX_NONE, AUX_WORD, AUX_INT, AUX_PAIR, AUX_OTHER = 0, 1, 2, 3, 4


AUX_ECRECOVER, AUX_ECADD, AUX_ECMUL, AUX_ECPAIRING = 5, 6, 7, 8
AUX_WIDE_CELLS = 12


def _is_word_obj(x):
    return hasattr(x, "lo") and hasattr(x, "hi")


def _is_fq_obj(x):
    return hasattr(x, "n") and not _is_word_obj(x)


def flatten_step_aux(steps):
    """StepState.aux_data (step.py:44, "auxiliary witness data needed by gadgets") -> cells + a kind per step:
    1 = a Word (CREATE: init-code hash), 2 = a non-negative int < 2^256 (ErrorOutOfGasSloadSstore: original value,
    lo/hi split), 3 = a pair of Python ints < p (CALL into a precompile: input / return length, callop.py:155-156),
    5 = ecRecover's [PrecompileAuxData, keccak_randomness] (ecrecover.py:15-23,38-44: msg_hash, sig_v, sig_r, sig_s as
    lo/hi, recovered_addr, input_rlc, output_rlc, randomness = 12 cells), 6 = ecAdd's [px, py, qx, qy, outx, outy]
    (ecadd.py:22-27, 10 cells), 7 = ecMul's [px, py, s, outx, outy] (ecmul.py:22-26, 8 cells), 8 = ecPairing's
    [input_rlc, input_pairs, is_valid_input, output] (ecpairing.py:26-29, 4 cells), 4 = anything else (gadgets that
    read it report ZK_UNSUPPORTED), 0 = absent.  Two cells per step unless a step needs the wide form (12 cells)."""
    cells, kinds = [], []
    for st in steps:
        a = getattr(st, "aux_data", None)
        c, k = [0, 0], AUX_NONE
        if a is None:
            pass
        elif _is_word_obj(a):
            c, k = [_n(a.lo), _n(a.hi)], AUX_WORD
        elif isinstance(a, int) and not isinstance(a, bool) and 0 <= a < (1 << 256):
            c, k = [a & ((1 << 128) - 1), a >> 128], AUX_INT
        elif isinstance(a, (list, tuple)):
            words = [_is_word_obj(x) for x in a]
            fqs = [_is_fq_obj(x) for x in a]
            if len(a) == 2 and all(isinstance(x, int) and not isinstance(x, bool) and 0 <= x < FR_MODULUS for x in a):
                c, k = [int(a[0]), int(a[1])], AUX_PAIR
            elif len(a) == 2 and hasattr(a[0], "msg_hash") and fqs[1]:
                d = a[0]
                if all(_is_word_obj(getattr(d, f)) for f in ("msg_hash", "sig_v", "sig_r", "sig_s")) and \
                        all(_is_fq_obj(getattr(d, f)) for f in ("recovered_addr", "input_rlc", "output_rlc")):
                    c = [_n(d.msg_hash.lo), _n(d.msg_hash.hi), _n(d.sig_v.lo), _n(d.sig_v.hi), _n(d.sig_r.lo), _n(d.sig_r.hi),
                         _n(d.sig_s.lo), _n(d.sig_s.hi), _n(d.recovered_addr), _n(d.input_rlc), _n(d.output_rlc), _n(a[1])]
                    k = AUX_ECRECOVER
                else:
                    k = AUX_OTHER
            elif len(a) == 6 and words == [True] * 4 + [False] * 2 and fqs[4] and fqs[5]:
                c = [v for w in a[:4] for v in (_n(w.lo), _n(w.hi))] + [_n(a[4]), _n(a[5])]
                k = AUX_ECADD
            elif len(a) == 5 and words == [True] * 3 + [False] * 2 and fqs[3] and fqs[4]:
                c = [v for w in a[:3] for v in (_n(w.lo), _n(w.hi))] + [_n(a[3]), _n(a[4])]
                k = AUX_ECMUL
            elif len(a) == 4 and all(fqs):
                c, k = [_n(x) for x in a], AUX_ECPAIRING
            else:
                k = AUX_OTHER
        else:
            k = AUX_OTHER
        cells.append(c)
        kinds.append(k)
    width = AUX_WIDE_CELLS if any(len(c) > 2 for c in cells) else 2
    cells = [c + [0] * (width - len(c)) for c in cells]
    return {"aux": rows_to_rowmajor(cells, width), "aux_kind": np.array(kinds, dtype=np.uint32)}


SIG_NCELLS = 9
ECC_NCELLS = 13


_SIG_T_SPEC = _spec("msg_hash.lo", "msg_hash.hi", "sig_v", "sig_r.lo", "sig_r.hi", "sig_s.lo", "sig_s.hi", "recovered_addr", "is_valid")
_ECC_T_SPEC = _spec("op_type", "px.lo", "px.hi", "py.lo", "py.hi", "qx.lo", "qx.hi", "qy.lo", "qy.hi", "input_rlc", "out_x", "out_y", "is_valid")
_WITHDRAWAL_SPEC = _spec("id", "validator_id", "address", "amount")


def flatten_sig_table(sig_table):
    """set of SigTableRow (evm_circuit/table.py:552-558) -> uint64[m, 9, 4]: msg_hash lo/hi, sig_v, sig_r lo/hi, sig_s lo/hi,
    recovered_addr, is_valid"""
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(sig_table), _SIG_T_SPEC)[0])[0]
    rows, _ = _dedup([([_n(r.msg_hash.lo), _n(r.msg_hash.hi), _n(r.sig_v), _n(r.sig_r.lo), _n(r.sig_r.hi), _n(r.sig_s.lo),
                        _n(r.sig_s.hi), _n(r.recovered_addr), _n(r.is_valid)], 0) for r in _iter_table(sig_table)])
    return rows_to_rowmajor(rows, SIG_NCELLS)


def flatten_ecc_table(ecc_table):
    """set of EccTableRow (evm_circuit/table.py:562-575) -> uint64[m, 13, 4]: op_type, px lo/hi, py lo/hi, qx lo/hi, qy lo/hi,
    input_rlc, out_x, out_y, is_valid"""
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(ecc_table), _ECC_T_SPEC)[0])[0]
    rows, _ = _dedup([([_n(r.op_type), _n(r.px.lo), _n(r.px.hi), _n(r.py.lo), _n(r.py.hi), _n(r.qx.lo), _n(r.qx.hi),
                        _n(r.qy.lo), _n(r.qy.hi), _n(r.input_rlc), _n(r.out_x), _n(r.out_y), _n(r.is_valid)], 0)
                      for r in _iter_table(ecc_table)])
    return rows_to_rowmajor(rows, ECC_NCELLS)


def flatten_withdrawal_table(withdrawal_table):
    """set of WithdrawalTableRow (evm_circuit/table.py:430-434: id, validator_id, address, amount) -> uint64[m, 4, 4],
    sorted by the cells (id first), which is the order end_block.py:152 walks them in"""
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(withdrawal_table), _WITHDRAWAL_SPEC)[0])[0]
    rows, _ = _dedup([([_n(r.id), _n(r.validator_id), _n(r.address), _n(r.amount)], 0) for r in _iter_table(withdrawal_table)])
    return rows_to_rowmajor(rows, 4)


def flatten_evm(tables, steps):
    """reference `Tables` (evm_circuit/table.py:578-671) + list of StepState -> dict of wire arrays.
    The copy / keccak / exp tables only exist on `Tables` built with those circuits (:614-619);
    absent ones travel as empty tables."""
    rw, rw_flags = flatten_rw_table(tables.rw_table)
    tx, tx_flags = flatten_tx_table(tables.tx_table)
    blk, blk_flags = flatten_block_table(tables.block_table)
    return {
        "steps": flatten_steps(steps),
        "rw": rw, "rw_flags": rw_flags,
        "bytecode": flatten_bytecode_table(tables.bytecode_table),
        "tx": tx, "tx_flags": tx_flags,
        "block": blk, "block_flags": blk_flags,
        **flatten_step_aux(steps),
        "withdrawals": flatten_withdrawal_table(getattr(tables, "withdrawal_table", None)),
        "copy": flatten_copy_table(getattr(tables, "copy_table", None)),
        "keccak": flatten_keccak_table(getattr(tables, "keccak_table", None)),
        "exp": flatten_exp_table(getattr(tables, "exp_table", None)),
        "sig": flatten_sig_table(getattr(tables, "sig_table", None)),
        "ecc": flatten_ecc_table(getattr(tables, "ecc_table", None)),
    }


# ---- Bytecode / Exp circuits ------------------------------------------------------------------------
BYTECODE_ROW_NCELLS = 12
KECCAK_NCELLS = 5
EXP_NCELLS = 21


def flatten_bytecode_rows(rows):
    """bytecode_circuit.Row (bytecode_circuit.py:15-26) -> uint64[12, n, 4] column-major"""
    cells = [[_n(r.q_first), _n(r.q_last), _n(r.hash.lo), _n(r.hash.hi), _n(r.tag), _n(r.index), _n(r.value),
              _n(r.is_code), _n(r.push_data_left), _n(r.value_rlc), _n(r.length), _n(r.push_data_size)] for r in rows]
    return rows_to_colmajor(cells, BYTECODE_ROW_NCELLS)


_KECCAK_T_SPEC = _spec("state_tag", "input_rlc", "input_len", "output.lo", "output.hi")


def flatten_keccak_table(keccak_table):
    """set of KeccakTableRow (evm_circuit/table.py:511-515) -> uint64[m, 5, 4]; rows already in wire
    form (e.g. from engine.keccak_table) pass through"""
    if isinstance(keccak_table, np.ndarray):
        return np.ascontiguousarray(keccak_table, dtype=np.uint64).reshape(-1, KECCAK_NCELLS, 4)
    if USE_EXT:
        return _dedup_rows(_pack(_iter_table(keccak_table), _KECCAK_T_SPEC)[0])[0]
    rows = sorted(set((_n(k.state_tag), _n(k.input_rlc), _n(k.input_len), _n(k.output.lo), _n(k.output.hi))
                      for k in _iter_table(keccak_table)))
    return rows_to_rowmajor([list(r) for r in rows], KECCAK_NCELLS)


def flatten_exp_rows(rows):
    """ExpCircuitRow (evm_circuit/table.py:519-535) -> uint64[21, n, 4] column-major"""
    cells = []
    for r in rows:
        c = [_n(r.q_usable), _n(r.is_step), _n(r.identifier), _n(r.is_last)]
        for w in (r.base, r.exponent, r.exponentiation, r.a, r.b, r.c, r.d, r.q):
            c += [_n(w.lo), _n(w.hi)]
        c.append(_n(r.r))
        cells.append(c)
    return rows_to_colmajor(cells, EXP_NCELLS)


# ---- Copy circuit --------------------------------------------------------------------------------------
COPY_NCELLS = 20


def flatten_copy_rows(rows):
    """CopyCircuitRow (evm_circuit/table.py:472-491) -> (uint64[20, n, 4] column-major, uint32 flags[n])"""
    cells, flags = [], []
    for r in rows:
        lo, hi, w = _word_cells(r.id)
        cells.append([_n(r.q_step), _n(r.is_first), _n(r.is_last), lo, hi, _n(r.tag), _n(r.addr), _n(r.src_addr_end),
                      _n(r.bytes_left), _n(r.value), _n(r.rlc_acc), _n(r.is_code), _n(r.is_pad), _n(r.rw_counter),
                      _n(r.rwc_inc_left), _n(r.is_memory), _n(r.is_bytecode), _n(r.is_tx_calldata), _n(r.is_tx_log),
                      _n(r.is_rlc_acc)])
        flags.append(1 if w else 0)
    return rows_to_colmajor(cells, COPY_NCELLS), np.array(flags, dtype=np.uint32)


# ---- Tx / Sig circuits ------------------------------------------------------------------------------------
SIGN_NBYTES_ROWS = 9
SIGN_NCELLS = 8


def _b32(x):
    """32 byte values, or None when the attribute is not a 32-byte bytes object (the reference's tests
    assign e.g. `Word(1)` to such attributes: every use then fails a type assert -> `malformed` bit)."""
    if not isinstance(x, (bytes, bytearray)) or len(x) != 32:
        return None
    return list(bytes(x))


def _byte_rows(fields):
    rows, malformed = [], 0
    for k, f in enumerate(fields):
        b = _b32(f)
        if b is None:
            malformed |= 1 << k
            b = [0] * 32
        rows.append(b)
    return rows, malformed


def _le_bytes_or_zero(get):
    try:
        return get().to_le_bytes()
    except Exception:  # noqa: BLE001 - tampered chips: the pre-computed ecdsa_status column carries the outcome
        return bytes(32)


def _ecdsa_status(call, returns_bool):
    """Outcome of the reference-style ECDSA chip's verify(): 0 verified, 1 not verified (returned False /
    asserted), otherwise (kind << 24) of the exception it raised — the pre-computed `ecdsa_status` column."""
    from .errors import kind_for_exception

    try:
        ok = call()
        return 0 if (ok or not returns_bool) else 1
    except AssertionError:
        return 1
    except Exception as e:  # noqa: BLE001 - third-party signature errors are data here
        return kind_for_exception(e) << 24


ECDSA_STATUS_PENDING = 0xFFFFFFFF  # meta[:, 0] placeholder of a verdict still to be computed: the kernel fails every unit left pending


def _b32_of(get):
    b = get()
    if not isinstance(b, (bytes, bytearray)) or len(b) != 32:
        raise ValueError("not a 32-byte string")
    return bytes(b)


def _chip_inputs_tx(e):
    """what `tx_circuit.ECDSAVerifyChip.verify` hands to eth_keys (tx_circuit.py:147-158): the chip's LIMBS (not its
    `*_bytes` attributes), as the packed layout of zk_ecdsa_verify (pk_x LE, pk_y LE, msg_hash BE, r LE, s LE); v = 0"""
    return [_b32_of(e.pub_key[0].to_le_bytes), _b32_of(e.pub_key[1].to_le_bytes), bytes(reversed(_b32_of(e.msg_hash.to_le_bytes))),
            _b32_of(e.signature[0].to_le_bytes), _b32_of(e.signature[1].to_le_bytes)], 0


def _chip_inputs_sig(e):
    """what `util.ec.ECDSAVerifyChip.verify` hands to eth_keys (util/ec.py:109-117)"""
    v = int.from_bytes(_b32_of(e.sig_v.to_le_bytes), "little")
    if v > 0xFFFFFFFF:
        raise ValueError("v does not fit the wire")
    return [bytes(reversed(_b32_of(e.pub_key[0].to_be_bytes))), bytes(reversed(_b32_of(e.pub_key[1].to_be_bytes))),
            _b32_of(e.msg_hash.to_be_bytes), _b32_of(e.sig_r.to_le_bytes), _b32_of(e.sig_s.to_le_bytes)], v


def _ecdsa_column(chips, inputs_of, host_call, returns_bool, on_device):
    """-> (status list, packed uint8[n, 5, 32], v uint32[n], deferred bool[n]).  With `on_device` the verdict of every chip
    whose five inputs can be read is left PENDING for the device pass (zk_ecdsa_verify over `packed`); chips whose
    attributes are malformed (the reference's tamper tests) are evaluated by calling them, as the reference does."""
    status, packed, vs, deferred = [], [], [], []
    for e in chips:
        rows, v, ok = [bytes(32)] * 5, 0, False
        if on_device:
            try:
                rows, v = inputs_of(e)
                ok = True
            except Exception:  # noqa: BLE001 - malformed chip: fall through to the chip's own verify()
                ok = False
        status.append(ECDSA_STATUS_PENDING if ok else _ecdsa_status(lambda e=e: host_call(e), returns_bool))
        packed.append([list(r) for r in rows])
        vs.append(v)
        deferred.append(ok)
    return (status, np.array(packed, dtype=np.uint8).reshape(-1, 5, 32), np.array(vs, dtype=np.uint32),
            np.array(deferred, dtype=bool))


def flatten_keccak_tuples(table):
    """tx_circuit.KeccakTable.table: set of (is_enabled, input_rlc, input_len, output Word) -> uint64[m, 5, 4]"""
    rows = sorted(set((_n(t[0]), _n(t[1]), _n(t[2]), _n(t[3].lo), _n(t[3].hi)) for t in table))
    return rows_to_rowmajor([list(r) for r in rows], KECCAK_NCELLS)


def flatten_tx_witness(witness, max_txs, ecdsa_on_device=False):
    """tx_circuit.Witness (rows, keccak_table, sign_verifications) -> dict of wire arrays.  ecdsa_on_device: leave the
    `ecdsa_status` column PENDING and return the chips' inputs (`ecdsa_packed`, `ecdsa_v`, `ecdsa_deferred`) instead of
    calling every chip's verify() on the host."""
    bts, cells, meta = [], [], []
    svs = witness.sign_verifications[:max_txs]
    col, packed, vs, deferred = _ecdsa_column([sv.ecdsa_chip for sv in svs], _chip_inputs_tx, lambda e: e.verify(""), False,
                                              ecdsa_on_device)
    for sv, st in zip(svs, col):
        e = sv.ecdsa_chip
        # rows 7, 8: the ECDSA chip's (r, s) as it hands them to eth_keys (tx_circuit.py:149-150); the Tx kernel does
        # not read them, the device ECDSA pass (zk_ecdsa_open, layout 1) does
        rows_, bad = _byte_rows([sv.pub_key_x_bytes, sv.pub_key_y_bytes, e.pub_key_x_bytes, e.pub_key_y_bytes,
                                 sv.msg_hash_bytes, e.msg_hash_bytes, sv.pub_key_hash,
                                 _le_bytes_or_zero(lambda e=e: e.signature[0]), _le_bytes_or_zero(lambda e=e: e.signature[1])])
        bts.append(rows_)
        cells.append([_n(sv.address), _n(sv.msg_hash.lo), _n(sv.msg_hash.hi), 0, 0, 0, 0, 0])
        meta.append([st, 1, bad, 0])
    pairs = []
    for r in witness.rows:
        lo, hi, w = _word_cells(r.value)
        pairs.append(([_n(r.tx_id), _n(r.tag), _n(r.index), lo, hi], 1 if w else 0))
    return {
        "bytes": np.array(bts, dtype=np.uint8).reshape(-1, SIGN_NBYTES_ROWS, 32),
        "cells": rows_to_colmajor(cells, SIGN_NCELLS),
        "meta": np.array(meta, dtype=np.uint32).reshape(-1, 4),
        "keccak": flatten_keccak_tuples(witness.keccak_table.table),
        "tx_rows": rows_to_rowmajor([c for c, _ in pairs], TX_NCELLS),
        "tx_flags": np.array([f for _, f in pairs], dtype=np.uint32),
        "ecdsa_packed": packed, "ecdsa_v": None, "ecdsa_deferred": deferred,
    }


def flatten_sig_witness(witness, ecdsa_on_device=False):
    """sig_circuit.Witness (rows, keccak_table) -> dict of wire arrays (ecdsa_on_device: see flatten_tx_witness)"""
    bts, cells, meta = [], [], []
    col, packed, vs, deferred = _ecdsa_column([row.ecdsa_chip for row in witness.rows], _chip_inputs_sig, lambda e: e.verify(), True,
                                              ecdsa_on_device)
    for row, st in zip(witness.rows, col):
        e = row.ecdsa_chip
        rows_, bad = _byte_rows([row.pub_key_x_bytes, row.pub_key_y_bytes, e.pub_key_x_bytes, e.pub_key_y_bytes,
                                 row.msg_hash_bytes, e.msg_hash_bytes, row.pub_key_hash, e.sig_r.le_bytes, e.sig_s.le_bytes])
        bts.append(rows_)
        cells.append([_n(row.recovered_addr), _n(row.msg_hash.lo), _n(row.msg_hash.hi), _n(row.sig_v), _n(row.sig_r.lo),
                      _n(row.sig_r.hi), _n(row.sig_s.lo), _n(row.sig_s.hi)])
        # meta[3]: the v the chip hands to eth_keys (util/ec.py:110; the Row's own sig_v cell can be tampered separately)
        chip_v = _le_bytes_or_zero(lambda e=e: e.sig_v)
        meta.append([st, int(bool(row.is_valid)), bad, min(int.from_bytes(chip_v, "little"), 0xFFFFFFFF)])
    return {
        "bytes": np.array(bts, dtype=np.uint8).reshape(-1, SIGN_NBYTES_ROWS, 32),
        "cells": rows_to_colmajor(cells, SIGN_NCELLS),
        "meta": np.array(meta, dtype=np.uint32).reshape(-1, 4),
        "keccak": flatten_keccak_tuples(witness.keccak_table.table),
        "tx_rows": np.zeros((0, TX_NCELLS, 4), dtype=np.uint64),
        "tx_flags": np.zeros(0, dtype=np.uint32),
        "ecdsa_packed": packed, "ecdsa_v": vs, "ecdsa_deferred": deferred,
    }


# ---- State-circuit witness assignment (ops wire) -----------------------------------------
OP_NSLOTS = 12


def state_op_slots(op):
    """reference state_circuit.Operation (:616-630) -> 12 slots + flags.  Slots 0-6 (rw_counter, rw, tag, id,
    address, field_tag, storage_key) are the op's Python ints as they are (U256, not reduced mod p); slots 7-11 are
    the field cells of value / initial_value / lexicographic_ordering_selector."""
    ints = [int(op.rw_counter), int(op.rw), int(op.tag), int(op.id), int(op.address), int(op.field_tag),
            int(op.storage_key)]
    for v in ints:
        if not 0 <= v < (1 << 256):
            raise OverflowError("state op field does not fit the 256-bit wire slot")
    vlo, vhi, vw = _word_cells(op.value)
    ilo, ihi, iw = _word_cells(op.initial_value)
    # `isinstance(field_tag, AccountFieldTag)` (state_circuit.py:915) is what selects the account proof types
    is_account_ft = type(op.field_tag).__name__ == "AccountFieldTag"
    flags = (1 if vw else 0) | (2 if iw else 0) | (4 if is_account_ft else 0)
    return ints + [vlo, vhi, ilo, ihi, _n(op.lexicographic_ordering_selector)], flags


def flatten_state_ops(ops):
    """-> (ops uint64[12, n, 4] column-major, flags uint32[n])"""
    sf = [state_op_slots(op) for op in ops]
    return rows_to_colmajor([s for s, _ in sf], OP_NSLOTS), np.array([f for _, f in sf], dtype=np.uint32)


# ---- Bytecode-circuit witness assignment (unrolled bytecodes wire) -----------------------
def flatten_unrolled_bytecodes(bytecodes):
    """Sequence of reference UnrolledBytecode (bytecode_circuit.py:31-33: `bytes` + the BytecodeTableRows of
    `Bytecode.table_assignments()`) -> (rows uint64[n, 6, 4] row-major in input order: hash lo/hi, tag, index, is_code,
    value; offsets uint64[m + 1] row offsets per bytecode; lengths uint64[m] = len(bytecode.bytes))."""
    cells, offsets, lengths = [], [0], []
    for b in bytecodes:
        for r in b.rows:
            cells.append([_n(r.bytecode_hash.lo), _n(r.bytecode_hash.hi), _n(r.field_tag), _n(r.index), _n(r.is_code), _n(r.value)])
        offsets.append(len(cells))
        lengths.append(len(b.bytes))
    return (rows_to_rowmajor(cells, BYTECODE_NCELLS), np.array(offsets, dtype=np.uint64), np.array(lengths, dtype=np.uint64))


# ---- Public-inputs (PI) circuit ------------------------------------------------------------------------
PI_NCELLS = 24


def flatten_pi_rows(rows):
    """pi_circuit.Row (pi_circuit.py:104-134) -> uint64[24, n, 4] column-major (cell order: oracle/pi_oracle.py)"""
    cells = [[_n(r.q_bytes_last), _n(r.q_tx_table), _n(r.q_tx_calldata), _n(r.q_tx_calldata_start), _n(r.q_rpi_keccak_lookup),
              _n(r.q_rpi_value_start), _n(r.tx_id_inv), _n(r.tx_value_lo_inv), _n(r.tx_id_diff_inv), _n(r.calldata_gas_cost), _n(r.is_final),
              _n(r.q_withdrawal_table), _n(r.rpi_bytes), _n(r.rpi_bytes_keccakrlc), _n(r.rpi_value_lc), _n(r.rpi_digest_word.lo),
              _n(r.rpi_digest_word.hi), _n(r.q_rpi_byte_enable), _n(r.tx_table.tx_id), _n(r.tx_table.tag), _n(r.tx_table.index),
              _n(r.tx_table.value.lo), _n(r.withdrawal_table.id), _n(r.withdrawal_table.amount)] for r in rows]
    return rows_to_colmajor(cells, PI_NCELLS)


def flatten_pi_gas_table(table):
    """set of TxCallDataGasCostAccRow (pi_circuit.py:64-68) -> uint64[m, 3, 4]"""
    rows = sorted(set((_n(g.tx_id), _n(g.is_final), _n(g.gas_cost_acc)) for g in table))
    return rows_to_rowmajor([list(r) for r in rows], 3)
