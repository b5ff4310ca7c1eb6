"""Witness objects rebuilt from wire arrays: the inverse of flatten.py.

The host mirrors (`verify_steps`, `verify_state_rows`, `verify_copy_table`, ...) take the reference's own objects
(`StepState`, `RWTableRow`, `state_circuit.Row`, ...) and only read attributes.  These light-weight stand-ins carry
the same attribute names (reference `evm_circuit/table.py:405-575`, `evm_circuit/step.py:16-75`,
`state_circuit.py:63-96`, `bytecode_circuit.py:15-26`), so a witness that exists as wire arrays (a golden fixture, the
output of a device-side assignment) can be handed to the same entry points a reference user calls.  Field values are
`FQ` stand-ins (an `int` in `.n`); nothing here does field arithmetic.
"""
from types import SimpleNamespace

from .evm_tables import ExecutionState
from .wire import cells_to_ints


class FQ:
    """Canonical field value: `.n` like py_ecc's FQ (util/arithmetic.py:41-63)."""

    __slots__ = ("n",)

    def __init__(self, n):
        self.n = int(n)

    def expr(self):
        return self

    def __int__(self):
        return self.n

    def __repr__(self):
        return f"FQ({self.n})"


class Word:
    """(lo, hi) 128-bit halves (util/arithmetic.py:99-168)."""

    is_word = True

    def __init__(self, lo, hi=None):
        if hi is None:  # Word(int)
            lo, hi = int(lo) & ((1 << 128) - 1), int(lo) >> 128
        self.lo, self.hi = FQ(lo), FQ(hi)


class WordOrValue(Word):
    """Word or single field value (util/arithmetic.py:171-195); a non-word value keeps whatever sits in its hi cell."""

    def __init__(self, lo, hi, is_word):
        super().__init__(lo, hi)
        self.is_word = bool(is_word)


def _rows(arr):
    """uint64[n, ncells, 4] (row-major) -> list of int lists"""
    n, nc = int(arr.shape[0]), int(arr.shape[1])
    flat = cells_to_ints(arr)
    return [flat[i * nc:(i + 1) * nc] for i in range(n)]


def _cols(arr):
    """uint64[ncells, n, 4] (column-major) -> list of int lists per row"""
    import numpy as np

    return _rows(np.ascontiguousarray(np.asarray(arr).transpose(1, 0, 2)))


def _aux_object(kind, a):
    """StepState.aux_data from its wire kind + cells (flatten.flatten_step_aux)"""
    W = Word
    if kind == 0:
        return None
    if kind == 1:
        return W(a[0], a[1])
    if kind == 2:
        return a[0] | (a[1] << 128)
    if kind == 3:
        return [a[0], a[1]]
    if kind == 5:
        d = SimpleNamespace(msg_hash=W(a[0], a[1]), sig_v=W(a[2], a[3]), sig_r=W(a[4], a[5]), sig_s=W(a[6], a[7]),
                            recovered_addr=FQ(a[8]), input_rlc=FQ(a[9]), output_rlc=FQ(a[10]))
        return [d, FQ(a[11])]
    if kind == 6:
        return [W(a[0], a[1]), W(a[2], a[3]), W(a[4], a[5]), W(a[6], a[7]), FQ(a[8]), FQ(a[9])]
    if kind == 7:
        return [W(a[0], a[1]), W(a[2], a[3]), W(a[4], a[5]), FQ(a[6]), FQ(a[7])]
    if kind == 8:
        return [FQ(a[0]), FQ(a[1]), FQ(a[2]), FQ(a[3])]
    return object()  # kind 4: a shape the wire does not carry


def steps_from_wire(steps, aux=None, aux_kind=None):
    out = []
    for i, c in enumerate(_rows(steps)):
        try:
            state = ExecutionState(c[0])
        except ValueError:
            state = c[0]
        s = SimpleNamespace(execution_state=state, rw_counter=FQ(c[1]), call_id=FQ(c[2]), is_root=bool(c[3]), is_create=bool(c[4]),
                            code_hash=Word(c[5], c[6]), program_counter=FQ(c[7]), stack_pointer=FQ(c[8]), gas_left=FQ(c[9]),
                            memory_word_size=FQ(c[10]), reversible_write_counter=FQ(c[11]), log_id=FQ(c[12]), aux_data=None)
        out.append(s)
    if aux is not None and aux_kind is not None:
        for s, a, k in zip(out, _rows(aux), aux_kind):
            s.aux_data = _aux_object(int(k), a)
    return out


def rw_table_from_wire(rw, rw_flags):
    return [SimpleNamespace(rw_counter=FQ(c[0]), rw=FQ(c[1]), key0=FQ(c[2]), id=FQ(c[3]), address=FQ(c[4]), field_tag=FQ(c[5]),
                            storage_key=Word(c[6], c[7]), value=WordOrValue(c[8], c[9], int(f) & 1),
                            value_prev=WordOrValue(c[10], c[11], int(f) & 2), aux0=Word(c[12], c[13]))
            for c, f in zip(_rows(rw), rw_flags)]


def bytecode_table_from_wire(bytecode):
    return [SimpleNamespace(bytecode_hash=Word(c[0], c[1]), field_tag=FQ(c[2]), index=FQ(c[3]), is_code=FQ(c[4]), value=FQ(c[5]))
            for c in _rows(bytecode)]


def tx_table_from_wire(tx, tx_flags):
    return [SimpleNamespace(tx_id=FQ(c[0]), field_tag=FQ(c[1]), call_data_index_or_zero=FQ(c[2]), value=WordOrValue(c[3], c[4], int(f) & 1))
            for c, f in zip(_rows(tx), tx_flags)]


def block_table_from_wire(block, block_flags):
    return [SimpleNamespace(field_tag=FQ(c[0]), block_number_or_zero=FQ(c[1]), value=WordOrValue(c[2], c[3], int(f) & 1))
            for c, f in zip(_rows(block), block_flags)]


def tables_from_wire(w):
    """wire dict (engine.open_evm's) -> object with the attributes of the reference's `Tables` (table.py:578-671)"""
    t = SimpleNamespace(
        rw_table=rw_table_from_wire(w["rw"], w["rw_flags"]),
        bytecode_table=bytecode_table_from_wire(w["bytecode"]),
        tx_table=tx_table_from_wire(w["tx"], w["tx_flags"]),
        block_table=block_table_from_wire(w["block"], w["block_flags"]),
    )
    if "withdrawals" in w:
        t.withdrawal_table = [SimpleNamespace(id=FQ(c[0]), validator_id=FQ(c[1]), address=FQ(c[2]), amount=FQ(c[3]))
                              for c in _rows(w["withdrawals"])]
    if "copy" in w:
        t.copy_table = [SimpleNamespace(is_first=FQ(c[0]), src_id=Word(c[1], c[2]), src_tag=FQ(c[3]), dst_id=Word(c[4], c[5]),
                                        dst_tag=FQ(c[6]), src_addr=FQ(c[7]), src_addr_end=FQ(c[8]), dst_addr=FQ(c[9]), length=FQ(c[10]),
                                        rlc_acc=FQ(c[11]), rw_counter=FQ(c[12]), rwc_inc=FQ(c[13])) for c in _rows(w["copy"])]
    if "keccak" in w:
        t.keccak_table = keccak_table_from_wire(w["keccak"])
    if "exp" in w:
        t.exp_table = [SimpleNamespace(is_step=FQ(c[0]), identifier=FQ(c[1]), is_last=FQ(c[2]), base_limb0=FQ(c[3]), base_limb1=FQ(c[4]),
                                       base_limb2=FQ(c[5]), base_limb3=FQ(c[6]), exponent=Word(c[7], c[8]),
                                       exponentiation=Word(c[9], c[10])) for c in _rows(w["exp"])]
    if "sig" in w:
        t.sig_table = [SimpleNamespace(msg_hash=Word(c[0], c[1]), sig_v=FQ(c[2]), sig_r=Word(c[3], c[4]), sig_s=Word(c[5], c[6]),
                                       recovered_addr=FQ(c[7]), is_valid=FQ(c[8])) for c in _rows(w["sig"])]
    if "ecc" in w:
        t.ecc_table = [SimpleNamespace(op_type=FQ(c[0]), px=Word(c[1], c[2]), py=Word(c[3], c[4]), qx=Word(c[5], c[6]), qy=Word(c[7], c[8]),
                                       input_rlc=FQ(c[9]), out_x=FQ(c[10]), out_y=FQ(c[11]), is_valid=FQ(c[12])) for c in _rows(w["ecc"])]
    return t


def evm_from_wire(w):
    """-> (tables, steps) for `verify_steps(tables, steps, ...)`"""
    return tables_from_wire(w), steps_from_wire(w["steps"], w.get("aux"), w.get("aux_kind"))


def keccak_table_from_wire(keccak):
    return [SimpleNamespace(state_tag=FQ(c[0]), input_rlc=FQ(c[1]), input_len=FQ(c[2]), output=Word(c[3], c[4])) for c in _rows(keccak)]


def state_rows_from_wire(cols, flags):
    """uint64[57, n, 4] + flags -> list of state_circuit.Row-like objects"""
    out = []
    for c, f in zip(_cols(cols), flags):
        out.append(SimpleNamespace(
            rw_counter=FQ(c[0]), is_write=FQ(c[1]), keys=(FQ(c[2]), FQ(c[3]), FQ(c[4]), FQ(c[5]), Word(c[6], c[7])),
            key2_limbs=tuple(FQ(x) for x in c[8:18]), key45_bytes=tuple(FQ(x) for x in c[18:50]),
            value=WordOrValue(c[50], c[51], int(f) & 1), initial_value=WordOrValue(c[52], c[53], int(f) & 2),
            root=Word(c[54], c[55]), lexicographic_ordering_selector=FQ(c[56])))
    return out


def mpt_table_from_wire(mpt):
    return [SimpleNamespace(address=FQ(c[0]), proof_type=FQ(c[1]), storage_key=Word(c[2], c[3]), root=Word(c[4], c[5]),
                            root_prev=Word(c[6], c[7]), value=Word(c[8], c[9]), value_prev=Word(c[10], c[11])) for c in _rows(mpt)]


def bytecode_rows_from_wire(cols):
    return [SimpleNamespace(q_first=FQ(c[0]), q_last=FQ(c[1]), hash=Word(c[2], c[3]), tag=FQ(c[4]), index=FQ(c[5]), value=FQ(c[6]),
                            is_code=FQ(c[7]), push_data_left=FQ(c[8]), value_rlc=FQ(c[9]), length=FQ(c[10]), push_data_size=FQ(c[11]))
            for c in _cols(cols)]


def exp_rows_from_wire(cols):
    out = []
    for c in _cols(cols):
        w = [Word(c[4 + 2 * k], c[5 + 2 * k]) for k in range(8)]
        out.append(SimpleNamespace(q_usable=FQ(c[0]), is_step=FQ(c[1]), identifier=FQ(c[2]), is_last=FQ(c[3]), base=w[0], exponent=w[1],
                                   exponentiation=w[2], a=w[3], b=w[4], c=w[5], d=w[6], q=w[7], r=FQ(c[20])))
    return out


def copy_rows_from_wire(cols, flags):
    out = []
    for c, f in zip(_cols(cols), flags):
        out.append(SimpleNamespace(q_step=FQ(c[0]), is_first=FQ(c[1]), is_last=FQ(c[2]), id=WordOrValue(c[3], c[4], int(f) & 1), tag=FQ(c[5]),
                                   addr=FQ(c[6]), src_addr_end=FQ(c[7]), bytes_left=FQ(c[8]), value=FQ(c[9]), rlc_acc=FQ(c[10]),
                                   is_code=FQ(c[11]), is_pad=FQ(c[12]), rw_counter=FQ(c[13]), rwc_inc_left=FQ(c[14]), is_memory=FQ(c[15]),
                                   is_bytecode=FQ(c[16]), is_tx_calldata=FQ(c[17]), is_tx_log=FQ(c[18]), is_rlc_acc=FQ(c[19])))
    return out


class CircuitRows:
    """`.table()` holder like the reference's CopyCircuit / ExpCircuit (evm_circuit/typing.py:868-880, :1153)"""

    def __init__(self, rows):
        self.rows = list(rows)

    def table(self):
        return self.rows


def pi_rows_from_wire(cols, keccak_table=None):
    """uint64[24, n, 4] -> pi_circuit.Row-like objects (pi_circuit.py:104-134)"""
    out = []
    for c in _cols(cols):
        out.append(SimpleNamespace(
            q_bytes_last=FQ(c[0]), q_tx_table=FQ(c[1]), q_tx_calldata=FQ(c[2]), q_tx_calldata_start=FQ(c[3]), q_rpi_keccak_lookup=FQ(c[4]),
            q_rpi_value_start=FQ(c[5]), tx_id_inv=FQ(c[6]), tx_value_lo_inv=FQ(c[7]), tx_id_diff_inv=FQ(c[8]), calldata_gas_cost=FQ(c[9]),
            is_final=FQ(c[10]), q_withdrawal_table=FQ(c[11]), rpi_bytes=FQ(c[12]), rpi_bytes_keccakrlc=FQ(c[13]), rpi_value_lc=FQ(c[14]),
            rpi_digest_word=Word(c[15], c[16]), q_rpi_byte_enable=FQ(c[17]), keccak_table=keccak_table,
            tx_table=SimpleNamespace(tx_id=FQ(c[18]), tag=FQ(c[19]), index=FQ(c[20]), value=WordOrValue(c[21], 0, False)),
            withdrawal_table=SimpleNamespace(id=FQ(c[22]), validator_id=FQ(0), address=Word(0, 0), amount=FQ(c[23]))))
    return out
