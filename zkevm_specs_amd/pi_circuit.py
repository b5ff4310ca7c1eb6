"""Host-side mirror of `zkevm_specs.pi_circuit.verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS)`
(pi_circuit.py:338-459).  Both halves run on the device: the copy constraints in front of the gates (:355-445) — block / tx /
withdrawal table cells and the public inputs' words against the byte strings of `witness.copy_constrains` — are listed here in
the reference's statement order (it consumes `witness.copy_constrains` with pop(0); so does this) and evaluated by one launch of
`zk_pi_copy_verify` (csrc/pi_circuit.hpp pi_copy_check: bytes_to_fq's length assert and the equality, per entry); the per-row
gates and lookups (`check_row`, :150-322) by one launch of `zk_pi_verify`.  Every copy-constraint failure is an AssertionError
in the reference; the gate pass reports the exception class of its first failing row."""
import numpy as np

from . import oneshot
from .errors import raise_for_code
from .flatten import _n, flatten_keccak_tuples, flatten_pi_gas_table, flatten_pi_rows

BLOCK_LEN = (8 + 256) * 2   # PUBLIC_INPUTS_BLOCK_LEN (util/param.py:126)
TX_LEN = 10                 # PUBLIC_INPUTS_TX_LEN (util/param.py:128)
MAX_N_BYTES = 31            # util/param.py: bytes_to_fq asserts len(value) <= MAX_N_BYTES (on the device: pi_copy_check site 1)
KECCAK_RAND = BYTE_POW_BASE = 255  # pi_circuit.py:834-836


PI_COPY_CELL = 0xFFFFFFFF  # csrc/pi_circuit.hpp: the 32 bytes are a canonical cell compared as is


class _Constraints:
    """the copy constraints of one witness, in the reference's statement order"""

    def __init__(self, cc):
        self.cc, self.cells, self.entries = cc, [], []

    def cell_eq(self, cell, other):       # `Word.__eq__`: lo / hi expressions compared (util/arithmetic.py:138-140)
        self.cells.append(_n(cell))
        self.entries.append((PI_COPY_CELL, int(_n(other)).to_bytes(32, "little")))

    def pop(self):                        # copy_constrains.pop(0)[::-1]: kept as popped, the device reads it big-endian
        return bytes(self.cc.pop(0))

    def eq(self, cell, entry):            # assert cell == bytes_to_fq(entry[::-1])
        self.cells.append(_n(cell))
        self.entries.append((len(entry), entry))

    def lo_hi(self, lo, hi, is_word):     # lo_le = pop; hi_le = pop if is_word else b""; then the two asserts (pops come first)
        lo_e = self.pop()
        hi_e = self.pop() if is_word else b""
        self.eq(lo, lo_e)
        self.eq(hi, hi_e)

    def wire(self):
        n = len(self.cells)
        cells = np.zeros((n, 4), dtype=np.uint64)
        data = np.zeros((n, 32), dtype=np.uint8)
        lens = np.zeros(n, dtype=np.uint32)
        for i, (c, (ln, e)) in enumerate(zip(self.cells, self.entries)):
            cells[i] = np.frombuffer(int(c).to_bytes(32, "little"), dtype="<u8")
            lens[i] = ln
            k = min(len(e), 32)  # an entry longer than its slot fails bytes_to_fq's length assert whatever its bytes
            data[i, :k] = np.frombuffer(e[:k], dtype=np.uint8)
        return cells, data, lens


def list_copy_constraints(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS):
    """-> (_Constraints, pending): the constraints listed until `witness.copy_constrains` ran out (`pending` = the IndexError the
    reference's pop(0) raises at that statement, raised after the constraints in front of it have passed), else all of them"""
    rows, public_inputs = witness.rows, witness.public_inputs
    block_table, tx_table, withdrawal_table = witness.block_table, witness.tx_table, witness.withdrawal_table
    C = _Constraints(witness.copy_constrains)
    try:
        # constrain witness rpi digest lo/hi equals pi input keccak lo/hi (:358)
        C.cell_eq(rows[0].rpi_digest_word.lo, public_inputs.pi_keccak.lo)
        C.cell_eq(rows[0].rpi_digest_word.hi, public_inputs.pi_keccak.hi)
        # block table word_or_value equals witness rpi bytes in vertical order (:361-372)
        for i in range(BLOCK_LEN // 2 + 1):
            block_row = block_table.table[i]
            C.lo_hi(block_row.lo, block_row.hi, block_row.is_word)
        # block_hash, state_root, state_root_prev (:374-393)
        for w in (public_inputs.block_hash, public_inputs.state_root, public_inputs.state_root_prev):
            C.lo_hi(w.lo, w.hi, True)
        # tx table id, index, value per row (:395-410)
        tx_len = TX_LEN * MAX_TXS + 1
        for i in range(tx_len):
            tx_row = tx_table.table[i]
            C.eq(tx_row.tx_id, C.pop())
            C.eq(tx_row.index, C.pop())
            C.lo_hi(tx_row.value.lo, tx_row.value.hi, tx_row.value.is_word)
        # tx calldata values (:412-423)
        for i in range(MAX_CALLDATA_BYTES):
            value = tx_table.table[tx_len + i].value
            C.lo_hi(value.lo, value.hi, value.is_word)
        # withdrawal table (:425-444)
        for i in range(MAX_WITHDRAWALS):
            wd = withdrawal_table.table[i]
            C.eq(wd.id, C.pop())
            C.eq(wd.validator_id, C.pop())
            C.lo_hi(wd.address.lo, wd.address.hi, True)
            C.eq(wd.amount, C.pop())
    except IndexError as e:  # pop(0) from an exhausted list / a table shorter than the circuit's shape
        return C, e
    return C, None


def verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS):
    rows = witness.rows
    C, pending = list_copy_constraints(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS)
    if C.cells:
        res, _ = oneshot.pi_copy_verify(*C.wire())  # zk_pi_copy_verify
        raise_for_code(res.first_fail_code, f"PI circuit copy constraint {res.first_fail_row}")
    if pending is not None:
        raise pending
    # gates (:447-459): one device pass over all rows
    res, _ = oneshot.pi_verify(flatten_pi_rows(rows), flatten_keccak_tuples(witness.keccak_table.table),
                               flatten_pi_gas_table(witness.calldata_gas_cost_table), int(witness.circuit_len), KECCAK_RAND, BYTE_POW_BASE)
    raise_for_code(res.first_fail_code, f"PI circuit row {res.first_fail_row}")
    return res
