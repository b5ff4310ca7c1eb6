"""Host-side mirror of `zkevm_specs.pi_circuit.verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS)`
(pi_circuit.py:338-459).  The per-row gates and lookups (`check_row`, :150-322) run as one device pass (`zk_pi_verify`); the
copy constraints in front of them (:355-445) — block / tx / withdrawal table cells against the byte strings of
`witness.copy_constrains` — are plain equality checks on a few thousand integers and stay on the host, statement by
statement like the reference (including that it consumes `witness.copy_constrains` with pop(0)).  Every failure is an
AssertionError there; the gate pass reports the exception class of its first failing row."""
from . import oneshot
from .errors import raise_for_code
from .flatten import FR_MODULUS, _n, flatten_keccak_tuples, flatten_pi_gas_table, flatten_pi_rows

BLOCK_LEN = (8 + 256) * 2   # PUBLIC_INPUTS_BLOCK_LEN (util/param.py:126)
TX_LEN = 10                 # PUBLIC_INPUTS_TX_LEN (util/param.py:128)
MAX_N_BYTES = 31            # util/param.py: bytes_to_fq asserts len(value) <= MAX_N_BYTES
KECCAK_RAND = BYTE_POW_BASE = 255  # pi_circuit.py:834-836


def _bytes_to_fq(value):
    assert len(value) <= MAX_N_BYTES  # util/arithmetic.py:227-229
    return int.from_bytes(value, "little") % FR_MODULUS


def _word_eq(a, b):
    return _n(a.lo) == _n(b.lo) and _n(a.hi) == _n(b.hi)


def verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS):
    rows = witness.rows
    public_inputs, block_table, tx_table, withdrawal_table = witness.public_inputs, witness.block_table, witness.tx_table, witness.withdrawal_table
    cc = witness.copy_constrains
    # constrain witness rpi digest lo/hi equals pi input keccak lo/hi (:358)
    assert _word_eq(rows[0].rpi_digest_word, public_inputs.pi_keccak)
    # block table word_or_value equals witness rpi bytes in vertical order (:361-372)
    for i in range(BLOCK_LEN // 2 + 1):
        block_row = block_table.table[i]
        lo_le = cc.pop(0)[::-1]
        hi_le = cc.pop(0)[::-1] if block_row.is_word else bytes(0)
        assert _n(block_row.lo) == _bytes_to_fq(lo_le)
        assert _n(block_row.hi) == _bytes_to_fq(hi_le)
    # block_hash, state_root, state_root_prev (:374-393)
    for w in (public_inputs.block_hash, public_inputs.state_root, public_inputs.state_root_prev):
        lo_le, hi_le = cc.pop(0)[::-1], cc.pop(0)[::-1]
        assert _n(w.lo) == _bytes_to_fq(lo_le)
        assert _n(w.hi) == _bytes_to_fq(hi_le)
    # tx table id, index, value per row (:395-410)
    tx_len = TX_LEN * MAX_TXS + 1
    for i in range(tx_len):
        tx_row = tx_table.table[i]
        assert _n(tx_row.tx_id) == _bytes_to_fq(cc.pop(0)[::-1])
        assert _n(tx_row.index) == _bytes_to_fq(cc.pop(0)[::-1])
        lo_le = cc.pop(0)[::-1]
        hi_le = cc.pop(0)[::-1] if tx_row.value.is_word else bytes(0)
        assert _n(tx_row.value.lo) == _bytes_to_fq(lo_le)
        assert _n(tx_row.value.hi) == _bytes_to_fq(hi_le)
    # tx calldata values (:412-423)
    for i in range(MAX_CALLDATA_BYTES):
        value = tx_table.table[tx_len + i].value
        lo_le = cc.pop(0)[::-1]
        hi_le = cc.pop(0)[::-1] if value.is_word else bytes(0)
        assert _n(value.lo) == _bytes_to_fq(lo_le)
        assert _n(value.hi) == _bytes_to_fq(hi_le)
    # withdrawal table (:425-444)
    for i in range(MAX_WITHDRAWALS):
        wd = withdrawal_table.table[i]
        assert _n(wd.id) == _bytes_to_fq(cc.pop(0)[::-1])
        assert _n(wd.validator_id) == _bytes_to_fq(cc.pop(0)[::-1])
        lo_le, hi_le = cc.pop(0)[::-1], cc.pop(0)[::-1]
        assert _n(wd.address.lo) == _bytes_to_fq(lo_le)
        assert _n(wd.address.hi) == _bytes_to_fq(hi_le)
        assert _n(wd.amount) == _bytes_to_fq(cc.pop(0)[::-1])
    # gates (:447-459): one device pass over all rows
    res, _ = oneshot.pi_verify(flatten_pi_rows(rows), flatten_keccak_tuples(witness.keccak_table.table),
                               flatten_pi_gas_table(witness.calldata_gas_cost_table), int(witness.circuit_len), KECCAK_RAND, BYTE_POW_BASE)
    raise_for_code(res.first_fail_code, f"PI circuit row {res.first_fail_row}")
    return res
