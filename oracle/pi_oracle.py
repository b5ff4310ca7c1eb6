"""Public-inputs (PI) circuit oracle (TEST INFRASTRUCTURE — see oracle/__init__.py).

Python-integer restatement of `check_row` (reference src/zkevm_specs/pi_circuit.py:150-322) over flattened wire rows.
Row cells (24): q_bytes_last, q_tx_table, q_tx_calldata, q_tx_calldata_start, q_rpi_keccak_lookup, q_rpi_value_start,
tx_id_inv, tx_value_lo_inv, tx_id_diff_inv, calldata_gas_cost, is_final, q_withdrawal_table, rpi_bytes,
rpi_bytes_keccakrlc, rpi_value_lc, rpi_digest lo, hi, q_rpi_byte_enable, tx_table.tx_id, .tag, .index, .value.lo,
withdrawal_table.id, .amount.  Tables: keccak rows (is_enabled, input_rlc, input_len, output lo, hi), calldata gas-cost
rows (tx_id, is_final, gas_cost_acc); the fixed u16 table is the range [0, 65536).
Site numbers follow the reference's evaluation order (csrc/pi_circuit.hpp uses the same).
Pinned to the reference by oracle/gen_golden_pi.py.
"""
from .codes import ASSERT, LOOKUP_AMBIGUOUS, LOOKUP_UNSAT, OK, Fail
from .wire import P

(Q_BYTES_LAST, Q_TX_TABLE, Q_TX_CALLDATA, Q_TX_CALLDATA_START, Q_KECCAK, Q_VALUE_START, TX_ID_INV, TX_LO_INV, TX_DIFF_INV, GAS_COST, IS_FINAL,
 Q_WD, RPI_BYTES, RPI_RLC, RPI_LC, DIGEST_LO, DIGEST_HI, Q_BYTE_EN, TX_ID, TX_TAG, TX_INDEX, TX_LO, WD_ID, WD_AMOUNT) = range(24)
NCELLS = 24
TAG_CALLDATA_LENGTH = 8          # tx_circuit.Tag.CallDataLength
GAS_NONZERO, GAS_ZERO = 16, 4    # GAS_COST_TX_CALL_DATA_PER_NON_ZERO_BYTE / _ZERO_BYTE (util/param.py)


def _a(cond, site):
    if not cond:
        raise Fail(ASSERT, site)


def check_row(rows, i, gas_table, keccak_table, circuit_len, keccak_rand=255, byte_pow_base=255):
    """gas_table: set of (tx_id, is_final, gas_cost_acc); keccak_table: set of 5-tuples"""
    r, nx = rows[i], rows[(i + 1) % len(rows)]
    try:
        en, last = r[Q_BYTE_EN], r[Q_BYTES_LAST]
        _a(en * last * (r[RPI_RLC] - r[RPI_BYTES]) % P == 0, 1)
        _a(en * (1 - last) * (r[RPI_RLC] - (nx[RPI_RLC] * keccak_rand + r[RPI_BYTES])) % P == 0, 2)
        _a(en * (1 - r[Q_VALUE_START]) * (r[RPI_LC] - (nx[RPI_LC] * byte_pow_base + r[RPI_BYTES])) % P == 0, 3)
        _a(en * r[Q_VALUE_START] * (r[RPI_LC] - r[RPI_BYTES]) % P == 0, 4)
        q = r[Q_KECCAK]
        d_lo, d_hi = r[DIGEST_LO] * q % P, r[DIGEST_HI] * q % P
        # rpi_digest_word.select(q) builds a checked Word (util/arithmetic.py:148-150, :110-112): an AssertionError too
        _a(d_lo < (1 << 128) and d_hi < (1 << 128) and (q, q * r[RPI_RLC] % P, q * circuit_len % P, d_lo, d_hi) in keccak_table, 5)
        if r[Q_TX_CALLDATA] != 0:
            tx_id, lo, nid = r[TX_ID], r[TX_LO], nx[TX_ID]
            _a(tx_id * (1 - r[TX_ID_INV] * tx_id) % P == 0, 6)
            _a(lo * (1 - r[TX_LO_INV] * lo) % P == 0, 7)
            _a((nid - tx_id) * (1 - r[TX_DIFF_INV] * (nid - tx_id)) % P == 0, 8)
            id_nz = tx_id * r[TX_ID_INV] % P
            id_next_nz = nid * nx[TX_ID_INV] % P
            id_z, id_next_z = (1 - id_nz) % P, (1 - id_next_nz) % P
            neq_next = (nid - tx_id) * r[TX_DIFF_INV] % P
            eq_next = (1 - neq_next) % P
            byte_nz = lo * r[TX_LO_INV] % P
            byte_next_nz = nx[TX_LO] * nx[TX_LO_INV] % P
            byte_z, byte_next_z = (1 - byte_nz) % P, (1 - byte_next_nz) % P
            for k, cons in enumerate((id_z * tx_id, id_z * nid, id_z * r[IS_FINAL], id_z * r[GAS_COST])):
                _a(cons % P == 0, 9 + k)
            gas = (GAS_NONZERO * byte_nz + GAS_ZERO * byte_z) % P
            gas_next = (GAS_NONZERO * byte_next_nz + GAS_ZERO * byte_next_z) % P
            v = neq_next * id_next_nz * (nid - tx_id - 1) % P
            if v >= 65536:
                raise Fail(LOOKUP_UNSAT, 13)
            idx_same = eq_next * (nx[TX_INDEX] - r[TX_INDEX] - 1)
            idx_next = (nid - tx_id) * nx[TX_INDEX]
            gas_same = eq_next * (nx[GAS_COST] - r[GAS_COST] - gas_next)
            gas_nexttx = id_next_nz * (nid - tx_id) * (nx[GAS_COST] - gas_next)
            gas_last = id_next_z * nx[GAS_COST]
            fin_same = eq_next * r[IS_FINAL]
            fin_next = (nid - tx_id) * (r[IS_FINAL] - 1)
            for k, cons in enumerate((idx_same, idx_next, gas_same, gas_nexttx, gas_last, fin_same, fin_next)):
                _a(id_nz * cons % P == 0, 14 + k)
            _a(r[Q_TX_CALLDATA_START] * id_nz * r[TX_INDEX] % P == 0, 21)
            _a(r[Q_TX_CALLDATA_START] * id_nz * (r[GAS_COST] - gas) % P == 0, 22)
        if r[Q_TX_TABLE] != 0:
            is_cdl = (r[TX_TAG] - TAG_CALLDATA_LENGTH) % P
            lo = r[TX_LO]
            _a(is_cdl * (1 - r[TX_ID_INV] * is_cdl) % P == 0, 23)
            _a(lo * (1 - r[TX_LO_INV] * lo) % P == 0, 24)
            cdl_row = (1 - is_cdl * r[TX_ID_INV]) % P
            len_nz = lo * r[TX_LO_INV] % P
            len_z = (1 - len_nz) % P
            cost = nx[TX_LO]
            _a(cdl_row * len_z * cost % P == 0, 25)
            cond = cdl_row * len_nz % P
            query = (r[TX_ID] * cond % P, cond, cost * cond % P)
            if query not in gas_table:
                raise Fail(LOOKUP_UNSAT, 26)
        if r[Q_WD] != 0:
            if nx[Q_WD] != 0:
                _a(nx[WD_ID] == (r[WD_ID] + 1) % P, 27)
            _a(r[WD_AMOUNT] != 0, 28)
    except Fail as f:
        return f.code
    return OK


def verify_rows(rows, gas_rows, keccak_rows, circuit_len, keccak_rand=255, byte_pow_base=255):
    gas = set(tuple(g) for g in gas_rows)
    kt = set(tuple(k) for k in keccak_rows)
    return [check_row(rows, i, gas, kt, circuit_len % P, keccak_rand, byte_pow_base) for i in range(len(rows))]


def copy_constraints_status(cells, data, lens):
    """Per-constraint status of the PI circuit's copy constraints (pi_circuit.py:355-445), in the wire form the mirror lists them
    (zkevm_specs_amd/pi_circuit.py): `assert cell == bytes_to_fq(entry[::-1])`, where bytes_to_fq asserts len(entry) <= 31 first
    (util/arithmetic.py:227-229); lens == 0xFFFFFFFF: the 32 bytes are a canonical cell compared as is (the word equality, :358).
    cells: list of ints, data: uint8[n][32], lens: ints.  Site 1 = the length assert, 2 = the equality."""
    from .codes import ASSERT, code

    out = []
    for c, d, ln in zip(cells, data, lens):
        ln = int(ln)
        raw = bytes(bytearray(int(x) for x in d))
        if ln == 0xFFFFFFFF:
            out.append(0 if int(c) == int.from_bytes(raw, "little") else code(ASSERT, 2))
            continue
        if ln > 31:
            out.append(code(ASSERT, 1))
            continue
        entry = raw[:ln]
        out.append(0 if int(c) == int.from_bytes(entry[::-1], "little") % P else code(ASSERT, 2))
    return out

