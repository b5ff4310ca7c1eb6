"""Host-side mirror of the reference's Bytecode-circuit interface, evaluated on the MI355X.

Reference seam (src/zkevm_specs/bytecode_circuit.py): `check_bytecode_row(cur, next, push_table,
keccak_table, keccak_randomness)` :37 is called per row by the driver loop of the reference's
tests (tests/test_bytecode_circuit.py:26-47, next row wraps modulo n).  `verify_bytecode_rows` is
that loop as one device pass.  A failing row raises AssertionError like the reference.
"""
from . import engine, oneshot
from .errors import KIND_ASSERT, exception_for_code, raise_for_code
from .flatten import _n, flatten_bytecode_rows, flatten_keccak_table, flatten_unrolled_bytecodes


def assign_keccak_table(bytecodes, keccak_randomness):
    """`assign_keccak_table(bytecodes, keccak_randomness)` (bytecode_circuit.py:182-186) on the device:
    one KeccakCircuit.add row per bytecode, as wire rows uint64[n, 5, 4] (accepted by every
    keccak_table argument of this package)."""
    from .errors import exception_for_code

    bytecodes = [bytes(b) for b in bytecodes]
    if not bytecodes:
        return engine.keccak_table([], 0)
    data, offsets = engine.pack_messages(bytecodes)
    res, _, rows = oneshot.keccak_table(data, offsets, _n(keccak_randomness), engine.KECCAK_MODE_CIRCUIT)  # zk_keccak_table
    if not res.ok:
        raise exception_for_code(res.first_fail_code, f"keccak table: message {res.first_fail_row}")
    return rows


def assign_bytecode_circuit(k, bytecodes, keccak_randomness):
    """`assign_bytecode_circuit(k, bytecodes, keccak_randomness)` (bytecode_circuit.py:104-167) on the device: bytecodes =
    sequence of reference-style UnrolledBytecode (or the (rows, offsets, lengths) wire arrays) -> the 2^k circuit rows
    in wire form uint64[12, 2^k, 4] (what verify_bytecode_rows / engine.open_bytecode take)."""
    in_rows, offsets, lengths = bytecodes if isinstance(bytecodes, tuple) else flatten_unrolled_bytecodes(bytecodes)
    return oneshot.bytecode_assign(in_rows, offsets, lengths, k, _n(keccak_randomness))[1]  # zk_bytecode_assign


def verify_bytecode_rows(rows, keccak_table, keccak_randomness, success=True):
    cols = rows if hasattr(rows, "shape") else flatten_bytecode_rows(rows)
    kt = keccak_table if hasattr(keccak_table, "shape") else flatten_keccak_table(keccak_table)
    res, _ = oneshot.bytecode_verify(cols, kt, _n(keccak_randomness))  # zk_bytecode_verify
    exception = None
    if not res.ok:
        exc = exception_for_code(res.first_fail_code, f"Bytecode circuit row {res.first_fail_row}")
        if res.first_fail_kind != KIND_ASSERT:
            raise exc
        exception = exc
    if success:
        if exception:
            raise exception
    else:
        assert exception is not None
    return res


def check_bytecode_row(cur, next, push_table, keccak_table, keccak_randomness):
    """Single-row form with the reference's signature (`push_table` is implied: opcode.py:432)."""
    cols = flatten_bytecode_rows([cur, next])
    kt = flatten_keccak_table(keccak_table)
    _, status = oneshot.bytecode_verify(cols, kt, _n(keccak_randomness))
    raise_for_code(int(status[0]), "Bytecode circuit row")
