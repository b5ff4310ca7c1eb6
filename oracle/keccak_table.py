"""Keccak table rows — CPU restatement of the reference's two row builders.  TEST INFRASTRUCTURE
(like everything under oracle/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.

Follows
  KeccakCircuit.add   src/zkevm_specs/evm_circuit/typing.py:854-865   (mode 0)
  KeccakTable.add     src/zkevm_specs/util/tables.py:18-27 (== tx_circuit.py:48-58)   (mode 1)
  RLC / linear_combine_bytes   src/zkevm_specs/util/arithmetic.py:9-24,69-96
  Word(int) / Word(bytes)      src/zkevm_specs/util/arithmetic.py:99-123
The digest is third-party in the reference (pycryptodome keccak / eth_utils.keccak): oracle/keccak.py.
Pinned by tests/golden/keccak_table.npz (generated from the unmodified reference by
oracle/gen_golden_keccak.py) and by the sponge cross-check against hashlib.sha3_256 in
tests/test_keccak_table.py.
"""
import numpy as np

from .codes import VALUE_ERROR
from .keccak import keccak256

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MODE_CIRCUIT, MODE_TABLE = 0, 1


def _cell(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype="<u8")


def input_rlc(data: bytes, r: int) -> int:
    """RLC(bytes(reversed(data)), r).expr(): sum data[len-1-i] * r^i  ==  Horner front to back."""
    acc = 0
    for b in data:
        acc = (acc * r + b) % P
    return acc


def row(data: bytes, r: int, mode: int):
    """-> (status code, [5 cell ints])"""
    if mode == MODE_TABLE and len(data) > 64:  # RLC(..., n_bytes=64) raises ValueError (arithmetic.py:82-83)
        return (VALUE_ERROR << 24) | 1, [0, 0, 0, 0, 0]
    digest = keccak256(data)
    if mode == MODE_TABLE:  # Word(bytes): lo = bytes[0:16], hi = bytes[16:32], little-endian
        lo, hi = int.from_bytes(digest[:16], "little"), int.from_bytes(digest[16:], "little")
    else:  # Word(int.from_bytes(digest, "big"))
        v = int.from_bytes(digest, "big")
        lo, hi = v & ((1 << 128) - 1), v >> 128
    return 0, [1 if mode == MODE_TABLE else 2, input_rlc(data, r), len(data), lo, hi]


def table_rows(messages, r: int, mode: int):
    """-> (rows uint64[n, 5, 4], status uint32[n])"""
    rows = np.zeros((len(messages), 5, 4), dtype=np.uint64)
    status = np.zeros(len(messages), dtype=np.uint32)
    for i, m in enumerate(messages):
        status[i], cells = row(bytes(m), r, mode)
        for k, c in enumerate(cells):
            rows[i, k] = _cell(c)
    return rows, status
