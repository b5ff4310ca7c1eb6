#!/usr/bin/env python3
"""Keccak-table golden vectors from the UNMODIFIED reference (build container only; run with
PYTHONPATH=oracle/refshim:/root/reference/src).

For a seeded batch of byte strings (every length 0..300, block-boundary lengths, bytecode-sized
inputs) and random 254-bit randomness, records what the reference's own builders produce:
  mode 0: KeccakCircuit.add   (evm_circuit/typing.py:854-865) and assign_keccak_table
          (bytecode_circuit.py:182-186)
  mode 1: KeccakTable.add     (util/tables.py:18-27 and the copy in tx_circuit.py:48-58)
including the ValueError the mode-1 builder raises for inputs longer than 64 bytes.
Output: tests/golden/keccak_table.npz.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle.gen_golden import kind_of_exception  # noqa: E402


def _cell(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype="<u8")


def main():
    from zkevm_specs.bytecode_circuit import assign_keccak_table
    from zkevm_specs.evm_circuit.typing import KeccakCircuit
    from zkevm_specs.tx_circuit import KeccakTable as TxKeccakTable
    from zkevm_specs.util import FQ
    from zkevm_specs.util.tables import KeccakTable as UtilKeccakTable

    rng = random.Random(20240807)
    lengths = list(range(0, 301)) + [135, 136, 137, 271, 272, 273, 407, 408, 409, 1000, 4096, 24576]
    messages = [bytes(rng.getrandbits(8) for _ in range(n)) for n in lengths]
    messages += [b"\x00" * 64, b"\xff" * 64, b"\xff" * 136, b"\x00" * 137, b"abc"]
    rs = [rng.randrange(1, FQ.field_modulus) for _ in range(2)] + [0, 1, FQ.field_modulus - 1]

    out = {"data": np.frombuffer(b"".join(messages), dtype=np.uint8),
           "offsets": np.cumsum([0] + [len(m) for m in messages]).astype(np.uint64),
           "randomness": np.stack([_cell(r) for r in rs])}
    for ri, r in enumerate(rs):
        rows0 = np.zeros((len(messages), 5, 4), dtype=np.uint64)
        rows1 = np.zeros((len(messages), 5, 4), dtype=np.uint64)
        st1 = np.zeros(len(messages), dtype=np.uint32)
        for i, m in enumerate(messages):
            row = KeccakCircuit().add(m, FQ(r)).rows[0]
            cells = [row.state_tag, row.input_rlc, row.input_len, row.output.lo, row.output.hi]
            for k, c in enumerate(cells):
                rows0[i, k] = _cell(c.expr().n)
            # assign_keccak_table builds the same rows (as a set)
            assert set(assign_keccak_table([m], FQ(r))) == {row}
            for cls in (UtilKeccakTable, TxKeccakTable):
                t = cls()
                try:
                    t.add(m, FQ(r))
                    kind = 0
                except Exception as e:  # noqa: BLE001
                    kind = kind_of_exception(e)
                if cls is UtilKeccakTable:
                    st1[i] = kind
                    if kind == 0:
                        (new,) = [x for x in t.table if x[0] == FQ(1)]  # the other row is the all-zeros one
                        en, rlc, ln, word = new
                        for k, c in enumerate([en, rlc, ln, word.lo, word.hi]):
                            rows1[i, k] = _cell(c.expr().n)
                else:
                    assert kind == st1[i]
        out[f"rows0_{ri}"] = rows0
        out[f"rows1_{ri}"] = rows1
        out[f"kind1_{ri}"] = st1
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, "keccak_table.npz"), **out)
    print("wrote keccak_table.npz:", len(messages), "messages x", len(rs), "randomness values")


if __name__ == "__main__":
    main()
