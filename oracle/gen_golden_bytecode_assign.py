#!/usr/bin/env python3
"""Bytecode-circuit witness-assignment golden vectors from the UNMODIFIED reference (build container only):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=oracle/refshim:/root/reference/src:/root/reference/tests \
        python3 oracle/gen_golden_bytecode_assign.py

Every `assign_bytecode_circuit(k, bytecodes, r)` call of the reference's tests/test_bytecode_circuit.py (re-run at k = 7
and k = 9 to stay small, plus the truncating k = 4), and seeded random bytecode sets incl. tampered unrolled rows.
Stored: the unrolled rows / offsets / lengths in wire form, k, r and the flattened rows the reference returned."""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    import test_bytecode_circuit as T
    from zkevm_specs.bytecode_circuit import UnrolledBytecode, assign_bytecode_circuit
    from zkevm_specs.util import FQ
    from zkevm_specs_amd.flatten import flatten_bytecode_rows, flatten_unrolled_bytecodes

    calls = []
    current = [None]

    def capture(k, bytecodes, randomness_keccak, success):
        calls.append((current[0], list(bytecodes), randomness_keccak))

    T.verify = capture
    T.verify_rows = lambda *a, **kw: None
    for name in sorted(dir(T)):
        if name.startswith("test_"):
            current[0] = name
            try:
                getattr(T, name)()
            except Exception:  # noqa: BLE001 - tests that post-process rows themselves
                pass
    rng = random.Random(4242)
    for i in range(12):
        codes = []
        for _ in range(rng.randrange(1, 5)):
            n = rng.choice([0, 1, 5, 33, 70, 200])
            code = bytes(rng.choice([rng.randrange(256), rng.randrange(0x60, 0x80), 0x7F, 0x60]) for _ in range(n))
            u = T.unroll(code, T.randomness_keccak)
            if i >= 6 and u.rows:  # tampered unrolled rows: the assignment trusts them (value drives the push tracking)
                rows = list(u.rows)
                j = rng.randrange(len(rows))
                import dataclasses
                rows[j] = dataclasses.replace(rows[j], value=FQ(rng.choice([0x7F, 0x60, 255, 256, rng.randrange(FQ.field_modulus)])))
                u = UnrolledBytecode(code if rng.random() < 0.5 else code + b"x", rows)
            codes.append(u)
        calls.append((f"random_{i:02d}", codes, FQ(rng.randrange(FQ.field_modulus))))
    out, names = {}, []
    for name, bytecodes, r in calls:
        total = sum(len(b.rows) for b in bytecodes)
        ks = sorted({4, 7, 9} if total < 400 else {9, 10})
        for k in ks:
            key = f"c{len(names):04d}"
            names.append(f"{name}@k{k}")
            rows, offsets, lengths = flatten_unrolled_bytecodes(bytecodes)
            out[key + "_in_rows"], out[key + "_offsets"], out[key + "_lengths"] = rows, offsets, lengths
            out[key + "_k"] = np.uint32(k)
            out[key + "_r"] = np.frombuffer(int(r.n).to_bytes(32, "little"), dtype="<u8").copy()
            out[key + "_rows"] = flatten_bytecode_rows(assign_bytecode_circuit(k, bytecodes, r))
    out["names"] = np.array(names)
    fn = os.path.join(GOLDEN, "bytecode_assign_cases.npz")
    np.savez_compressed(fn, **out)
    print(f"bytecode assign: {len(names)} cases -> {os.path.getsize(fn)//1024} KiB")


if __name__ == "__main__":
    main()
