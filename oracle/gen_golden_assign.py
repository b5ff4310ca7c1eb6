#!/usr/bin/env python3
"""State-circuit witness-assignment golden vectors from the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=oracle/refshim:/root/reference/src:/root/reference/tests \
        python3 oracle/gen_golden_assign.py

Cases: every list of Operations the reference's tests/test_state_circuit.py hands to its driver,
plus seeded random op lists (sorted and unsorted, repeated MPT keys, enum and plain-int field
tags, over-wide addresses, value cells whose Word() sanity assert fails).  Stored per case: the
ops in wire form and what `assign_state_circuit` / `mpt_table_from_ops` return (flattened rows,
sorted MPT rows) or the exception class they raise.
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


def harvest():
    import test_state_circuit as T
    from zkevm_specs.state_circuit import Operation

    cases = []
    current = [None]

    def capture(ops_or_rows, tables, success=True):
        if isinstance(ops_or_rows[0], Operation):
            cases.append((current[0], list(ops_or_rows)))

    T.verify = capture
    for name in sorted(dir(T)):
        if name.startswith("test_"):
            current[0] = name
            try:
                getattr(T, name)()
            except Exception:  # noqa: BLE001 - tests that assign rows themselves still ran their capture
                pass
    return cases


def random_ops(rng, n, hostile):
    from zkevm_specs.evm_circuit.table import RW, AccountFieldTag, CallContextFieldTag, TxLogFieldTag, TxReceiptFieldTag
    from zkevm_specs.state_circuit import (AccountOp, CallContextOp, MemoryOp, Operation, StackOp, StartOp, StorageOp,
                                           Tag, TxAccessListAccountOp, TxAccessListAccountStorageOp, TxLogOp,
                                           TxReceiptOp, TxRefundOp)
    from zkevm_specs.util import FQ, U256, Word, WordOrValue

    addrs = [rng.randrange(1 << 160) for _ in range(4)] + [0, 1, (1 << 160) - 1]
    keys = [rng.randrange(1 << 256) for _ in range(3)] + [0, 1, (1 << 256) - 1, 1 << 128]
    ops = [StartOp(0, RW.Read, lexicographic_ordering_selector=0)]
    for k in range(1, n):
        rw = rng.choice([RW.Read, RW.Write])
        c = rng.randrange(13)
        w = Word(rng.randrange(1 << 256))
        if c == 0:
            op = MemoryOp(k, rw, rng.randrange(1, 5), rng.randrange(1 << 32), rng.randrange(256))
        elif c == 1:
            op = StackOp(k, rw, rng.randrange(1, 5), rng.randrange(1024), w)
        elif c in (2, 3, 4):
            op = StorageOp(k, rw, rng.randrange(1, 3), rng.choice(addrs), rng.choice(keys), w,
                           Word(rng.randrange(1 << 256)))
        elif c == 5:
            op = CallContextOp(k, rw, rng.randrange(1, 5), rng.choice(list(CallContextFieldTag)), FQ(rng.randrange(1 << 64)))
        elif c in (6, 7, 8):
            ft = rng.choice(list(AccountFieldTag))
            v = FQ(rng.randrange(1 << 64)) if ft == AccountFieldTag.Nonce else w
            op = AccountOp(k, rw, rng.choice(addrs), ft, v, FQ(rng.randrange(1 << 64)) if ft == AccountFieldTag.Nonce else Word(rng.randrange(1 << 256)))
        elif c == 9:
            op = TxRefundOp(k, rw, rng.randrange(1, 5), w)
        elif c == 10:
            op = rng.choice([TxAccessListAccountOp(k, rw, 1, rng.choice(addrs), FQ(rng.randrange(2))),
                             TxAccessListAccountStorageOp(k, rw, 1, rng.choice(addrs), rng.choice(keys), FQ(rng.randrange(2)))])
        elif c == 11:
            op = TxLogOp(k, RW.Write, 1, rng.randrange(4), rng.choice(list(TxLogFieldTag)), rng.randrange(4), w)
        else:
            op = TxReceiptOp(k, rw, rng.randrange(1, 5), rng.choice(list(TxReceiptFieldTag)), FQ(rng.randrange(1 << 32)))
        if hostile and rng.random() < 0.2:
            m = rng.randrange(7)
            if m == 0:  # address wider than 160 bits: op2row's to_bytes(20) overflows
                op = op._replace(address=U256(rng.choice([1 << 160, rng.randrange(1 << 256), FQ.field_modulus + 5])))
            elif m == 1:  # malformed word cells: Word(int_value()) asserts inside _mock_mpt_updates
                bad = WordOrValue(Word((FQ(rng.randrange(FQ.field_modulus)), FQ(rng.randrange(1 << 128, FQ.field_modulus))), check=False))
                op = op._replace(value=bad) if rng.random() < 0.5 else op._replace(initial_value=bad)
            elif m == 2:  # Account tag with a plain-int field tag: isinstance() is False -> StorageMod
                op = Operation(k, rw, U256(Tag.Account), U256(0), U256(rng.choice(addrs)), U256(rng.randrange(6)), U256(0),
                               WordOrValue(w), WordOrValue(Word(7)), FQ(1))
            elif m == 3:  # Storage tag carrying an AccountFieldTag enum
                op = Operation(k, rw, U256(Tag.Storage), U256(1), U256(rng.choice(addrs)), rng.choice(list(AccountFieldTag)),
                               U256(rng.choice(keys)), WordOrValue(w), WordOrValue(w), FQ(1))
            elif m == 4:  # unreduced tag / id / rw_counter / field_tag
                op = op._replace(id=U256(op.id + FQ.field_modulus), rw_counter=op.rw_counter + FQ.field_modulus)
            elif m == 5:  # rw outside the enum counts as a write; odd selector
                op = op._replace(rw=rng.randrange(2, 9), lexicographic_ordering_selector=FQ(rng.randrange(FQ.field_modulus)))
            else:  # tag == Account + p is not an MPT-keyed op (raw comparison), but its cell is Account
                op = op._replace(tag=U256(int(Tag.Account) + FQ.field_modulus))
        ops.append(op)
    return ops


def main():
    from zkevm_specs.state_circuit import Tables, assign_state_circuit, mpt_table_from_ops
    from zkevm_specs_amd.flatten import flatten_mpt_table, flatten_state_ops, flatten_state_rows

    cases = harvest()
    rng = random.Random(20260926)
    for k in range(60):
        n = rng.choice([2, 3, 8, 40, 150])
        ops = random_ops(rng, n, hostile=k >= 20)
        if k % 3 == 0:  # sorted like a real trace: tag, id, address, field_tag, storage_key, rw_counter
            ops = [ops[0]] + sorted(ops[1:], key=lambda o: (o.tag, o.id, o.address, o.field_tag, o.storage_key, o.rw_counter))
        cases.append((f"random_{k:02d}", ops))
    out, names = {}, []
    n_err = 0
    for idx, (name, ops) in enumerate(cases):
        key = f"c{idx:03d}"
        names.append(name)
        wire_ops, wire_flags = flatten_state_ops(ops)
        out[key + "_ops"], out[key + "_opflags"] = wire_ops, wire_flags
        try:
            rows = assign_state_circuit(ops)
            out[key + "_rows"], out[key + "_rowflags"] = flatten_state_rows(rows)
            out[key + "_kind"] = np.uint8(0)
        except Exception as e:  # noqa: BLE001
            out[key + "_kind"] = np.uint8(kind_of_exception(e))
            n_err += 1
        try:
            out[key + "_mpt"] = flatten_mpt_table(Tables(mpt_table_from_ops(ops)).mpt_table)
            out[key + "_mpt_kind"] = np.uint8(0)
        except Exception as e:  # noqa: BLE001
            out[key + "_mpt_kind"] = np.uint8(kind_of_exception(e))
    out["names"] = np.array(names)
    path = os.path.join(GOLDEN, "assign_cases.npz")
    np.savez_compressed(path, **out)
    print(f"assign: {len(names)} cases ({n_err} raising) -> {path}")


if __name__ == "__main__":
    main()
