#!/usr/bin/env python3
"""Copy-circuit golden vectors from the UNMODIFIED reference (build container only).

Replays the reference's opcode tests that drive `verify_copy_table` (tests/evm/test_{sha3,codecopy,
calldatacopy,returndatacopy,logs,extcodecopy,return_revert}.py), records the flattened copy circuit
+ RW / bytecode / tx tables and the reference's per-row outcome.  The per-row outcome is obtained by
calling the reference's own `verify_copy_table` on a rotated single-row view of the table, so
exactly the window (i, i+1, i+2) and that row's lookups are evaluated.
"""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle.gen_golden import kind_of_exception  # noqa: E402
from oracle.wire import P, colmajor_to_rows, rowmajor_to_rows  # noqa: E402

FILES = "sha3 codecopy calldatacopy returndatacopy logs extcodecopy return_revert".split()


class _OneRowView:
    """Sequence whose item j is table[(i + j) % n] and whose iteration yields only item 0."""

    def __init__(self, table, i):
        self.t, self.i = table, i

    def __len__(self):
        return len(self.t)

    def __getitem__(self, j):
        return self.t[(self.i + j) % len(self.t)]

    def __iter__(self):
        yield self[0]


class _FakeCircuit:
    def __init__(self, view):
        self.view = view

    def table(self):
        return self.view


def ref_copy_outcomes(table, tables, r):
    from zkevm_specs.copy_circuit import verify_copy_table

    out = []
    for i in range(len(table)):
        try:
            verify_copy_table(_FakeCircuit(_OneRowView(table, i)), tables, r)
            out.append(0)
        except Exception as e:  # noqa: BLE001
            out.append(kind_of_exception(e))
    return out


class Harvest:
    def __init__(self):
        self.cases = []

    def pytest_runtest_setup(self, item):
        real = sys.modules["zkevm_specs.copy_circuit"].verify_copy_table
        harvest = self

        def capture(copy_circuit, tables, r):
            harvest.cases.append((item.nodeid.split("/")[-1], list(copy_circuit.table()), tables, r))
            return real(copy_circuit, tables, r)

        item.module.verify_copy_table = capture


def unflatten_copy(cols, flags):
    from zkevm_specs.evm_circuit import CopyCircuitRow
    from zkevm_specs.util import FQ, Word, WordOrValue

    rows = []
    for c, f in zip(colmajor_to_rows(cols), flags):
        if f & 1:
            idv = WordOrValue(Word((FQ(c[3]), FQ(c[4])), check=False))
        else:
            idv = WordOrValue(FQ(c[3]))
            idv.hi = FQ(c[4])
        rows.append(CopyCircuitRow(FQ(c[0]), FQ(c[1]), FQ(c[2]), idv, *[FQ(v) for v in c[5:]]))
    return rows


def main():
    from oracle.gen_golden_evm import unflatten as unflatten_evm
    from zkevm_specs_amd.flatten import flatten_bytecode_table, flatten_copy_rows, flatten_rw_table, flatten_tx_table

    rng = random.Random(515)
    out, names = {}, []
    for name in FILES:
        h = Harvest()
        rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", "--rootdir=/tmp", "-c", "/dev/null",
                          f"/root/reference/tests/evm/test_{name}.py"], plugins=[h])
        assert rc == 0, name
        cases = [c for c in h.cases if 0 < len(c[1]) <= 200]
        if len(cases) > 10:
            cases = rng.sample(cases, 10)
        for tid, table, tables, r in cases:
            cols, flags = flatten_copy_rows(table)
            rw, rw_flags = flatten_rw_table(tables.rw_table)
            bc = flatten_bytecode_table(tables.bytecode_table)
            tx, tx_flags = flatten_tx_table(tables.tx_table)
            kinds = ref_copy_outcomes(table, tables, r)
            assert not any(kinds), (tid, kinds)
            wire_tables = {"steps": np.zeros((1, 13, 4), dtype=np.uint64), "rw": rw, "rw_flags": rw_flags, "bytecode": bc,
                           "tx": tx, "tx_flags": tx_flags, "block": np.zeros((0, 4, 4), dtype=np.uint64),
                           "block_flags": np.zeros(0, dtype=np.uint32)}
            wire_tables["steps"][0, 0, 0] = 3  # a dummy EndBlock step so the tables can be rebuilt
            t2, _ = unflatten_evm(wire_tables)
            assert ref_copy_outcomes(unflatten_copy(cols, flags), t2, r) == kinds, tid
            variants = [("", cols, flags, rw, kinds)]
            for k in range(4):
                fc, ff, frw = cols.copy(), flags.copy(), rw.copy()
                for _ in range(rng.choice([1, 2, 3])):
                    if rng.random() < 0.75:
                        c, i = rng.randrange(20), rng.randrange(fc.shape[1])
                        old = int.from_bytes(fc[c, i].tobytes(), "little")
                        new = rng.choice([old + 1, old - 1, 0, 1, 2, rng.randrange(P), 1 << 40, old ^ 1]) % P
                        fc[c, i] = np.frombuffer(new.to_bytes(32, "little"), dtype="<u8")
                    elif rng.random() < 0.5 and frw.shape[0]:
                        i, c = rng.randrange(frw.shape[0]), rng.randrange(10)
                        old = int.from_bytes(frw[i, c].tobytes(), "little")
                        frw[i, c] = np.frombuffer(((old + 1) % P).to_bytes(32, "little"), dtype="<u8")
                    else:
                        i = rng.randrange(len(ff))
                        ff[i] ^= np.uint32(1)
                wt = dict(wire_tables, rw=frw)
                t3, _ = unflatten_evm(wt)
                variants.append((f"#fuzz{k}", fc, ff, frw, ref_copy_outcomes(unflatten_copy(fc, ff), t3, r)))
            for suffix, c_, f_, rw_, kd in variants:
                key = f"c{len(names):04d}"
                names.append(f"{name}:{tid}{suffix}")
                out[key + "_rows"], out[key + "_flags"] = c_, f_
                out[key + "_rw"], out[key + "_rw_flags"] = rw_, rw_flags
                out[key + "_bytecode"], out[key + "_tx"], out[key + "_tx_flags"] = bc, tx, tx_flags
                out[key + "_r"] = np.frombuffer(int(r.n).to_bytes(32, "little"), dtype="<u8").copy()
                out[key + "_ref_kind"] = np.array(kd, dtype=np.uint8)
    out["names"] = np.array(names)
    fn = os.path.join(GOLDEN, "copy_cases.npz")
    np.savez_compressed(fn, **out)
    nf = sum(int(out[f"c{i:04d}_ref_kind"].any()) for i in range(len(names)))
    print(f"copy: {len(names)} cases ({nf} with failing rows) -> {os.path.getsize(fn)//1024} KiB")


if __name__ == "__main__":
    main()
