#!/usr/bin/env python3
"""Differential fuzz of the HIP Copy-circuit kernel against the oracle (GPU box).  Event mixes of random size around the kernel's
62-row wavefront period, assigned on the host by the oracle, then random damage to circuit cells, RW cells, bytecode / tx table cells
and type bits; dense-RW-index and generic-index sessions, whole tables and random row ranges (zk_set_range); per-row status words
compared bit for bit.  usage: python tests/gpu_fuzz_copy.py [N=120] [seed=1]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import copy_assign_oracle as CA, copy_oracle as co, wire
from zkevm_specs_amd import engine
from zkevm_specs_amd.synth import synth_copy_events

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
P = wire.P
bad = rows_total = fails_total = 0


def put(arr, idx, val):
    arr[idx] = np.frombuffer(int(val % P).to_bytes(32, "little"), dtype="<u8")


for case in range(n_cases):
    target = rng.choice([2, 20, 58, 60, 62, 64, 120, 124, 126, 250, 500, 1000, 3000, 9000])
    w = synth_copy_events(target, seed=rng.randrange(1 << 30), max_len=rng.choice([8, 33, 70, 192]))
    rows, rf, table, rw, rwf = CA.assign(wire.rowmajor_to_rows(w["events"]), [int(f) for f in w["flags"]], w["data"], w["offsets"], w["r"])
    n = len(rows)
    h_rows = wire.rows_to_colmajor(rows)
    h_rf = np.array(rf, dtype=np.uint32)
    h_rw = wire.rows_to_rowmajor(rw, 14) if rw else np.zeros((0, 14, 4), dtype=np.uint64)
    h_rwf = np.array(rwf, dtype=np.uint32)
    bc, tx, txf = w["bytecode"].copy(), w["tx"].copy(), w["tx_flags"].copy()
    for _ in range(rng.choice([0, 1, 2, 4, 8, 16])):
        what = rng.random()
        if what < 0.6:
            c, i = rng.randrange(20), rng.choice([rng.randrange(n), n - 1, 0, min(n - 1, 61), min(n - 1, 62), max(0, n - 2)])
            old = int.from_bytes(h_rows[c, i].tobytes(), "little")
            put(h_rows, (c, i), rng.choice([old + 1, old - 1, 0, 1, 2, old ^ 1, rng.randrange(P), 1 << 40, 1 << 130, 5, 256]))
        elif what < 0.75 and h_rw.shape[0]:
            i, c = rng.randrange(h_rw.shape[0]), rng.randrange(10)
            old = int.from_bytes(h_rw[i, c].tobytes(), "little")
            put(h_rw, (i, c), rng.choice([old + 1, 0, rng.randrange(P)]))
        elif what < 0.85 and bc.shape[0]:
            i, c = rng.randrange(bc.shape[0]), rng.randrange(6)
            old = int.from_bytes(bc[i, c].tobytes(), "little")
            put(bc, (i, c), rng.choice([old + 1, 0, old ^ 1]))
        elif what < 0.92 and tx.shape[0]:
            i, c = rng.randrange(tx.shape[0]), rng.randrange(5)
            old = int.from_bytes(tx[i, c].tobytes(), "little")
            put(tx, (i, c), rng.choice([old + 1, 0]))
        elif what < 0.96:
            h_rf[rng.randrange(n)] ^= np.uint32(1)
        elif h_rwf.shape[0]:
            h_rwf[rng.randrange(h_rwf.shape[0])] ^= np.uint32(1)
    T = co.CopyTables(wire.rowmajor_to_rows(h_rw), h_rwf.tolist(), wire.rowmajor_to_rows(bc), wire.rowmajor_to_rows(tx), txf)
    exp = co.verify_rows(wire.colmajor_to_rows(h_rows), h_rf.tolist(), T, w["r"])
    generic = rng.random() < 0.3
    with engine.open_copy(h_rows, h_rf, w["r"], h_rw, h_rwf, bc, tx, txf, generic_index=generic) as s:
        res = s.run()
        st = s.read_status().tolist()
        ok = st == exp and res.fail_count == sum(1 for e in exp if e)
        if n > 4:
            lo = rng.randrange(n - 1)
            hi = rng.randrange(lo + 1, n + 1)
            s.set_range(lo, hi)
            r2 = s.run()
            ok = ok and s.read_status()[lo:hi].tolist() == exp[lo:hi] and r2.fail_count == sum(1 for e in exp[lo:hi] if e)
    rows_total += n
    fails_total += sum(1 for e in exp if e)
    if not ok:
        bad += 1
        diff = [(i, hex(a), hex(b)) for i, (a, b) in enumerate(zip(st, exp)) if a != b][:5]
        print("MISMATCH case", case, "n", n, "generic", generic, diff, flush=True)
print(f"copy fuzz: {n_cases} cases, {rows_total} rows, {fails_total} failing rows, {bad} mismatching cases")
sys.exit(1 if bad else 0)
