#!/usr/bin/env python3
"""Wide differential fuzz of the HIP State path against the oracle (GPU box): random sizes around the wavefront tiling, random cell
damage in the rows and the MPT table (weighted towards Storage / Account rows, whose lookup path is GPU-only code), per-row statuses
compared.  usage: python tests/gpu_fuzz_state.py [N=200] [seed=1]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import state_oracle, wire
from zkevm_specs_amd import engine
from zkevm_specs_amd.synth import synth_state_witness

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
P = wire.P
bad = 0
rows_total = 0
fails_total = 0
for case in range(n_cases):
    n = rng.choice([64, 65, 100, 126, 127, 128, 189, 190, 500, 1000, 2048, 4097, 8191])
    cols, flags, mpt = synth_state_witness(n, seed=rng.randrange(1 << 30))
    tags = cols[2, :, 0]
    sa = np.nonzero((tags == 4) | (tags == 6))[0]
    for _ in range(rng.choice([0, 1, 2, 4, 8])):
        kind = rng.random()
        if kind < 0.35 and mpt.shape[0]:
            r, c = rng.randrange(mpt.shape[0]), rng.randrange(12)
            old = int.from_bytes(mpt[r, c].tobytes(), "little")
            new = rng.choice([old + 1, old ^ (1 << rng.randrange(250)), 0, rng.randrange(P)]) % P
            mpt[r, c] = np.frombuffer(new.to_bytes(32, "little"), dtype="<u8")
        else:
            i = int(rng.choice(sa)) if (kind < 0.7 and len(sa)) else rng.randrange(n)
            c = rng.randrange(57)
            old = int.from_bytes(cols[c, i].tobytes(), "little")
            new = rng.choice([old + 1, old - 1, old ^ (1 << rng.randrange(250)), 0, 1, 255, 256, 65535, 65536, rng.randrange(P)]) % P
            cols[c, i] = np.frombuffer(new.to_bytes(32, "little"), dtype="<u8")
        if rng.random() < 0.1:
            flags[rng.randrange(n)] ^= np.uint32(rng.choice([1, 2]))
    with engine.open_state(cols, flags, mpt) as s:
        res = s.run()
        st = s.read_status().tolist()
    exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
    rows_total += n
    fails_total += sum(1 for c in exp if c)
    if st != exp or res.fail_count != sum(1 for c in exp if c):
        bad += 1
        k = next(j for j in range(n) if st[j] != exp[j]) if st != exp else -1
        print(f"MISMATCH case {case} n={n} row {k}: gpu {st[k]:#x} oracle {exp[k]:#x}", flush=True)
print(f"state fuzz: {n_cases} cases, {rows_total} rows, {fails_total} failing rows, {bad} mismatching cases")
sys.exit(1 if bad else 0)
