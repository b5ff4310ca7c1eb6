#!/usr/bin/env python3
"""Wide differential fuzz of the HIP EVM path against the oracle (GPU box): every golden case, N fuzzed variants each,
biased towards the step cells (the LDS-staged pair and the 64-bit transition tail are GPU-only code).
usage: python tests/gpu_fuzz_evm.py [N=20] [seed=1]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests.evm_cases import fuzz_wire, golden_files, load_cases, oracle_status
from zkevm_specs_amd import engine

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def put(arr, idx, val):
    arr[idx] = np.frombuffer(int(val % P).to_bytes(32, "little"), dtype="<u8")


def cur(arr, idx):
    return int.from_bytes(arr[idx].tobytes(), "little")


def step_fuzz(w):
    w = {k: v.copy() for k, v in w.items()}
    for _ in range(rng.choice([1, 1, 2])):
        i, c = rng.randrange(w["steps"].shape[0]), rng.randrange(1, 13)
        old = cur(w["steps"], (i, c))
        put(w["steps"], (i, c), rng.choice([old + 1, old - 1, 0, 1, 2**64 - 1, 2**64, 2**64 + old, 2**128 - 1, 2**128, P - 1, P - old if old else 0,
                                            old ^ (1 << rng.randrange(70)), rng.randrange(2**64), rng.randrange(P)]))
    return w


tot = fail = bad = 0
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for fn in golden_files(gd):
    cases = [c for c in load_cases(fn) if "#fuzz" not in c[0]][:10]
    for name, w, opts, _ in cases:
        for k in range(n_fuzz):
            fw = step_fuzz(w) if k % 2 == 0 else fuzz_wire(w, rng)
            exp = oracle_status(fw, opts)
            with engine.open_evm(fw, bool(opts[0]), bool(opts[1])) as s:
                s.run()
                got = s.read_status().tolist()
            tot += len(exp)
            fail += sum(1 for e in exp if e)
            if got != exp:
                bad += 1
                if bad <= 10:
                    print("MISMATCH", os.path.basename(fn), name, got, exp, flush=True)
print(f"fuzzed step pairs: {tot}, failing in the oracle: {fail}, mismatching cases: {bad}")
