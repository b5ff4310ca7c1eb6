#!/usr/bin/env python3
"""In-process differential fuzz: oracle/evm_oracle.py vs the UNMODIFIED reference (build container).

For every witness the reference's own EVM tests produce, fuzz cells N times and compare the
exception class of every step pair.  Nothing is stored; this is the wide net behind the
committed golden subset.  Usage (same env as gen_golden.py):
    python3 oracle/fuzz_vs_reference.py N [test names...]
"""
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import codes, evm_oracle as eo, wire  # noqa: E402
from oracle.gen_golden_evm import PRECOMPILE_TESTS, REF_TESTS, TEST_FILES, Harvest, fuzz_wire, ref_step_outcomes, unflatten  # noqa: E402


def to_witness(w):
    return eo.EvmWitness(wire.rowmajor_to_rows(w["steps"]), wire.rowmajor_to_rows(w["rw"]), w["rw_flags"],
                         wire.rowmajor_to_rows(w["bytecode"]), wire.rowmajor_to_rows(w["tx"]), w["tx_flags"],
                         wire.rowmajor_to_rows(w["block"]), w["block_flags"], wire.rowmajor_to_rows(w["copy"]),
                         wire.rowmajor_to_rows(w["keccak"]), wire.rowmajor_to_rows(w["exp"]),
                         wire.rowmajor_to_rows(w["aux"]), w["aux_kind"], wire.rowmajor_to_rows(w["withdrawals"]),
                         wire.rowmajor_to_rows(w["sig"]), wire.rowmajor_to_rows(w["ecc"]))


def main():
    from zkevm_specs_amd.flatten import flatten_evm

    n_fuzz = int(sys.argv[1])
    only = sys.argv[2:]
    rng = random.Random(12345)
    bad = tot = fails = 0
    for name in TEST_FILES:
        if only and name not in only:
            continue
        if name == "end_block_padding":
            from oracle.gen_golden_evm import end_block_padding_cases

            all_cases = end_block_padding_cases()
        elif name == "error_oog_precompile_custom":
            from oracle.gen_golden_evm import error_oog_precompile_cases

            all_cases = error_oog_precompile_cases()
        else:
            h = Harvest()
            rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", "--rootdir=/tmp", "-c", "/dev/null",
                              os.path.join(REF_TESTS, "precompiles" if name in PRECOMPILE_TESTS else "", f"test_{name}.py")], plugins=[h])
            assert rc == 0
            all_cases = h.cases
        cases = all_cases if len(all_cases) <= 60 else rng.sample(all_cases, 60)
        nb = 0
        for tid, tables, steps, begin, end, _ in cases:
            wire0 = flatten_evm(tables, steps)
            for _ in range(n_fuzz):
                fw = fuzz_wire(wire0, rng)
                t3, s3 = unflatten(fw)
                ref = ref_step_outcomes(t3, s3, begin, end)
                got = eo.verify_steps(to_witness(fw), begin, end)
                for j, (a, b) in enumerate(zip(got, ref)):
                    tot += 1
                    fails += b != 0
                    if codes.kind_of(a) != codes.UNSUPPORTED and codes.kind_of(a) != b:
                        nb += 1
                        if nb <= 5:
                            print("MISMATCH", name, tid, j, codes.KIND_NAMES[codes.kind_of(a)], codes.site_of(a), "ref",
                                  codes.KIND_NAMES[int(b)], flush=True)
        bad += nb
        print(f"{name}: {nb} mismatches", flush=True)
    print(f"TOTAL {bad} mismatches of {tot} step evaluations ({fails} failing in the reference)")


if __name__ == "__main__":
    main()
