#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference source.

Runs only in the build container (needs /root/reference; the GPU box has no reference):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=oracle/refshim:/root/reference/src:/root/reference/tests \
        python3 oracle/gen_golden.py [state|evm|bytecode|exp|copy|all]

For every case it stores the flattened wire inputs plus the reference's own outcome per row
(exception class of `check_*`/`verify_step` evaluated on that row alone, 0 = pass), so the
oracle and the HIP path can both be compared with what the reference actually does.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle.codes import NAME_TO_KIND  # noqa: E402


def kind_of_exception(e):
    name = type(e).__name__
    if name in NAME_TO_KIND:
        return NAME_TO_KIND[name]
    for base in type(e).__mro__:
        if base.__name__ in NAME_TO_KIND:
            return NAME_TO_KIND[base.__name__]
    raise RuntimeError(f"unmapped exception class {name}: {e!r}")


# --------------------------------------------------------------------------------------
# State circuit
# --------------------------------------------------------------------------------------
def ref_state_outcomes(rows, tables):
    from zkevm_specs.state_circuit import check_state_row

    out = []
    n = len(rows)
    for idx, row in enumerate(rows):
        try:
            check_state_row(row, rows[(idx - 1) % n], rows[(idx + 1) % n], tables)
            out.append(0)
        except Exception as e:  # noqa: BLE001 - we record the class
            out.append(kind_of_exception(e))
    return out


def harvest_state_tests():
    """Replay every test of the reference's tests/test_state_circuit.py, capturing the rows it
    hands to its `verify` driver (:17-38)."""
    import test_state_circuit as T
    from zkevm_specs.state_circuit import Operation, assign_state_circuit

    cases = []

    def capture(ops_or_rows, tables, success=True):
        rows = ops_or_rows
        if isinstance(ops_or_rows[0], Operation):
            rows = assign_state_circuit(ops_or_rows)
        cases.append((current[0], list(rows), tables, success))

    current = [None]
    T.verify = capture
    for name in sorted(dir(T)):
        if name.startswith("test_"):
            current[0] = name
            getattr(T, name)()
    return cases


def mutate_state_rows(rows, rng, n_mut):
    """Tampered variants: overwrite random cells with boundary / random field values."""
    from zkevm_specs.util import FQ, Word, WordOrValue

    P = FQ.field_modulus
    rows = list(rows)
    n = len(rows)

    def rnd_val(old):
        c = rng.randrange(12)
        return [0, 1, 2, 255, 256, 65535, 65536, 2**32, (old + 1) % P, (old - 1) % P,
                rng.randrange(P), P - 1][c]

    for _ in range(n_mut):
        i = rng.randrange(n)
        r = rows[i]
        f = rng.randrange(12)
        if f == 0:
            r = r._replace(rw_counter=FQ(rnd_val(r.rw_counter.n)))
        elif f == 1:
            r = r._replace(is_write=FQ(rnd_val(r.is_write.n)))
        elif f == 2:
            k = rng.randrange(4)
            keys = list(r.keys)
            keys[k] = FQ(rnd_val(keys[k].n) if rng.random() < 0.7 else rng.randrange(1, 13))
            r = r._replace(keys=tuple(keys))
        elif f == 3:
            keys = list(r.keys)
            keys[4] = Word((FQ(rnd_val(keys[4].lo.n)), FQ(rnd_val(keys[4].hi.n))), check=False)
            r = r._replace(keys=tuple(keys))
        elif f == 4:
            limbs = list(r.key2_limbs)
            k = rng.randrange(10)
            limbs[k] = FQ(rnd_val(limbs[k].n))
            r = r._replace(key2_limbs=tuple(limbs))
        elif f == 5:
            bs = list(r.key45_bytes)
            k = rng.randrange(32)
            bs[k] = FQ(rnd_val(bs[k].n))
            r = r._replace(key45_bytes=tuple(bs))
        elif f in (6, 7):
            fld = "value" if f == 6 else "initial_value"
            old = getattr(r, fld)
            lo, hi = old.lo.expr().n, old.hi.expr().n
            mode = rng.randrange(4)
            if mode == 0:  # flip type bit, keep cells
                new = WordOrValue(FQ(lo)) if old.is_word else WordOrValue(Word((FQ(lo), FQ(hi)), check=False))
                if not old.is_word:
                    pass
                else:
                    new.hi = FQ(hi)
            elif mode == 1:
                new = WordOrValue(Word((FQ(rnd_val(lo)), FQ(hi)), check=False))
            elif mode == 2:
                new = WordOrValue(Word((FQ(lo), FQ(rnd_val(hi))), check=False))
            else:
                new = WordOrValue(FQ(rnd_val(lo)))
            r = r._replace(**{fld: new})
        elif f == 8:
            r = r._replace(root=Word((FQ(rnd_val(r.root.lo.n)), FQ(r.root.hi.n)), check=False))
        elif f == 9:
            r = r._replace(lexicographic_ordering_selector=FQ(rnd_val(r.lexicographic_ordering_selector.n)))
        elif f == 10:  # duplicate the previous row's keys (breaks ordering / makes keys equal)
            r = r._replace(keys=rows[(i - 1) % n].keys, key2_limbs=rows[(i - 1) % n].key2_limbs,
                           key45_bytes=rows[(i - 1) % n].key45_bytes)
        else:  # swap with neighbour
            j = (i + 1) % n
            rows[i], rows[j] = rows[j], rows[i]
            continue
        rows[i] = r
    return rows


def gen_state():
    from zkevm_specs.state_circuit import Tables
    from zkevm_specs_amd.flatten import flatten_mpt_table, flatten_state_rows

    cases = harvest_state_tests()
    out = {}
    names = []
    rng = random.Random(20240807)
    base_ok = [c for c in cases if c[0] == "test_state_ok"][0]
    # fuzz variants of the big positive case
    for k in range(120):
        rows = mutate_state_rows(base_ok[1], rng, rng.choice([1, 1, 1, 2, 3, 6]))
        cases.append((f"fuzz_{k:03d}", rows, base_ok[2], None))
    for idx, (name, rows, tables, success) in enumerate(cases):
        kinds = ref_state_outcomes(rows, tables)
        if success is not None:
            # the reference driver stops at the first AssertionError; cross-check its verdict
            first = next((k for k in kinds if k != 0), 0)
            assert (first == 0) == success, (name, kinds)
        cols, flags = flatten_state_rows(rows)
        key = f"c{idx:03d}"
        names.append(name)
        out[key + "_rows"] = cols
        out[key + "_flags"] = flags
        out[key + "_mpt"] = flatten_mpt_table(tables.mpt_table)
        out[key + "_ref_kind"] = np.array(kinds, dtype=np.uint8)
    out["names"] = np.array(names)
    path = os.path.join(GOLDEN, "state_cases.npz")
    np.savez_compressed(path, **out)
    nfail = sum(int((out[f"c{i:03d}_ref_kind"] != 0).any()) for i in range(len(names)))
    print(f"state: {len(names)} cases ({nfail} with failing rows) -> {path}")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(GOLDEN, exist_ok=True)
    if what in ("state", "all"):
        gen_state()
    if what in ("evm", "all"):
        from oracle import gen_golden_evm

        gen_golden_evm.main()


if __name__ == "__main__":
    main()
