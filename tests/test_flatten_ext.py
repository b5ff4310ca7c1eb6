"""zkevm_specs_amd/_flatten_ext (csrc/flatten_ext.c) against the Python loops of zkevm_specs_amd/flatten.py, which define the wire: the same
arrays, bit for bit, for every table and for the steps — on the package's mirror objects rebuilt from synthetic wires (duplicates, shuffled
rows, non-Word values, plain ints and negative counters included) and, where the reference can be imported (the build container),
on the reference's own objects."""
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import pytest

from zkevm_specs_amd import flatten, objects
from zkevm_specs_amd.synth import synth_state_witness
from zkevm_specs_amd.synth_evm import synth_evm_trace

pytestmark = pytest.mark.skipif(flatten._ext is None, reason="zkevm_specs_amd/_flatten_ext.so is not built (csrc/build.sh)")


def both(fn, *args):
    """fn through the extension and through the loops -> the extension's result, after asserting the two are identical"""
    was = flatten.USE_EXT
    try:
        flatten.USE_EXT = True
        a = fn(*args)
        flatten.USE_EXT = False
        b = fn(*args)
    finally:
        flatten.USE_EXT = was
    for x, y in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
        if isinstance(x, dict):
            assert x.keys() == y.keys()
            for k in x:
                assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])) and np.asarray(x[k]).dtype == np.asarray(y[k]).dtype, k
        else:
            assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y)
    return a


def test_evm_wire_of_mirror_objects():
    w = dict(synth_evm_trace(1 << 10, seed=9))
    tables, steps = objects.evm_from_wire(w)
    rng = random.Random(4)
    # sets in the reference: order must not matter, duplicates collapse (first occurrence's type bits)
    rw = list(tables.rw_table)
    rw += [rw[rng.randrange(len(rw))] for _ in range(50)]
    rng.shuffle(rw)
    tables.rw_table = rw
    bc = list(tables.bytecode_table)
    rng.shuffle(bc)
    tables.bytecode_table = bc + bc[:100]
    out = both(flatten.flatten_evm, tables, steps)
    # (the generator's RW table is already in the wire's order — ascending cells —, its bytecode table is grouped by contract)
    assert np.array_equal(out["steps"], w["steps"]) and np.array_equal(out["rw"], w["rw"]) and np.array_equal(out["rw_flags"], w["rw_flags"])
    assert out["bytecode"].shape == w["bytecode"].shape


def test_odd_values():
    FQ, Word, WV = objects.FQ, objects.Word, objects.WordOrValue
    rows = [
        SimpleNamespace(rw_counter=-1, rw=True, key0=FQ(3), id=7, address=FQ(1 << 159), field_tag=0, storage_key=Word(5, (1 << 128) - 1),
                        value=WV(9, 0, 0), value_prev=FQ(4), aux0=Word(0, 0)),                      # plain ints, a negative counter, a bare FQ value
        SimpleNamespace(rw_counter=FQ(2), rw=FQ(0), key0=FQ(3), id=FQ(7), address=FQ(0), field_tag=FQ(0), storage_key=Word(0, 0),
                        value=WV(9, 1, 1), value_prev=WV(1, 2, 1), aux0=Word(3, 4)),
    ]
    rw, flags = both(flatten.flatten_rw_table, rows + rows[:1])
    assert rw.shape == (2, 14, 4) and sorted(flags.tolist()) == [0, 3]
    # the same cells with other type bits are ONE row: the first occurrence's bits stay
    twin = SimpleNamespace(**{**rows[1].__dict__, "value": WV(9, 1, 0)})
    _, f2 = both(flatten.flatten_rw_table, [twin, rows[1]])
    assert f2.tolist() == [2]
    for n in (0, 1):
        both(flatten.flatten_rw_table, rows[:n])
        both(flatten.flatten_steps, [])
    with pytest.raises(OverflowError):
        flatten.flatten_withdrawal_table([SimpleNamespace(id=FQ(1), validator_id=SimpleNamespace(n=1 << 256), address=FQ(0), amount=FQ(0))])
    with pytest.raises(AttributeError):
        flatten.flatten_withdrawal_table([SimpleNamespace(id=FQ(1))])


def test_state_rows_and_small_tables():
    cols, flags, mpt = synth_state_witness(700, seed=3)
    rows = objects.state_rows_from_wire(cols, flags)
    c2, f2 = both(flatten.flatten_state_rows, rows)
    assert np.array_equal(c2, cols) and np.array_equal(f2, flags)
    kt = objects.keccak_table_from_wire(np.random.default_rng(1).integers(0, 1 << 60, size=(40, 5, 4), dtype=np.uint64) & np.uint64(0xFFFFFFF))
    both(flatten.flatten_keccak_table, kt + kt[:5])


def test_dedup_rows_order_is_the_integer_order():
    rng = random.Random(8)
    ints = [[rng.choice([0, 1, 5, 1 << 64, (1 << 64) + 1, 1 << 200, (1 << 256) - 1]) for _ in range(3)] for _ in range(400)]
    want = sorted(set(tuple(r) for r in ints))
    from zkevm_specs_amd.wire import cells_to_ints, rows_to_rowmajor

    got, _ = flatten._dedup_rows(rows_to_rowmajor(ints, 3))
    flat = cells_to_ints(got)
    assert [tuple(flat[i : i + 3]) for i in range(0, len(flat), 3)] == want


REF = "/root/reference/src"
SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "refshim")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference lives in the build container only")
def test_reference_objects():
    """the reference's own FQ / Word / RLC / IntEnum cells (child process: PYTHONPATH = oracle/refshim + the reference): real
    RWTableRow / BytecodeTableRow / BlockTableRow / TxTableRow / StepState objects through both paths"""
    import subprocess

    script = r'''
import sys
import numpy as np
from zkevm_specs.evm_circuit import Bytecode, RWDictionary, StepState, Tables, Block, Transaction, ExecutionState, CallContextFieldTag
from zkevm_specs.util import Word, FQ, U64
from zkevm_specs_amd import flatten
bytecode = Bytecode().push(0x1234, n_bytes=2).push(7, n_bytes=1).add().stop()
tx = Transaction(id=1, gas=U64(21000), call_data=bytes([1, 2, 3]))
rw = (RWDictionary(1).call_context_read(1, CallContextFieldTag.TxId, 1).stack_read(1, 1022, Word(7)).stack_read(1, 1023, Word(0x1234))
      .stack_write(1, 1023, Word(0x123B)).tx_refund_read(1, 5))
tables = Tables(block_table=set(Block().table_assignments()), tx_table=set(tx.table_assignments()), withdrawal_table=set(),
                bytecode_table=set(bytecode.table_assignments()), rw_table=set(rw.rws))
steps = [StepState(execution_state=ExecutionState.ADD, rw_counter=2, call_id=1, is_root=True, is_create=False, code_hash=Word(bytecode.hash()),
                   program_counter=5, stack_pointer=1022, gas_left=3),
         StepState(execution_state=ExecutionState.STOP, rw_counter=5, call_id=1, is_root=True, is_create=False, code_hash=Word(bytecode.hash()),
                   program_counter=6, stack_pointer=1023, gas_left=0)]
assert flatten._ext is not None
flatten.USE_EXT = True
a = flatten.flatten_evm(tables, steps)
flatten.USE_EXT = False
b = flatten.flatten_evm(tables, steps)
assert a.keys() == b.keys()
for k in a:
    assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])) and np.asarray(a[k]).dtype == np.asarray(b[k]).dtype, k
assert a["rw"].shape[0] == len(rw.rws) and a["steps"].shape == (2, 13, 4) and a["tx"].shape[0] > 3 and a["block"].shape[0] > 3
print("ok", {k: tuple(np.asarray(v).shape) for k, v in a.items()})
'''
    root = os.path.dirname(SHIM.rstrip("/").rsplit("/oracle", 1)[0] + "/x")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIM, REF, root]))
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]
