#!/usr/bin/env python3
"""EVM-circuit golden vectors from the UNMODIFIED reference (build container only).

Replays the reference's own opcode tests (tests/evm/test_*.py) under pytest with
`verify_steps` intercepted: for every call we record the flattened witness and, per step pair,
the exception class the reference's `verify_step` raises on that pair alone (0 = pass).
A second pass fuzzes the recorded witnesses at the cell level, rebuilds reference objects from
the wire cells and records the reference's outcome again (tampered-witness parity).
"""
import os
import random
import sys

os.environ.setdefault("ZKEVM_SHIM_SEED", "20240807")  # oracle/refshim/Crypto/Random: the reference's tests draw unseeded operands

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_TESTS = "/root/reference/tests/evm"

from oracle.gen_golden import kind_of_exception  # noqa: E402

TEST_FILES = """add_sub mul_div_mod comparator slt_sgt iszero not bitwise byte signextend push pop shl_shr addmod
mulmod memory caller callvalue address calldatasize returndatasize origin gasprice selfbalance block_ctx gas
msize codesize jump jumpi sload sstore stop sar sdiv_smod balance extcodesize extcodehash blockhash
calldataload error_invalid_opcode error_stack error_oog_constant error_invalid_jump sha3 codecopy calldatacopy
returndatacopy extcodecopy exp error_oog_static_memory_expansion error_oog_dynamic_memory_expansion
error_oog_memory_copy error_oog_account_access error_oog_log error_oog_exp error_oog_sha3
error_return_data_out_of_bound error_write_protection logs return_revert
error_invalild_creation_code error_code_store end_block_padding end_tx begin_tx callop error_oog_call error_oog_sload_store create end_block dataCopy error_oog_precompile_custom error_oog_create error_gas_uint_overflow ecRecover ecAdd ecMul ecPairing wide_cells""".split()
PRECOMPILE_TESTS = ("ecRecover", "ecAdd", "ecMul", "ecPairing")  # tests/evm/precompiles/
MAX_CASES_PER_FILE = 48  # (wide_cells: 80 harvested cases, each with six wide-cell variants)


def ref_step_outcomes(tables, steps, begin, end):
    from zkevm_specs.evm_circuit.instruction import Instruction
    from zkevm_specs.evm_circuit.main import verify_step

    out = []
    n = len(steps)
    for idx in range(n - 1):
        try:
            verify_step(Instruction(tables=tables, curr=steps[idx], next=steps[idx + 1],
                                    is_first_step=begin and idx == 0, is_last_step=end and idx == n - 2))
            out.append(0)
        except Exception as e:  # noqa: BLE001
            out.append(kind_of_exception(e))
    return out


def ref_driver_outcomes(tables, steps, begin, end):
    """What the reference's own driver does with the witness (evm_circuit/main.py:14-44): the exception class
    `verify_steps(..., success=True)` and `verify_steps(..., success=False)` raise (0 = returned normally).
    `steps` already carries the dummy EndBlock step when `end`; the driver appends its own, so it is cut off here."""
    from zkevm_specs.evm_circuit.main import verify_steps

    out = []
    for success in (True, False):
        st = list(steps[:-1] if end else steps)
        try:
            verify_steps(tables, st, begin, end, success)
            out.append(0)
        except Exception as e:  # noqa: BLE001
            out.append(kind_of_exception(e))
    return out


def reseed(name):
    """every test file starts from its own seed: a file's goldens do not depend on which other files were harvested"""
    seed = sum(map(ord, name)) * 7919 + 20240807
    random.seed(seed)
    np.random.seed(seed % (1 << 32))
    import Crypto.Random

    Crypto.Random._rng.seed(seed)


class Harvest:
    def __init__(self):
        self.cases = []
        self.current = None

    def pytest_runtest_setup(self, item):
        self.current = item.nodeid.split("/")[-1]
        real = sys.modules["zkevm_specs.evm_circuit.main"].verify_steps
        harvest = self

        def capture(tables, steps, begin_with_first_step=False, end_with_last_step=False, success=True):
            from zkevm_specs.evm_circuit.main import DUMMY_STEP_STATE

            steps_l = list(steps) + ([DUMMY_STEP_STATE] if end_with_last_step else [])
            harvest.cases.append((harvest.current, tables, steps_l, begin_with_first_step, end_with_last_step, success))
            return real(tables, steps, begin_with_first_step, end_with_last_step, success)

        item.module.verify_steps = capture


def unflatten(wire):
    """wire dict -> reference (Tables, steps): used to ask the reference about fuzzed cells."""
    from zkevm_specs.evm_circuit import (BlockTableRow, BytecodeTableRow, ExecutionState, RWTableRow, StepState,
                                         Tables, TxTableRow)
    from zkevm_specs.evm_circuit.table import CopyTableRow, ExpTableRow, KeccakTableRow
    from zkevm_specs.util import FQ, Word, WordOrValue

    from oracle.wire import colmajor_to_rows, rowmajor_to_rows

    def wov(lo, hi, is_word):
        if is_word:
            return WordOrValue(Word((FQ(lo), FQ(hi)), check=False))
        v = WordOrValue(FQ(lo))
        v.hi = FQ(hi)
        return v

    W = lambda lo, hi: Word((FQ(lo), FQ(hi)), check=False)  # noqa: E731
    steps = []
    for c in rowmajor_to_rows(wire["steps"]):
        s = StepState(ExecutionState(c[0]), 0, code_hash=W(c[5], c[6]))
        s.rw_counter, s.call_id = FQ(c[1]), FQ(c[2])
        s.is_root, s.is_create = bool(c[3]), bool(c[4])
        s.program_counter, s.stack_pointer, s.gas_left = FQ(c[7]), FQ(c[8]), FQ(c[9])
        s.memory_word_size, s.reversible_write_counter, s.log_id = FQ(c[10]), FQ(c[11]), FQ(c[12])
        steps.append(s)
    if "aux" in wire:  # StepState.aux_data (kinds: zkevm_specs_amd/flatten.py flatten_step_aux)
        for s, a, k in zip(steps, rowmajor_to_rows(wire["aux"]), wire["aux_kind"]):
            k = int(k)
            if k == 4:  # a shape the wire does not carry (e.g. test_extcodesize.py's bool): only gadgets that never read it may see it
                s.aux_data = object()
            elif k == 5:
                from zkevm_specs.evm_circuit.execution.precompiles.ecrecover import PrecompileAuxData

                s.aux_data = [PrecompileAuxData(W(a[0], a[1]), W(a[2], a[3]), W(a[4], a[5]), W(a[6], a[7]), FQ(a[8]), FQ(a[9]),
                                                FQ(a[10])), FQ(a[11])]
            elif k == 6:
                s.aux_data = [W(a[0], a[1]), W(a[2], a[3]), W(a[4], a[5]), W(a[6], a[7]), FQ(a[8]), FQ(a[9])]
            elif k == 7:
                s.aux_data = [W(a[0], a[1]), W(a[2], a[3]), W(a[4], a[5]), FQ(a[6]), FQ(a[7])]
            elif k == 8:
                s.aux_data = [FQ(a[0]), FQ(a[1]), FQ(a[2]), FQ(a[3])]
            else:
                s.aux_data = None if k == 0 else (W(a[0], a[1]) if k == 1 else (a[0] | (a[1] << 128) if k == 2 else [a[0], a[1]]))
    rw = set()
    for c, f in zip(rowmajor_to_rows(wire["rw"]), wire["rw_flags"]):
        rw.add(RWTableRow(FQ(c[0]), FQ(c[1]), FQ(c[2]), FQ(c[3]), FQ(c[4]), FQ(c[5]), W(c[6], c[7]),
                          wov(c[8], c[9], f & 1), wov(c[10], c[11], f & 2), W(c[12], c[13])))
    bc = set(BytecodeTableRow(W(c[0], c[1]), FQ(c[2]), FQ(c[3]), FQ(c[4]), FQ(c[5])) for c in rowmajor_to_rows(wire["bytecode"]))
    tx = set(TxTableRow(FQ(c[0]), FQ(c[1]), FQ(c[2]), wov(c[3], c[4], f & 1))
             for c, f in zip(rowmajor_to_rows(wire["tx"]), wire["tx_flags"]))
    blk = set(BlockTableRow(FQ(c[0]), FQ(c[1]), wov(c[2], c[3], f & 1))
              for c, f in zip(rowmajor_to_rows(wire["block"]), wire["block_flags"]))
    from zkevm_specs.evm_circuit.table import WithdrawalTableRow

    wds = set(WithdrawalTableRow(FQ(c[0]), FQ(c[1]), FQ(c[2]), FQ(c[3])) for c in rowmajor_to_rows(wire["withdrawals"])) \
        if "withdrawals" in wire else set()
    tables = Tables(block_table=blk, tx_table=tx, withdrawal_table=wds, bytecode_table=bc, rw_table=rw)
    WV = lambda lo, hi: WordOrValue(W(lo, hi))  # noqa: E731  (ids match on lo/hi only)
    tables.copy_table = set(CopyTableRow(FQ(c[0]), WV(c[1], c[2]), FQ(c[3]), WV(c[4], c[5]), FQ(c[6]), FQ(c[7]), FQ(c[8]),
                                         FQ(c[9]), FQ(c[10]), FQ(c[11]), FQ(c[12]), FQ(c[13]))
                            for c in rowmajor_to_rows(wire["copy"]))
    tables.keccak_table = set(KeccakTableRow(FQ(c[0]), FQ(c[1]), FQ(c[2]), W(c[3], c[4])) for c in rowmajor_to_rows(wire["keccak"]))
    tables.exp_table = set(ExpTableRow(FQ(c[0]), FQ(c[1]), FQ(c[2]), FQ(c[3]), FQ(c[4]), FQ(c[5]), FQ(c[6]), W(c[7], c[8]),
                                       W(c[9], c[10])) for c in rowmajor_to_rows(wire["exp"]))
    from zkevm_specs.evm_circuit.table import EccTableRow, SigTableRow

    if "sig" in wire:
        tables.sig_table = set(SigTableRow(W(c[0], c[1]), FQ(c[2]), W(c[3], c[4]), W(c[5], c[6]), FQ(c[7]), FQ(c[8]))
                               for c in rowmajor_to_rows(wire["sig"]))
    if "ecc" in wire:
        tables.ecc_table = set(EccTableRow(FQ(c[0]), W(c[1], c[2]), W(c[3], c[4]), W(c[5], c[6]), W(c[7], c[8]), FQ(c[9]), FQ(c[10]),
                                           FQ(c[11]), FQ(c[12])) for c in rowmajor_to_rows(wire["ecc"]))
    return tables, steps


def fuzz_wire(wire, rng):
    """Overwrite 1..3 random cells of the steps / rw / bytecode tables (canonical values)."""
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    w = {k: v.copy() for k, v in wire.items()}

    def put(arr, idx, val):
        arr[idx] = np.frombuffer(int(val % P).to_bytes(32, "little"), dtype="<u8")

    def cur(arr, idx):
        return int.from_bytes(arr[idx].tobytes(), "little")

    for _ in range(rng.choice([1, 1, 2, 3])):
        which = rng.choice(["steps", "steps", "rw", "rw", "rw", "bytecode", "flags"])
        aux = [k for k in ("copy", "keccak", "exp", "sig", "ecc") if k in w and w[k].shape[0]]
        wide = "aux" in w and w["aux"].shape[1] > 2
        if wide and rng.random() < 0.3:  # precompile gadgets read most of their inputs from aux_data
            i = rng.choice([j for j in range(w["aux"].shape[0]) if w["aux_kind"][j]] or [0])
            c = rng.randrange(w["aux"].shape[1])
            old = cur(w["aux"], (i, c))
            put(w["aux"], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(P), old ^ (1 << rng.randrange(128)), 27, 28]))
        elif aux and rng.random() < 0.25:
            k = rng.choice(aux)
            i, c = rng.randrange(w[k].shape[0]), rng.randrange(w[k].shape[1])
            old = cur(w[k], (i, c))
            put(w[k], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(P), old ^ (1 << rng.randrange(64))]))
        elif which == "steps":
            c, i = rng.randrange(1, 13), rng.randrange(w["steps"].shape[0])
            if c in (3, 4):
                put(w["steps"], (i, c), rng.randrange(2))
            else:
                old = cur(w["steps"], (i, c))
                put(w["steps"], (i, c), rng.choice([old + 1, old - 1, 0, rng.randrange(P), old ^ 1, 2**64, 2**128 + old]))
        elif which == "rw" and w["rw"].shape[0]:
            i, c = rng.randrange(w["rw"].shape[0]), rng.randrange(14)
            old = cur(w["rw"], (i, c))
            put(w["rw"], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(P), old ^ (1 << rng.randrange(128)),
                                              2**128, 2**255 % P, old + 2**128, rng.randrange(2**128)]))
        elif which == "bytecode" and w["bytecode"].shape[0]:
            i, c = rng.randrange(w["bytecode"].shape[0]), rng.randrange(2, 6)
            old = cur(w["bytecode"], (i, c))
            put(w["bytecode"], (i, c), rng.choice([old + 1, old - 1, 0, 1, rng.randrange(256), rng.randrange(P)]))
        elif w["rw_flags"].shape[0]:
            i = rng.randrange(w["rw_flags"].shape[0])
            w["rw_flags"][i] ^= np.uint32(rng.choice([1, 2]))
    return w


WIDE_SOURCES = ("mul_div_mod", "shl_shr", "sar", "sdiv_smod", "addmod", "mulmod", "create", "ecRecover")


def wide_fuzz_wire(wire, rng):
    """Malformed *word cells*: one to three lo / hi cells of the stack operands (or of ecRecover's aux words) set to values
    >= 2^128, where the reference's witness code computes with unbounded Python ints (`Word.int_value()`:
    mul_div_mod.py:23-41, shl_shr.py:103-127, sar.py:158, sdiv_smod.py:85-99, addmod.py:32-41, mulmod.py:41-50,
    instruction.py:1349-1350, precompiles/ecrecover.py:49-52)."""
    from zkevm_specs_amd.evm_tables import Target

    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    w = {k: v.copy() for k, v in wire.items()}

    def put(arr, idx, val):
        arr[idx] = np.frombuffer(int(val % P).to_bytes(32, "little"), dtype="<u8")

    def cur(arr, idx):
        return int.from_bytes(arr[idx].tobytes(), "little")

    def wide_value(old):
        return rng.choice([1 << 128, (1 << 128) + rng.randrange(1, 1 << 20), P - 1, rng.randrange(1 << 128, P), 1 << 253,
                           (1 << 128) + old, (1 << 128) - 1 + (1 << 128), rng.randrange(1 << 128, 1 << 130), (1 << 127) << 1,
                           ((1 << 127) + rng.randrange(1 << 64)) << 1])

    stack_rows = [i for i in range(w["rw"].shape[0]) if cur(w["rw"], (i, 2)) == int(Target.Stack)]
    aux_words = "aux" in w and w["aux"].shape[1] > 2 and any(int(k) == 5 for k in w["aux_kind"])
    for _ in range(rng.choice([1, 1, 2, 3])):
        if aux_words and rng.random() < 0.6:
            i = rng.choice([j for j in range(w["aux"].shape[0]) if int(w["aux_kind"][j]) == 5])
            c = rng.randrange(8)
            put(w["aux"], (i, c), wide_value(cur(w["aux"], (i, c))))
        elif stack_rows:
            i, c = rng.choice(stack_rows), rng.choice([8, 9])
            put(w["rw"], (i, c), wide_value(cur(w["rw"], (i, c))))
            if rng.random() < 0.3:  # keep the other cell small so that the witness arithmetic goes deep
                put(w["rw"], (i, 17 - c), rng.choice([0, 1, rng.randrange(1 << 64)]))
    return w


def wide_cell_cases():
    """The reference's own tests of the big-int gadgets, re-harvested, each with wide-cell variants (labelled by the unmodified
    reference like every other golden case)."""
    cases = []
    for name in WIDE_SOURCES:
        path = os.path.join(REF_TESTS, "precompiles" if name in PRECOMPILE_TESTS else "", f"test_{name}.py")
        h = Harvest()
        reseed(name)
        rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", "--rootdir=/tmp", "-c", "/dev/null", path], plugins=[h])
        assert rc == 0, (name, rc)
        rng = random.Random(sum(map(ord, name)) + 777)
        got = h.cases if len(h.cases) <= 10 else rng.sample(h.cases, 10)
        cases += [(f"{name}::{c[0]}",) + c[1:] for c in got]
    return cases


def check_tables_against_reference():
    """zkevm_specs_amd/evm_tables.py restates the reference's enum numberings and opcode metadata as
    data; assert every entry against the imported reference so a drift fails golden generation."""
    from zkevm_specs.evm_circuit import execution_state as res, opcode as rop, table as rt
    from zkevm_specs.evm_circuit.execution import EXECUTION_STATE_IMPL
    from zkevm_specs_amd import evm_tables as T

    assert {s.name: int(s) for s in res.ExecutionState} == {s.name: int(s) for s in T.ExecutionState}
    assert {o.name: int(o) for o in rop.Opcode} == {o.name: int(o) for o in T.Opcode}
    for o in rop.valid_opcodes():
        info = rop.OPCODE_INFO_MAP[o]
        assert T.OPCODES[o.name][1:] == (info.constant_gas_cost, int(info.has_dynamic_gas)), o
        assert T.STACK_BOUNDS[o.name] == (info.min_stack_pointer, info.max_stack_pointer), o
    assert sorted(s.name for s in res.ExecutionState if s not in EXECUTION_STATE_IMPL) == sorted(T.REFERENCE_UNIMPLEMENTED)
    for st, ops in T.RESPONSIBLE.items():
        got = res.ExecutionState[st].responsible_opcode()
        assert sorted(int(o) for o in got) == sorted(int(T.Opcode[o]) for o in ops), st
    for mine, ref in [(T.Target, rt.Target), (T.CallContextFieldTag, rt.CallContextFieldTag),
                      (T.AccountFieldTag, rt.AccountFieldTag), (T.TxContextFieldTag, rt.TxContextFieldTag),
                      (T.BlockContextFieldTag, rt.BlockContextFieldTag), (T.BytecodeFieldTag, rt.BytecodeFieldTag),
                      (T.FixedTableTag, rt.FixedTableTag), (T.RW, rt.RW)]:
        assert {e.name: int(e) for e in mine} == {e.name: int(e) for e in ref}, mine
    assert [s.name for s in res.ExecutionState if s.halts_in_success()] == T.HALTS_IN_SUCCESS or \
        sorted(s.name for s in res.ExecutionState if s.halts_in_success()) == sorted(T.HALTS_IN_SUCCESS)
    assert sorted(s.name for s in res.ExecutionState if s.halts_in_exception()) == sorted(T.HALTS_IN_EXCEPTION)


def end_block_padding_cases():
    """EndBlock steps that pad the circuit (end_block.py, `else` branch: rw_counter and call_id propagate).
    The reference's own EndBlock tests only exercise the is_last_step branch, which needs whole-table
    aggregates and the withdrawal table; these cases pin the padding rule the engine does evaluate."""
    from zkevm_specs.evm_circuit import ExecutionState, StepState, Tables

    cases = []
    for rwc, call_id in ((1, 0), (23, 1), (2**40, 77)):
        tables = Tables(block_table=set(), tx_table=set(), withdrawal_table=set(), bytecode_table=set(), rw_table=set())
        steps = [StepState(ExecutionState.EndBlock, rw_counter=rwc, call_id=call_id) for _ in range(4)]
        cases.append((f"end_block_padding[{rwc}-{call_id}]", tables, steps, False, False, True))
    return cases


def error_oog_precompile_cases():
    """ErrorOutOfGasPrecompile (precompiles/error_oog_precompile.py) has no test in the reference's suite: these
    witnesses are evaluated by the unmodified reference here, like every other golden case.  The state is not in the
    reference's halts_in_exception list, so a root call cannot move on to EndTx (instruction.py:189-204): the valid
    cases are internal calls that restore the caller's context."""
    from zkevm_specs.evm_circuit import CallContextFieldTag as C, ExecutionState, RWDictionary, StepState, Tables
    from zkevm_specs.util import Word

    cases = []
    code_hash = Word(0x1234567890ABCDEF << 130 | 0x42)
    for address, calldata_len, gas_left, rev, is_root in ((4, 64, 10, 0, False), (4, 0, 14, 3, False), (8, 192, 78999, 0, False),
                                                          (8, 384, 112999, 1, False), (1, 128, 2999, 0, False), (6, 128, 149, 2, False),
                                                          (4, 64, 21, 0, False), (12, 64, 1, 0, False), (4, 64, 10, 0, True)):
        rw = (RWDictionary(9).call_context_read(2, C.CalleeAddress, Word(address)).call_context_read(2, C.CallDataLength, calldata_len)
              .call_context_read(2, C.IsSuccess, 0))
        if is_root:
            nxt = StepState(ExecutionState.EndTx, rw_counter=12 + rev, call_id=2, is_root=True)
        else:
            rw = (rw.call_context_read(2, C.CallerId, 1).call_context_read(1, C.IsRoot, 1).call_context_read(1, C.IsCreate, 0)
                  .call_context_read(1, C.CodeHash, code_hash).call_context_read(1, C.ProgramCounter, 77)
                  .call_context_read(1, C.StackPointer, 1000).call_context_read(1, C.GasLeft, 5000)
                  .call_context_read(1, C.MemorySize, 3).call_context_read(1, C.ReversibleWriteCounter, 6)
                  .call_context_write(1, C.LastCalleeId, 2).call_context_write(1, C.LastCalleeReturnDataOffset, 0)
                  .call_context_write(1, C.LastCalleeReturnDataLength, 0))
            nxt = StepState(ExecutionState.PUSH, rw_counter=9 + 3 + rev + 12, call_id=1, is_root=True, is_create=False,
                            code_hash=code_hash, program_counter=77, stack_pointer=1000, gas_left=5000, memory_word_size=3,
                            reversible_write_counter=6)
        tables = Tables(block_table=set(), tx_table=set(), withdrawal_table=set(), bytecode_table=set(), rw_table=set(rw.rws))
        steps = [StepState(ExecutionState.ErrorOutOfGasPrecompile, rw_counter=9, call_id=2, is_root=is_root, gas_left=gas_left,
                           reversible_write_counter=rev), nxt]
        cases.append((f"error_oog_precompile[{address}-{calldata_len}-{gas_left}-{rev}-{is_root}]", tables, steps, False, False, None))
    return cases


def main():
    from zkevm_specs_amd.flatten import flatten_evm

    os.makedirs(GOLDEN, exist_ok=True)
    check_tables_against_reference()
    total = 0
    only = sys.argv[1:] if len(sys.argv) > 1 and sys.argv[0].endswith("gen_golden_evm.py") else None
    for name in TEST_FILES:
        if only and name not in only:
            continue
        rng = random.Random(hash(name) % 1000 + 20240807) if False else random.Random(sum(map(ord, name)) + 20240807)
        if name == "end_block_padding":
            cases = end_block_padding_cases()
        elif name == "error_oog_precompile_custom":
            cases = error_oog_precompile_cases()
        elif name == "wide_cells":
            cases = wide_cell_cases()
        else:
            path = os.path.join(REF_TESTS, "precompiles" if name in PRECOMPILE_TESTS else "", f"test_{name}.py")
            h = Harvest()
            reseed(name)
            rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", "--rootdir=/tmp", "-c", "/dev/null", path], plugins=[h])
            assert rc == 0, (name, rc)
            cases = h.cases
        if len(cases) > MAX_CASES_PER_FILE and name != "wide_cells":
            cases = rng.sample(cases, MAX_CASES_PER_FILE)
        out, names = {}, []
        n_fuzz_fail = 0
        for tid, tables, steps, begin, end, success in cases:
            wire = flatten_evm(tables, steps)
            kinds = ref_step_outcomes(tables, steps, begin, end)
            assert success is None or success != any(kinds), (tid, kinds)  # a test expects a failure iff a step pair fails
            # sanity: the unflattened witness behaves identically
            t2, s2 = unflatten(wire)
            assert ref_step_outcomes(t2, s2, begin, end) == kinds, tid
            driver = ref_driver_outcomes(t2, s2, begin, end)
            assert driver == ref_driver_outcomes(tables, steps, begin, end), tid
            variants = [("", wire, kinds, driver)]
            for k in range(6 if name == "wide_cells" else 3):
                fw = wide_fuzz_wire(wire, rng) if name == "wide_cells" else fuzz_wire(wire, rng)
                t3, s3 = unflatten(fw)
                fk = ref_step_outcomes(t3, s3, begin, end)
                n_fuzz_fail += any(fk)
                variants.append((f"#fuzz{k}", fw, fk, ref_driver_outcomes(t3, s3, begin, end)))
            for suffix, w, kd, drv in variants:
                key = f"c{len(names):04d}"
                names.append(tid + suffix)
                for k, v in w.items():
                    out[f"{key}_{k}"] = v
                out[f"{key}_opts"] = np.array([int(begin), int(end)], dtype=np.uint8)
                out[f"{key}_ref_kind"] = np.array(kd, dtype=np.uint8)
                out[f"{key}_ref_driver"] = np.array(drv, dtype=np.uint8)  # verify_steps(success=True / False): exception kind
        out["names"] = np.array(names)
        fn = os.path.join(GOLDEN, f"evm_{name}.npz")
        np.savez_compressed(fn, **out)
        total += len(names)
        print(f"evm/{name}: {len(names)} cases ({n_fuzz_fail} fuzzed with failures) -> {os.path.getsize(fn)//1024} KiB", flush=True)
    print("total evm cases", total)


if __name__ == "__main__":
    main()
