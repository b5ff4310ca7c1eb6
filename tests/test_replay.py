"""Failure replay (zkevm_specs_amd/replay.py) in the build container.  Since round 4 replay is an opt-in
diagnostic (ZK_REPLAY=always: the reference's own message): every golden pair — word cells >= 2^128 included — has a verdict of
its own, and the mirror's `verify_steps` must raise the class the unmodified reference raises without executing the reference.  The device is stood in for by the oracle (as in tests/test_dropin_cpu.py); the replay
itself runs the reference (needs /root/reference + oracle/refshim: skipped elsewhere, e.g. on the GPU box)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is not on this machine")


@pytest.fixture()
def reference_on_path(monkeypatch):
    os.environ.setdefault("ZKEVM_SHIM_SEED", "20240807")
    for p in (REF, os.path.join(ROOT, "oracle", "refshim")):
        if p not in sys.path:
            monkeypatch.syspath_prepend(p)
    yield


def test_wide_cell_pairs_need_no_replay(reference_on_path, monkeypatch, golden_dir):
    """Round 4: word cells >= 2^128 (where the reference computes with unbounded Python ints) have verdicts of their own
    (csrc/bigz.hpp), so with the default ZK_REPLAY=never the mirror's `verify_steps` must raise the class the unmodified
    reference's driver raised on the reference-labelled wide-cell goldens — and, the reference being loaded in this process,
    the reference's OWN exception class objects (errors._boundary_exception)."""
    from oracle import codes
    from oracle.gen_golden_evm import unflatten
    from tests import dropin_cases as D
    from tests.evm_cases import load_cases, oracle_status
    from tests.test_dropin_cpu import _result
    from zkevm_specs_amd import oneshot, replay
    from zkevm_specs_amd.evm_circuit import verify_steps

    def evm_verify(w, begin=False, end=False, opts=0, device=None):
        st = oracle_status(w, (int(begin), int(end)))
        return _result(st), np.array(st, dtype=np.uint32)

    monkeypatch.setattr(oneshot, "evm_verify", evm_verify)
    monkeypatch.delenv("ZK_REPLAY", raising=False)
    assert replay.replay_mode() == "never"
    called = []
    monkeypatch.setattr(replay, "replay_step", lambda *a, **k: called.append(a))
    fn = os.path.join(golden_dir, "evm_wide_cells.npz")
    g = np.load(fn)
    n = 0
    for ci, (name, w, opts, ref_kind) in enumerate(load_cases(fn)):
        if ci % 5 or "#fuzz" not in name:
            continue
        assert not any(codes.kind_of(c) == codes.UNSUPPORTED for c in oracle_status(w, opts))
        driver = g[f"c{ci:04d}_ref_driver"].tolist()
        begin, end = bool(opts[0]), bool(opts[1])
        for success, kind in zip((True, False), driver):
            tables, steps = unflatten(w)
            st = list(steps[:-1] if end else steps)
            D.expect_outcome(kind, lambda: verify_steps(tables, st, begin, end, success))  # noqa: B023
        n += 1
    assert n >= 70 and not called


def test_mirror_raises_the_references_own_classes(reference_on_path):
    """errors.exception_for_code with the reference loaded: the reference's class objects, constructed with its signatures
    (evm_circuit/table.py:363-378, instruction.py:53, util/constraint_system.py:7)."""
    import zkevm_specs.evm_circuit.instruction as ins
    import zkevm_specs.evm_circuit.table as tab
    import zkevm_specs.util.constraint_system as cs
    from zkevm_specs_amd import errors

    assert type(errors.exception_for_code((2 << 24) | 5, "EVM circuit step 3")) is ins.ConstraintUnsatFailure
    assert type(errors.exception_for_code((2 << 24) | 5, "Exp circuit row 3")) is cs.ConstraintUnsatFailure
    assert type(errors.exception_for_code((3 << 24) | 1, "EVM circuit step 0")) is tab.LookupUnsatFailure
    assert type(errors.exception_for_code((4 << 24) | 1, "EVM circuit step 0")) is tab.LookupAmbiguousFailure
    assert type(errors.exception_for_code((5 << 24) | 1, "EVM circuit step 0")) is tab.WrongQueryKey
    e = errors.exception_for_code((1 << 24) | 9, "EVM circuit step 1")
    assert type(e) is AssertionError and type(e.args[0]) is ins.ConstraintUnsatFailure
    assert "site 9" in errors.exception_for_code((3 << 24) | 9, "x").message


def test_replay_always_gives_the_references_own_exception(reference_on_path, monkeypatch, golden_dir):
    """ZK_REPLAY=always: every failure is re-raised by the reference itself (its class AND message), the device's kind being the
    cross-check — a sample of ordinary failing golden cases."""
    from oracle.gen_golden_evm import unflatten
    from tests.evm_cases import golden_files, load_cases, oracle_status
    from tests.test_dropin_cpu import _result
    from zkevm_specs_amd import errors, oneshot
    from zkevm_specs_amd.evm_circuit import verify_steps

    def evm_verify(w, begin=False, end=False, opts=0, device=None):
        st = oracle_status(w, (int(begin), int(end)))
        return _result(st), np.array(st, dtype=np.uint32)

    monkeypatch.setattr(oneshot, "evm_verify", evm_verify)
    monkeypatch.setenv("ZK_REPLAY", "always")
    n = 0
    for fn in golden_files(golden_dir)[::6]:
        for ci, (name, w, opts, ref_kind) in enumerate(load_cases(fn)):
            if ci % 9 or not any(ref_kind.tolist()):
                continue
            tables, steps = unflatten(w)
            begin, end = bool(opts[0]), bool(opts[1])
            first_kind = next(k for k in ref_kind.tolist() if k)
            try:
                verify_steps(tables, list(steps[:-1] if end else steps), begin, end, True)
            except Exception as e:  # noqa: BLE001
                assert errors.kind_for_exception(e) == first_kind, (name, type(e))
                # the reference's own exception object, not the mirror's mapped one
                assert "constraint site" not in str(e)
            else:
                raise AssertionError(f"{name}: expected a failure")
            n += 1
    assert n >= 10
