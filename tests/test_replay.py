"""Failure replay (zkevm_specs_amd/replay.py) in the build container: the golden step pairs for which the device
has no verdict (word cells >= 2^128 in MUL/DIV/MOD, SHL/SHR, SAR, SDIV/SMOD, the ecRecover aux data: kind UnsupportedOnDevice)
must come out of the mirror's `verify_steps` with the exception class the unmodified reference raises, once the caller hands
over the reference's own objects.  The device is stood in for by the oracle (as in tests/test_dropin_cpu.py); the replay
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


def test_unsupported_pairs_report_the_references_class(reference_on_path, monkeypatch, golden_dir):
    from oracle import codes
    from oracle.gen_golden_evm import unflatten
    from tests import dropin_cases as D
    from tests.evm_cases import golden_files, load_cases, oracle_status
    from tests.test_dropin_cpu import _result
    from zkevm_specs_amd import errors, oneshot, replay
    from zkevm_specs_amd.evm_circuit import verify_steps

    def evm_verify(w, begin=False, end=False, opts=0, device=None):
        st = oracle_status(w, (int(begin), int(end)))
        return _result(st), np.array(st, dtype=np.uint32)

    monkeypatch.setattr(oneshot, "evm_verify", evm_verify)
    assert replay.reference_available()
    n = 0
    for fn in golden_files(golden_dir):
        g = np.load(fn)
        for ci, (name, w, opts, ref_kind) in enumerate(load_cases(fn)):
            first = next((c for c in oracle_status(w, opts) if c), 0)
            if codes.kind_of(first) != codes.UNSUPPORTED:
                continue
            driver = g[f"c{ci:04d}_ref_driver"].tolist()
            begin, end = bool(opts[0]), bool(opts[1])
            for success, kind in zip((True, False), driver):
                tables, steps = unflatten(w)
                assert replay.is_reference_tables(tables)
                st = list(steps[:-1] if end else steps)
                D.expect_outcome(kind, lambda: verify_steps(tables, st, begin, end, success))  # noqa: B023
            # without the replay the mirror says where the device stopped
            monkeypatch.setenv("ZK_REPLAY", "never")
            tables, steps = unflatten(w)
            with pytest.raises(errors.UnsupportedOnDevice):
                verify_steps(tables, list(steps[:-1] if end else steps), begin, end, True)
            monkeypatch.delenv("ZK_REPLAY")
            n += 1
    assert n == 8


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
