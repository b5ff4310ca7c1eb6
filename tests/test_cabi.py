"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/zkevm_hip.h declares; the host-side error mapping mirrors the reference."""
import ctypes
import os
import re

import pytest

from zkevm_specs_amd import _lib, errors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkevm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 10
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/zkevm_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_no_gpu_fails_loudly():
    """Without a usable GPU the engine must raise, never fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EngineError):
        _lib.init(0)


def test_error_mapping_classes():
    assert isinstance(errors.exception_for_code((1 << 24) | 5), AssertionError)
    assert isinstance(errors.exception_for_code((2 << 24) | 5), errors.ConstraintUnsatFailure)
    assert isinstance(errors.exception_for_code((3 << 24) | 5), errors.LookupUnsatFailure)
    assert isinstance(errors.exception_for_code((4 << 24) | 5), errors.LookupAmbiguousFailure)
    assert isinstance(errors.exception_for_code((6 << 24) | 5), NotImplementedError)
    assert isinstance(errors.exception_for_code((9 << 24) | 5), ValueError)
    errors.raise_for_code(0)
