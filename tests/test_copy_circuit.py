"""Copy-circuit tests.  CPU: oracle vs the reference's recorded outcomes and the kernel's row logic
(hostsim, both index modes) vs the oracle; GPU (marked): the same through the C ABI."""
import ctypes
import os

import numpy as np
import pytest

from oracle import codes, copy_oracle as co, wire

KEYS = ("rows", "flags", "rw", "rw_flags", "bytecode", "tx", "tx_flags", "r", "ref_kind")


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "copy_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), {key: np.ascontiguousarray(g[f"{k}_{key}"]) for key in KEYS}


def _oracle(c):
    T = co.CopyTables(wire.rowmajor_to_rows(c["rw"]), c["rw_flags"], wire.rowmajor_to_rows(c["bytecode"]),
                      wire.rowmajor_to_rows(c["tx"]), c["tx_flags"])
    return co.verify_rows(wire.colmajor_to_rows(c["rows"]), c["flags"], T, wire.cells_to_ints(c["r"])[0])


def test_oracle_reference_and_kernel_logic(golden_dir, hostsim):
    vp = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
    u64 = ctypes.c_uint64
    n = n_fail = 0
    for name, c in _cases(golden_dir):
        exp = _oracle(c)
        assert [codes.kind_of(e) for e in exp] == c["ref_kind"].tolist(), name
        for generic in (0, 1):
            st = np.zeros(len(exp), dtype=np.uint32)
            hostsim.sim_copy_verify(vp(c["rows"]), vp(c["flags"]), u64(len(exp)), vp(c["r"]), vp(c["rw"]), vp(c["rw_flags"]),
                                    u64(c["rw"].shape[0]), vp(c["bytecode"]), u64(c["bytecode"].shape[0]), vp(c["tx"]),
                                    vp(c["tx_flags"]), u64(c["tx"].shape[0]), ctypes.c_uint32(generic), vp(st))
            assert st.tolist() == exp, (name, generic)
        n += len(exp)
        n_fail += sum(1 for e in exp if e)
    assert n > 10000 and n_fail > 300


@pytest.mark.gpu
def test_gpu_goldens(golden_dir):
    from zkevm_specs_amd import engine

    for idx, (name, c) in enumerate(_cases(golden_dir)):
        exp = _oracle(c)
        for generic in ((False, True) if idx % 5 == 0 else (False,)):
            with engine.open_copy(c["rows"], c["flags"], c["r"], c["rw"], c["rw_flags"], c["bytecode"], c["tx"], c["tx_flags"],
                                  generic_index=generic) as s:
                res = s.run()
                status = s.read_status().tolist()
            assert status == exp, (name, generic)
            assert [e >> 24 for e in status] == c["ref_kind"].tolist(), name
            fails = [j for j, e in enumerate(exp) if e]
            assert res.fail_count == len(fails)
            if fails:
                assert res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]
