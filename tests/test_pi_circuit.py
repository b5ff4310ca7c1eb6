"""Public-inputs circuit.  CPU: oracle vs the unmodified reference's per-row outcomes (witnesses of
`public_data2witness` + cell-level fuzz, tests/golden/pi_cases.npz) and the kernel's row logic (hostsim) vs the oracle;
GPU (marked): the same through the C ABI (session and one-shot)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import codes, pi_oracle as PO, wire

vp = lambda x: ctypes.c_void_p(np.ascontiguousarray(x).ctypes.data)  # noqa: E731


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "pi_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        if k + "_rows" in g.files:  # a witness as the reference produced it
            base = (np.ascontiguousarray(g[k + "_rows"]), np.ascontiguousarray(g[k + "_gas"]), np.ascontiguousarray(g[k + "_keccak"]))
            rows = base[0]
        else:                       # a fuzz variant: the cells that differ from its base witness
            rows = base[0].copy()
            idx = g[k + "_diff_idx"]
            rows[idx[:, 0], idx[:, 1]] = g[k + "_diff_val"]
        yield str(nm), rows, base[1], base[2], int(g[k + "_circuit_len"][0]), g[k + "_ref_kind"].tolist()


def _oracle(cols, gas, keccak, circuit_len):
    return PO.verify_rows(wire.colmajor_to_rows(cols), wire.rowmajor_to_rows(gas), wire.rowmajor_to_rows(keccak), circuit_len)


def _cell(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype="<u8").copy()


def _sim(lib, cols, gas, keccak, circuit_len):
    st = np.zeros(cols.shape[1], dtype=np.uint32)
    lib.sim_pi_verify(vp(cols), ctypes.c_uint64(cols.shape[1]), vp(keccak), ctypes.c_uint64(keccak.shape[0]), vp(gas), ctypes.c_uint64(gas.shape[0]),
                      ctypes.c_uint64(circuit_len), vp(_cell(255)), vp(_cell(255)), vp(st))
    return st.tolist()


def test_oracle_reference_and_kernel_logic(golden_dir, hostsim):
    n = n_fail = 0
    sites = set()
    for name, cols, gas, keccak, circuit_len, ref_kind in _cases(golden_dir):
        exp = _oracle(cols, gas, keccak, circuit_len)
        assert [codes.kind_of(e) for e in exp] == ref_kind, name
        assert _sim(hostsim, cols, gas, keccak, circuit_len) == exp, name
        n += len(exp)
        n_fail += sum(1 for e in exp if e)
        sites |= {codes.site_of(e) for e in exp if e}
    assert n > 100000 and n_fail > 40 and len(sites) >= 8, (n, n_fail, sorted(sites))


@pytest.mark.gpu
def test_gpu_goldens(golden_dir):
    from zkevm_specs_amd import engine, oneshot

    for idx, (name, cols, gas, keccak, circuit_len, ref_kind) in enumerate(_cases(golden_dir)):
        exp = _oracle(cols, gas, keccak, circuit_len)
        with engine.open_pi(cols, keccak, gas, circuit_len) as s:
            res = s.run()
            status = s.read_status().tolist()
        assert status == exp, name
        fails = [j for j, e in enumerate(exp) if e]
        assert res.fail_count == len(fails)
        if fails:
            assert res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]
        if idx % 4 == 0:
            res1, st1 = oneshot.pi_verify(cols, keccak, gas, circuit_len)
            assert st1.tolist() == exp and res1.fail_count == len(fails)
