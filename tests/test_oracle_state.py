"""CPU tests: State-circuit oracle vs the reference's recorded outcomes, and the kernels'
per-row logic (hostsim build of csrc/state_circuit.hpp) vs the oracle."""
import ctypes
import os

import numpy as np
import pytest

from oracle import codes, state_oracle, wire
from zkevm_specs_amd.synth import synth_state_witness


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "state_cases.npz"))
    for i, name in enumerate(g["names"]):
        k = f"c{i:03d}"
        yield str(name), g[k + "_rows"], g[k + "_flags"], g[k + "_mpt"], g[k + "_ref_kind"]


def _hostsim_state(lib, cols, flags, mpt):
    cols, flags, mpt = map(np.ascontiguousarray, (cols, flags, mpt))
    n = cols.shape[1]
    st = np.zeros(n, dtype=np.uint32)
    vp = ctypes.c_void_p
    lib.sim_state_verify(vp(cols.ctypes.data), vp(flags.ctypes.data), ctypes.c_uint64(n), vp(mpt.ctypes.data),
                         ctypes.c_uint64(mpt.shape[0]), vp(st.ctypes.data))
    return st


def test_oracle_matches_reference_outcomes(golden_dir):
    """Every row of every golden case: oracle's exception class == the reference's."""
    n_rows = n_fail = 0
    for name, cols, flags, mpt, ref_kind in _cases(golden_dir):
        got = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
        kinds = [codes.kind_of(c) for c in got]
        assert kinds == ref_kind.tolist(), name
        n_rows += len(kinds)
        n_fail += sum(k != 0 for k in kinds)
    assert n_rows > 4000 and n_fail > 250


def test_kernel_logic_matches_oracle_on_goldens(golden_dir, hostsim):
    for name, cols, flags, mpt, _ in _cases(golden_dir):
        exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
        got = _hostsim_state(hostsim, cols, flags, mpt)
        assert got.tolist() == exp, name


@pytest.mark.parametrize("n", [64, 1000, 4096])
def test_synthetic_witness_is_valid(n):
    cols, flags, mpt = synth_state_witness(n, seed=2)
    got = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
    assert not any(got)


def test_kernel_logic_on_tampered_synthetic(hostsim):
    """Flip random cells of a 4096-row synthetic witness; statuses must equal the oracle's."""
    rng = np.random.default_rng(7)
    cols, flags, mpt = synth_state_witness(4096, seed=3)
    for _ in range(200):
        c, i = int(rng.integers(0, 57)), int(rng.integers(0, 4096))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            cols[c, i, 0] ^= np.uint64(1)
        elif mode == 1:
            cols[c, i, int(rng.integers(0, 4))] = np.uint64(rng.integers(0, 2**62))
            cols[c, i, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)  # keep the cell canonical (< p)
        elif mode == 2:
            cols[c, i] = 0
        else:
            flags[i] ^= np.uint32(1 << int(rng.integers(0, 2)))
    exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
    got = _hostsim_state(hostsim, cols, flags, mpt)
    assert got.tolist() == exp
    assert sum(1 for e in exp if e) > 100
