"""State-circuit witness assignment: oracle vs the reference's recorded
`assign_state_circuit` / `mpt_table_from_ops` outputs, the device functions' logic (hostsim) vs the
oracle, and the HIP path vs the oracle (gpu)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import assign_oracle, codes, state_oracle, wire
from zkevm_specs_amd.synth import synth_state_ops


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "assign_cases.npz"))
    for i, name in enumerate(g["names"]):
        k = f"c{i:03d}"
        yield str(name), g, k


def _oracle(ops, flags):
    return assign_oracle.assign(wire.colmajor_to_rows(ops), flags.tolist())


def _mpt_set(mpt):
    return sorted(set(tuple(r) for r in mpt))


def test_oracle_matches_reference(golden_dir):
    """rows, type bits, MPT rows and exception classes equal what the unmodified reference produced."""
    n_cases = n_raise = 0
    for name, g, k in _cases(golden_dir):
        rows, rflags, mpt, status = _oracle(g[k + "_ops"], g[k + "_opflags"])
        err = assign_oracle.first_error(status)
        assert (codes.kind_of(err[1]) if err else 0) == int(g[k + "_kind"]), name
        mock_err = next((c for c in status if c and codes.site_of(c) in (1, 2, 4)), 0)
        assert codes.kind_of(mock_err) == int(g[k + "_mpt_kind"]), name
        if not err:
            assert rows == wire.colmajor_to_rows(g[k + "_rows"]), name
            assert rflags == g[k + "_rowflags"].tolist(), name
        else:
            n_raise += 1
        if not mock_err:
            assert _mpt_set(mpt) == [tuple(r) for r in wire.rowmajor_to_rows(g[k + "_mpt"])], name
        n_cases += 1
    assert n_cases >= 80 and n_raise >= 15


def _hostsim_assign(lib, ops, flags):
    ops, flags = np.ascontiguousarray(ops), np.ascontiguousarray(flags)
    n = ops.shape[1]
    rows = np.zeros((57, n, 4), dtype=np.uint64)
    rflags = np.zeros(n, dtype=np.uint32)
    mpt = np.zeros((n, 12, 4), dtype=np.uint64)
    status = np.zeros(n, dtype=np.uint32)
    m = ctypes.c_uint64()
    vp = ctypes.c_void_p
    lib.sim_state_assign(vp(ops.ctypes.data), vp(flags.ctypes.data), ctypes.c_uint64(n), vp(rows.ctypes.data),
                         vp(rflags.ctypes.data), vp(mpt.ctypes.data), ctypes.byref(m), vp(status.ctypes.data))
    return rows, rflags, mpt[: m.value], status


def _check_against_oracle(got, ops, flags, tag):
    rows, rflags, mpt, status = got
    e_rows, e_rflags, e_mpt, e_status = _oracle(ops, flags)
    assert status.tolist() == e_status, tag
    assert wire.rowmajor_to_rows(mpt) == e_mpt, tag  # first-occurrence order
    assert rflags.tolist() == e_rflags, tag
    assert wire.colmajor_to_rows(rows) == e_rows, tag


def test_kernel_logic_matches_oracle_on_goldens(golden_dir, hostsim):
    for name, g, k in _cases(golden_dir):
        _check_against_oracle(_hostsim_assign(hostsim, g[k + "_ops"], g[k + "_opflags"]), g[k + "_ops"], g[k + "_opflags"], name)


def _shuffled_hostile(n, seed):
    """synthetic ops, shuffled (first occurrences no longer lead their group), with repeated keys and bad cells"""
    rng = np.random.default_rng(seed)
    ops, flags, *_ = synth_state_ops(n, seed)
    perm = rng.permutation(n)
    ops, flags = np.ascontiguousarray(ops[:, perm]), flags[perm]
    for _ in range(n // 16):
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        m = int(rng.integers(0, 5))
        if m == 0:
            ops[4, i, 2] |= np.uint64(1 << 40)  # address >= 2^160
        elif m == 1:
            ops[8, i, 2] = np.uint64(5)  # value.hi >= 2^128
        elif m == 2:
            ops[2:7, i] = ops[2:7, j]  # same keys as another op
        elif m == 3:
            ops[10, i, 3] = np.uint64(1)  # initial_value.hi >= 2^192
        else:
            ops[4, i] = np.uint64(0xFFFFFFFFFFFFFFFF)  # address = 2^256 - 1: reduced mod p in the row, raw in the limbs
    return ops, flags


def test_kernel_logic_on_shuffled_hostile_ops(hostsim):
    ops, flags = _shuffled_hostile(3000, 21)
    got = _hostsim_assign(hostsim, ops, flags)
    _check_against_oracle(got, ops, flags, "hostile")
    assert sum(1 for c in got[3] if c) > 50


def test_synthetic_ops_assign_to_a_valid_witness():
    """oracle-assigned rows of the synthetic ops == the generator's rows, and they satisfy the State circuit"""
    ops, flags, cols, rflags, mpt = synth_state_ops(2048, 9)
    rows, got_flags, got_mpt, status = _oracle(ops, flags)
    assert not any(status)
    assert rows == wire.colmajor_to_rows(cols) and got_flags == rflags.tolist()
    assert got_mpt == wire.rowmajor_to_rows(mpt)
    assert not any(state_oracle.verify_rows(rows, got_flags, got_mpt))


# ---- GPU --------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_matches_oracle_on_goldens(golden_dir):
    from zkevm_specs_amd import engine

    for name, g, k in _cases(golden_dir):
        ops, flags = g[k + "_ops"], g[k + "_opflags"]
        with engine.open_state_assign(ops, flags) as s:
            res = s.run()
            status = s.read_status()
            rows, rflags, mpt = s.read()
        _check_against_oracle((rows, rflags, mpt, status), ops, flags, name)
        assert res.fail_count == int((status != 0).sum()), name


@pytest.mark.gpu
@pytest.mark.parametrize("n", [255, 256, 257, 5000, 70001])
def test_hip_matches_oracle_on_shuffled_hostile_ops(n):
    """block / wave boundaries of the rank scan and of the next-keyed-op suffix-min; contended MPT keys"""
    from zkevm_specs_amd import engine

    ops, flags = _shuffled_hostile(n, 100 + n)
    with engine.open_state_assign(ops, flags) as s:
        s.run()
        status = s.read_status()
        rows, rflags, mpt = s.read()
    _check_against_oracle((rows, rflags, mpt, status), ops, flags, f"n={n}")


@pytest.mark.gpu
def test_hip_one_key_repeated():
    """every op hits the same storage slot: one MPT row, all lanes contend on one index slot"""
    from zkevm_specs_amd import engine

    ops, flags, *_ = synth_state_ops(4096, 3)
    tag = ops[2, :, 0]
    src = int(np.nonzero(tag == 4)[0][0])
    ops[2:7, 1:] = ops[2:7, src][:, None]
    with engine.open_state_assign(ops, flags) as s:
        s.run()
        status = s.read_status()
        rows, rflags, mpt = s.read()
    assert mpt.shape[0] == 1
    _check_against_oracle((rows, rflags, mpt, status), ops, flags, "one key")


@pytest.mark.gpu
def test_hip_host_mirror_raises_like_the_reference(golden_dir):
    """assign_state_circuit / mpt_table_from_ops on the wire arrays raise the reference's exception classes"""
    from zkevm_specs_amd import errors, state_circuit

    n_raise = 0
    for name, g, k in _cases(golden_dir):
        wire_ops = (g[k + "_ops"], g[k + "_opflags"])
        for fn, kind_key in ((state_circuit.assign_state_circuit, "_kind"), (state_circuit.mpt_table_from_ops, "_mpt_kind")):
            kind = int(g[k + kind_key])
            if kind == 0:
                fn(wire_ops)
            else:
                with pytest.raises(type(errors.exception_for_code(kind << 24))):
                    fn(wire_ops)
                n_raise += 1
    assert n_raise >= 20


@pytest.mark.gpu
def test_assign_then_verify_at_config_size_on_device():
    """BASELINE config 2 size: 2^16 ops -> rows + mock MPT table assigned in HBM -> State circuit evaluated on the
    same buffers: equals the generator's witness bit for bit and satisfies every constraint."""
    import torch

    from zkevm_specs_amd import engine

    n = 1 << 16
    ops, flags, cols, rflags, mpt = synth_state_ops(n, 2)
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
    d_ops, d_flags = dev(ops), dev(flags)
    d_rows = torch.empty((57, n, 4), dtype=torch.int64, device="cuda")
    d_rflags = torch.empty(n, dtype=torch.int32, device="cuda")
    d_mpt = torch.empty((n, 12, 4), dtype=torch.int64, device="cuda")
    with engine.open_state_assign(d_ops, d_flags, d_rows, d_rflags, d_mpt) as s:
        res = s.run()
        m = s.n_mpt()
    assert res.ok and m == mpt.shape[0]
    assert np.array_equal(d_rows.cpu().numpy().view(np.uint64), cols)
    assert np.array_equal(d_rflags.cpu().numpy().view(np.uint32), rflags)
    assert np.array_equal(d_mpt[:m].cpu().numpy().view(np.uint64), mpt)
    with engine.open_state(d_rows, d_rflags, d_mpt[:m]) as s:
        res = s.run()
    assert res.ok and res.rows_evaluated == n
