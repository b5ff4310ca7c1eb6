"""Tx- and Sig-circuit tests.  CPU: oracle vs the reference's recorded per-unit outcomes and the
kernel's logic (hostsim) vs the oracle; GPU (marked): through the C ABI, incl. BASELINE config 4's
2^14 synthetic tx slots."""
import ctypes
import os
import random

import numpy as np
import pytest

from oracle import codes, sign_oracle as so, wire
from zkevm_specs_amd.synth import synth_tx_witness

FIELDS = ("bytes", "cells", "meta", "keccak", "tx_rows", "tx_flags", "r", "is_sig", "ref_kind")


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "sign_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), {f: np.ascontiguousarray(g[f"{k}_{f}"]) for f in FIELDS}


def _oracle(c, r=None, is_sig=None):
    r = wire.cells_to_ints(c["r"])[0] if r is None else r
    is_sig = int(c["is_sig"][0]) if is_sig is None else is_sig
    return so.verify_units(c["bytes"], c["cells"], c["meta"], wire.rowmajor_to_rows(c["keccak"]), r, is_sig,
                           wire.rowmajor_to_rows(c["tx_rows"]), c["tx_flags"])


def _sim(lib, c, r_cells, is_sig):
    vp = lambda x: ctypes.c_void_p(np.ascontiguousarray(x).ctypes.data)  # noqa: E731
    u64 = ctypes.c_uint64
    n = c["bytes"].shape[0]
    st = np.zeros(n, dtype=np.uint32)
    lib.sim_sign_verify(vp(c["bytes"]), vp(c["cells"]), vp(c["meta"]), u64(n), vp(c["keccak"]), u64(c["keccak"].shape[0]),
                        vp(c["tx_rows"]), vp(c["tx_flags"]), u64(c["tx_rows"].shape[0]), vp(r_cells), ctypes.c_uint32(is_sig), vp(st))
    return st.tolist()


def test_oracle_reference_and_kernel_logic(golden_dir, hostsim):
    n = n_fail = 0
    for name, c in _cases(golden_dir):
        exp = _oracle(c)
        assert [codes.kind_of(e) for e in exp] == c["ref_kind"].tolist(), name
        assert _sim(hostsim, c, c["r"], int(c["is_sig"][0])) == exp, name
        n += len(exp)
        n_fail += sum(1 for e in exp if e)
    assert n > 300 and n_fail > 80


def _tamper(w, rng, k):
    w = {key: v.copy() for key, v in w.items()}
    n = w["bytes"].shape[0]
    for _ in range(k):
        i, what = rng.randrange(n), rng.randrange(6)
        if what == 0:
            w["bytes"][i, rng.randrange(7), rng.randrange(32)] ^= 1
        elif what == 1:
            w["cells"][rng.randrange(3), i, 0] ^= np.uint64(1)
        elif what == 2:
            w["meta"][i, 0] = rng.choice([1, 9 << 24])
        elif what == 3:
            w["tx_rows"][i * 12 + rng.choice([3, 11]), 3, 0] ^= np.uint64(2)
        elif what == 4:
            w["tx_flags"][i * 12 + 3] ^= np.uint32(1)
        else:
            w["keccak"][rng.randrange(w["keccak"].shape[0]), 1, 0] ^= np.uint64(1)
    return w


def test_synthetic_tx_witness_and_tampering(hostsim):
    r = random.Random(9).getrandbits(253)
    rc = np.frombuffer(int(r).to_bytes(32, "little"), dtype="<u8").copy()
    w = synth_tx_witness(200, r, seed=4, padding=8)
    exp = _oracle(w, r, 0)
    assert len(exp) == 208 and not any(exp)
    assert _sim(hostsim, w, rc, 0) == exp
    rng = random.Random(10)
    tw = _tamper(w, rng, 60)
    exp = _oracle(tw, r, 0)
    assert _sim(hostsim, tw, rc, 0) == exp and sum(1 for e in exp if e) > 30


@pytest.mark.gpu
def test_gpu_goldens_and_config4(golden_dir):
    from zkevm_specs_amd import engine

    for name, c in _cases(golden_dir):
        exp = _oracle(c)
        with engine.open_sign(c, c["r"], bool(c["is_sig"][0])) as s:
            res = s.run()
            status = s.read_status().tolist()
        assert status == exp and [e >> 24 for e in status] == c["ref_kind"].tolist(), name
        fails = [j for j, e in enumerate(exp) if e]
        assert res.fail_count == len(fails)
        if fails:
            assert res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]
    # BASELINE configs[3]: 2^14 synthetic txs
    r = random.Random(11).getrandbits(253)
    w = synth_tx_witness(1 << 14, r, seed=4)
    with engine.open_sign(w, r, False) as s:
        res = s.run()
    assert res.ok and res.rows_evaluated == 1 << 14
    tw = _tamper(w, random.Random(12), 300)
    with engine.open_sign(tw, r, False) as s:
        res = s.run()
        status = s.read_status().tolist()
    exp = _oracle(tw, r, 0)
    assert status == exp and res.fail_count == sum(1 for e in exp if e) > 150
