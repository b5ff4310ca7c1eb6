"""GPU parity tests of the State-circuit kernel, through the C ABI."""
import os

import numpy as np
import pytest

from oracle import codes, state_oracle, wire
from zkevm_specs_amd import engine
from zkevm_specs_amd.synth import synth_state_witness

pytestmark = pytest.mark.gpu
P = wire.P


def _run(cols, flags, mpt):
    with engine.open_state(cols, flags, mpt) as s:
        res = s.run()
        return res, s.read_status()


def test_fr_ops_known_answers():
    """Device Fr add/sub/mul/montmul/neg/inv/div vs Python big-int (bit-exact)."""
    import random

    rng = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, 2**128, 2**255 % P, 2**64 - 1, 2**64, 2**253, pow(8, -1, P)]
    A = [rng.choice(edge) if rng.random() < 0.3 else rng.randrange(P) for _ in range(20000)]
    B = [rng.choice(edge) if rng.random() < 0.3 else rng.randrange(P) for _ in range(20000)]
    a, b = wire.ints_to_cells(A), wire.ints_to_cells(B)
    rinv = pow(1 << 256, -1, P)
    inv = lambda x: pow(x, -1, P) if x else 0  # noqa: E731  (py_ecc's prime_field_inv(0) == 0)
    fns = [lambda x, y: (x + y) % P, lambda x, y: (x - y) % P, lambda x, y: x * y % P,
           lambda x, y: x * y * rinv % P, lambda x, y: (-x) % P, lambda x, y: inv(x), lambda x, y: x * inv(y) % P]
    for op, f in enumerate(fns):
        got = wire.cells_to_ints(engine.fr_op(op, a, b))
        assert got == [f(x, y) for x, y in zip(A, B)], f"op {op}"


def test_golden_cases_match_reference_and_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "state_cases.npz"))
    for i, name in enumerate(g["names"]):
        k = f"c{i:03d}"
        cols, flags, mpt, ref_kind = g[k + "_rows"], g[k + "_flags"], g[k + "_mpt"], g[k + "_ref_kind"]
        res, status = _run(cols, flags, mpt)
        exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
        assert status.tolist() == exp, name
        assert [c >> 24 for c in status.tolist()] == ref_kind.tolist(), name
        fails = [j for j, c in enumerate(exp) if c]
        assert res.fail_count == len(fails)
        if fails:
            assert res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]
        else:
            assert res.first_fail_row is None


def test_full_size_valid_witness_passes():
    """BASELINE config 2: 2^16 RW rows, all constraints satisfied."""
    cols, flags, mpt = synth_state_witness(1 << 16, seed=2)
    res, status = _run(cols, flags, mpt)
    assert res.ok and res.rows_evaluated == 1 << 16 and not status.any()


@pytest.mark.parametrize("k", [1, 16, 256])
def test_full_size_tampered_matches_oracle(k):
    """Flip k cells at 2^16 rows: per-row statuses must be bit-identical to the oracle's."""
    n = 1 << 16
    rng = np.random.default_rng(100 + k)
    cols, flags, mpt = synth_state_witness(n, seed=2)
    for _ in range(k):
        c, i = int(rng.integers(0, 57)), int(rng.integers(0, n))
        cols[c, i, int(rng.integers(0, 2))] ^= np.uint64(1 << int(rng.integers(0, 60)))
    res, status = _run(cols, flags, mpt)
    exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
    assert status.tolist() == exp
    fails = [j for j, c in enumerate(exp) if c]
    assert res.fail_count == len(fails) and len(fails) >= 1
    assert res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]


def test_device_pointer_path_and_idempotence():
    """Inputs already resident in HBM (torch tensors); repeated passes give the same tally."""
    import torch

    cols, flags, mpt = synth_state_witness(1 << 14, seed=5)
    cols[1, 77, 0] = np.uint64(2)
    d = [torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda() for x in (cols, flags, mpt)]
    with engine.open_state(*d) as s:
        s.launch()
        s.launch()
        r1 = s.collect()
        r2 = s.run()
    assert r1.launches == 2 and r2.launches == 1
    assert (r1.fail_count, r1.first_fail_row, r1.first_fail_code) == (r2.fail_count, r2.first_fail_row, r2.first_fail_code)
    assert r1.fail_count >= 1 and r1.first_fail_row <= 78


def test_sharded_ranges_with_halo_match_full_pass():
    """Multi-GPU sharding on one GPU: two row ranges with halos give the full pass's statuses."""
    from zkevm_specs_amd import distributed

    n = 1 << 13
    cols, flags, mpt = synth_state_witness(n, seed=6)
    cols[1, 5000, 0] = np.uint64(2)
    cols[50, n // 2 - 1, 0] ^= np.uint64(1)
    cols[0, n // 2, 0] = np.uint64(0)
    _, full = _run(cols, flags, mpt)
    got = np.zeros(n, dtype=np.uint32)
    total = 0
    for rank in range(2):
        lc, lf, lo, hi, off = distributed.shard_state(cols, flags, rank, 2)
        with engine.open_state(lc, lf, mpt) as s:
            s.set_range(lo, hi)
            res = s.run()
            st = s.read_status()
        assert res.rows_evaluated == hi - lo
        got[off:off + hi - lo] = st[lo:hi]
        total += res.fail_count
    assert np.array_equal(got, full) and total == int((full != 0).sum()) >= 2


def _tamper_mpt_and_rows(n, seed):
    """a valid witness with the MPT table and the rows that look it up damaged in every way the lookup has to notice"""
    rng = np.random.default_rng(seed)
    cols, flags, mpt = synth_state_witness(n, seed=seed)
    m = mpt.shape[0]
    for c in range(12):  # every MPT column, hashed (0..4) or only compared (5..11), both halves of the cell
        r = int(rng.integers(0, m))
        mpt[r, c, int(rng.integers(0, 4))] ^= np.uint64(1 << int(rng.integers(0, 60)))
    tags = cols[2, :, 0]
    for t in (4, 6):  # Storage / Account rows: value, initial value, root, storage key, address
        rows = np.nonzero(tags == t)[0]
        for c in (50, 52, 54, 55, 6, 4):
            if len(rows):
                cols[c, int(rng.choice(rows)), 0] ^= np.uint64(1 << int(rng.integers(0, 30)))
    return cols, flags, mpt


@pytest.mark.parametrize("n", [64, 65, 126, 127, 1000, 4097])
def test_mpt_lookup_and_ragged_sizes_match_oracle(n):
    """Row counts around the 63-rows-per-wavefront tiling; damaged MPT rows / Storage / Account rows: statuses bit-identical to the oracle's."""
    cols, flags, mpt = _tamper_mpt_and_rows(n, seed=40 + n)
    res, status = _run(cols, flags, mpt)
    exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
    assert status.tolist() == exp
    assert res.fail_count == sum(1 for c in exp if c)


def test_lane_quad_form_matches_oracle():
    """ZK_STATE_DMA=0 selects the all-register lane-quad kernel (the comparison point of the LDS-ring kernel): same statuses."""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import state_oracle, wire\n"
        "from zkevm_specs_amd import engine\n"
        "from tests.test_state_gpu import _tamper_mpt_and_rows\n"
        "for n in (65, 4097):\n"
        "    cols, flags, mpt = _tamper_mpt_and_rows(n, seed=7 + n)\n"
        "    with engine.open_state(cols, flags, mpt) as s:\n"
        "        s.run(); st = s.read_status()\n"
        "    exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))\n"
        "    assert st.tolist() == exp and any(exp), n\n"
        "print('quad ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZK_STATE_DMA="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b"quad ok" in p.stdout, p.stderr.decode()[-2000:]
