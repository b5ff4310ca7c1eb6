"""RW table -> State-circuit operations (the sort half; include/zkevm_hip.h zk_state_ops_from_rw*):
the C-ABI entry against the checker oracle/rw_state_oracle.py — ops, op flags, order and per-row status bit for bit.
CPU suite: through libzkevm_cpu.so (the same per-row functions and compact-key plan, std::stable_sort).  GPU suite: the HIP
path (radix passes on the device) on the same cases, 200 fuzzed tables, and the full 2^18-step block."""
import os
import random

import numpy as np
import pytest

from oracle import rw_state_oracle, wire
from zkevm_specs_amd import oneshot
from zkevm_specs_amd.wire import rows_to_rowmajor

M256 = (1 << 256) - 1
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def rand_rw_table(rng, n, wide=0.0, bad=0.0, dup=0.3):
    """n RW rows (14 ints) over every target, with repeated keys, unsorted / repeated rw_counters, CallContext field tags on both
    sides of 24, packed TxLog address cells, and — with probability `wide` per row — cells as wide as the wire allows; `bad`: rows
    the entry rejects (no such target, storage_key hi >= 2^128)."""
    rows, flags = [], []
    addrs = [rng.getrandbits(160) for _ in range(3)]
    keys = [rng.getrandbits(256) for _ in range(4)] + [0, 1, (1 << 128), (1 << 128) - 1]
    for i in range(n):
        if rows and rng.random() < dup:  # an earlier row's key again (the stable order decides)
            c = list(rng.choice(rows))
            if rng.random() < 0.7:
                c[0] = rng.randrange(1, 4 * n)
            c[8] = rng.getrandbits(128)
            rows.append(c)
            flags.append(rng.getrandbits(2))
            continue
        t = rng.choice([2, 3, 4, 5, 6, 7, 7, 8, 8, 8, 9, 9, 9, 10, 11, 1])
        c = [0] * 14
        c[0] = rng.randrange(0, 4 * n)
        c[1] = rng.getrandbits(1)
        c[2] = t
        c[3] = rng.randrange(0, 40)
        if t in (8,):
            c[4] = rng.randrange(900, 1024)
        elif t == 9:
            c[4] = rng.randrange(0, 1 << rng.choice([5, 12, 33]))
        elif t == 7:
            c[4] = rng.choice([1, 2, 5, 17, 20, 23, 24, 25, 26, 300])
        elif t == 10:
            c[4] = (rng.randrange(0, 6) << 48) | (rng.randrange(0, 5) << 32) | rng.randrange(0, 70)
        elif t in (2, 3, 5, 6):
            c[4] = rng.choice(addrs)
        if t in (5, 11):
            c[5] = rng.randrange(1, 5)
        if t in (3, 6):
            k = rng.choice(keys)
            c[6], c[7] = k & ((1 << 128) - 1), k >> 128
        c[8], c[9] = rng.getrandbits(128), rng.getrandbits(rng.choice([0, 128]))
        c[10], c[11] = rng.getrandbits(128), rng.getrandbits(rng.choice([0, 128]))
        c[12], c[13] = rng.getrandbits(128), rng.getrandbits(rng.choice([0, 128]))
        if rng.random() < wide:
            j = rng.choice([0, 3, 4, 5, 6, 7, 8, 12])
            c[j] = rng.choice([rng.getrandbits(256), P - 1, M256, rng.getrandbits(200), (1 << 128) | rng.getrandbits(20)])
            if j == 7 and rng.random() < 0.5:
                c[7] &= (1 << 128) - 1
        if rng.random() < bad:
            if rng.random() < 0.5:
                c[2] = rng.choice([0, 12, 13, 255, 1 << 40, P - 1])
            else:
                c[7] = (1 << 128) | rng.getrandbits(100)
        rows.append(c)
        flags.append(rng.getrandbits(2))
    return rows, flags


def check_against_oracle(rows, flags, device):
    rw = rows_to_rowmajor(rows, 14)
    fl = np.array(flags, dtype=np.uint32)
    res, status, ops, op_flags = oneshot.state_ops_from_rw(rw, fl, device=device)
    e_ops, e_flags, e_status = rw_state_oracle.rw_to_state_ops(rows, flags, strict=False)
    assert status.tolist() == e_status
    assert res.fail_count == sum(1 for s in e_status if s)
    assert ops.shape[1] == len(e_ops)
    got = wire.colmajor_to_rows(ops)
    for j, (g, e) in enumerate(zip(got, e_ops)):
        assert g == e, (j, g, e)
    assert op_flags.tolist() == e_flags
    return len(e_ops)


CASES = [(1, 0.0, 0.0), (2, 0.0, 0.0), (63, 0.0, 0.0), (64, 0.0, 0.0), (65, 0.1, 0.0), (257, 0.0, 0.05), (1000, 0.02, 0.01), (4097, 0.0, 0.0),
         (5000, 0.3, 0.1), (9000, 0.001, 0.0)]


@pytest.mark.parametrize("n,wide,bad", CASES)
def test_cpu_backend_matches_checker(n, wide, bad):
    rng = random.Random(1000 * n + 7)
    rows, flags = rand_rw_table(rng, n, wide, bad)
    check_against_oracle(rows, flags, "cpu")


def test_cpu_backend_without_ranks_matches(monkeypatch):
    """the generic path (no rank compression: every varying bit of the wide fields is a key bit)"""
    monkeypatch.setenv("ZK_REKEY_NO_RANKS", "1")
    rng = random.Random(5)
    rows, flags = rand_rw_table(rng, 700, 0.2, 0.02)
    check_against_oracle(rows, flags, "cpu")


def test_checker_matches_trace_generator():
    """the block generator's RW table: every kept row becomes one op, the order is the State circuit's, and the derived witness is
    one the State circuit accepts (oracle/state_oracle.py over the oracle's own assignment)"""
    from oracle import assign_oracle, state_oracle
    from zkevm_specs_amd.synth_block import synth_block_trace

    w = synth_block_trace(600, seed=3, seg_len=96, n_contracts=2)
    rows = wire.rowmajor_to_rows(w["rw"])
    ops, flags, status = rw_state_oracle.rw_to_state_ops(rows, w["rw_flags"].tolist())
    assert not any(status) and len(ops) > 1000
    keys = [(o[2], o[3], o[4], o[5], o[6], o[0]) for o in ops[1:]]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    st_rows, st_flags, mpt, st = assign_oracle.assign(ops, flags)
    assert not any(st)
    assert not any(state_oracle.verify_rows(st_rows, st_flags, mpt))


def fused_equals_two_steps(rows, flags, device):
    """zk_state_assign_from_rw (ops read straight from the RW rows) == zk_state_ops_from_rw followed by zk_state_assign"""
    from zkevm_specs_amd import engine

    rw = rows_to_rowmajor(rows, 14)
    fl = np.array(flags, dtype=np.uint32)
    _, _, ops, op_flags = oneshot.state_ops_from_rw(rw, fl, device=device)
    r2, st2, rows2, rf2, mpt2 = oneshot.state_assign(ops, op_flags, device=device)
    with engine.open_state_assign_from_rw(rw, fl, device=device) as a:
        assert a.n == ops.shape[1]
        r1 = a.run()
        st1 = a.read_status()
        rows1, rf1, mpt1 = a.read()
    assert (r1.fail_count, r1.first_fail_row, r1.first_fail_code) == (r2.fail_count, r2.first_fail_row, r2.first_fail_code)
    assert st1.tolist() == st2.tolist()
    assert np.array_equal(rows1, rows2) and np.array_equal(rf1, rf2) and np.array_equal(mpt1, mpt2)


def test_cpu_backend_fused_assign():
    rng = random.Random(77)
    rows, flags = rand_rw_table(rng, 1500, 0.0, 0.0)
    fused_equals_two_steps(rows, flags, "cpu")


@pytest.mark.gpu
def test_gpu_fused_assign():
    rng = random.Random(78)
    for n in (1, 70, 1500, 6000):
        rows, flags = rand_rw_table(rng, n, 0.0, 0.0)
        fused_equals_two_steps(rows, flags, None)


@pytest.mark.gpu
@pytest.mark.parametrize("n,wide,bad", CASES)
def test_gpu_matches_checker(n, wide, bad):
    rng = random.Random(1000 * n + 7)
    rows, flags = rand_rw_table(rng, n, wide, bad)
    check_against_oracle(rows, flags, None)


@pytest.mark.gpu
def test_gpu_200_fuzzed_tables():
    rng = random.Random(20260101)
    total = 0
    for t in range(200):
        n = rng.choice([3, 40, 130, 700, 2100, 4500])
        rows, flags = rand_rw_table(rng, n, wide=rng.choice([0.0, 0.0, 0.05, 0.5]), bad=rng.choice([0.0, 0.02]), dup=rng.choice([0.1, 0.6]))
        total += check_against_oracle(rows, flags, None)
    assert total > 100000


@pytest.mark.gpu
def test_gpu_without_ranks(monkeypatch):
    monkeypatch.setenv("ZK_REKEY_NO_RANKS", "1")
    rng = random.Random(6)
    rows, flags = rand_rw_table(rng, 3000, 0.2, 0.02)
    check_against_oracle(rows, flags, None)


@pytest.mark.gpu
def test_gpu_generic_sort_path(monkeypatch):
    """keys wider than 64 bits take the generic path (index-only passes, digits gathered from the key words): forced here for
    tables the fast path would sort, and reached naturally by the un-ranked wide tables of test_gpu_without_ranks"""
    monkeypatch.setenv("ZK_REKEY_NO_FAST", "1")
    rng = random.Random(8)
    for n in (5, 900, 4097, 9000):
        rows, flags = rand_rw_table(rng, n, 0.01, 0.01)
        check_against_oracle(rows, flags, None)


@pytest.mark.gpu
def test_gpu_full_block_2p18():
    """BASELINE configs[4]'s block: the 2^18-step trace's RW table (about 7.6e5 rows) re-keyed and sorted on the device, device
    pointers in and out, against the checker; then straight into the device-side State assignment and the State circuit."""
    import torch

    from zkevm_specs_amd import engine
    from zkevm_specs_amd.synth_block import synth_block_trace

    w = synth_block_trace(1 << 18, seed=5)
    rw, fl = w["rw"], w["rw_flags"]
    n = int(rw.shape[0])
    dev = torch.device("cuda:0")
    d_rw = torch.from_numpy(rw.view(np.int64)).to(dev)
    d_fl = torch.from_numpy(fl.view(np.int32)).to(dev)
    d_ops = torch.empty(48 * (n + 1), dtype=torch.int64, device=dev)
    d_of = torch.empty(n + 1, dtype=torch.int32, device=dev)
    with engine.open_state_ops_from_rw(d_rw, d_fl, d_ops, d_of) as s:
        res = s.run()
        m = s.n_ops
    assert res.ok
    rows = wire.rowmajor_to_rows(rw)
    e_ops, e_flags, _ = rw_state_oracle.rw_to_state_ops(rows, fl.tolist())
    assert m == len(e_ops)
    got = d_ops[: 48 * m].view(12, m, 4).cpu().numpy().view(np.uint64)
    exp = np.ascontiguousarray(rows_to_rowmajor(e_ops, 12).transpose(1, 0, 2))
    assert np.array_equal(got, exp)
    assert d_of[:m].cpu().numpy().view(np.uint32).tolist() == e_flags
    # ... and on into the State circuit without leaving the device
    ops_t = d_ops[: 48 * m].view(12, m, 4)
    st_rows = torch.empty((57, m, 4), dtype=torch.int64, device=dev)
    st_flags = torch.empty(m, dtype=torch.int32, device=dev)
    mpt = torch.empty((m, 12, 4), dtype=torch.int64, device=dev)
    with engine.open_state_assign(ops_t, d_of[:m], st_rows, st_flags, mpt) as a:
        assert a.run().ok
        n_mpt = a.n_mpt()
    with engine.open_state(st_rows, st_flags, mpt[:n_mpt]) as st:
        r = st.run()
    assert r.ok and r.rows_evaluated == m
    # the fused session (no op list in memory) produces the same witness
    f_rows = torch.empty(57 * 4 * (n + 1), dtype=torch.int64, device=dev)
    f_flags = torch.empty(n + 1, dtype=torch.int32, device=dev)
    f_mpt = torch.empty(48 * (n + 1), dtype=torch.int64, device=dev)
    with engine.open_state_assign_from_rw(d_rw, d_fl, f_rows, f_flags, f_mpt) as a:
        assert a.n == m and a.run().ok
        assert a.n_mpt() == n_mpt
    assert torch.equal(f_rows[: 57 * 4 * m].view(57, m, 4), st_rows) and torch.equal(f_flags[:m], st_flags)
    assert torch.equal(f_mpt[: 48 * n_mpt].view(n_mpt, 12, 4), mpt[:n_mpt])


def compact_equals_full(rows, flags, device, tamper=0):
    """ZK_OPT_STATE_COMPACT: the 15-cell witness of the assignment is columns 0..7 and 50..56 of the 57-cell one, and the State circuit
    gives every row the status it gives the 57-cell row — also after damage to cells both forms carry"""
    from zkevm_specs_amd import engine

    rw = rows_to_rowmajor(rows, 14)
    fl = np.array(flags, dtype=np.uint32)
    with engine.open_state_assign_from_rw(rw, fl, device=device) as a:
        assert a.run().ok
        full, rf, mpt = a.read()
    with engine.open_state_assign_from_rw(rw, fl, device=device, compact=True) as a:
        assert a.run().ok
        comp, rf_c, mpt_c = a.read()
    assert comp.shape == (15, full.shape[1], 4) and np.array_equal(comp, np.concatenate([full[:8], full[50:]])) and np.array_equal(rf, rf_c)
    assert np.array_equal(mpt, mpt_c)
    rng = random.Random(len(rows) + tamper)
    decomposed = set()  # rows whose address / storage-key cells were damaged: the cells the limb / byte columns decompose
    for _ in range(tamper):
        c, i = rng.randrange(15), rng.randrange(full.shape[1])
        v = rng.choice([np.uint64(1), np.uint64(1) << np.uint64(40)])
        w = rng.randrange(4)  # (words 1..3: also values the derived limbs cannot express — address >= 2^160, key halves >= 2^128)
        comp[c, i, w] ^= v
        full[c if c < 8 else c + 42, i, w] ^= v
        if c in (4, 6, 7):
            decomposed.add(i)
    # the 57-cell form of the damaged witness: its limb / byte cells are the ones assigned from the UNDAMAGED address and key, so rows whose
    # address / key cells were hit fail a recomposition check there (sites 5 / 7) exactly where the compact form reports them
    with engine.open_state(full, rf, mpt, device=device) as s:
        r_full = s.run()
        st_full = s.read_status()
    with engine.open_state(comp, rf, mpt, device=device, compact=True) as s:
        r_comp = s.run()
        st_comp = s.read_status()
    bad = [j for j in range(len(st_full)) if st_full[j] != st_comp[j]]
    # only damage to the address / key cells may be reported differently, and only at the damaged row and its two neighbours (the
    # ordering checks compare a row's packed key with the previous row's, the last-access check looks at the next row's): the 57-cell
    # row's stored limbs / bytes still spell the UNDAMAGED address / key — it fails 0.1 / 0.2 (sites 5 / 7) and packs the old bytes —,
    # the compact row derives them from the damaged cell.  Damage to any other cell is reported identically.
    n_rows = len(st_full)
    for j in bad:
        assert any(((j + d) % n_rows) in decomposed for d in (-1, 0, 1)), (j, hex(st_full[j]), hex(st_comp[j]))
    for j in decomposed:  # (the compact row with another address / key may well be a row the State circuit accepts: it is another witness)
        assert st_full[j] != 0, (j, hex(st_full[j]), hex(st_comp[j]))
    if not tamper:
        assert not bad and r_full.fail_count == r_comp.fail_count  # (the fuzz tables are no valid State witnesses: equal verdicts, not clean ones)
    return len(bad), int(r_full.fail_count), int(r_comp.fail_count)


def _valid_block_rw(n_steps):
    from zkevm_specs_amd.synth_block import synth_block_trace

    w = synth_block_trace(n_steps, seed=4, seg_len=96, n_contracts=2)
    return wire.rowmajor_to_rows(w["rw"]), w["rw_flags"].tolist()


def test_cpu_backend_compact_state_rows():
    rng = random.Random(91)
    rows, flags = rand_rw_table(rng, 1200, 0.0, 0.0)
    compact_equals_full(rows, flags, "cpu")
    rows, flags = _valid_block_rw(500)  # a consistent trace: the derived witness satisfies the State circuit in both forms
    _, f_full, f_comp = compact_equals_full(rows, flags, "cpu")
    assert f_full == 0 and f_comp == 0
    _, f_full, f_comp = compact_equals_full(rows, flags, "cpu", tamper=60)
    assert f_full >= 20 and f_comp >= 20


@pytest.mark.gpu
def test_gpu_compact_state_rows():
    rng = random.Random(92)
    for n in (3, 200, 5000):
        rows, flags = rand_rw_table(rng, n, 0.0, 0.0)
        compact_equals_full(rows, flags, None)
        compact_equals_full(rows, flags, None, tamper=max(3, n // 25))
    rows, flags = _valid_block_rw(3000)
    _, f_full, f_comp = compact_equals_full(rows, flags, None)
    assert f_full == 0 and f_comp == 0
    _, f_full, f_comp = compact_equals_full(rows, flags, None, tamper=300)
    assert f_full >= 100 and f_comp >= 100
