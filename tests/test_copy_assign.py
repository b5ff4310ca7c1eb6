"""Copy-circuit witness assignment (`CopyCircuit.copy`, evm_circuit/typing.py:1010-1151).  CPU: the
restatement oracle/copy_assign_oracle.py against what the unmodified reference produced for every `copy()` call of its own
tests (rows, RW rows, copy-table row), the device functions' logic (hostsim) against the same, and a synthetic event mix
assigned and then checked by the Copy circuit.  GPU (marked): zk_copy_assign / zk_copy_assign_open against all of it, and the
chain events -> rows + RW rows (device) -> Copy circuit (device) at >= 2^16 rows, valid and tampered, every row vs the oracle."""
import ctypes
import os
import random

import numpy as np
import pytest

from oracle import copy_assign_oracle as CA, copy_oracle as co, wire
from zkevm_specs_amd.synth import synth_copy_events

vp = lambda x: None if x is None else ctypes.c_void_p(np.ascontiguousarray(x).ctypes.data)  # noqa: E731


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "copy_assign_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), {f: g[f"{k}_{f}"] for f in ("event", "flags", "data", "r", "rows", "row_flags", "rw", "rw_flags", "table")}


def _oracle(events, flags, data, offsets, r):
    rows, rf, table, rw, rwf = CA.assign(wire.rowmajor_to_rows(events), [int(f) for f in flags], data, offsets, r)
    return rows, rf, table, rw, rwf


def _hostsim(lib, events, flags, data, offsets, r):
    events, flags = np.ascontiguousarray(events), np.ascontiguousarray(flags, dtype=np.uint32)
    data, offsets = np.ascontiguousarray(data, dtype=np.uint16), np.ascontiguousarray(offsets, dtype=np.uint64)
    ev = wire.rowmajor_to_rows(events)
    n_rows = sum(2 * e[9] for e in ev)
    n_table = sum(1 for e in ev if e[9])
    n_rw = sum((CA.n_real(e) if e[2] == 2 else 0) + (e[9] if e[5] in (2, 4) else 0) for e in ev)
    rows, rf = np.zeros((20, n_rows, 4), dtype=np.uint64), np.zeros(n_rows, dtype=np.uint32)
    table = np.zeros((n_table, 14, 4), dtype=np.uint64)
    rw, rwf = np.zeros((max(n_rw, 1), 14, 4), dtype=np.uint64), np.zeros(max(n_rw, 1), dtype=np.uint32)
    rc = np.frombuffer(int(r).to_bytes(32, "little"), dtype="<u8").copy()
    assert lib.sim_copy_assign(vp(events), vp(flags), ctypes.c_uint64(len(ev)), vp(data if len(data) else np.zeros(1, np.uint16)), vp(offsets), vp(rc),
                               vp(rows), vp(rf), vp(table), vp(rw), vp(rwf), ctypes.c_uint64(n_rows)) == 0
    return rows, rf, table, rw[:n_rw], rwf[:n_rw]


def _same(got, exp):
    rows, rf, table, rw, rwf = got
    e_rows, e_rf, e_table, e_rw, e_rwf = exp
    assert wire.colmajor_to_rows(rows) == e_rows and rf.tolist() == e_rf
    assert wire.rowmajor_to_rows(table) == e_table
    assert wire.rowmajor_to_rows(rw) == e_rw and rwf.tolist() == e_rwf


def _batch(golden_dir):
    """all golden events as ONE batch (what a block's copy circuit is), under the first event's randomness"""
    evs, fls, data, offs = [], [], [], [0]
    r = None
    for _, c in _cases(golden_dir):
        rr = wire.cells_to_ints(c["r"])[0]
        if r is None:
            r = rr
        evs.append(c["event"][0])
        fls.append(int(c["flags"][0]))
        data.extend(c["data"].tolist())
        offs.append(len(data))
    return np.stack(evs), np.array(fls, dtype=np.uint32), np.array(data, dtype=np.uint16), np.array(offs, dtype=np.uint64), r


def test_oracle_and_kernel_logic_match_the_reference(golden_dir, hostsim):
    n = n_rlc = n_pad = 0
    for name, c in _cases(golden_dir):
        r = wire.cells_to_ints(c["r"])[0]
        offsets = np.array([0, len(c["data"])], dtype=np.uint64)
        exp = (wire.colmajor_to_rows(c["rows"]), c["row_flags"].tolist(), wire.rowmajor_to_rows(c["table"]), wire.rowmajor_to_rows(c["rw"]),
               c["rw_flags"].tolist())
        assert _oracle(c["event"], c["flags"], c["data"], offsets, r) == exp, name
        _same(_hostsim(hostsim, c["event"], c["flags"], c["data"], offsets, r), exp)
        ev = wire.rowmajor_to_rows(c["event"])[0]
        n += 1
        n_rlc += ev[5] == 5
        n_pad += ev[9] > CA.n_real(ev)
    assert n >= 150 and n_rlc >= 15 and n_pad >= 10
    evs, fls, data, offs, r = _batch(golden_dir)
    assert evs.shape[0] >= 50
    _same(_hostsim(hostsim, evs, fls, data, offs, r), _oracle(evs, fls, data, offs, r))


def _copy_status(rows, rf, w, rw, rwf):
    T = co.CopyTables(rw, rwf, wire.rowmajor_to_rows(w["bytecode"]), wire.rowmajor_to_rows(w["tx"]), w["tx_flags"])
    return co.verify_rows(rows, rf, T, w["r"])


def test_synthetic_events_assign_to_a_valid_copy_witness(hostsim):
    w = synth_copy_events(6000, seed=6)
    exp = _oracle(w["events"], w["flags"], w["data"], w["offsets"], w["r"])
    _same(_hostsim(hostsim, w["events"], w["flags"], w["data"], w["offsets"], w["r"]), exp)
    rows, rf, table, rw, rwf = exp
    assert len(rows) == w["n_rows"] >= 6000 and len({tuple(t) for t in table}) == len(table)
    st = _copy_status(rows, rf, w, rw, rwf)
    assert not any(st), [(i, s >> 24, s & 0xFFFFFF) for i, s in enumerate(st) if s][:5]
    kinds = {(wire.rowmajor_to_rows(w["events"])[i][2], wire.rowmajor_to_rows(w["events"])[i][5]) for i in range(w["events"].shape[0])}
    assert {(1, 2), (3, 2), (2, 2), (2, 4), (2, 5), (2, 1)} <= kinds


# ---- GPU --------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_matches_the_reference_rows(golden_dir):
    from zkevm_specs_amd import engine, oneshot

    for name, c in _cases(golden_dir):
        r = wire.cells_to_ints(c["r"])[0]
        offsets = np.array([0, len(c["data"])], dtype=np.uint64)
        res, rows, rf, table, rw, rwf = oneshot.copy_assign(c["event"], c["flags"], c["data"], offsets, r)
        assert res.ok
        _same((rows, rf, table, rw, rwf), (wire.colmajor_to_rows(c["rows"]), c["row_flags"].tolist(), wire.rowmajor_to_rows(c["table"]),
                                            wire.rowmajor_to_rows(c["rw"]), c["rw_flags"].tolist()))
    evs, fls, data, offs, r = _batch(golden_dir)
    with engine.open_copy_assign(evs, fls, data, offs, r) as s:
        assert s.run().ok
        _same(s.read(), _oracle(evs, fls, data, offs, r))


@pytest.mark.gpu
def test_events_to_verified_copy_circuit_on_device_valid_and_tampered():
    """>= 2^16 circuit rows: events -> rows + RW rows in HBM (zk_copy_assign_open) -> Copy circuit over the same buffers
    (zk_copy_open); then ~300 tampered cells: every row's status against the oracle."""
    import torch

    from zkevm_specs_amd import engine

    w = synth_copy_events(1 << 16, seed=8)
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x.view(np.int16)).cuda()  # noqa: E731
    d = {k: dev(w[k]) for k in ("events", "flags", "data", "offsets", "bytecode", "tx", "tx_flags")}
    n_rows, n_table, n_rw = engine.copy_assign_sizes(w["events"], w["flags"], w["data"], w["offsets"])
    assert n_rows == w["n_rows"] >= 1 << 16
    rows = torch.empty((20, n_rows, 4), dtype=torch.int64, device="cuda")
    rf = torch.empty(n_rows, dtype=torch.int32, device="cuda")
    table = torch.empty((n_table, 14, 4), dtype=torch.int64, device="cuda")
    rw = torch.empty((n_rw, 14, 4), dtype=torch.int64, device="cuda")
    rwf = torch.empty(n_rw, dtype=torch.int32, device="cuda")
    with engine.open_copy_assign(d["events"], d["flags"], d["data"], d["offsets"], w["r"], rows, rf, table, rw, rwf) as s:
        assert s.run().ok
    torch.cuda.synchronize()
    exp = _oracle(w["events"], w["flags"], w["data"], w["offsets"], w["r"])
    got = (rows.cpu().numpy().view(np.uint64), rf.cpu().numpy().view(np.uint32), table.cpu().numpy().view(np.uint64),
           rw.cpu().numpy().view(np.uint64), rwf.cpu().numpy().view(np.uint32))
    _same(got, exp)
    with engine.open_copy(rows, rf, w["r"], rw, rwf, d["bytecode"], d["tx"], d["tx_flags"]) as s:
        res = s.run()
        assert res.ok and res.rows_evaluated == n_rows
    # tamper: circuit cells, RW cells, type bits
    h_rows, h_rf, h_rw, h_rwf = got[0].copy(), got[1].copy(), got[3].copy(), got[4].copy()
    rng = random.Random(5)
    for _ in range(300):
        what = rng.randrange(10)
        if what < 7:
            c, i = rng.randrange(20), rng.randrange(n_rows)
            old = int.from_bytes(h_rows[c, i].tobytes(), "little")
            new = rng.choice([old + 1, old - 1, 0, 1, 2, old ^ 1, rng.randrange(wire.P), 1 << 40]) % wire.P
            h_rows[c, i] = np.frombuffer(new.to_bytes(32, "little"), dtype="<u8")
        elif what < 9:
            i, c = rng.randrange(n_rw), rng.randrange(10)
            old = int.from_bytes(h_rw[i, c].tobytes(), "little")
            h_rw[i, c] = np.frombuffer(((old + 1) % wire.P).to_bytes(32, "little"), dtype="<u8")
        else:
            h_rf[rng.randrange(n_rows)] ^= np.uint32(1)
    # the tampered rows include the last row and both sides of a wavefront boundary of the kernel (a wavefront evaluates 62 rows and
    # takes rows i + 1 / i + 2 from lanes + 1 / + 2; the table's last rows take them from rows 0 / 1)
    for i, c in ((n_rows - 1, 6), (n_rows - 2, 13), (0, 3), (1, 6), (61, 6), (62, 13), (63, 9), (62 * 40 - 1, 10), (62 * 40, 6), (62 * 40 + 1, 8)):
        old = int.from_bytes(h_rows[c, i].tobytes(), "little")
        h_rows[c, i] = np.frombuffer(((old + 1) % wire.P).to_bytes(32, "little"), dtype="<u8")
    e_st = _copy_status(wire.colmajor_to_rows(h_rows), h_rf.tolist(), w, wire.rowmajor_to_rows(h_rw), h_rwf.tolist())
    with engine.open_copy(h_rows, h_rf, w["r"], h_rw, h_rwf, w["bytecode"], w["tx"], w["tx_flags"]) as s:
        res = s.run()
        status = s.read_status().tolist()
        # row ranges (zk_set_range: what a rank of a sharded run evaluates), cut at and off the 62-row wavefront period
        for lo, hi in ((0, 62), (61, 125), (62 * 40 - 3, 62 * 40 + 3), (n_rows - 70, n_rows), (12345, 12346), (1000, 30000)):
            s.set_range(lo, hi)
            r2 = s.run()
            part = s.read_status()[lo:hi].tolist()
            assert part == e_st[lo:hi], (lo, hi)
            pf = [j for j in range(lo, hi) if e_st[j]]
            assert r2.rows_evaluated == hi - lo and r2.fail_count == len(pf), (lo, hi)
            if pf:
                assert r2.first_fail_row == pf[0] and r2.first_fail_code == e_st[pf[0]], (lo, hi)
    assert status == e_st
    fails = [j for j, c in enumerate(e_st) if c]
    assert res.fail_count == len(fails) >= 200 and res.first_fail_row == fails[0] and res.first_fail_code == e_st[fails[0]]
