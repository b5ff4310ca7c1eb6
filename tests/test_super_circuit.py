"""Super circuit (BASELINE config 5): the EVM, State, Bytecode and Tx kernels over one witness set.
CPU: every part of the synthetic super witness satisfies its circuit according to the oracles; the contracts'
keccak table ties the EVM trace, the Bytecode rows and the table builder together.  GPU (marked): the same
through the C ABI, tally = sum over the circuits, tampering one cell per circuit is found where it was put."""
import numpy as np
import pytest

from oracle import assign_oracle, bytecode_assign_oracle, keccak_table, row_oracles, sign_oracle, state_oracle, wire
from tests.evm_cases import oracle_status
from zkevm_specs_amd.super_circuit import CIRCUITS, synth_super



def _block_state_ops(p):
    """the block's State ops as the CHECKER derives them from the block's RW table (oracle/rw_state_oracle.py): what SuperCircuit
    computes on the device (zk_state_assign_from_rw) — lists of ints, ready for assign_oracle.assign"""
    from oracle import rw_state_oracle

    ops, flags, status = rw_state_oracle.rw_to_state_ops(wire.rowmajor_to_rows(p["evm"]["rw"]), p["evm"]["rw_flags"].tolist())
    assert not any(status)
    return ops, flags


def _oracle_keccak(codes, r):
    return keccak_table.table_rows(codes, r, keccak_table.MODE_CIRCUIT)[0]


def test_super_witness_parts_are_valid_and_consistent():
    p = synth_super(13, seed=7, keccak_rows_of=_oracle_keccak)
    assert sum(p["rows"].values()) == (1 << 13) - 1 and set(p["rows"]) == set(CIRCUITS)
    # EVM trace (its bytecode table carries the keccak digests of the contracts)
    assert not any(oracle_status(dict(p["evm"])))
    bc_rows, keccak, r = p["bytecode"]
    hashes = {tuple(wire.cells_to_ints(keccak[i, 3:5])) for i in range(keccak.shape[0])}
    evm_hashes = {tuple(row[0:2]) for row in wire.rowmajor_to_rows(p["evm"]["bytecode"])}
    assert evm_hashes == hashes
    # Bytecode circuit over the same contracts, looking up the same keccak table
    st = row_oracles.bytecode_verify_rows(wire.colmajor_to_rows(bc_rows), wire.rowmajor_to_rows(keccak), r)
    assert not any(st)
    # ... and the rows the device assignment produces from the EVM circuit's own bytecode table satisfy it too
    ub_rows, ub_off, ub_len, k = p["bytecode_unrolled"]
    assigned = bytecode_assign_oracle.assign(k, wire.rowmajor_to_rows(ub_rows), ub_off, ub_len, r)
    assert not any(row_oracles.bytecode_verify_rows(assigned, wire.rowmajor_to_rows(keccak), r))
    # State rows: assigned from the op list, then checked
    ops, flags = p["state_ops"]
    rows, rflags, mpt, status = assign_oracle.assign(wire.colmajor_to_rows(ops), flags.tolist())
    assert not any(status) and not any(state_oracle.verify_rows(rows, rflags, mpt))
    # Tx units
    tx, r_tx = p["tx"]
    st = sign_oracle.verify_units(tx["bytes"], tx["cells"], tx["meta"], wire.rowmajor_to_rows(tx["keccak"]), r_tx, 0,
                                  wire.rowmajor_to_rows(tx["tx_rows"]), tx["tx_flags"])
    assert not any(st)


@pytest.mark.gpu
def test_super_circuit_on_device_and_tamper_localisation():
    import torch

    from zkevm_specs_amd.super_circuit import SuperCircuit

    p = synth_super(16, seed=3)  # keccak table of the contracts built on the GPU
    assert sum(p["rows"].values()) == (1 << 16) - 1
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
    with SuperCircuit(p, to_device=dev) as sc:
        assert sc.rows == p["rows"]
        sc.launch()
        results, total, first = sc.collect()
        assert total == 0 and first is None and all(r.ok for r in results.values())
    # the device-built keccak table equals the oracle's
    assert np.array_equal(p["bytecode"][1], _oracle_keccak(p["codes"], p["bytecode"][2]))
    # one tampered cell per circuit (host copies -> staged by the library)
    p["evm"]["steps"][100, 7, 0] += np.uint64(1)                 # program counter of step 100
    p["state_ops"][0][7, 50, 0] ^= np.uint64(1)                  # value.lo of op 50
    p["bytecode_unrolled"][0][20, 5, 0] ^= np.uint64(1)          # byte value of unrolled bytecode row 20 (Bytecode circuit's copy)
    p["tx"][0]["cells"][0, 5, 0] ^= np.uint64(1)                 # address of tx 5
    with SuperCircuit(p) as sc:
        sc.launch()
        results, total, first = sc.collect()
    assert not results["evm"].ok and results["evm"].first_fail_row in (99, 100)
    assert not results["state"].ok and results["state"].first_fail_row in (50, 51)
    assert not results["bytecode"].ok  # the re-assigned value_rlc no longer matches the keccak table at the end of that contract
    assert not results["tx"].ok and results["tx"].first_fail_row == 5
    assert total == sum(r.fail_count for r in results.values()) >= 4 and first[0] == "evm"


def test_block_witness_is_one_consistent_witness():
    """config 5 as stated: ONE block.  The State rows ARE the EVM trace's RW table (re-keyed, re-sorted) and satisfy
    the State circuit; the trace satisfies the EVM circuit; the Bytecode rows are the executed contracts; the Copy circuit's rows
    are the copy events of the trace's own SHA3 / CODECOPY steps (execution/sha3.py:20-34, codecopy.py:26) and look up the block's
    own RW / bytecode tables; the copy table those steps look up is the table of the same events; the keccak table they look up is
    built from the SHA3 inputs; the Exp circuit's rows and the exp table are the trace's EXP steps (exp.py:31-33)."""
    from oracle import copy_assign_oracle, copy_oracle
    from zkevm_specs_amd.super_circuit import BLOCK_CIRCUITS, synth_super_block
    from zkevm_specs_amd.wire import rows_to_rowmajor

    p = synth_super_block(15, seed=3, keccak_rows_of=_oracle_keccak)
    assert set(p["rows"]) == set(BLOCK_CIRCUITS) and 0.7 * (1 << 15) < sum(p["rows"].values()) < 1.7 * (1 << 15)
    m = p["meta"]
    assert m["copy_exp_rows_from_trace"] and m["copy_events"] >= 20 and m["sha3_steps"] >= 1 and p["copy_events"]["from_trace"]
    ce = p["copy_events"]
    c_rows, c_rf, c_table, c_rw, c_rwf = copy_assign_oracle.assign(wire.rowmajor_to_rows(ce["events"]), ce["flags"].tolist(), ce["data"], ce["offsets"], ce["r"])
    evm = dict(p["evm"])
    evm["copy"] = rows_to_rowmajor(c_table, 14)  # what SuperCircuit hands the EVM session (zk_copy_assign's table output)
    assert not any(oracle_status(evm))
    states = set(int(x) for x in evm["steps"][:, 0, 0])
    from zkevm_specs_amd import evm_tables as T
    assert int(T.ExecutionState.SHA3) in states and int(T.ExecutionState.CODECOPY) in states
    # State circuit over the trace's own RW rows
    assert p["state_ops"] is None  # derived on the device by SuperCircuit; here by the checker
    ops, flags = _block_state_ops(p)
    assert len(ops) == p["rows"]["state"] == evm["rw"].shape[0] + 1 - int((evm["rw"][:, 2, 0] == 7).astype(int) @ (evm["rw"][:, 4, 0] > 24).astype(int))
    rows, rflags, mpt, status = assign_oracle.assign(ops, flags)
    assert not any(status) and not any(state_oracle.verify_rows(rows, rflags, mpt))
    by_rwc = {r[0]: r for r in rows}
    for c in wire.rowmajor_to_rows(evm["rw"][:: 37]):
        if c[2] == 7 and c[4] > 24:
            continue
        s_row = by_rwc[c[0]]
        assert s_row[1] == c[1] and (s_row[50], s_row[51]) == (c[8], c[9])
    bc_rows, keccak, r = p["bytecode"]
    assert not any(row_oracles.bytecode_verify_rows(wire.colmajor_to_rows(bc_rows), wire.rowmajor_to_rows(keccak), r))
    # Copy circuit: its rows against the BLOCK's tables; the RW rows the events imply are rows of the block's RW table
    blk_rw = wire.rowmajor_to_rows(evm["rw"])
    T_ = copy_oracle.CopyTables(blk_rw, evm["rw_flags"], wire.rowmajor_to_rows(evm["bytecode"]), wire.rowmajor_to_rows(evm["tx"]), evm["tx_flags"])
    assert len(c_rows) == p["rows"]["copy"] and not any(copy_oracle.verify_rows(c_rows, c_rf, T_, ce["r"]))
    in_block = {tuple(x) for x in blk_rw}
    assert c_rw and all(tuple(x) in in_block for x in c_rw)
    # Exp circuit: the EXP steps' traces (another seed so that the small block has some)
    p2 = synth_super_block(13, seed=7, keccak_rows_of=_oracle_keccak)
    assert p2["rows"]["exp"] >= 10 and not any(row_oracles.exp_verify_rows(wire.colmajor_to_rows(p2["exp_rows"])))
    ce2 = p2["copy_events"]
    evm2 = dict(p2["evm"])
    evm2["copy"] = rows_to_rowmajor(copy_assign_oracle.assign(wire.rowmajor_to_rows(ce2["events"]), ce2["flags"].tolist(), ce2["data"], ce2["offsets"], ce2["r"])[2], 14)
    assert not any(oracle_status(evm2)) and int(T.ExecutionState.EXP) in set(int(x) for x in evm2["steps"][:, 0, 0])


@pytest.mark.gpu
def test_block_super_circuit_on_device():
    """the block through the C ABI on the device, then tampering of SHARED data: every circuit that sees the cell must notice"""
    import torch

    from zkevm_specs_amd.super_circuit import BLOCK_CIRCUITS, SuperCircuit, synth_super_block

    p = synth_super_block(16, seed=3)
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
    with SuperCircuit(p, to_device=dev) as sc:
        assert set(sc.rows) == set(BLOCK_CIRCUITS) and sc.rows == p["rows"]
        sc.launch()
        results, total, first = sc.collect()
        assert total == 0 and first is None and all(r.ok for r in results.values()), {k: (r.fail_count, r.first_fail_row, r.first_fail_code) for k, r in results.items()}
    assert p["rows"]["copy"] >= 1000 and p["rows"]["exp"] >= 16
    rw = p["evm"]["rw"]

    def run(pp):
        with SuperCircuit(pp) as sc:
            sc.launch()
            res = sc.collect()[0]
        # the same block with no State witness in HBM (state_fused: rows evaluated from the RW table where they are computed; two resident
        # passes): every circuit's verdict — count, first failing row, code — is the 57-cell form's
        with SuperCircuit(pp, to_device=dev, state_fused=True) as sf:
            assert sf.state_fused and sf.rows == sc.rows
            for _ in range(2):
                sf.launch()
                rf = sf.collect()[0]
                assert {k: (r.fail_count, r.first_fail_row, r.first_fail_code) for k, r in rf.items()} == \
                       {k: (r.fail_count, r.first_fail_row, r.first_fail_code) for k, r in res.items()}
        return res

    # (1) a Stack read's value: the EVM circuit (the step that looks the row up) and the State circuit (read consistency)
    i = next(j for j in range(2000, rw.shape[0]) if int(rw[j, 2, 0]) == 8 and int(rw[j, 1, 0]) == 0)
    rw[i, 8, 0] ^= np.uint64(1)
    res = run(p)
    assert not res["evm"].ok and not res["state"].ok and res["bytecode"].ok and res["copy"].ok and res["exp"].ok
    rw[i, 8, 0] ^= np.uint64(1)
    # (2) a Memory byte a SHA3 / CODECOPY step's copy event moves: the Copy circuit (its RW lookup) and the State circuit; the EVM
    # circuit does not look at the byte itself (only at the copy table)
    ce = p["copy_events"]
    ev = ce["events"]
    k = next(j for j in range(ev.shape[0]) if int(ev[j, 9, 0]) >= 4 and int(ev[j, 2, 0]) == 2)  # a SHA3 step's event: its Memory rows are READS
    first_rwc = int(ev[k, 11, 0])
    j = int(np.searchsorted(rw[:, 0, 0].astype(np.int64), first_rwc + 1))
    from zkevm_specs_amd import evm_tables as ET

    assert int(rw[j, 0, 0]) == first_rwc + 1 and int(rw[j, 2, 0]) == int(ET.Target.Memory)  # a Memory row of that event
    rw[j, 8, 0] ^= np.uint64(1)
    res = run(p)
    assert not res["copy"].ok and not res["state"].ok and res["bytecode"].ok and res["exp"].ok
    rw[j, 8, 0] ^= np.uint64(1)
    # (3) the exp table row an EXP step looks up: the EVM circuit only (the Exp circuit's own rows are untouched)
    p["evm"]["exp"][0, 9, 0] ^= np.uint64(1)
    res = run(p)
    assert not res["evm"].ok and res["exp"].ok and res["state"].ok and res["copy"].ok
    p["evm"]["exp"][0, 9, 0] ^= np.uint64(1)
    assert all(r.ok for r in run(p).values())


@pytest.mark.gpu
def test_full_size_block_every_row_of_every_circuit_vs_oracles():
    """VERDICT r3 #2b: BASELINE config 5 at its full size (the 2^18-step trace inside a ~1.17 M-row block), per-row / per-pair
    statuses of ALL SIX circuits read back from the device and compared with their oracles — the valid block (every status 0,
    tally 0) and the block with 600 tampered cells of SHARED data (RW rows: EVM + State + Copy; bytecode table: EVM + Bytecode +
    Copy; copy events' bytes: Copy + EVM's copy table; step cells, exp rows / table, keccak rows, Tx cells)."""
    import random

    import torch

    from oracle import copy_assign_oracle, copy_oracle
    from zkevm_specs_amd import evm_tables as ET
    from zkevm_specs_amd.super_circuit import BLOCK_CIRCUITS, SuperCircuit, synth_super_block
    from zkevm_specs_amd.wire import rows_to_rowmajor

    p = synth_super_block(20, seed=5)
    assert sum(p["rows"].values()) > 1_100_000 and p["rows"]["evm"] == (1 << 18) - 1
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
    with SuperCircuit(p, to_device=dev) as sc:
        assert set(sc.rows) == set(BLOCK_CIRCUITS)
        sc.launch()
        results, total, first = sc.collect()
        assert total == 0 and first is None
        for k, s in sc.sessions.items():
            assert not s.read_status().any(), k
    # ---- tamper shared data ------------------------------------------------------------------------------------------
    rng = random.Random(2024)
    evm = p["evm"]
    rw, bt = evm["rw"], evm["bytecode"]
    tag = rw[:, 2, 0].astype(np.int64)
    plain = np.nonzero((tag == int(ET.Target.Stack)) | (tag == int(ET.Target.Memory)))[0]
    for j in rng.sample(plain.tolist(), 260):          # a Stack / Memory row's value: the step that looks it up, the State circuit, the Copy circuit
        rw[j, 8, 0] ^= np.uint64(1 << rng.randrange(8))
    ub = p["bytecode_unrolled"][0]
    assert np.array_equal(ub, bt)
    for j in rng.sample(range(bt.shape[0]), 120):      # a byte of a contract: opcode / push-data lookups, the Bytecode circuit, CODECOPY's source
        if int(bt[j, 2, 0]) == 2:
            bt[j, 5, 0] ^= np.uint64(1 << rng.randrange(8))
            ub[j, 5, 0] = bt[j, 5, 0]
    for _ in range(120):                               # step cells: the EVM circuit alone
        evm["steps"][rng.randrange(evm["steps"].shape[0]), rng.choice([1, 7, 8, 9, 10, 11]), 0] ^= np.uint64(1)
    ce = p["copy_events"]
    for j in rng.sample(range(ce["data"].shape[0]), 40):  # a copied byte: the Copy circuit's rows AND the copy table the SHA3 / CODECOPY steps look up
        ce["data"][j] ^= np.uint16(1)
    for _ in range(20):
        evm["exp"][rng.randrange(evm["exp"].shape[0]), rng.randrange(11), 0] ^= np.uint64(1)
        p["exp_rows"][rng.randrange(21), rng.randrange(p["exp_rows"].shape[1]), 0] ^= np.uint64(1)
    for _ in range(10):
        evm["keccak"][rng.randrange(evm["keccak"].shape[0]), rng.randrange(5), 0] ^= np.uint64(1)
    tx, r_tx = p["tx"]
    for _ in range(30):
        tx["cells"][rng.randrange(tx["cells"].shape[0]), rng.randrange(tx["cells"].shape[1]), 0] ^= np.uint64(1)
    with SuperCircuit(p) as sc:
        sc.launch()
        results, total, first = sc.collect()
        got = {k: s.read_status().tolist() for k, s in sc.sessions.items()}
    # ---- the oracles over the same block -----------------------------------------------------------------------------
    c_rows, c_rf, c_table, c_rw, c_rwf = copy_assign_oracle.assign(wire.rowmajor_to_rows(ce["events"]), ce["flags"].tolist(), ce["data"], ce["offsets"], ce["r"])
    exp = {"evm": oracle_status(dict(evm, copy=rows_to_rowmajor(c_table, 14)))}
    ops, flags = _block_state_ops(p)
    rows, rflags, mpt, a_status = assign_oracle.assign(ops, flags)
    assert not any(a_status)
    exp["state"] = state_oracle.verify_rows(rows, rflags, mpt)
    ub_rows, ub_off, ub_len, k = p["bytecode_unrolled"]
    _, keccak, r = p["bytecode"]
    exp["bytecode"] = row_oracles.bytecode_verify_rows(bytecode_assign_oracle.assign(k, wire.rowmajor_to_rows(ub_rows), ub_off, ub_len, r),
                                                       wire.rowmajor_to_rows(keccak), r)
    T_ = copy_oracle.CopyTables(wire.rowmajor_to_rows(rw), evm["rw_flags"], wire.rowmajor_to_rows(bt), wire.rowmajor_to_rows(evm["tx"]), evm["tx_flags"])
    exp["copy"] = copy_oracle.verify_rows(c_rows, c_rf, T_, ce["r"])
    exp["tx"] = sign_oracle.verify_units(tx["bytes"], tx["cells"], tx["meta"], wire.rowmajor_to_rows(tx["keccak"]), r_tx, 0,
                                         wire.rowmajor_to_rows(tx["tx_rows"]), tx["tx_flags"])
    exp["exp"] = row_oracles.exp_verify_rows(wire.colmajor_to_rows(p["exp_rows"]))
    n_fail = {}
    for k in BLOCK_CIRCUITS:
        assert len(got[k]) == len(exp[k]), k
        bad = [i for i, (a, b) in enumerate(zip(got[k], exp[k])) if a != b]
        assert not bad, (k, len(bad), bad[:5], [(got[k][i], exp[k][i]) for i in bad[:5]])
        n_fail[k] = sum(1 for e in exp[k] if e)
        assert results[k].fail_count == n_fail[k], k
    assert all(n_fail[k] >= 5 for k in BLOCK_CIRCUITS), n_fail
    assert total == sum(n_fail.values())


@pytest.mark.gpu
def test_block_one_shot_from_raw_inputs():
    """block.BlockVerifier: keccak table, Bytecode / Copy / State assignments (State from the RW table) and the six circuits derived and
    evaluated from the block's raw device-resident inputs in one call — clean block, then a tampered RW value: the same circuits notice
    as with the resident SuperCircuit"""
    import torch

    from zkevm_specs_amd.block import BlockVerifier, stage_block
    from zkevm_specs_amd.super_circuit import BLOCK_CIRCUITS, synth_super_block

    p = synth_super_block(16, seed=3)
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
    bv = BlockVerifier(0)
    try:
        for _ in range(3):
            results, total = bv.verify(stage_block(p, dev))
            assert total == 0 and set(results) == set(BLOCK_CIRCUITS)
            assert {k: r.rows_evaluated for k, r in results.items()} == p["rows"]
        rw = p["evm"]["rw"]
        i = next(j for j in range(2000, rw.shape[0]) if int(rw[j, 2, 0]) == 8 and int(rw[j, 1, 0]) == 0)
        rw[i, 8, 0] ^= np.uint64(1)  # a Stack read's value: the EVM circuit and the State circuit
        results, total = bv.verify(stage_block(p, dev))
        assert not results["evm"].ok and not results["state"].ok and results["bytecode"].ok and results["copy"].ok and results["exp"].ok and results["tx"].ok
        rw[i, 8, 0] ^= np.uint64(1)
    finally:
        bv.close()


@pytest.mark.gpu
def test_block_one_shot_native_entry():
    """zk_block_verify (the chains on threads inside the library) == block.BlockVerifier (the same chains from Python), full and compact
    State rows, clean and tampered block"""
    import torch

    from zkevm_specs_amd.block import BlockVerifier, stage_block, verify_block_native
    from zkevm_specs_amd.super_circuit import BLOCK_CIRCUITS, synth_super_block

    p = synth_super_block(16, seed=3)
    dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
    bv = BlockVerifier(0)
    try:
        for compact, rows in ((False, False), (False, True), (True, False)):  # State rows fused (default) / 57-cell witness / 15-cell witness
            for _ in range(2):
                results, total, ends = verify_block_native(stage_block(p, dev), 0, compact, rows)
                assert total == 0 and set(results) == set(BLOCK_CIRCUITS) and {k: r.rows_evaluated for k, r in results.items()} == p["rows"]
                assert all(e > 0 for e in ends)
        rw = p["evm"]["rw"]
        i = next(j for j in range(2000, rw.shape[0]) if int(rw[j, 2, 0]) == 8 and int(rw[j, 1, 0]) == 0)
        rw[i, 8, 0] ^= np.uint64(1)
        b = stage_block(p, dev)
        want, _ = bv.verify(b)
        for rows in (False, True):
            got, total, _ = verify_block_native(stage_block(p, dev), 0, False, rows)
            assert total == sum(r.fail_count for r in want.values()) > 0
            for k in BLOCK_CIRCUITS:
                assert (got[k].fail_count, got[k].first_fail_row, got[k].first_fail_code) == (want[k].fail_count, want[k].first_fail_row, want[k].first_fail_code), k
        rw[i, 8, 0] ^= np.uint64(1)
        # a block whose State witness cannot be assigned (an RW row with no Target in its tag cell): an error return with its text, every
        # chain ended (no thread left waiting on another), and the next block verifies again
        from zkevm_specs_amd._lib import EngineError

        old = int(rw[3000, 2, 0])
        rw[3000, 2, 0] = np.uint64(99)
        with pytest.raises(EngineError, match="State witness assignment"):
            verify_block_native(stage_block(p, dev), 0, False)
        rw[3000, 2, 0] = np.uint64(old)
        _, total, _ = verify_block_native(stage_block(p, dev), 0, False)
        assert total == 0
    finally:
        bv.close()
