"""Bodies of the drop-in boundary tests, shared by tests/test_dropin_gpu.py (the HIP library, `-m gpu`) and
tests/test_dropin_cpu.py (the same bodies with `zkevm_specs_amd.oneshot` swapped for oracle-backed stand-ins, so the
host logic — object marshalling, success / exception semantics — is exercised in the GPU-less container).

Part 1: every one-shot C entry INTEGRATION.md's stub binds, on every golden case, status arrays vs the oracle.
Part 2: the Python mirrors of the names the reference's tests import, driven with witness OBJECTS (rebuilt from the
golden wires with zkevm_specs_amd.objects) and compared with what the reference's own drivers did.
"""
import os

import numpy as np
import pytest

from oracle import assign_oracle, bytecode_assign_oracle, codes, copy_oracle, ecdsa_oracle, keccak_table as KT
from oracle import row_oracles as ro, sign_oracle as so, state_oracle, wire
from tests.evm_cases import golden_files, load_cases, oracle_status
from zkevm_specs_amd import errors, objects, oneshot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_tally(res, exp):
    fails = [j for j, c in enumerate(exp) if c]
    assert res.fail_count == len(fails)
    if fails:
        assert res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]
    else:
        assert res.first_fail_row is None


def exc_class(kind):
    return type(errors.exception_for_code(int(kind) << 24))


def expect_outcome(kind, call):
    """`call()` must raise the exception class of `kind` (0 = return normally)"""
    if kind == 0:
        call()
    else:
        with pytest.raises(Exception) as ei:  # noqa: B017, PT011
            call()
        # by class NAME: with the reference loaded in the process the mirror raises the reference's own class objects, of which
        # ConstraintUnsatFailure exists twice (errors._boundary_exception)
        assert type(ei.value).__name__ == exc_class(kind).__name__, (type(ei.value), kind)


# ---------------------------------------------------------------------------------------------------------------------
# golden loaders
# ---------------------------------------------------------------------------------------------------------------------
def state_cases():
    g = np.load(os.path.join(GOLDEN, "state_cases.npz"))
    for i, name in enumerate(g["names"]):
        k = f"c{i:03d}"
        yield str(name), g[k + "_rows"], g[k + "_flags"], g[k + "_mpt"], g[k + "_ref_kind"].tolist()


def bytecode_cases():
    g = np.load(os.path.join(GOLDEN, "bytecode_cases.npz"))
    r = wire.cells_to_ints(g["r"])[0]
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), np.ascontiguousarray(g[k + "_rows"]), np.ascontiguousarray(g[k + "_keccak"]), g[k + "_ref_kind"].tolist(), r


def exp_cases():
    g = np.load(os.path.join(GOLDEN, "exp_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), np.ascontiguousarray(g[k + "_rows"]), g[k + "_ref_kind"].tolist()


COPY_KEYS = ("rows", "flags", "rw", "rw_flags", "bytecode", "tx", "tx_flags", "r", "ref_kind")


def copy_cases():
    g = np.load(os.path.join(GOLDEN, "copy_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), {key: np.ascontiguousarray(g[f"{k}_{key}"]) for key in COPY_KEYS}


SIGN_FIELDS = ("bytes", "cells", "meta", "keccak", "tx_rows", "tx_flags", "r", "is_sig", "ref_kind")


def sign_cases():
    g = np.load(os.path.join(GOLDEN, "sign_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        yield str(nm), {f: np.ascontiguousarray(g[f"{k}_{f}"]) for f in SIGN_FIELDS}


def copy_oracle_status(c):
    T = copy_oracle.CopyTables(wire.rowmajor_to_rows(c["rw"]), c["rw_flags"], wire.rowmajor_to_rows(c["bytecode"]),
                               wire.rowmajor_to_rows(c["tx"]), c["tx_flags"])
    return copy_oracle.verify_rows(wire.colmajor_to_rows(c["rows"]), c["flags"], T, wire.cells_to_ints(c["r"])[0])


def sign_oracle_status(c):
    return so.verify_units(c["bytes"], c["cells"], c["meta"], wire.rowmajor_to_rows(c["keccak"]), wire.cells_to_ints(c["r"])[0],
                           int(c["is_sig"][0]), wire.rowmajor_to_rows(c["tx_rows"]), c["tx_flags"])


# ---------------------------------------------------------------------------------------------------------------------
# Part 1: the one-shot C entries
# ---------------------------------------------------------------------------------------------------------------------
def oneshot_state():
    n = 0
    for name, cols, flags, mpt, ref_kind in state_cases():
        res, status = oneshot.state_verify(cols, flags, mpt)
        exp = state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt))
        assert status.tolist() == exp, name
        assert [c >> 24 for c in exp] == ref_kind, name
        check_tally(res, exp)
        assert res.rows_evaluated == len(exp) and res.launches == 1
        n += 1
    assert n >= 140


def oneshot_evm():
    n = 0
    for fn in golden_files(GOLDEN):
        for name, w, opts, ref_kind in load_cases(fn):
            exp = oracle_status(w, opts)
            res, status = oneshot.evm_verify(w, bool(opts[0]), bool(opts[1]))
            assert status.tolist() == exp, (os.path.basename(fn), name)
            check_tally(res, exp)
            if n % 11 == 0:
                res2, status2 = oneshot.evm_verify(w, bool(opts[0]), bool(opts[1]), opts=4)  # ZK_OPT_GENERIC_INDEX
                assert status2.tolist() == exp, (os.path.basename(fn), name)
            n += 1
    assert n > 2000


def oneshot_bytecode_exp_copy_sign():
    n = 0
    for name, cols, keccak, ref_kind, r in bytecode_cases():
        res, status = oneshot.bytecode_verify(cols, keccak, r)
        exp = ro.bytecode_verify_rows(wire.colmajor_to_rows(cols), wire.rowmajor_to_rows(keccak), r)
        assert status.tolist() == exp, name
        check_tally(res, exp)
        n += 1
    assert n >= 120
    n = 0
    for name, cols, ref_kind in exp_cases():
        res, status = oneshot.exp_verify(cols)
        exp = ro.exp_verify_rows(wire.colmajor_to_rows(cols))
        assert status.tolist() == exp, name
        check_tally(res, exp)
        n += 1
    assert n >= 20
    n = 0
    for idx, (name, c) in enumerate(copy_cases()):
        exp = copy_oracle_status(c)
        for opts in ((0, 4) if idx % 4 == 0 else (0,)):
            res, status = oneshot.copy_verify(c["rows"], c["flags"], c["r"], c["rw"], c["rw_flags"], c["bytecode"], c["tx"], c["tx_flags"],
                                              opts=opts)
            assert status.tolist() == exp, (name, opts)
            check_tally(res, exp)
        n += 1
    assert n >= 200
    n = 0
    for name, c in sign_cases():
        exp = sign_oracle_status(c)
        res, status = oneshot.sign_verify(c, c["r"], int(c["is_sig"][0]))
        assert status.tolist() == exp, name
        check_tally(res, exp)
        n += 1
    assert n >= 30


def oneshot_keccak_assign_ecdsa():
    z = np.load(os.path.join(GOLDEN, "keccak_table.npz"))
    rs = [int.from_bytes(c.tobytes(), "little") for c in z["randomness"]]
    for ri, r in enumerate(rs):
        for mode in (0, 1):
            res, status, rows = oneshot.keccak_table(z["data"], z["offsets"], r, mode)
            assert np.array_equal(rows, z[f"rows{mode}_{ri}"])
            if mode == 1:
                assert np.array_equal(status >> 24, z[f"kind1_{ri}"])
            else:
                assert not status.any() and res.ok
    g = np.load(os.path.join(GOLDEN, "assign_cases.npz"))
    for i, name in enumerate(g["names"]):
        k = f"c{i:03d}"
        ops, flags = g[k + "_ops"], g[k + "_opflags"]
        res, status, rows, rflags, mpt = oneshot.state_assign(ops, flags)
        o_rows, o_flags, o_mpt, o_status = assign_oracle.assign(wire.colmajor_to_rows(ops), flags.tolist())
        assert status.tolist() == o_status, name
        assert wire.rowmajor_to_rows(mpt) == o_mpt, name  # first-occurrence order
        assert rflags.tolist() == o_flags and wire.colmajor_to_rows(rows) == o_rows, name
        assert res.fail_count == sum(1 for c in o_status if c), name
    g = np.load(os.path.join(GOLDEN, "bytecode_assign_cases.npz"))
    for i, nm in enumerate(g["names"]):
        k = f"c{i:04d}"
        r = wire.cells_to_ints(g[k + "_r"])[0]
        res, rows = oneshot.bytecode_assign(g[k + "_in_rows"], g[k + "_offsets"], g[k + "_lengths"], int(g[k + "_k"]), r)
        assert res.ok and np.array_equal(rows, g[k + "_rows"]), str(nm)
        assert bytecode_assign_oracle.assign(int(g[k + "_k"]), wire.rowmajor_to_rows(g[k + "_in_rows"]), g[k + "_offsets"], g[k + "_lengths"],
                                             r) == wire.colmajor_to_rows(rows)
    g = np.load(os.path.join(GOLDEN, "ecdsa_cases.npz"))
    for v in (np.ascontiguousarray(g["v"]), None):
        res, status = oneshot.ecdsa_verify(g["sigs"], v)
        exp = ecdsa_oracle.verify_packed(g["sigs"], v)
        assert status.tolist() == exp
        assert res.fail_count == sum(1 for e in exp if e)


# ---------------------------------------------------------------------------------------------------------------------
# Part 2: the host mirrors, driven with witness objects
# ---------------------------------------------------------------------------------------------------------------------
def mirror_evm_verify_steps():
    """`verify_steps(tables, steps, begin_with_first_step, end_with_last_step, success)` with success True and False must
    do what the reference's driver did with the same witness (recorded by oracle/gen_golden_evm.py as `ref_driver`)."""
    from zkevm_specs_amd.evm_circuit import verify_steps
    from zkevm_specs_amd.flatten import flatten_evm

    n = n_raise = n_unsupported = 0
    for fn in golden_files(GOLDEN):
        g = np.load(fn)
        for ci, (name, w, opts, ref_kind) in enumerate(load_cases(fn)):
            if ci % 2 and "#fuzz" in name:
                continue  # half of the fuzz variants: the object path is per-cell Python
            driver = g[f"c{ci:04d}_ref_driver"].tolist()
            begin, end = bool(opts[0]), bool(opts[1])
            tables, steps = objects.evm_from_wire(w)
            if ci % 5 == 0 and "#fuzz" not in name:  # the objects carry exactly the recorded witness (fuzzed tables are no longer sorted)
                back = flatten_evm(tables, steps)
                for k in ("steps", "rw", "rw_flags", "bytecode", "tx", "tx_flags", "block", "block_flags", "copy", "keccak", "exp", "aux_kind"):
                    assert np.array_equal(back[k], w[k]), (name, k)
            exp = oracle_status(w, opts)
            first = next((c for c in exp if c), 0)
            if codes.kind_of(first) == codes.UNSUPPORTED:  # declared domain limit: the mirror raises UnsupportedOnDevice there
                with pytest.raises(errors.UnsupportedOnDevice):
                    verify_steps(tables, list(steps[:-1] if end else steps), begin, end, True)
                n_unsupported += 1
                continue
            for success, kind in zip((True, False), driver):
                st = list(steps[:-1] if end else steps)
                n_before = len(st)
                expect_outcome(kind, lambda: verify_steps(tables, st, begin, end, success))  # noqa: B023
                assert len(st) == n_before + (1 if end else 0)  # the reference appends its dummy step to the caller's list
                n_raise += kind != 0
            n += 1
    assert n > 1200 and n_raise > 800 and n_unsupported < n // 20


def _first_kind(ref_kind):
    return next((k for k in ref_kind if k), 0)


def mirror_state():
    from types import SimpleNamespace

    from zkevm_specs_amd.state_circuit import check_state_row, verify_state_rows

    n = 0
    for name, cols, flags, mpt, ref_kind in state_cases():
        rows = objects.state_rows_from_wire(cols, flags)
        tables = SimpleNamespace(mpt_table=objects.mpt_table_from_wire(mpt))
        k = _first_kind(ref_kind)
        # the reference's driver (tests/test_state_circuit.py:17-38): AssertionError -> ok = False; `assert ok == success`
        expect_outcome(k if k else 0, lambda: verify_state_rows(rows, tables, success=True))  # noqa: B023
        expect_outcome((0 if k == codes.ASSERT else k) if k else codes.ASSERT, lambda: verify_state_rows(rows, tables, success=False))  # noqa: B023
        if n % 6 == 0:  # single-row form, reference signature (state_circuit.py:492)
            m = len(rows)
            for i in sorted({0, m - 1, next((j for j, x in enumerate(ref_kind) if x), 0)}):
                expect_outcome(ref_kind[i], lambda: check_state_row(rows[i], rows[(i - 1) % m], rows[(i + 1) % m], tables))  # noqa: B023
        n += 1
    assert n >= 140


def mirror_bytecode():
    from zkevm_specs_amd.bytecode_circuit import check_bytecode_row, verify_bytecode_rows

    n = 0
    for name, cols, keccak, ref_kind, r in bytecode_cases():
        if n % 3:
            n += 1
            continue
        rows = objects.bytecode_rows_from_wire(cols)
        kt = objects.keccak_table_from_wire(keccak)
        k = _first_kind(ref_kind)
        # tests/test_bytecode_circuit.py:31-50: AssertionError is caught; success -> re-raised, else must have happened
        expect_outcome(k, lambda: verify_bytecode_rows(rows, kt, objects.FQ(r), success=True))  # noqa: B023
        expect_outcome((0 if k == codes.ASSERT else k) if k else codes.ASSERT,
                       lambda: verify_bytecode_rows(rows, kt, objects.FQ(r), success=False))  # noqa: B023
        i = next((j for j, x in enumerate(ref_kind) if x), 0)
        expect_outcome(ref_kind[i], lambda: check_bytecode_row(rows[i], rows[(i + 1) % len(rows)], None, kt, objects.FQ(r)))  # noqa: B023
        n += 1
    assert n >= 120


def mirror_copy_exp():
    from types import SimpleNamespace

    from zkevm_specs_amd.copy_circuit import verify_copy_table
    from zkevm_specs_amd.exp_circuit import verify_exp_circuit

    n = 0
    for name, c in copy_cases():
        tables = SimpleNamespace(rw_table=objects.rw_table_from_wire(c["rw"], c["rw_flags"]),
                                 bytecode_table=objects.bytecode_table_from_wire(c["bytecode"]),
                                 tx_table=objects.tx_table_from_wire(c["tx"], c["tx_flags"]))
        circuit = objects.CircuitRows(objects.copy_rows_from_wire(c["rows"], c["flags"]))
        expect_outcome(_first_kind(c["ref_kind"].tolist()), lambda: verify_copy_table(circuit, tables, objects.FQ(wire.cells_to_ints(c["r"])[0])))  # noqa: B023
        n += 1
    assert n >= 200
    n = 0
    for name, cols, ref_kind in exp_cases():
        circuit = objects.CircuitRows(objects.exp_rows_from_wire(cols))
        expect_outcome(_first_kind(ref_kind), lambda: verify_exp_circuit(circuit))  # noqa: B023
        n += 1
    assert n >= 20
    assert verify_exp_circuit(objects.CircuitRows([])) is None and verify_copy_table(objects.CircuitRows([]), None, 0) is None


class _Limb:
    """stand-in of the reference's WrongFieldInteger / Secp256k1*Field: only the byte views the chips read"""

    def __init__(self, le):
        self._le = bytes(le)

    def to_le_bytes(self):
        return self._le

    def to_be_bytes(self):
        return self._le[::-1]


def _recorded_verify(status, returns_bool):
    def verify(*_):
        if status == 0:
            return True
        if status == 1:
            if returns_bool:
                return False
            raise AssertionError("ecdsa_verify failed")
        raise errors.exception_for_code(status)
    return verify


def sign_witness_from_wire(c, name):
    """Tx / Sig witness objects from a golden unit set.  The chips' limbs are the units' own byte rows; chips the
    reference's tests tampered with (limbs and `*_bytes` attributes differ, or an attribute is not a byte string) keep the
    recorded verdict of the reference's chip behind `verify()`, which is what the marshalling falls back to for them."""
    from types import SimpleNamespace

    is_sig = int(c["is_sig"][0])
    bts, meta = c["bytes"], c["meta"]
    cells = wire.colmajor_to_rows(c["cells"])
    units = []
    for i in range(bts.shape[0]):
        b = [bytes(bts[i, k].tolist()) for k in range(9)]
        bad = int(meta[i, 2])
        attr = lambda k: objects.Word(1) if bad & (1 << k) else b[k]  # noqa: E731  a non-bytes attribute, like the reference's tamper tests
        status = int(meta[i, 0])
        from_rows = (bad & 0x1AC) == 0 and "tamper" not in name
        chip = SimpleNamespace(pub_key_x_bytes=attr(2), pub_key_y_bytes=attr(3), msg_hash_bytes=attr(5))
        if is_sig:
            chip.sig_r, chip.sig_s = SimpleNamespace(le_bytes=b[7]), SimpleNamespace(le_bytes=b[8])
            chip.sig_r.to_le_bytes, chip.sig_s.to_le_bytes = (lambda x=b[7]: x), (lambda x=b[8]: x)
            chip.sig_v = _Limb(int(meta[i, 3]).to_bytes(32, "little"))
            chip.msg_hash = _Limb(b[5][::-1])  # the Sig chip keeps msg_hash_bytes big-endian
        else:
            chip.signature = (_Limb(b[7]), _Limb(b[8]))
            chip.msg_hash = _Limb(b[5])
        chip.pub_key = (_Limb(b[2]), _Limb(b[3])) if from_rows else None  # None: reading the limbs raises -> chip.verify()
        chip.verify = _recorded_verify(status, bool(is_sig))
        u = SimpleNamespace(pub_key_x_bytes=attr(0), pub_key_y_bytes=attr(1), msg_hash_bytes=attr(4), pub_key_hash=attr(6), ecdsa_chip=chip,
                            msg_hash=objects.Word(cells[i][1], cells[i][2]))
        if is_sig:
            u.recovered_addr, u.sig_v = objects.FQ(cells[i][0]), objects.FQ(cells[i][3])
            u.sig_r, u.sig_s = objects.Word(cells[i][4], cells[i][5]), objects.Word(cells[i][6], cells[i][7])
            u.is_valid = bool(meta[i, 1])
        else:
            u.address = objects.FQ(cells[i][0])
        units.append(u)
    keccak = SimpleNamespace(table=[(objects.FQ(x[0]), objects.FQ(x[1]), objects.FQ(x[2]), objects.Word(x[3], x[4]))
                                    for x in wire.rowmajor_to_rows(c["keccak"])])
    if is_sig:
        return SimpleNamespace(rows=units, keccak_table=keccak)
    tx_rows = [SimpleNamespace(tx_id=objects.FQ(x[0]), tag=objects.FQ(x[1]), index=objects.FQ(x[2]), value=objects.WordOrValue(x[3], x[4], int(f) & 1))
               for x, f in zip(wire.rowmajor_to_rows(c["tx_rows"]), c["tx_flags"])]
    return SimpleNamespace(rows=tx_rows, keccak_table=keccak, sign_verifications=units)


def mirror_tx_sig():
    from zkevm_specs_amd import sig_circuit, tx_circuit

    n = n_dev = 0
    for name, c in sign_cases():
        witness = sign_witness_from_wire(c, name)
        r = objects.FQ(wire.cells_to_ints(c["r"])[0])
        kind = _first_kind(c["ref_kind"].tolist())
        if int(c["is_sig"][0]):
            expect_outcome(kind, lambda: sig_circuit.verify_circuit(witness, r))  # noqa: B023
        else:
            expect_outcome(kind, lambda: tx_circuit.verify_circuit(witness, c["bytes"].shape[0], 1 << 20, r))  # noqa: B023
        n += 1
        n_dev += "tamper" not in name
    assert n >= 30 and n_dev >= 10


def pi_witness_from_driver_fixture():
    """the PI witness of the reference's own test configuration, rebuilt as objects from tests/golden/pi_driver.npz"""
    from types import SimpleNamespace

    g = np.load(os.path.join(GOLDEN, "pi_driver.npz"))
    ints = lambda a: wire.cells_to_ints(a)  # noqa: E731
    kt = SimpleNamespace(table=[(objects.FQ(x[0]), objects.FQ(x[1]), objects.FQ(x[2]), objects.Word(x[3], x[4])) for x in wire.rowmajor_to_rows(g["keccak"])])
    cc, off = [], 0
    raw = g["copy_constrains"].tobytes()
    for ln in g["copy_lengths"].tolist():
        cc.append(raw[off:off + ln])
        off += ln
    blk = [objects.WordOrValue(*ints(b)) for b in g["block_table"]]
    txs = []
    for t in g["tx_table"]:
        v = ints(t)
        txs.append(SimpleNamespace(tx_id=objects.FQ(v[0]), tag=objects.FQ(v[1]), index=objects.FQ(v[2]), value=objects.WordOrValue(v[3], v[4], v[5])))
    wds = []
    for t in g["withdrawal_table"]:
        v = ints(t)
        wds.append(SimpleNamespace(id=objects.FQ(v[0]), validator_id=objects.FQ(v[1]), address=objects.Word(v[2], v[3]), amount=objects.FQ(v[4])))
    pis = [objects.Word(*ints(x)) for x in g["public_inputs"]]
    gas = [SimpleNamespace(tx_id=objects.FQ(x[0]), is_final=objects.FQ(x[1]), gas_cost_acc=objects.FQ(x[2])) for x in wire.rowmajor_to_rows(g["gas"])]
    w = SimpleNamespace(rows=objects.pi_rows_from_wire(g["rows"], kt), keccak_table=kt, calldata_gas_cost_table=gas,
                        public_inputs=SimpleNamespace(pi_keccak=pis[0], block_hash=pis[1], state_root=pis[2], state_root_prev=pis[3]),
                        block_table=SimpleNamespace(table=blk), tx_table=SimpleNamespace(table=txs), withdrawal_table=SimpleNamespace(table=wds),
                        circuit_len=int(g["circuit_len"][0]), copy_constrains=cc)
    return w, tuple(int(x) for x in g["shape"]), [str(x) for x in g["tamper_names"]], g["driver_kind"].tolist()


def mirror_pi_verify_circuit():
    """`pi_circuit.verify_circuit(witness, MAX_TXS, MAX_CALLDATA_BYTES, MAX_WITHDRAWALS)` on the reference's own test witness and
    its seven tampering tests (tests/test_public_inputs.py:150-207), plus gate failures from tampered row cells"""
    import copy

    from zkevm_specs_amd import pi_circuit

    w0, shape, names, kinds = pi_witness_from_driver_fixture()
    word123 = objects.WordOrValue(123, 0, True)
    tampers = {"valid": lambda w: None,
               "bad_block_table": lambda w: w.block_table.table.__setitem__(5, word123),
               "bad_tx_table_tx_id": lambda w: setattr(w.tx_table.table[5], "tx_id", objects.FQ(123)),
               "bad_tx_table_index": lambda w: setattr(w.tx_table.table[5], "index", objects.FQ(123)),
               "bad_tx_table_value": lambda w: setattr(w.tx_table.table[5], "value", word123),
               "bad_keccak_digest": lambda w: setattr(w.public_inputs, "pi_keccak", objects.Word(123, 0)),
               "bad_state_root": lambda w: setattr(w.public_inputs, "state_root", objects.Word(123, 0)),
               "bad_state_root_prev": lambda w: setattr(w.public_inputs, "state_root_prev", objects.Word(123, 0))}
    for name, kind in zip(names, kinds):
        w = copy.deepcopy(w0)
        tampers[name](w)
        expect_outcome(kind, lambda: pi_circuit.verify_circuit(w, *shape))  # noqa: B023
    # a gate failure: break the keccak-RLC chain in the middle -> AssertionError from the device pass
    w = copy.deepcopy(w0)
    w.rows[len(w.rows) // 2].rpi_bytes_keccakrlc = objects.FQ(7)
    expect_outcome(codes.ASSERT, lambda: pi_circuit.verify_circuit(w, *shape))
    # a lookup failure: the CallDataLength row's gas cost is not in the gas-cost table -> LookupUnsatFailure
    w = copy.deepcopy(w0)
    w.calldata_gas_cost_table = w.calldata_gas_cost_table[:1]
    expect_outcome(codes.LOOKUP_UNSAT, lambda: pi_circuit.verify_circuit(w, *shape))
