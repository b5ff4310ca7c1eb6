"""The host side of the drop-in boundary in the GPU-less container: tests/dropin_cases.py's mirror bodies with the
one-shot C entries (`zkevm_specs_amd.oneshot`) replaced by oracle-backed stand-ins of the same signatures.  What this
checks is everything ABOVE the C ABI — witness objects -> wire arrays (objects.py / flatten.py round trip), the
`success` / exception semantics of the reference's drivers, the deferred ECDSA column — so that the `-m gpu` run of
the very same bodies (tests/test_dropin_gpu.py) only adds the HIP library underneath."""
import numpy as np
import pytest

from oracle import assign_oracle, bytecode_assign_oracle, ecdsa_oracle, keccak_table as KT, row_oracles as ro
from oracle import sign_oracle as so, state_oracle, wire
from tests import dropin_cases as D
from tests.evm_cases import oracle_status
from zkevm_specs_amd import _lib, engine, oneshot


def _result(status, launches=1):
    z = _lib.ZkResult()
    fails = [j for j, c in enumerate(status) if c]
    z.fail_count = len(fails)
    z.first_fail_row = fails[0] if fails else 0xFFFFFFFFFFFFFFFF
    z.first_fail_code = int(status[fails[0]]) if fails else 0
    z.launches, z.rows_evaluated, z.kernel_ms = launches, len(status), 0.0
    return engine.Result(z)


def _r(randomness):
    return randomness if isinstance(randomness, int) else wire.cells_to_ints(np.asarray(randomness))[0]


@pytest.fixture()
def oracle_backend(monkeypatch):
    def state_verify(rows, flags, mpt, device=None):
        st = state_oracle.verify_rows(wire.colmajor_to_rows(rows), flags, wire.rowmajor_to_rows(mpt))
        return _result(st), np.array(st, dtype=np.uint32)

    def evm_verify(w, begin=False, end=False, opts=0, device=None):
        st = oracle_status(w, (int(begin), int(end)))
        return _result(st), np.array(st, dtype=np.uint32)

    def bytecode_verify(rows, keccak, randomness, device=None):
        st = ro.bytecode_verify_rows(wire.colmajor_to_rows(rows), wire.rowmajor_to_rows(keccak), _r(randomness))
        return _result(st), np.array(st, dtype=np.uint32)

    def exp_verify(rows, device=None):
        st = ro.exp_verify_rows(wire.colmajor_to_rows(rows))
        return _result(st), np.array(st, dtype=np.uint32)

    def copy_verify(rows, row_flags, randomness, rw, rw_flags, bytecode, tx, tx_flags, opts=0, device=None):
        st = D.copy_oracle_status({"rows": rows, "flags": row_flags, "rw": rw, "rw_flags": rw_flags, "bytecode": bytecode, "tx": tx,
                                   "tx_flags": tx_flags, "r": wire.ints_to_cells([_r(randomness)])})
        return _result(st), np.array(st, dtype=np.uint32)

    def sign_verify(w, randomness, is_sig, device=None):
        st = so.verify_units(w["bytes"], w["cells"], w["meta"], wire.rowmajor_to_rows(w["keccak"]), _r(randomness), int(bool(is_sig)),
                             wire.rowmajor_to_rows(w["tx_rows"]), w["tx_flags"])
        return _result(st), np.array(st, dtype=np.uint32)

    def keccak_table(data, offsets, randomness, mode=0, device=None):
        data = np.asarray(data, dtype=np.uint8)
        msgs = [data[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(len(offsets) - 1)]
        rows, st = KT.table_rows(msgs, _r(randomness), mode)
        return _result(st.tolist()), st, rows

    def state_assign(ops, op_flags, device=None):
        rows, rflags, mpt, st = assign_oracle.assign(wire.colmajor_to_rows(ops), list(op_flags))
        return (_result(st), np.array(st, dtype=np.uint32), wire.rows_to_colmajor(rows), np.array(rflags, dtype=np.uint32),
                wire.rows_to_rowmajor(mpt, 12))

    def bytecode_assign(in_rows, offsets, lengths, k, randomness, device=None):
        rows = bytecode_assign_oracle.assign(int(k), wire.rowmajor_to_rows(in_rows), offsets, lengths, _r(randomness))
        return _result([0] * (1 << int(k))), wire.rows_to_colmajor(rows)

    def pi_verify(rows, keccak, gas, circuit_len, keccak_rand=255, byte_pow_base=255, device=None):
        from oracle import pi_oracle

        st = pi_oracle.verify_rows(wire.colmajor_to_rows(rows), wire.rowmajor_to_rows(gas), wire.rowmajor_to_rows(keccak), circuit_len,
                                   keccak_rand, byte_pow_base)
        return _result(st), np.array(st, dtype=np.uint32)

    def pi_copy_verify(cells, data, lens, device=None):
        from oracle import pi_oracle

        st = pi_oracle.copy_constraints_status(wire.cells_to_ints(np.asarray(cells)), np.asarray(data), [int(x) for x in lens])
        return _result(st), np.array(st, dtype=np.uint32)

    def copy_assign(events, flags, data, offsets, randomness, device=None):
        from oracle import copy_assign_oracle as CA

        rows, rf, table, rw, rwf = CA.assign(wire.rowmajor_to_rows(np.asarray(events)), [int(f) for f in flags], data, offsets, _r(randomness))
        return (_result([0] * len(rows)), wire.rows_to_colmajor(rows) if rows else np.zeros((20, 0, 4), np.uint64), np.array(rf, dtype=np.uint32),
                wire.rows_to_rowmajor(table, 14), wire.rows_to_rowmajor(rw, 14), np.array(rwf, dtype=np.uint32))

    def ecdsa_verify(sig_bytes, v=None, layout=0, v_stride=1, device=None):
        assert layout == 0
        st = ecdsa_oracle.verify_packed(np.asarray(sig_bytes), v)
        return _result(st), np.array(st, dtype=np.uint32)

    for name, fn in list(locals().items()):
        if callable(fn) and hasattr(oneshot, name):
            monkeypatch.setattr(oneshot, name, fn)


def test_every_oneshot_entry_has_a_stand_in(oracle_backend):
    import inspect

    real = [n for n, f in vars(oneshot).items() if inspect.isfunction(f) and f.__module__ == __name__]
    assert sorted(real) == sorted(["state_verify", "evm_verify", "bytecode_verify", "exp_verify", "copy_verify", "sign_verify",
                                   "keccak_table", "state_assign", "bytecode_assign", "ecdsa_verify", "pi_verify", "copy_assign", "pi_copy_verify"])


def test_mirror_verify_steps_host_logic(oracle_backend):
    D.mirror_evm_verify_steps()


def test_mirror_state_host_logic(oracle_backend):
    D.mirror_state()


def test_mirror_bytecode_host_logic(oracle_backend):
    D.mirror_bytecode()


def test_mirror_copy_exp_host_logic(oracle_backend):
    D.mirror_copy_exp()


def test_mirror_tx_sig_host_logic(oracle_backend):
    D.mirror_tx_sig()


def test_mirror_pi_verify_circuit_host_logic(oracle_backend):
    D.mirror_pi_verify_circuit()


def test_copy_events_builder_host_logic(oracle_backend, golden_dir):
    """CopyEvents.copy() takes the reference's arguments and reproduces the rows of its `CopyCircuit.copy` calls"""
    import os

    from zkevm_specs_amd.copy_circuit import CopyEvents

    g = np.load(os.path.join(golden_dir, "copy_assign_cases.npz"))
    for i in range(0, len(g["names"]), 7):
        k = f"c{i:04d}"
        ev = wire.rowmajor_to_rows(g[k + "_event"])[0]
        fl = int(g[k + "_flags"][0])
        data = g[k + "_data"].tolist()
        n_real = max(0, min(ev[9], ev[7] - ev[6]))
        src_data = {ev[6] + j: ((data[j] & 0xFF, data[j] >> 8) if (ev[2] == 1 or ev[5] == 1) else data[j] & 0xFF) for j in range(n_real)}
        mk = lambda lo, hi, w: D.objects.Word(lo, hi) if w else lo  # noqa: E731
        b = CopyEvents(wire.cells_to_ints(g[k + "_r"])[0])
        end = b.copy(ev[11], mk(ev[0], ev[1], fl & 1), ev[2], mk(ev[3], ev[4], fl & 2), ev[5], ev[6], ev[7], ev[8], ev[9], src_data, ev[10])
        rows, rf, table, rw, rwf = b.assign()
        assert np.array_equal(rows, g[k + "_rows"]) and np.array_equal(rf, g[k + "_row_flags"])
        assert np.array_equal(rw, g[k + "_rw"]) and np.array_equal(table, g[k + "_table"])
        assert end == ev[11] + rw.shape[0]


def test_state_assign_mirror_host_logic(oracle_backend, golden_dir):
    import os

    from zkevm_specs_amd import errors, state_circuit

    g = np.load(os.path.join(golden_dir, "assign_cases.npz"))
    n_raise = 0
    for i, name in enumerate(g["names"]):
        k = f"c{i:03d}"
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
