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


def _constraint_cases():
    """the copy constraints (pi_circuit.py:355-445) of the reference's own PI test witness (tests/golden/pi_driver.npz), as the mirror
    lists them, valid and under the reference's seven tampering tests; then cell / byte / length fuzz of the valid list"""
    import copy
    import random

    from tests import dropin_cases as D
    from zkevm_specs_amd import objects
    from zkevm_specs_amd.pi_circuit import list_copy_constraints

    w0, shape, names, kinds = D.pi_witness_from_driver_fixture()
    word123 = objects.WordOrValue(123, 0, True)
    tampers = {"valid": lambda w: None,
               "bad_block_table": lambda w: w.block_table.table.__setitem__(5, word123),
               "bad_tx_table_tx_id": lambda w: setattr(w.tx_table.table[5], "tx_id", objects.FQ(123)),
               "bad_tx_table_index": lambda w: setattr(w.tx_table.table[5], "index", objects.FQ(123)),
               "bad_tx_table_value": lambda w: setattr(w.tx_table.table[5], "value", word123),
               "bad_keccak_digest": lambda w: setattr(w.public_inputs, "pi_keccak", objects.Word(123, 0)),
               "bad_state_root": lambda w: setattr(w.public_inputs, "state_root", objects.Word(123, 0)),
               "bad_state_root_prev": lambda w: setattr(w.public_inputs, "state_root_prev", objects.Word(123, 0))}
    base = None
    for name, kind in zip(names, kinds):
        w = copy.deepcopy(w0)
        tampers[name](w)
        C, pending = list_copy_constraints(w, *shape)
        cells, data, lens = C.wire()
        if name == "valid":
            base = (cells, data, lens)
        yield name, cells, data, lens, (kind, pending)
    rng = random.Random(6)
    for k in range(40):
        cells, data, lens = (a.copy() for a in base)
        for _ in range(rng.choice([1, 2, 5])):
            i = rng.randrange(len(lens))
            what = rng.choice(["cell", "byte", "len", "len32"])
            if what == "cell":
                cells[i, rng.randrange(4)] ^= np.uint64(1 << rng.randrange(64))
            elif what == "byte":
                data[i, rng.randrange(32)] ^= np.uint8(1 << rng.randrange(8))
            elif what == "len" and int(lens[i]) != 0xFFFFFFFF:
                lens[i] = rng.randrange(0, 32)
            elif int(lens[i]) != 0xFFFFFFFF:
                lens[i] = 32
        yield f"fuzz{k}", cells, data, lens, None


def test_copy_constraints_oracle_and_kernel_logic(hostsim):
    n = n_fail = 0
    for name, cells, data, lens, ref in _constraint_cases():
        exp = PO.copy_constraints_status(wire.cells_to_ints(cells), data, lens.tolist())
        st = np.zeros(len(exp), dtype=np.uint32)
        hostsim.sim_pi_copy_verify(vp(cells), vp(data), vp(lens), ctypes.c_uint64(len(exp)), vp(st))
        assert st.tolist() == exp, name
        if ref is not None:  # the reference driver's recorded outcome of this tampering: an AssertionError iff a constraint fails
            kind, pending = ref
            # (a table entry turned into a word makes the reference pop one entry more: the list runs out at the end, after the
            # failing assert — `pending` is the IndexError the mirror raises only if every listed constraint passed)
            if name == "valid":
                assert pending is None and not any(exp)
            elif kind == codes.ASSERT and name.startswith("bad_"):
                assert any(exp), name
        n += len(exp)
        n_fail += sum(1 for e in exp if e)
    assert n > 10000 and n_fail > 40


@pytest.mark.gpu
def test_gpu_copy_constraints():
    from zkevm_specs_amd import oneshot

    for name, cells, data, lens, _ in _constraint_cases():
        exp = PO.copy_constraints_status(wire.cells_to_ints(cells), data, lens.tolist())
        res, st = oneshot.pi_copy_verify(cells, data, lens)
        assert st.tolist() == exp, name
        fails = [j for j, e in enumerate(exp) if e]
        assert res.fail_count == len(fails) and (not fails or (res.first_fail_row == fails[0] and res.first_fail_code == exp[fails[0]]))

