"""The CPU backend behind the same C ABI (libzkevm_cpu.so, ZK_BACKEND=cpu; csrc/cpu_backend.cpp): BASELINE configs[0] — the
Bytecode circuit over a 256-byte contract, "pure CPU path (plumbing, no GPU)" — and a slice of every other circuit, through
the real boundary (ctypes -> C ABI -> the kernels' own per-row functions on the host cores), against the oracle.  Runs in a
child process because the backend is chosen when zkevm_specs_amd._lib is first imported."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["ZK_ROOT"])
from zkevm_specs_amd import _lib, engine, oneshot
assert _lib.BACKEND == "cpu" and _lib.LIB_PATH.endswith("libzkevm_cpu.so")
from oracle import row_oracles as ro, state_oracle, wire, sign_oracle as so, ecdsa_oracle
from tests.evm_cases import golden_files, load_cases, oracle_status
from zkevm_specs_amd.synth import synth_bytecode_witness, synth_state_witness, synth_tx_witness, synth_exp_witness
out = {}

# BASELINE configs[0]: 256-byte contract, k = 9 -> 512 rows (257 real), random keccak randomness; valid and tampered
code = bytes(np.random.default_rng(1).integers(0, 256, 256, dtype=np.uint8))
r = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % (1 << 253)
cols, keccak = synth_bytecode_witness([code], 9, r)
res, st = oneshot.bytecode_verify(cols, keccak, r)
assert res.ok and res.rows_evaluated == 512 and not st.any()
cols[6, 17, 0] ^= 1
cols[9, 300, 0] ^= 1
res, st = oneshot.bytecode_verify(cols, keccak, r)
exp = ro.bytecode_verify_rows(wire.colmajor_to_rows(cols), wire.rowmajor_to_rows(keccak), r)
assert st.tolist() == exp and res.fail_count == sum(1 for c in exp if c) >= 1 and res.first_fail_row == next(i for i, c in enumerate(exp) if c)
out["bytecode_rows"] = 512

# State circuit, sessions + ranges
c, f, m = synth_state_witness(4096, seed=5)
c[1, 1000, 0] = 2
with engine.open_state(c, f, m) as s:
    res = s.run()
    st = s.read_status()
    exp = state_oracle.verify_rows(wire.colmajor_to_rows(c), f, wire.rowmajor_to_rows(m))
    assert st.tolist() == exp and res.fail_count == sum(1 for x in exp if x) >= 1
    s.set_range(2000, 3000)
    assert s.run().ok and s.collect().rows_evaluated == 1000
out["state_rows"] = 4096

# EVM circuit: a slice of the golden cases (kind and site) through zk_evm_verify
n = 0
for fn in golden_files(os.path.join(os.environ["ZK_ROOT"], "tests", "golden"))[::5]:
    for name, w, opts, ref_kind in list(load_cases(fn))[::7]:
        res, st = oneshot.evm_verify(w, bool(opts[0]), bool(opts[1]))
        assert st.tolist() == oracle_status(w, opts), (fn, name)
        n += 1
out["evm_cases"] = n

# Exp circuit
e = synth_exp_witness(256, seed=3)
res, st = oneshot.exp_verify(e)
assert st.tolist() == ro.exp_verify_rows(wire.colmajor_to_rows(e))

# Tx circuit with the ECDSA column computed by the same library
R = 0x1F2E3D4C5B6A79881726354433221100FFEEDDCCBBAA99887766554433221
tx = synth_tx_witness(24, R, seed=4, signed=True)
res_e, st_e = oneshot.ecdsa_verify(tx["bytes"], None, layout=1)
assert not st_e.any()
tx["meta"][:, 0] = st_e
tx["cells"][1, 5, 0] ^= 1
res, st = oneshot.sign_verify(tx, R, False)
exp = so.verify_units(tx["bytes"], tx["cells"], tx["meta"], wire.rowmajor_to_rows(tx["keccak"]), R, 0, wire.rowmajor_to_rows(tx["tx_rows"]), tx["tx_flags"])
assert st.tolist() == exp and res.fail_count == 1

# Fr ops incl. inv / div
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
A = [0, 1, 8, P - 1, 2**200 + 12345]; B = [5, 7, P - 2, 3, 2**128]
a, b = wire.ints_to_cells(A), wire.ints_to_cells(B)
inv = lambda x: pow(x, -1, P) if x else 0
assert wire.cells_to_ints(engine.fr_op(5, a, b)) == [inv(x) for x in A]
assert wire.cells_to_ints(engine.fr_op(6, a, b)) == [x * inv(y) % P for x, y in zip(A, B)]
assert wire.cells_to_ints(engine.fr_op(2, a, b)) == [x * y % P for x, y in zip(A, B)]

# witness assignment through the CPU backend (round 4: it used to refuse these entries): the golden op lists / unrolled bytecodes /
# copy events of the reference's own assign_state_circuit / assign_bytecode_circuit / CopyCircuit.copy, outputs vs the oracles
from oracle import assign_oracle, bytecode_assign_oracle, copy_assign_oracle
g = np.load(os.path.join(os.environ["ZK_ROOT"], "tests", "golden", "assign_cases.npz"))
n_as = 0
for i in range(0, len(g["names"]), 3):
    k = f"c{i:03d}"
    ops, fl = g[k + "_ops"], g[k + "_opflags"]
    rows, rflags, mpt, status = assign_oracle.assign(wire.colmajor_to_rows(ops), fl.tolist())
    res, st, d_rows, d_rf, d_mpt = oneshot.state_assign(ops, fl)
    assert st.tolist() == list(status), g["names"][i]
    if not any(status):
        assert wire.colmajor_to_rows(d_rows) == rows and d_rf.tolist() == rflags and wire.rowmajor_to_rows(d_mpt) == mpt, g["names"][i]
    n_as += 1
g = np.load(os.path.join(os.environ["ZK_ROOT"], "tests", "golden", "bytecode_assign_cases.npz"))
n_bc = 0
for i in range(0, len(g["names"]), 5):
    k = f"c{i:04d}"
    in_rows, off, ln, kk = g[k + "_in_rows"], g[k + "_offsets"], g[k + "_lengths"], int(g[k + "_k"])
    r = wire.cells_to_ints(g[k + "_r"])[0]
    res, d_rows = oneshot.bytecode_assign(in_rows, off, ln, kk, r)
    assert wire.colmajor_to_rows(d_rows) == bytecode_assign_oracle.assign(kk, wire.rowmajor_to_rows(in_rows), off, ln, r), g["names"][i]
    n_bc += 1
g = np.load(os.path.join(os.environ["ZK_ROOT"], "tests", "golden", "copy_assign_cases.npz"))
n_cp = 0
for i in range(0, len(g["names"]), 6):
    k = f"c{i:04d}"
    ev, fl, da = g[k + "_event"], g[k + "_flags"], g[k + "_data"]
    off = np.array([0, len(da)], dtype=np.uint64)
    r = wire.cells_to_ints(g[k + "_r"])[0]
    exp = copy_assign_oracle.assign(wire.rowmajor_to_rows(ev), fl.tolist(), da, off, r)
    res, c_rows, c_rf, c_table, c_rw, c_rwf = oneshot.copy_assign(ev, fl, da, off, r)
    if not len(exp[0]):
        continue  # (an event that copies nothing has no rows: the C entry wants at least the row buffers)
    assert wire.colmajor_to_rows(c_rows) == exp[0] and c_rf.tolist() == list(exp[1]), g["names"][i]
    assert wire.rowmajor_to_rows(c_table) == exp[2] and wire.rowmajor_to_rows(c_rw) == exp[3] and c_rwf.tolist() == list(exp[4]), g["names"][i]
    n_cp += 1
out["assign_cases"] = [n_as, n_bc, n_cp]

# zk_dist_* on the CPU backend: one process = world 1, the identity with the row offset applied; a larger world is refused
from zkevm_specs_amd import distributed
cols, flags, mpt = synth_state_witness(512, seed=4)
cols[1, 77, 0] = np.uint64(2)
with engine.open_state(cols, flags, mpt) as s:
    res = s.run()
with distributed.RcclTally(0, 1) as t:
    assert t.reduce(res, row_offset=1000) == (res.fail_count, 1077, res.first_fail_code) and res.fail_count >= 1
    class Clean: fail_count = 0; first_fail_row = None; first_fail_code = 0
    assert t.reduce(Clean, row_offset=5) == (0, None, 0)
import ctypes
lib = _lib.load()
h = ctypes.c_void_p()
assert lib.zk_dist_init((ctypes.c_uint8 * 128)(), 1, 2, ctypes.byref(h)) != 0 and b"world must be 1" in lib.zk_last_error()
out["dist"] = "ok"

# the block's six circuits over one consistent witness, with the 57-cell State witness and with none (state_fused: the State rows evaluated from
# the RW table where they are computed, two passes over the same session): equal verdicts, clean and with a damaged RW value
from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block
p = synth_super_block(13, seed=3)
rw = p["evm"]["rw"]
i = next(j for j in range(500, rw.shape[0]) if int(rw[j, 2, 0]) == 8 and int(rw[j, 1, 0]) == 0)  # a Stack read
tallies = {}
for damaged in (False, True):
    if damaged:
        rw[i, 8, 0] ^= np.uint64(1)
    for fused in (False, True):
        with SuperCircuit(p, state_fused=fused) as sc:
            assert sc.state_fused == fused
            for _ in range(2):
                sc.launch()
                res, total, first = sc.collect()
                tallies[(damaged, fused, _)] = {k: (r.fail_count, r.first_fail_row, r.first_fail_code) for k, r in res.items()}
    assert tallies[(damaged, True, 0)] == tallies[(damaged, True, 1)] == tallies[(damaged, False, 0)], damaged
assert all(v == (0, None, 0) for v in tallies[(False, True, 0)].values())
assert tallies[(True, True, 0)]["state"][0] >= 1 and tallies[(True, True, 0)]["evm"][0] >= 1
out["super_fused"] = "ok"
print("RESULT " + json.dumps(out))
'''


def test_cpu_backend_through_the_c_abi(tmp_path):
    so = os.path.join(ROOT, "zkevm_specs_amd", "libzkevm_cpu.so")
    if not os.path.exists(so):
        import __graft_entry__

        __graft_entry__.build()
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, ZK_BACKEND="cpu", ZK_ROOT=ROOT, OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("RESULT ")]
    out = json.loads(line[0][7:])
    assert out["bytecode_rows"] == 512 and out["evm_cases"] >= 20
    assert out["assign_cases"][0] >= 25 and out["assign_cases"][1] >= 15 and out["assign_cases"][2] >= 30


def test_cpu_library_exports_the_declared_abi():
    import ctypes
    import re

    so = os.path.join(ROOT, "zkevm_specs_amd", "libzkevm_cpu.so")
    if not os.path.exists(so):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(so)
    text = open(os.path.join(ROOT, "include", "zkevm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for sym in sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text))):
        assert hasattr(lib, sym), f"{sym} declared in include/zkevm_hip.h but not exported by libzkevm_cpu.so"
