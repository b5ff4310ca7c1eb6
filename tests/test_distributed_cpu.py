"""World-size-2 gloo test of the multi-GPU path's host logic: row sharding with halos, replicated
tables, and the tally all-reduce — State (±1 halo), EVM (+1 step), Tx units (no halo, their tx-table rows travel with
them) and Copy rows (+2 halo).  Each rank evaluates its shard with the CPU logic harness
(hostsim) in place of the GPU; the reduced tally must equal the single-process one."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["ZK_ROOT"])
from tests.evm_cases import hostsim_status
from zkevm_specs_amd import distributed
from zkevm_specs_amd.synth import synth_state_witness
from zkevm_specs_amd.synth_evm import synth_evm_trace

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(os.path.join(os.environ["ZK_ROOT"], "tests", "hostsim", "libhostsim.so"))
vp = lambda x: ctypes.c_void_p(x.ctypes.data)

# ZK_ODD=1 (the world-size-8 run): row counts not divisible by the world size, a tampered cell on both sides of EVERY shard boundary,
# and the EVM driver's begin_with_first_step / end_with_last_step flags (only rank 0 / the last rank may apply them)
odd = os.environ.get("ZK_ODD") == "1"
N_STATE = 4099 if odd else 4096
# ---- State circuit: one global witness, tampered around the shard boundaries ------------------
cols, flags, mpt = synth_state_witness(N_STATE, seed=9)
cols[1, 3000, 0] = 2      # is_write not boolean (rank 1's range)
cols[50, 2047, 0] ^= 1    # value of the last row of rank 0's range
cols[0, 2048, 0] = 0      # rw_counter of the first row of rank 1's range (halo of rank 0)
if odd:
    for r in range(1, world):
        b = distributed.shard_bounds(N_STATE, r, world)[0]
        cols[50, b - 1, 0] ^= 1   # last row of rank r - 1 (read as `prev` by rank r's first row)
        cols[0, b, 0] ^= 4        # first row of rank r (read as `next` by rank r - 1's last row)
lc, lf, elo, ehi, off = distributed.shard_state(cols, flags, rank, world)
st = np.zeros(lc.shape[1], dtype=np.uint32)
lib.sim_state_verify_range(vp(lc), vp(lf), ctypes.c_uint64(lc.shape[1]), vp(mpt), ctypes.c_uint64(mpt.shape[0]),
                           ctypes.c_uint64(elo), ctypes.c_uint64(ehi), vp(st))
local = st[elo:ehi]
fails = np.nonzero(local)[0]
res = distributed.reduce_tally(len(fails), int(fails[0]) if len(fails) else None, int(local[fails[0]]) if len(fails) else 0, off)
full = np.zeros(N_STATE, dtype=np.uint32)
lib.sim_state_verify(vp(cols), vp(flags), ctypes.c_uint64(N_STATE), vp(mpt), ctypes.c_uint64(mpt.shape[0]), vp(full))
ff = np.nonzero(full)[0]
assert res == (len(ff), int(ff[0]), int(full[ff[0]])), (rank, res, len(ff), ff[:4])
assert np.array_equal(local, full[off:off + len(local)])

# ---- EVM circuit: one global trace, tampered near the shard boundary ------------------------
w = synth_evm_trace(1501, seed=12)
w.pop("meta")
w["steps"][750, 7, 0] += 1     # program counter of the boundary step
w["rw"][40, 8, 0] ^= 1
if odd:
    n_pairs = w["steps"].shape[0] - 1
    for r in range(1, world):
        b_ = distributed.shard_bounds(n_pairs, r, world)[0]
        w["steps"][b_, 8, 0] += 1   # stack pointer of the step both neighbouring shards read (last pair of r - 1, first pair of r)
lw, b, e, off = distributed.shard_evm(w, rank, world, begin_with_first_step=odd, end_with_last_step=odd)
assert (b, e) == (odd and rank == 0, odd and rank == world - 1)
local = np.array(hostsim_status(lib, lw, (b, e)), dtype=np.uint32)
fails = np.nonzero(local)[0]
res = distributed.reduce_tally(len(fails), int(fails[0]) if len(fails) else None, int(local[fails[0]]) if len(fails) else 0, off)
full = np.array(hostsim_status(lib, w, (odd, odd)), dtype=np.uint32)
if odd:  # the trace neither begins with BeginTx nor ends in EndBlock: exactly the first and the last pair notice the flags
    plain = hostsim_status(lib, w)
    assert full[0] != plain[0] and full[-1] != plain[-1] and full[1:-1].tolist() == plain[1:-1]
ff = np.nonzero(full)[0]
assert len(ff) >= 2
assert res == (len(ff), int(ff[0]), int(full[ff[0]])), (rank, res)
assert np.array_equal(local, full[off:off + len(local)])
# ---- Tx units: one global witness, no halo; the twelve fixed tx-table rows travel with their unit -------------
from zkevm_specs_amd.synth import synth_tx_witness, synth_copy_events
R = 0x1F2E3D4C5B6A79881726354433221100FFEEDDCCBBAA99887766554433221
r4 = np.frombuffer(R.to_bytes(32, "little"), dtype="<u8").copy()
tx = synth_tx_witness(101, R, seed=21)
tx["cells"][0, 70, 0] ^= 1            # address cell of unit 70 (rank 1)
tx["tx_rows"][12 * 13 + 11, 3, 0] ^= 1  # TxSignHash row of unit 13 (rank 0)
def sign_status(w):
    n = w["bytes"].shape[0]
    st = np.zeros(n, dtype=np.uint32)
    a = {k: np.ascontiguousarray(v) for k, v in w.items()}
    lib.sim_sign_verify(vp(a["bytes"]), vp(a["cells"]), vp(a["meta"]), ctypes.c_uint64(n), vp(a["keccak"]), ctypes.c_uint64(a["keccak"].shape[0]),
                        vp(a["tx_rows"]), vp(a["tx_flags"]), ctypes.c_uint64(a["tx_rows"].shape[0]), vp(r4), ctypes.c_uint32(0), vp(st))
    return st
lw, off = distributed.shard_units(tx, rank, world)
local = sign_status(lw)
fails = np.nonzero(local)[0]
res = distributed.reduce_tally(len(fails), int(fails[0]) if len(fails) else None, int(local[fails[0]]) if len(fails) else 0, off)
full = sign_status(tx)
ff = np.nonzero(full)[0]
assert ff.tolist() == [13, 70], ff
assert res == (2, 13, int(full[13])), (rank, res)
assert np.array_equal(local, full[off:off + len(local)])

# ---- Copy circuit: rows sharded with a 2-row halo (copy_circuit.py:92-130 reads rows i + 1 and i + 2) ------
ce = synth_copy_events(600, seed=22)
n_rows = int(ce["n_rows"])
evl = [[int(c[0]) for c in e] for e in ce["events"]]  # every event cell of the generator fits 64 bits
n_real = lambda e: max(0, min(e[9], e[7] - e[6]))
n_rw = sum((n_real(e) if e[2] == 2 else 0) + (e[9] if e[5] in (2, 4) else 0) for e in evl)
rows = np.zeros((20, n_rows, 4), dtype=np.uint64); rf = np.zeros(n_rows, dtype=np.uint32)
table = np.zeros((len(evl), 14, 4), dtype=np.uint64)
rw = np.zeros((max(n_rw, 1), 14, 4), dtype=np.uint64); rwf = np.zeros(max(n_rw, 1), dtype=np.uint32)
rc4 = np.frombuffer(int(ce["r"]).to_bytes(32, "little"), dtype="<u8").copy()
ev, fl, da, of = (np.ascontiguousarray(ce[k]) for k in ("events", "flags", "data", "offsets"))
assert lib.sim_copy_assign(vp(ev), vp(fl), ctypes.c_uint64(ev.shape[0]), vp(da), vp(of), vp(rc4), vp(rows), vp(rf), vp(table), vp(rw), vp(rwf),
                           ctypes.c_uint64(n_rows)) == 0
half = n_rows // 2
rows[9, half - 1, 0] ^= 1   # a cell of the last row of rank 0's range
rows[9, half + 1, 0] ^= 1   # ... and of a halo row of rank 0 = row 1 of rank 1's range
if odd:
    for r in range(1, world):
        b_ = distributed.shard_bounds(n_rows, r, world)[0]
        rows[9, b_ + 1, 0] ^= 2   # second halo row of rank r - 1
def copy_status(c, f):
    st = np.zeros(c.shape[1], dtype=np.uint32)
    bc, txr, txf = (np.ascontiguousarray(ce[k]) for k in ("bytecode", "tx", "tx_flags"))
    lib.sim_copy_verify(vp(c), vp(f), ctypes.c_uint64(c.shape[1]), vp(rc4), vp(rw), vp(rwf), ctypes.c_uint64(n_rw), vp(bc), ctypes.c_uint64(bc.shape[0]),
                        vp(txr), vp(txf), ctypes.c_uint64(txr.shape[0]), ctypes.c_uint32(0), vp(st))
    return st
lc, lf, elo, ehi, off = distributed.shard_rows(rows, rf, rank, world, "copy")
assert lc.shape[1] == (ehi - elo) + 2 and elo == 0
local = copy_status(lc, lf)[elo:ehi]
fails = np.nonzero(local)[0]
res = distributed.reduce_tally(len(fails), int(fails[0]) if len(fails) else None, int(local[fails[0]]) if len(fails) else 0, off)
full = copy_status(rows, rf)
ff = np.nonzero(full)[0]
assert len(ff) >= 2 and ff[0] < half <= ff[-1], ff
assert res == (len(ff), int(ff[0]), int(full[ff[0]])), (rank, res, ff)
assert np.array_equal(local, full[off:off + len(local)])
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _run_world(tmp_path, world, port, odd):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, ZK_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), ZK_ODD="1" if odd else "0",
               OMP_NUM_THREADS="1")
    procs = []
    for rank in range(world):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o}"
        assert f"rank {rank} ok" in o


def test_two_rank_sharding_and_tally_allreduce(hostsim, tmp_path):
    _run_world(tmp_path, 2, 29613, odd=False)


def test_eight_rank_uneven_shards_and_driver_flags(hostsim, tmp_path):
    """The shape the driver's 8-GPU run has: world size 8, row counts not divisible by 8 (State 4,099 rows, 1,500
    step pairs, 101 Tx units, the Copy rows of 600 events), tampered cells on both sides of every shard boundary, and
    begin_with_first_step / end_with_last_step applied by rank 0 / rank 7 only — the reduced tally (SUM of the counts, MIN of the
    first failing global row) equals the single-process one for all four workloads."""
    _run_world(tmp_path, 8, 29641, odd=True)
