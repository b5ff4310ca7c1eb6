"""World-size-2 gloo test of the multi-GPU path's host logic: row sharding with halos, replicated
tables, and the tally all-reduce.  Each rank evaluates its shard with the CPU logic harness
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

# ---- State circuit: one global 4096-row witness, tampered in both halves ------------------
cols, flags, mpt = synth_state_witness(4096, seed=9)
cols[1, 3000, 0] = 2      # is_write not boolean (rank 1's range)
cols[50, 2047, 0] ^= 1    # value of the last row of rank 0's range
cols[0, 2048, 0] = 0      # rw_counter of the first row of rank 1's range (halo of rank 0)
lc, lf, elo, ehi, off = distributed.shard_state(cols, flags, rank, world)
st = np.zeros(lc.shape[1], dtype=np.uint32)
lib.sim_state_verify_range(vp(lc), vp(lf), ctypes.c_uint64(lc.shape[1]), vp(mpt), ctypes.c_uint64(mpt.shape[0]),
                           ctypes.c_uint64(elo), ctypes.c_uint64(ehi), vp(st))
local = st[elo:ehi]
fails = np.nonzero(local)[0]
res = distributed.reduce_tally(len(fails), int(fails[0]) if len(fails) else None, int(local[fails[0]]) if len(fails) else 0, off)
full = np.zeros(4096, dtype=np.uint32)
lib.sim_state_verify(vp(cols), vp(flags), ctypes.c_uint64(4096), vp(mpt), ctypes.c_uint64(mpt.shape[0]), vp(full))
ff = np.nonzero(full)[0]
assert res == (len(ff), int(ff[0]), int(full[ff[0]])), (rank, res, len(ff), ff[:4])
assert np.array_equal(local, full[off:off + len(local)])

# ---- EVM circuit: one global trace, tampered near the shard boundary ------------------------
w = synth_evm_trace(1501, seed=12)
w.pop("meta")
w["steps"][750, 7, 0] += 1     # program counter of the boundary step
w["rw"][40, 8, 0] ^= 1
lw, b, e, off = distributed.shard_evm(w, rank, world)
local = np.array(hostsim_status(lib, lw, (b, e)), dtype=np.uint32)
fails = np.nonzero(local)[0]
res = distributed.reduce_tally(len(fails), int(fails[0]) if len(fails) else None, int(local[fails[0]]) if len(fails) else 0, off)
full = np.array(hostsim_status(lib, w), dtype=np.uint32)
ff = np.nonzero(full)[0]
assert len(ff) >= 2
assert res == (len(ff), int(ff[0]), int(full[ff[0]])), (rank, res)
assert np.array_equal(local, full[off:off + len(local)])
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_sharding_and_tally_allreduce(hostsim, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, ZK_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o}"
        assert f"rank {rank} ok" in o
