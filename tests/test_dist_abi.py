"""The C ABI's multi-rank tally (`zk_dist_unique_id / _init / _tally / _close`, include/zkevm_hip.h) with MORE THAN
ONE RANK and no GPUs: every rank is a process that loads libzkevm_cpu.so (ZK_BACKEND=cpu) and reaches the collective library
through the same binding as the HIP library (csrc/dist_tally.hpp: dlopen + ncclGetUniqueId / ncclCommInitRank / ncclAllGather),
with ZK_RCCL_LIB naming tests/fakerccl's stand-in (an all-gather over a shared-memory file).  World 2 and world 8 with uneven
shards: `zk_dist_tally` on every rank == the single-process tally == zkevm_specs_amd.distributed.reduce_tally's definition
(SUM of the counts, first failing GLOBAL row with its code, rows SUM, kernel_ms MAX).  The GPU leg of the same test
(tests/test_bench_multi_gpu_dryrun.py) runs the HIP library's entries with two ranks on one GPU through libfakerccl_hip.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_DIR = os.path.join(ROOT, "tests", "fakerccl")


@pytest.fixture(scope="module")
def fake_host():
    so, src = os.path.join(FAKE_DIR, "libfakerccl_host.so"), os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([os.path.join(FAKE_DIR, "build.sh")])
    return so


WORKER = r'''
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["ZK_ROOT"])
from zkevm_specs_amd import _lib, distributed, engine
from zkevm_specs_amd.synth import synth_state_witness
assert _lib.BACKEND == "cpu"
rank, world, idf = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["ZK_ID_FILE"]
lib = _lib.init("cpu")
# the communicator id travels by a file here (a host with no torch.distributed: exactly the caller these entries are for)
ident = (ctypes.c_uint8 * 128)()
if rank == 0:
    engine.check(lib.zk_dist_unique_id(ident), "zk_dist_unique_id", lib)
    open(idf + ".tmp", "wb").write(bytes(ident)); os.rename(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 60; time.sleep(0.01)
    ident = (ctypes.c_uint8 * 128).from_buffer_copy(open(idf, "rb").read())
h = ctypes.c_void_p()
engine.check(lib.zk_dist_init(ident, rank, world, ctypes.byref(h)), "zk_dist_init", lib)

N = 4099  # not divisible by 2 or 8
cols, flags, mpt = synth_state_witness(N, seed=9)
def tamper(c):
    c = c.copy()
    for r in range(1, world):  # a damaged cell on both sides of every shard boundary
        b = distributed.shard_bounds(N, r, world)[0]
        c[50, b - 1, 0] ^= 1
        c[0, b, 0] ^= 4
    c[1, 3000, 0] = 2
    return c
for case, cc in (("clean", cols), ("tampered", tamper(cols)), ("one_rank_fails", None)):
    if cc is None:
        cc = cols.copy(); cc[1, N - 7, 0] = 2  # only the last rank sees a failure
    with engine.open_state(cc, flags, mpt) as s:  # the whole witness, this rank's range (halo rows are read, not evaluated)
        lo, hi = distributed.shard_bounds(N, rank, world)
        s.set_range(lo, hi)
        local = s.run()
        s.set_range(0, N)
        full = s.run()
    raw, out = _lib.ZkResult(), _lib.ZkResult()
    raw.fail_count = local.fail_count
    raw.first_fail_row = 0xFFFFFFFFFFFFFFFF if local.first_fail_row is None else local.first_fail_row - lo  # row of the shard
    raw.first_fail_code = local.first_fail_code or 0
    raw.rows_evaluated = local.rows_evaluated
    raw.kernel_ms = 1.0 + rank
    engine.check(lib.zk_dist_tally(h, ctypes.byref(raw), lo, ctypes.byref(out)), "zk_dist_tally", lib)
    got = (out.fail_count, None if out.first_fail_row == 0xFFFFFFFFFFFFFFFF else out.first_fail_row, out.first_fail_code)
    assert got == (full.fail_count, full.first_fail_row, full.first_fail_code or 0), (case, rank, got, full.fail_count, full.first_fail_row)
    assert out.rows_evaluated == N and out.kernel_ms == float(world), (case, out.rows_evaluated, out.kernel_ms)
    if case == "tampered":
        assert full.fail_count >= world
    if case == "one_rank_fails":
        assert full.fail_count >= 1 and (local.fail_count > 0) == (rank == world - 1)
    # the Python mirror's wrapper object over the same handle-less path agrees (RcclTally.reduce packs the same words)
assert lib.zk_dist_close(h) == 0
print("rank", rank, "ok")
'''


def _run_world(tmp_path, world, fake):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, ZK_ROOT=ROOT, ZK_BACKEND="cpu", ZK_RCCL_LIB=fake, WORLD_SIZE=str(world), ZK_ID_FILE=str(tmp_path / f"id_{world}"),
               OMP_NUM_THREADS="1", ZK_CPU_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o[-3000:]}"
        assert f"rank {rank} ok" in o


@pytest.mark.parametrize("world", [2, 8])
def test_zk_dist_tally_multi_rank_through_the_collective_binding(tmp_path, fake_host, world):
    _run_world(tmp_path, world, fake_host)


def test_cpu_backend_refuses_world_2_without_a_host_collective(tmp_path):
    code = ("import ctypes, os, sys; sys.path.insert(0, os.environ['ZK_ROOT']); from zkevm_specs_amd import _lib; lib = _lib.init('cpu'); "
            "i = (ctypes.c_uint8 * 128)(); h = ctypes.c_void_p(); assert lib.zk_dist_unique_id(i) == 0; "
            "assert lib.zk_dist_init(i, 0, 2, ctypes.byref(h)) < 0 and b'ZK_RCCL_LIB' in lib.zk_last_error(); "
            "assert lib.zk_dist_init(i, 0, 1, ctypes.byref(h)) == 0 and lib.zk_dist_close(h) == 0; print('ok')")
    env = {k: v for k, v in os.environ.items() if k != "ZK_RCCL_LIB"}
    p = subprocess.run([sys.executable, "-c", code], env=dict(env, ZK_ROOT=ROOT, ZK_BACKEND="cpu"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0 and b"ok" in p.stdout, p.stdout.decode()[-2000:]
