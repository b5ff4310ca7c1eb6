#!/usr/bin/env python3
"""Per-wavefront clocks of the warm EVM instantiation (SHA3 / *COPY / LOG / EXP steps) on a block trace (tuning aid, ZK_EVM_PROF): the
warm launch runs after the hot one and is a grid of <= 1024 wavefronts, so slots 0..4 of the first 1024 profile records are its
stamps (gadget entry, after the common checks, after the gadget, end, execution state)."""
import ctypes, os, sys, collections
os.environ["ZK_EVM_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zkevm_specs_amd import _lib, engine, synth_block
from zkevm_specs_amd import evm_tables as T
from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block

dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
lib = _lib.init(0)
p = synth_super_block(int(os.environ.get("LOGT", "20")), seed=5)
with SuperCircuit(p, to_device=dev) as sc:
    ev = sc.sessions["evm"]
    ev.run()
    for _ in range(3):
        ev.launch()
    r = ev.collect()
    buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
    assert lib.zk_debug_read_prof(ev._h, ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(-1, 8).astype(np.int64)[:1024]
names = {int(e): e.name for e in T.ExecutionState}
warm = [i for i in range(len(t)) if t[i, 3] > 0 and names.get(int(t[i, 4]), "").upper() in ("SHA3", "CODECOPY", "EXP", "CALLDATACOPY", "RETURNDATACOPY", "EXTCODECOPY", "LOG")]
print("evm span ms", round(r.kernel_ms, 4), "warm wavefront records", len(warm))
by = collections.defaultdict(list)
for i in warm:
    by[names[int(t[i, 4])]].append((t[i, 3] - t[i, 0], t[i, 1] - t[i, 0], t[i, 2] - t[i, 1], t[i, 3] - t[i, 2]))
print("state        waves  median total  common  gadget  tail   max total (core clocks)")
for k, v in by.items():
    a = np.array(v)
    print(f"{k:12s} {len(v):5d} {int(np.median(a[:, 0])):10d} {int(np.median(a[:, 1])):8d} {int(np.median(a[:, 2])):8d} {int(np.median(a[:, 3])):6d} {int(a[:, 0].max()):10d}")

t2 = buf.reshape(-1, 8).astype(np.int64)[2048:3072]
sh = [i for i in warm if names[int(t[i, 4])] == "SHA3" and t2[i, 7] > 0]
if sh:
    d = np.array([[t2[i, k + 1] - t2[i, k] for k in range(7)] for i in sh])
    print("SHA3 (ZK_WARM_STAMPS build) median clocks: opcode_lookup, stack x3, offsets, copy_lookup, keccak_lookup, gas, tail:", [int(x) for x in np.median(d, axis=0)])
