#!/usr/bin/env python3
"""Which pairs of bench.py's config-3 trace does the fast (hot) EVM kernel hand to the general build?  (tuning aid)"""
import collections, ctypes, os, sys
os.environ["ZK_EVM_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd import evm_tables as T
from zkevm_specs_amd.synth_evm import synth_evm_trace

n = 1 << int(os.environ.get("LOGN", "18"))
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
lib = _lib.init(0)
w = synth_evm_trace(n, seed=3)
w.pop("meta")
with engine.open_evm({k: to_dev(v) for k, v in w.items()}) as s:
    r = s.run()
    cnt = ctypes.c_uint32()
    lst = np.zeros(n, dtype=np.uint32)
    assert lib.zk_debug_read_deferred(s._h, ctypes.byref(cnt), ctypes.c_void_p(lst.ctypes.data), ctypes.c_uint32(n)) == 0
    buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
    if lib.zk_debug_read_prof(s._h, ctypes.c_void_p(buf.ctypes.data)) == 0:
        print("reasons (ZK_DEFER_DEBUG builds, ZK_EVM_PROF=1):", {k: int(buf[4095 * 8 - 16 + k]) for k in range(16) if buf[4095 * 8 - 16 + k]})
print("result", r, "deferred", cnt.value)
states = collections.Counter(int(w["steps"][i, 0, 0]) for i in lst[: cnt.value])
for st, c in states.most_common():
    name = T.ExecutionState(st).name if st in [int(e) for e in T.ExecutionState] else str(st)
    print(f"{name:16s} {c}")
