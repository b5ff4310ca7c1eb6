#!/usr/bin/env python3
"""Phase timestamps inside the EVM kernel (tuning aid): cycles spent per wave in
[load step cells + prelude] [gadget body] [transition tail] for single-opcode traces."""
import ctypes, os, sys
os.environ["ZK_EVM_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth_evm import synth_evm_trace

n = 1 << int(os.environ.get("LOGN", "16"))
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
lib = _lib.init(0)
for kind in sys.argv[1:] or ["PUSH1", "ADDSUB", "PUSH32", "MEMORY"]:
    partner = "POP" if kind in ("PUSH1", "PUSH32", "READER") else "PUSH1"
    w = synth_evm_trace(n, seed=1, seg_len=200, mix=[(3, kind), (1, partner)])
    w.pop("meta")
    with engine.open_evm({k: to_dev(v) for k, v in w.items()}) as s:
        for _ in range(3):
            s.launch()
        r = s.collect()
        buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
        assert lib.zk_debug_read_prof(s._h, ctypes.c_void_p(buf.ctypes.data)) == 0
    t = buf.reshape(-1, 8).astype(np.int64)
    t = t[t[:, 3] > 0]
    d = np.diff(t[:, :4], axis=1)
    span = (t[:, 3].max() - t[:, 0].min())
    print(kind, "waves", len(t), "median cycles: load+prelude", int(np.median(d[:, 0])), "gadget", int(np.median(d[:, 1])),
          "tail", int(np.median(d[:, 2])), "| per-wave total", int(np.median(t[:, 3] - t[:, 0])), "| first-start to last-end", int(span),
          "| kernel_ms", round(r.kernel_ms, 4))
