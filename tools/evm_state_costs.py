#!/usr/bin/env python3
"""Per-gadget cost of the EVM kernel: time single-opcode traces on the GPU (tuning aid)."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth_evm import synth_evm_trace

KINDS = sys.argv[1:] or ["PUSH1", "PUSH32", "POP", "ADDSUB", "MULDIVMOD", "CMP", "SCMP", "BITWISE", "NOT", "ISZERO", "BYTE",
                         "SIGNEXTEND", "SHIFT", "ADDMOD", "MULMOD", "MEMORY", "SLOAD", "SSTORE", "READER"]
n = 1 << int(os.environ.get("LOGN", "16"))
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
_lib.init(0)
out = {}
for kind in KINDS:
    # pair every opcode with a stack-neutralising partner so the stack pointer stays in range
    partner = "POP" if kind in ("PUSH1", "PUSH32", "READER") else "PUSH1"
    w = synth_evm_trace(n, seed=1, seg_len=200, mix=[(3, kind), (1, partner)] if kind not in ("POP",) else [(1, "POP"), (1, "PUSH1")])
    meta = w.pop("meta")
    with engine.open_evm({k: to_dev(v) for k, v in w.items()}) as s:
        for _ in range(3):
            s.launch()
        s.collect()
        for _ in range(10):
            s.launch()
        r = s.collect()
        assert r.ok, (kind, r)
    out[kind] = {"ms": round(r.kernel_ms, 4), "ns_per_step": round(r.kernel_ms * 1e6 / n, 2), "rw_rows": meta["n_rw"],
                 "GBps": round(meta["algorithmic_bytes"] / r.kernel_ms / 1e6, 1)}
    print(kind, out[kind], flush=True)
print(json.dumps(out))
