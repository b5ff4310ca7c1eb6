#!/usr/bin/env python3
"""The State circuit's resident pass over an RW table (zk_state_verify_from_rw_open kept open: the evaluation kernel alone) on the
2^18-step block trace, next to the 57-cell form over the rows the device assigns from the same table (tuning aid)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_specs_amd import engine  # noqa: E402
from zkevm_specs_amd.synth_block import synth_block_trace  # noqa: E402

dev = torch.device("cuda:0")
up = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).to(dev)  # noqa: E731
w = synth_block_trace(1 << 18, seed=5)
rw, fl = up(w["rw"]), up(w["rw_flags"])
out = {"rw_rows": int(rw.shape[0])}
with engine.open_state_verify_from_rw(rw, fl) as s:
    first = s.run()
    assert first.ok
    ms = sorted(s.run().kernel_ms for _ in range(int(os.environ.get("PASSES", "20"))))
    out["n_ops"] = s.n
    out["first_pass_ms"] = first.kernel_ms
    out["resident_pass_ms"] = {"min": ms[0], "median": ms[len(ms) // 2], "max": ms[-1]}
print(json.dumps(out))
