#!/usr/bin/env python3
"""Time zk_state_ops_from_rw on the RW table of the 2^18-step block trace (device pointers in and out):
open (class scan + plan, one host round trip) and the pass (pack + radix passes + op list), wall clock and device time."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from zkevm_specs_amd import engine  # noqa: E402
from zkevm_specs_amd.synth_block import synth_block_trace  # noqa: E402

log_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = synth_block_trace(1 << log_steps, seed=5)
rw, fl = w["rw"], w["rw_flags"]
n = int(rw.shape[0])
dev = torch.device("cuda:0")
d_rw = torch.from_numpy(rw.view(np.int64)).to(dev)
d_fl = torch.from_numpy(fl.view(np.int32)).to(dev)
d_ops = torch.empty(48 * (n + 1), dtype=torch.int64, device=dev)
d_of = torch.empty(n + 1, dtype=torch.int32, device=dev)
out = {"rw_rows": n, "reps": reps, "open_ms": [], "pass_wall_ms": [], "pass_kernel_ms": [], "total_ms": []}
for r in range(reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = engine.open_state_ops_from_rw(d_rw, d_fl, d_ops, d_of)
    t1 = time.perf_counter()
    res = s.run()
    t2 = time.perf_counter()
    s.close()
    assert res.ok
    if r >= 2:
        out["open_ms"].append((t1 - t0) * 1e3)
        out["pass_wall_ms"].append((t2 - t1) * 1e3)
        out["pass_kernel_ms"].append(res.kernel_ms)
        out["total_ms"].append((t2 - t0) * 1e3)
out["n_ops"] = s.n_ops
for k in ("open_ms", "pass_wall_ms", "pass_kernel_ms", "total_ms"):
    v = sorted(out[k])
    out[k] = {"median": v[len(v) // 2], "min": v[0], "max": v[-1]}
print(json.dumps(out))
