#!/usr/bin/env python3
"""State witness assignment timings (device pointers in and out): zk_state_assign over 2^20 synthetic ops, and
zk_state_assign_from_rw over the RW table of the 2^18-step block trace (re-keying + sort + assignment in one session),
zk_state_verify_from_rw over the same table (rows evaluated where they are computed) next to the two-step form it replaces."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_specs_amd import engine  # noqa: E402
from zkevm_specs_amd.synth import synth_state_ops  # noqa: E402
from zkevm_specs_amd.synth_block import synth_block_trace  # noqa: E402

dev = torch.device("cuda:0")
up = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).to(dev)  # noqa: E731
out = {}
ops, oflags, *_ = synth_state_ops(1 << 20, seed=2)
n = ops.shape[1]
d_ops, d_of = up(ops), up(oflags)
rows, rfl, mpt = torch.empty((57, n, 4), dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 12, 4), dtype=torch.int64, device=dev)
with engine.open_state_assign(d_ops, d_of, rows, rfl, mpt) as a:
    for _ in range(3):
        a.run()
    ms = sorted(a.run().kernel_ms for _ in range(10))
out["state_assign_2p20_ms"] = ms[len(ms) // 2]
w = synth_block_trace(1 << 18, seed=5)
rw, fl = up(w["rw"]), up(w["rw_flags"])
m = int(rw.shape[0])
rows_b, fl_b, mpt_b = torch.empty(57 * 4 * (m + 1), dtype=torch.int64, device=dev), torch.empty(m + 1, dtype=torch.int32, device=dev), torch.empty(48 * (m + 1), dtype=torch.int64, device=dev)
opens, passes = [], []
for r in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = engine.open_state_assign_from_rw(rw, fl, rows_b, fl_b, mpt_b)
    t1 = time.perf_counter()
    res = a.run()
    a.close()
    if r >= 2:
        opens.append((t1 - t0) * 1e3)
        passes.append(res.kernel_ms)
out["from_rw_2p18"] = {"rw_rows": m, "n_ops": a.n, "open_wall_ms": sorted(opens)[len(opens) // 2], "pass_kernel_ms": sorted(passes)[len(passes) // 2]}
# the State circuit's verdict straight from the RW table (rows evaluated where they are computed), and the two-step form it replaces
opens, passes, walls = [], [], []
for r in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = engine.open_state_verify_from_rw(rw, fl)
    t1 = time.perf_counter()
    res = s.run()
    s.close()
    t2 = time.perf_counter()
    assert res.ok
    if r >= 2:
        opens.append((t1 - t0) * 1e3)
        passes.append(res.kernel_ms)
        walls.append((t2 - t0) * 1e3)
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
out["verify_from_rw_2p18"] = {"n_ops": s.n, "open_wall_ms": med(opens), "pass_kernel_ms": med(passes), "call_wall_ms": med(walls)}
walls = []
for r in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with engine.open_state_assign_from_rw(rw, fl, rows_b, fl_b, mpt_b) as a:
        assert a.run().ok
        k, nm = a.n, a.n_mpt()
    with engine.open_state(rows_b[: 57 * 4 * k].view(57, k, 4), fl_b[:k], mpt_b[: 48 * nm].view(nm, 12, 4)) as s:
        assert s.run().ok
    if r >= 2:
        walls.append((time.perf_counter() - t0) * 1e3)
out["assign_then_verify_2p18"] = {"call_wall_ms": med(walls)}
print(json.dumps(out))
