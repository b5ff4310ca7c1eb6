#!/usr/bin/env python3
"""Per-row clocks of the State kernel's Storage / Account branch (needs a -DZK_STATE_PROF=2 build of k_state.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth import synth_state_witness
n = 1 << int(os.environ.get("LOGN", "16"))
_lib.init(0)
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
rows, flags, mpt = synth_state_witness(n, seed=2)
with engine.open_state(to_dev(rows), to_dev(flags), to_dev(mpt)) as s:
    for _ in range(3): s.launch()
    s.collect()
    st = s.read_status()
tags = rows[2, :, 0]
for t in (4, 6):
    m = tags == t
    a, b = (st[m] & 0xffff).astype(np.int64) * 16, (st[m] >> 16).astype(np.int64) * 16
    print(f"tag {t}: rows {int(m.sum())}; next-row compare clocks median {int(np.median(a))} p90 {int(np.percentile(a, 90))} max {int(a.max())};"
          f" lookups {int((b > 0).sum())}: clocks median {int(np.median(b[b > 0]))} p90 {int(np.percentile(b[b > 0], 90))} max {int(b.max())}")
