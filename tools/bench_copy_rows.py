#!/usr/bin/env python3
"""Copy-circuit kernels alone (tuning runs: block size through ZK_COPY_BLOCK, alternative builds through ZK_HIP_LIB): copy_assign +
copy_rows at 2^15 and 2^19 events' worth of rows, the same cases as tools/bench_row_kernels.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth import synth_copy_events

to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
_lib.init(0)
out = {"lib": os.path.basename(_lib.LIB_PATH), "ZK_COPY_BLOCK": os.environ.get("ZK_COPY_BLOCK")}
for kk in (15, 19):
    ce = synth_copy_events(1 << kk, seed=6)
    ev, fl, da, of = to_dev(ce["events"]), to_dev(ce["flags"]), torch.from_numpy(ce["data"].view(np.int16)).cuda(), to_dev(ce["offsets"])
    n_rows, n_table, n_rw = engine.copy_assign_sizes(ce["events"], ce["flags"], ce["data"], ce["offsets"])
    c_rows = torch.empty((20, n_rows, 4), dtype=torch.int64, device="cuda")
    c_rf = torch.empty(n_rows, dtype=torch.int32, device="cuda")
    c_rw = torch.empty((n_rw, 14, 4), dtype=torch.int64, device="cuda")
    c_rwf = torch.empty(n_rw, dtype=torch.int32, device="cuda")
    with engine.open_copy_assign(ev, fl, da, of, ce["r"], c_rows, c_rf, None, c_rw, c_rwf) as s:
        assert s.run().ok
    with engine.open_copy(c_rows, c_rf, ce["r"], c_rw, c_rwf, to_dev(ce["bytecode"]), to_dev(ce["tx"]), to_dev(ce["tx_flags"])) as s:
        for _ in range(3):
            s.launch()
        s.collect()
        for _ in range(20):
            s.launch()
        r = s.collect()
        assert r.ok
    out[f"copy_rows_2p{kk}"] = {"rows": n_rows, "kernel_us": round(r.kernel_ms * 1e3, 1), "frac_of_8TBps": round(n_rows * (20 + 14) * 32 / r.kernel_ms / 1e6 / 8000, 3)}
print(json.dumps(out))
