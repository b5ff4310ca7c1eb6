#!/usr/bin/env python3
"""Does the power-of-two column stride of the State witness (57 columns n * 32 B apart) cost HBM efficiency?  The same
kernel on n = 2^20 and on nearby row counts that are not powers of two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth import synth_state_witness
_lib.init(0)
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
for n in ((1 << 20), (1 << 20) + 8 * 63, (1 << 20) - 64 * 1000 + 77, 1000003):
    rows, flags, mpt = synth_state_witness(n, seed=2)
    with engine.open_state(to_dev(rows), to_dev(flags), to_dev(mpt)) as s:
        for _ in range(3): s.launch()
        s.collect()
        for _ in range(20): s.launch()
        r = s.collect()
        assert r.ok
        print(n, round(r.kernel_ms, 4), "ms", round(n * 57 * 32 / r.kernel_ms / 1e9, 3), "TB/s algorithmic", flush=True)
