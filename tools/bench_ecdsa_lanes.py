#!/usr/bin/env python3
"""ECDSA verification kernel time by batch size and lanes per signature (ZK_ECDSA_LANES = 1 | 2 | 4; tuning aid)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_specs_amd import engine  # noqa: E402
from zkevm_specs_amd.synth import synth_signatures  # noqa: E402

n0 = 1 << 11
sigs = synth_signatures(n0, 9)
packed = np.frombuffer(b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") + z.to_bytes(32, "big") + rr.to_bytes(32, "little") +
                                ss.to_bytes(32, "little") for x, y, z, rr, ss in sigs), dtype=np.uint8).reshape(n0, 5, 32).copy()
out = {}
for logn in (11, 12, 13, 14, 15):
    d = torch.from_numpy(np.tile(packed, (1 << (logn - 11), 1, 1))).cuda()
    for lanes in ("1", "2", "4"):
        os.environ["ZK_ECDSA_LANES"] = lanes
        with engine.open_ecdsa(d) as s:
            for _ in range(2):
                s.launch()
            s.collect()
            for _ in range(6):
                s.launch()
            r = s.collect()
            assert r.ok, (logn, lanes, r)
        out[f"2p{logn}_L{lanes}"] = round(r.kernel_ms, 4)
        print(f"2^{logn} signatures, {lanes} lane(s) per signature: {r.kernel_ms:.4f} ms", flush=True)
print(json.dumps(out))
