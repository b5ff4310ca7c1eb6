#!/usr/bin/env python3
"""Does a resident SuperCircuit in the same process slow zk_block_verify down?  one-shot | open SuperCircuit + passes | one-shot | close | one-shot"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zkevm_specs_amd.block import stage_block, verify_block_native  # noqa: E402
from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block  # noqa: E402

parts = synth_super_block(20, seed=5)
dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
blocks = [stage_block(parts, dev) for _ in range(3)]


def oneshot(tag):
    t = []
    for r in range(11):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, total, ends = verify_block_native(blocks[r % 3], 0)
        t.append((time.perf_counter() - t0) * 1e3)
        assert total == 0
    t = sorted(t[3:])
    print(f"{tag}: median {t[len(t) // 2]:.3f} min {t[0]:.3f}  chain ends {[round(e, 3) for e in ends[:4]]}", flush=True)


oneshot("fresh process")
sc = SuperCircuit(parts, device=0, to_device=dev)
for _ in range(10):
    sc.launch()
    sc.collect()
oneshot("SuperCircuit open")
sc.close()
oneshot("SuperCircuit closed")
