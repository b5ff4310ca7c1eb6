#!/usr/bin/env python3
"""Wall clock of block.BlockVerifier.verify on BASELINE config 5's block: everything derived on the device (keccak table, the
Bytecode / Copy / State assignments incl. the RW -> State sort, six opens), one pass of each circuit, collects, closes.
Rotates over `copies` device-resident copies of the inputs so that no call finds its block in the caches."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_specs_amd import _lib  # noqa: E402

_lib.load()
from zkevm_specs_amd.block import BlockVerifier, stage_block, verify_block_native  # noqa: E402
from zkevm_specs_amd.super_circuit import synth_super_block  # noqa: E402

log_total = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
copies = int(sys.argv[3]) if len(sys.argv) > 3 else 3
parts = synth_super_block(log_total, seed=5)
dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()  # noqa: E731
blocks = [stage_block(parts, dev) for _ in range(copies)]
compact = os.environ.get("ZK_STATE_COMPACT", "0") == "1"
state_rows = os.environ.get("ZK_BLOCK_STATE_ROWS", "0") == "1"  # native entry: write the 57-cell State witness and read it back (round 6's first form)
native = os.environ.get("ZK_BLOCK_NATIVE", "0") == "1"  # zk_block_verify (the chains on threads inside the library) instead of block.py's Python threads
bv = BlockVerifier(0, state_compact=compact)
last_trace = []
times = []
for r in range(reps + 3):
    b = blocks[r % copies]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if native:
        results, total, ends = verify_block_native(b, 0, compact, state_rows)
    else:
        results, total = bv.verify(b)
    t1 = time.perf_counter()
    assert total == 0, {k: (v.fail_count, v.first_fail_row, v.first_fail_code) for k, v in results.items()}
    if r >= 3:
        times.append((t1 - t0) * 1e3)
        last_trace = [("native", f"chain {c} start {ends[4 + k]:.3f} end", t) for k, (c, t) in enumerate(zip(("state", "keccak", "copy", "rest"), ends[:4]))] + [("native", "all chains ended", ends[8]), ("native", "return", ends[9])] if native else sorted(bv.trace, key=lambda e: e[2])
times.sort()
rows = {k: v.rows_evaluated for k, v in results.items()}
print(json.dumps({"block_rows": rows, "total_rows": sum(rows.values()), "reps": reps, "copies": copies,
                  "oneshot_ms": {"median": times[len(times) // 2], "min": times[0], "max": times[-1]},
                  "kernel_ms": {k: v.kernel_ms for k, v in results.items()}}))
for c, what, t in last_trace:
    print(f"{t:8.3f} ms  {c:7s} {what}")
bv.close()
