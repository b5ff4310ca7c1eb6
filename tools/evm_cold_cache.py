#!/usr/bin/env python3
"""Diagnostic: EVM hot-kernel time with warm vs cold caches.  bench.py repeats the pass over the same witness, so
part of the RW / step rows can still sit in the 256 MB Infinity Cache from the previous pass; a fresh witness (or the
super circuit, whose other kernels stream > 1 GB in between) does not get that.  Between two passes this tool
streams `--flush-mb` of unrelated data on the same stream; kernel_ms (HIP events around the EVM kernels only) is
then the cold-cache figure."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth_evm import synth_evm_trace

ap = argparse.ArgumentParser()
ap.add_argument("--log-rows", type=int, default=18)
ap.add_argument("--flush-mb", type=int, default=2048)
ap.add_argument("--passes", type=int, default=20)
ap.add_argument("--clean", action="store_true", help="flush with reads only (no dirty lines to write back)")
args = ap.parse_args()
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()  # noqa: E731
_lib.init(0)
_st = torch.cuda.Stream()  # a real stream shared with torch: the flush and the pass are ordered on it
torch.cuda.set_stream(_st)
_lib.check(_lib.load().zk_set_stream(_st.cuda_stream), "zk_set_stream")
w = synth_evm_trace(1 << args.log_rows, seed=3)
meta = w.pop("meta")
dev_w = {k: to_dev(v) for k, v in w.items()}
flush = torch.zeros(args.flush_mb << 18, dtype=torch.int32, device="cuda")
out = {}
for sort in (True,):
  sess = engine.open_evm(dev_w, state_sort=sort)
  for mode in ("warm", "cold"):
    for _ in range(3):
        sess.launch()
    sess.collect()
    for _ in range(args.passes):
        if mode == "cold":
            flush.sum() if args.clean else flush.add_(1)
        sess.launch()
    r = sess.collect()
    assert r.ok
    out[("sorted_" if sort else "trace_order_") + mode] = {"kernel_ms": r.kernel_ms, "algorithmic_GBps": meta["algorithmic_bytes"] / r.kernel_ms / 1e6}
  sess.close()
print(json.dumps(out))
