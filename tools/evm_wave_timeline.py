#!/usr/bin/env python3
"""Per-wavefront timeline of the hot EVM kernel on bench.py's own config-3 trace (tuning aid): which execution states the
wavefronts' time goes to, how long a wavefront lives, how full the chip is over the kernel's span."""
import ctypes, os, sys, collections
os.environ["ZK_EVM_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd import evm_tables as T
from zkevm_specs_amd.synth_evm import synth_evm_trace

n = 1 << int(os.environ.get("LOGN", "18"))
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
lib = _lib.init(0)
w = synth_evm_trace(n, seed=3)
w.pop("meta")
single = os.environ.get("SINGLE_PASS", "0") == "1"  # the one-shot form: staged from the 416-byte step rows (no packed step records)
with engine.open_evm({k: to_dev(v) for k, v in w.items()}, single_pass=single) as s:
    for _ in range(1 if single else 5):
        s.launch()
    r = s.collect()
    buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
    assert lib.zk_debug_read_prof(s._h, ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(-1, 8).astype(np.int64)
t = t[t[:, 3] > 0]
# slots: 0 gadget-driver entry, 1 after the common checks, 2 after the gadget, 3 end (core clock); 4 execution state;
# 5 kernel entry (core clock); 6 / 7 kernel entry / exit on the 100 MHz wall clock (comparable across XCDs)
dur = t[:, 3] - t[:, 5]
w0 = t[:, 6].min()
ws, we = (t[:, 6] - w0) * 10, (t[:, 7] - w0) * 10  # ns
span = we.max()
print(f"kernel_ms {r.kernel_ms:.4f}  waves {len(t)}  first entry -> last exit {span / 1e3:.1f} us   mean wave {dur.mean():.0f} core ticks = {(we - ws).mean() / 1e3:.1f} us"
      f"  core clock ~{(dur / np.maximum(we - ws, 10)).mean():.2f} GHz   wave-us / span = {(we - ws).sum() / span:.0f} waves in flight on average")
by = collections.defaultdict(list)
for row, d, a, b in zip(t, dur, ws, we):
    by[int(row[4])].append((d, a, b, row[0] - row[5], row[1] - row[0], row[2] - row[1], row[3] - row[2]))
print("state            waves  median  stage  common  gadget  tail   share   first-start .. last-end (us)")
tot = dur.sum()
for st, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    a = np.array(v)
    name = T.ExecutionState(st).name if st in [int(e) for e in T.ExecutionState] else str(st)
    print(f"{name:16s} {len(v):5d} {int(np.median(a[:, 0])):7d} {int(np.median(a[:, 3])):6d} {int(np.median(a[:, 4])):7d} {int(np.median(a[:, 5])):7d} {int(np.median(a[:, 6])):5d}"
          f"  {a[:, 0].sum() / tot:6.1%}   {a[:, 1].min() / 1e3:7.1f} .. {a[:, 2].max() / 1e3:7.1f}")
edges = np.linspace(0, span, 21)
alive = [int(((ws < e1) & (we > e0)).sum()) for e0, e1 in zip(edges[:-1], edges[1:])]
print("waves alive per 5% slice of the span:", alive)
starts = np.sort(ws)
print("wave entry times (us) percentiles 0/25/50/75/100:", [round(float(np.percentile(starts, q)) / 1e3, 1) for q in (0, 25, 50, 75, 100)])
order = np.argsort(-we)[:16]
print("last wavefronts to exit: (state, entry us, exit us, core ticks)")
for i in order:
    st = int(t[i, 4])
    name = T.ExecutionState(st).name if st in [int(e) for e in T.ExecutionState] else str(st)
    print(f"  {name:12s} {ws[i] / 1e3:7.1f} {we[i] / 1e3:7.1f} {int(dur[i]):8d}")
for st, v in sorted(by.items(), key=lambda kv: -max(x[0] for x in kv[1]))[:8]:
    a = np.array(v)
    name = T.ExecutionState(st).name if st in [int(e) for e in T.ExecutionState] else str(st)
    print(f"  {name:12s} ticks min/median/max {int(a[:, 0].min())} {int(np.median(a[:, 0]))} {int(a[:, 0].max())}")
