#!/usr/bin/env python3
"""Per-wavefront timeline of the State kernel (tuning aid; needs a -DZK_STATE_PROF build of k_state.hip, ZK_HIP_LIB=...): when each
wavefront starts and ends, how long its load phase and its whole evaluation take in core clocks, which CU it ran on."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkevm_specs_amd import _lib, engine
from zkevm_specs_amd.synth import synth_state_witness
n = 1 << int(os.environ.get("LOGN", "16"))
_lib.init(0)
to_dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).cuda()
rows, flags, mpt = synth_state_witness(n, seed=2)
with engine.open_state(to_dev(rows), to_dev(flags), to_dev(mpt)) as s:
    for _ in range(5): s.launch()
    r = s.collect()
    st = s.read_status()
w = np.arange(0, n - 13, 63)
w = w[st[w + 8] == 0xabcd1234]
t0 = st[w].astype(np.int64) | (st[w + 1].astype(np.int64) << 32)
t1 = st[w + 2].astype(np.int64) | (st[w + 3].astype(np.int64) << 32)
load, total, hw, xcc = st[w + 4].astype(np.int64), st[w + 5].astype(np.int64), st[w + 6], st[w + 7] & 0xf
s0, s1, s2 = st[w + 9].astype(np.int64), st[w + 10].astype(np.int64), st[w + 11].astype(np.int64)  # check entry, before / after the tag switch
base = t0.min()
ws, we = (t0 - base) * 10, (t1 - base) * 10  # ns (100 MHz wall clock)
print(f"kernel_ms {r.kernel_ms:.4f} waves {len(w)}  first entry -> last exit {we.max() / 1e3:.1f} us; mean wave {(we - ws).mean() / 1e3:.1f} us "
      f"= {total.mean():.0f} core clocks ({(total / np.maximum(we - ws, 10)).mean():.2f} GHz), load phase {load.mean():.0f} clocks")
print("entry time percentiles (us) 0/10/25/50/75/90/100:", [round(float(np.percentile(ws, q)) / 1e3, 1) for q in (0, 10, 25, 50, 75, 90, 100)])
print("exit  time percentiles (us) 0/10/25/50/75/90/100:", [round(float(np.percentile(we, q)) / 1e3, 1) for q in (0, 10, 25, 50, 75, 90, 100)])
cu = (xcc.astype(np.int64) << 16) | ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 8)
per = collections.Counter(cu.tolist())
print("distinct (xcc, se, sh, cu):", len(per), " waves per CU min/median/max:", min(per.values()), int(np.median(list(per.values()))), max(per.values()))
simd = (hw >> 4) & 3
print("waves per SIMD id:", collections.Counter(simd.tolist()))
edges = np.linspace(0, we.max(), 21)
print("waves alive per 5% slice:", [int(((ws < e1) & (we > e0)).sum()) for e0, e1 in zip(edges[:-1], edges[1:])])
tags = rows[2, :, 0].astype(np.int64)
order = np.argsort(-we)[:24]
print("slowest wavefronts: first row, exit us, load clocks, total clocks, tags of its rows")
for k in order:
    f = int(w[k])
    print(f"  row {f:7d}  exit {we[k] / 1e3:6.1f}  load {int(load[k]):7d}  check-entry {int(s0[k]):7d} switch {int(s1[k]):7d} after {int(s2[k]):7d} total {int(total[k]):7d}  tags {dict(collections.Counter(tags[f:f + 63].tolist()))}")
ex = [st[w + 9 + k].astype(np.int64) for k in (3, 4, 5, 6)]  # after next-row keys, before lookup, after lookup, after hash
print("by dominant tag: waves, median total clocks, max")
dom = np.array([collections.Counter(tags[int(f):int(f) + 63].tolist()).most_common(1)[0][0] for f in w])
for t in sorted(set(dom.tolist())):
    m = dom == t
    print(f"  tag {t}: {int(m.sum()):5d} waves  median {int(np.median(total[m])):7d}  max {int(total[m].max()):7d}  median load {int(np.median(load[m])):7d}  check-entry {int(np.median(s0[m])):7d} switch {int(np.median(s1[m])):7d} after {int(np.median(s2[m])):7d}" + (f"  | keys {int(np.median(ex[0][m]))} q {int(np.median(ex[1][m]))} hash {int(np.median(ex[3][m]))} lookup-done {int(np.median(ex[2][m]))}" if t in (4, 6) else ""))
