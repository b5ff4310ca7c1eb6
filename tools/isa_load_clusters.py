#!/usr/bin/env python3
"""How a kernel's global loads are grouped between s_waitcnt vmcnt(...) instructions (from `hipcc -S --offload-device-only` output):
a cluster of 1 is a load whose latency nothing else shares.  usage: isa_load_clusters.py file.s [kernel-name-substring]"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur, inside = None, False
stats = collections.defaultdict(lambda: collections.Counter())
pending = 0
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur, inside, pending = m.group(1), want in m.group(1), 0
        continue
    if not inside:
        continue
    t = ln.strip()
    if t.startswith("global_load") or t.startswith("flat_load") or t.startswith("buffer_load"):
        pending += 1
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
        if pending:
            stats[cur][pending] += 1
        pending = 0
    elif t.startswith("s_endpgm"):
        inside = False
for k, c in stats.items():
    tot = sum(n * v for n, v in c.items())
    print(k[:70], "loads", tot, "waits-with-loads", sum(c.values()))
    print("   cluster size: count ->", dict(sorted(c.items())))
