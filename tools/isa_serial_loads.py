#!/usr/bin/env python3
"""Which source lines own the global loads that sit alone (or as one 32-byte cell = two loads) in front of a `s_waitcnt vmcnt(0)`?
Input: `hipcc -S --offload-device-only -gline-tables-only` output.  A load cluster of 1-2 followed by a full wait is a memory
latency nothing else shares; on a hot path it is worth an explicit batch.  usage: isa_serial_loads.py file.s [kernel-substring]"""
import re, sys, collections
want = sys.argv[2] if len(sys.argv) > 2 else ""
files, inside, cur = {}, False, None
pending = []  # (line of first load, loc)
loc = None
hits = collections.Counter()
inl = collections.Counter()
for ln in open(sys.argv[1]):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        inside = want in m.group(1)
        pending = []
        continue
    if not inside:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = ln.strip()
    if t.startswith("global_load") or t.startswith("flat_load"):
        pending.append(loc)
    elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
        if 0 < len(pending) <= 2:
            hits[pending[0]] += 1
        pending = []
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
        pending = []
    elif t.startswith("s_endpgm"):
        inside = False
for (f, l), n in hits.most_common(40):
    print(f"{n:5d}  {f}:{l}")
print("total serial clusters:", sum(hits.values()))
