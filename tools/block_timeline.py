#!/usr/bin/env python3
"""Kernel timeline of the LAST block one-shot in a rocprofv3 --kernel-trace CSV of tools/bench_block_oneshot.py: start / end of every
kernel relative to the first kernel of that block (us), with its stream / queue — where the chains wait for each other."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last block starts at the last rwk_scan_kernel
i0 = max(i for i, r in enumerate(rows) if "rwk_scan_kernel" in r["Kernel_Name"])
# walk back to the earliest kernel within 400 us before it (the other chains may have started first)
t_scan = int(rows[i0]["Start_Timestamp"])
j = i0
while j > 0 and t_scan - int(rows[j - 1]["Start_Timestamp"]) < 400000:
    j -= 1
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
