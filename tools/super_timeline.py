"""Per-queue timeline of the Super circuit's back-to-back passes from a rocprofv3 kernel trace (run on the GPU box):
for every HIP stream (Queue_Id) the busy time, the span and the gaps between consecutive dispatches over the timed passes.
usage: python tools/super_timeline.py <dir with *kernel_trace.csv> [n_last_passes]"""
import collections
import csv
import glob
import sys

d, n_last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
state = [r for r in rows if "state_rows_dma_kernel" in r["Kernel_Name"]]
if len(state) < n_last + 1:
    sys.exit("not enough State dispatches")
t_lo = int(state[-n_last - 1]["End_Timestamp"])   # the last n passes of the State stream
t_hi = int(state[-1]["End_Timestamp"])
print(f"window: last {n_last} State dispatches, {(t_hi - t_lo) / n_last / 1e3:.1f} us per pass")
byq = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t_lo and e <= t_hi + 400_000:
        byq[r["Queue_Id"]].append((s, e, r["Kernel_Name"][:44]))
for q, ev in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in ev)
    names = collections.Counter(n for _, _, n in ev)
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
    big = sorted(gaps)[-3:] if gaps else []
    print(f"queue {q}: {len(ev)} dispatches, busy {busy / n_last / 1e3:.1f} us/pass, median gap {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, "
          f"largest gaps {[round(g / 1e3, 1) for g in big]} us; kernels: " + ", ".join(f"{n} x{c}" for n, c in names.most_common(4)))
# the State queue in detail: start-to-start of consecutive passes
ss = [int(r["Start_Timestamp"]) for r in state[-n_last - 1:]]
ee = [int(r["End_Timestamp"]) for r in state[-n_last - 1:]]
print("State start-to-start us:", [round((ss[i + 1] - ss[i]) / 1e3, 1) for i in range(n_last)])
print("State durations us:     ", [round((ee[i] - ss[i]) / 1e3, 1) for i in range(n_last + 1)])
print("State end -> next start:", [round((ss[i + 1] - ee[i]) / 1e3, 1) for i in range(n_last)])
