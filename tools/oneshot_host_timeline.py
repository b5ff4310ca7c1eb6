"""Where the host-visible time of a one-shot zk_evm_verify goes, from a rocprofv3 --hip-trace --kernel-trace run of bench.py:
per step (one evm_open_fill_kernel dispatch each): first HIP call of the step -> fill kernel start, kernel busy time, gaps between
kernels, last kernel end -> end of the step's hipStreamSynchronize.  usage: python tools/oneshot_host_timeline.py <trace dir>"""
import csv
import glob
import statistics
import sys

d = sys.argv[1]
kern, api = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    kern += list(csv.DictReader(open(f)))
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    api += list(csv.DictReader(open(f)))
kern.sort(key=lambda r: int(r["Start_Timestamp"]))
api.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(kern), "kernel dispatches,", len(api), "HIP calls")
fills = [i for i, r in enumerate(kern) if "evm_open_fill_kernel" in r["Kernel_Name"]]
rows = []
for a, b in zip(fills[5:-1], fills[6:]):  # skip the warm-up steps
    ks = [r for r in kern[a:b] if "rocclr" not in r["Kernel_Name"]]
    t0, t1 = int(ks[0]["Start_Timestamp"]), int(ks[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
    # the step's HIP calls: from the last hipStreamSynchronize that ended before t0 to the first one that ends after t1
    prev_sync_end = max((int(r["End_Timestamp"]) for r in api if r["Function"] == "hipStreamSynchronize" and int(r["End_Timestamp"]) <= t0), default=None)
    sync = next((r for r in api if r["Function"] == "hipStreamSynchronize" and int(r["End_Timestamp"]) >= t1), None)
    first_call = next((r for r in api if prev_sync_end is not None and int(r["Start_Timestamp"]) >= prev_sync_end and r["Function"] not in ("hipStreamSynchronize",)), None)
    launch_fill = next((r for r in api if "Launch" in r["Function"] and int(r["Start_Timestamp"]) >= (prev_sync_end or 0)), None)
    copyk = [r for r in kern[a:b] if "rocclr" in r["Kernel_Name"]]
    rows.append({
        "first_call_to_fill_start": (t0 - int(first_call["Start_Timestamp"])) / 1e3 if first_call else None,
        "fill_launch_call_to_fill_start": (t0 - int(launch_fill["Start_Timestamp"])) / 1e3 if launch_fill else None,
        "span": (t1 - t0) / 1e3, "busy": busy / 1e3, "gaps": (t1 - t0 - busy) / 1e3,
        "last_kernel_to_copy_start": (int(copyk[0]["Start_Timestamp"]) - t1) / 1e3 if copyk else None,
        "copy_kernel": (int(copyk[0]["End_Timestamp"]) - int(copyk[0]["Start_Timestamp"])) / 1e3 if copyk else None,
        "last_kernel_to_sync_return": (int(sync["End_Timestamp"]) - t1) / 1e3 if sync else None,
        "sync_return_to_next_first_call": None,
    })
for k in rows[0]:
    v = [r[k] for r in rows if r[k] is not None]
    if v:
        print(f"{k:34s} median {statistics.median(v):8.1f} us   min {min(v):8.1f}   max {max(v):8.1f}")
