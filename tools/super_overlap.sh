#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of the super-circuit bench, summarised as
#   * per-kernel average duration (kernel_stats.csv)
#   * busy time of the circuit kernels: sum of durations vs union of their [start, end] intervals - the difference
#     is what the per-circuit HIP streams overlap (zkevm_specs_amd/super_circuit.py)
# usage: tools/super_overlap.sh <tag> [bench.py args...]
set -u
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --workload super --no-cpu-baseline --no-cold-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $cmd > "$out/trace.log" 2>&1
python - "$out" <<'PY'
import csv, glob, json, shutil, sys
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, out + "/kernel_stats.csv")
names = ("evm_steps_kernel", "evm_state_hist_kernel", "evm_state_scatter_kernel", "state_rows_dma_kernel", "bytecode_rows_kernel", "sign_units_kernel")
iv = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(n in r["Kernel_Name"] for n in names):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
total = sum(e - s for s, e in iv)
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
res = {"dispatches": len(iv), "sum_of_kernel_durations_ms": total / 1e6, "union_of_kernel_intervals_ms": union / 1e6,
       "overlapped_ms": (total - union) / 1e6, "overlap_fraction_of_sum": (total - union) / total if total else None}
json.dump(res, open(out + "/overlap.json", "w"), indent=1)
print(json.dumps(res))
PY
tail -1 "$out/trace.log"
head -10 "$out/kernel_stats.csv" | cut -d, -f1-4
rm -rf "$out/trace"
