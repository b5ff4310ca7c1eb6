#!/bin/bash
# Run on the GPU box (via gpurun): collect the rocprofv3 evidence bench.py's roofline block cites.
#   pass 1: --kernel-trace --stats          -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE                -> HBM-side read traffic per dispatch
#   pass 3: --pmc WRITE_SIZE                -> HBM-side write traffic per dispatch
#   pass 4: --pmc SQ_* (8 SQ slots)         -> VALU instructions / VALU-active and wave quad-cycles per dispatch
# (counter passes are separate from the trace pass, as the MI355X guide prescribes; FETCH_SIZE and WRITE_SIZE do not
#  fit one pass).  Writes gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_*_summary.csv, summary.json}.
# usage: tools/profile_bench.sh <tag> [bench.py args...]
set -u
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --no-cold-leg --no-fresh-leg $*"
if [ -n "${PROFILE_CMD:-}" ]; then cmd="$PROFILE_CMD"; fi  # e.g. PROFILE_CMD="python tools/bench_row_kernels.py" (run from the repo root)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $cmd > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $cmd > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $cmd > "$out/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d "$out/pmc_sq" -- $cmd > "$out/pmc_sq.log" 2>&1
python - "$out" "$tag" "$*" <<'PY'
import csv, glob, sys, collections, json
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
summary = {"tag": tag, "command": f"tools/profile_bench.sh {tag} {args}".strip(), "bench_line": None, "kernels": {}}
for line in open(out + "/trace.log"):
    if line.startswith("{") and '"metric"' in line:
        summary["bench_line"] = json.loads(line)
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    import shutil; shutil.copy(f, out + "/kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        summary["kernels"].setdefault(r["Name"], {})["trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                                    "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
for name in ("fetch", "write", "sq"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(out + f"/pmc_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    with open(out + f"/pmc_{name}_summary.csv", "w") as g:
        g.write("kernel,counter,dispatches,sum,avg_per_dispatch\n")
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            g.write(f"\"{k}\",{c},{n},{s},{s/n}\n")
            summary["kernels"].setdefault(k, {}).setdefault("pmc", {})[c] = {"dispatches": n, "avg_per_dispatch": s / n}
# keep the kernels that matter (>= 1 % of the traced time) to keep the file small
summary["kernels"] = {k: v for k, v in summary["kernels"].items() if v.get("trace", {}).get("pct", 0) >= (0.2 if "row_kernels" in tag else 1.0)}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
PY
tail -1 "$out/trace.log"
head -8 "$out/kernel_stats.csv"
head -6 "$out/pmc_fetch_summary.csv"
head -6 "$out/pmc_write_summary.csv"
head -12 "$out/pmc_sq_summary.csv"
# keep only the summaries (the raw traces are large)
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq"
