#!/bin/bash
# Run on the GPU box (via gpurun): collect the rocprofv3 evidence bench.py's roofline block cites.
#   pass 1: --kernel-trace --stats          -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE                -> HBM read traffic per dispatch
#   pass 3: --pmc WRITE_SIZE                -> HBM write traffic per dispatch
# (counter passes are separate from the trace pass, as the MI355X guide prescribes)
# usage: tools/profile_bench.sh <tag> [bench.py args...]
set -u
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --no-cold-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $cmd > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $cmd > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $cmd > "$out/pmc_write.log" 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    res["kernel_stats"] = list(csv.DictReader(open(f)))
    import shutil; shutil.copy(f, out + "/kernel_stats.csv")
for name in ("fetch", "write"):
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
PY
tail -1 "$out/trace.log"
head -8 "$out/kernel_stats.csv"
head -6 "$out/pmc_fetch_summary.csv"
head -6 "$out/pmc_write_summary.csv"
# keep only the summaries (the raw traces are large)
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write"
