#!/bin/bash
set -u
out=$PWD/gpurun_out/r4za; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $root/bench.py --workload super --no-cpu-baseline --steps 5 --warmup 2 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); grep -E "keccak|bca_|assign_rows|rpow|cpa_" $f | cut -c1-160
rm -rf $out/trace
