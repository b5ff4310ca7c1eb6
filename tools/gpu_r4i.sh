#!/bin/bash
set -u
out=$PWD/gpurun_out/r4i; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do
ZK_EVM_WIRE=$w rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_w$w -- python $root/bench.py --no-other-configs --no-cpu-baseline --no-fresh-leg --no-cold-leg --no-session-leg --no-batch-leg --steps 20 --warmup 5 > $out/trace_w$w.log 2>&1
f=$(find $out/trace_w$w -name "*kernel_stats.csv" | head -1); echo "== wire=$w"; head -12 $f | cut -c1-200
grep '"metric"' $out/trace_w$w.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_ms'], r['open_ms'], r['pass_kernel_ms'])"
rm -rf $out/trace_w$w
done
