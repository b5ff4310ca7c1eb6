#!/bin/bash
out=gpurun_out/diag; mkdir -p $out
for only in "" "0" "1" "2"; do
cd /tmp && export TMPDIR=/tmp
ONLY="$only," timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/tools/evm_cold_cost.py > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
grep "steps" $out/trace.log
f=$(find $out/trace -name '*kernel_stats.csv' | head -1); grep "evm_steps_kernel\|evm_deferred" "$f" | cut -d, -f1-4,6,7 ; rm -rf $out/trace
done
