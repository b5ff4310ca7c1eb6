#!/bin/bash
set -u
out=$PWD/gpurun_out/r4u; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $out/trace -- python $root/bench.py --no-cpu-baseline --no-other-configs --no-session-leg --no-batch-leg --no-cold-leg --no-fresh-leg --steps 30 --warmup 5 > $out/trace.log 2>&1
ls $out/trace/*/ | head
python $root/tools/oneshot_host_timeline.py $out/trace
rm -rf $out/trace
