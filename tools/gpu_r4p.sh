#!/bin/bash
set -u
out=$PWD/gpurun_out/r4p; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $root/bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 > $out/trace.log 2>&1
python $root/tools/super_timeline.py $out/trace 12
rm -rf $out/trace
