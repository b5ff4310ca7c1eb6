#!/bin/bash
# round-3 GPU check A: new tests, default bench line, kernel trace of a fresh-witness run
set -u
out=gpurun_out/r3a; mkdir -p $out
timeout 1500 python -m pytest tests/test_evm_gpu.py tests/test_bench_multi_gpu_dryrun.py tests/test_super_circuit.py tests/test_dropin_gpu.py tests/test_copy_circuit.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -c 1500 $out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-cold-leg --no-other-configs > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/trace
head -40 $out/kernel_stats.csv
