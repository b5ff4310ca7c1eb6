#!/bin/bash
# round-3 GPU check C: EVM parity tests, wave timeline, default bench line (no other configs), kernel trace
set -u
out=gpurun_out/r3c; mkdir -p $out
timeout 1500 python -m pytest tests/test_evm_gpu.py tests/test_dropin_gpu.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python tools/evm_wave_timeline.py > $out/timeline.txt 2>&1; head -32 $out/timeline.txt
timeout 900 python bench.py --no-other-configs --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -c 600 $out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-cold-leg --no-other-configs > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/trace
grep -v "at::native" $out/kernel_stats.csv | head -12
