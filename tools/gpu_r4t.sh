#!/bin/bash
set -u
out=gpurun_out/r4t; mkdir -p $out
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -x -q -k "tally or independent" > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $out/pytest.log | tail -5
timeout 600 python bench.py --workload state --tally abi --no-cpu-baseline --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json; grep -v "^$" $out/bench.err | grep -iv "rccl\|version\|hostname\|librccl" | tail -5
