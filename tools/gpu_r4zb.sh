#!/bin/bash
set -u
out=$PWD/gpurun_out/r4zb; mkdir -p $out; root=$PWD
timeout 1200 python -m pytest tests/test_keccak_table.py tests/test_bytecode_assign.py tests/test_copy_assign.py tests/test_sign_circuit.py tests/test_row_circuits.py tests/test_super_circuit.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $out/pytest.log | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $root/bench.py --workload super --no-cpu-baseline --steps 5 --warmup 2 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); grep -E "keccak|bca_|assign_|rpow|cpa_|slots_fill" $f | cut -d, -f1-4 | cut -c1-120
rm -rf $out/trace
