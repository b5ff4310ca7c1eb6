#!/bin/bash
set -u
out=gpurun_out/r4z; mkdir -p $out
timeout 900 python -m pytest tests/test_keccak_table.py tests/test_bytecode_assign.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $out/pytest.log | tail -6
for ng in 0 1; do
if [ $ng = 1 ]; then export ZK_KECCAK_NO_GROUPS=1; fi
python tools/bench_row_kernels.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if 'keccak' in k: print('no_groups=$ng', k, v['kernel_ms'], v['units_per_s'])
"
done
