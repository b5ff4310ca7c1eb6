#!/bin/bash
set -u
out=gpurun_out/r4v; mkdir -p $out
timeout 900 python -m pytest tests/test_ecdsa.py tests/test_sign_circuit.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log | tail -2
python tools/bench_row_kernels.py > $out/row_kernels.txt 2>&1; tail -1 $out/row_kernels.txt > $out/row_kernels.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4v/row_kernels.json'))
for k,v in d.items():
    if 'ecdsa' in k.lower(): print(k, v)
PY
for rep in 1 2; do timeout 600 python bench.py --workload tx --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tx ms', round(d['ms_per_step'],4), 'txs/s', round(d['value']), 'ecdsa kernel ms', round(d['roofline']['kernel_ms'],4))"; done
