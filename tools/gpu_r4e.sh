#!/bin/bash
set -u
out=gpurun_out/r4e; mkdir -p $out
timeout 900 python -m pytest tests/test_ecdsa.py tests/test_sign_circuit.py -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 600 python bench.py --workload tx --no-cpu-baseline --steps 10 --warmup 2 > $out/bench_tx.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench_tx.json'))
print("tx value", d['value'], "ms/step", d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms'), d['roofline'].get('sig_circuit'))
PY
python tools/bench_row_kernels.py > $out/row_kernels.txt 2>&1; tail -12 $out/row_kernels.txt
