#!/bin/bash
set -u
out=gpurun_out/r4g; mkdir -p $out
timeout 900 python -m pytest tests/test_ecdsa.py tests/test_sign_circuit.py tests/test_bench_multi_gpu_dryrun.py -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 600 python bench.py --workload tx --no-cpu-baseline --steps 10 --warmup 2 > $out/bench_tx.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 500 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4g/bench_tx.json'))
print("tx value", d['value'], "ms/step", d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms'), d['roofline'].get('sig_circuit'))
PY
