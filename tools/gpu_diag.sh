#!/bin/bash
out=gpurun_out/diag; mkdir -p $out
for q in 4 8 12; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q python bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['value'], d['ms_per_step'], d['config'].get('per_circuit_kernel_ms'))
"
done
echo "== tx"
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --workload tx --no-cpu-baseline --steps 10 --warmup 2 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['value'], d['ms_per_step'])
"
done
