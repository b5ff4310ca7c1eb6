#!/bin/bash
export PYTHONPATH=$PWD
run() { for i in 1 2 3; do python bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('  ', round(d['value']/1e9,3), round(d['ms_per_step'],4), {k: round(v['kernel_ms'], 3) for k, v in d['roofline']['per_circuit'].items()})
"; done; }
echo "== fork=1 dma=1"; run
echo "== fork=0 dma=1"; ZK_EVM_FORK=0 run
echo "== fork=1 dma=0"; ZK_STATE_DMA=0 run
echo "== fork=0 dma=0"; ZK_EVM_FORK=0 ZK_STATE_DMA=0 run
