#!/bin/bash
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_state_gpu.py tests/test_state_assign.py tests/test_super_circuit.py tests/test_cpu_backend.py -m gpu -x -q 2>&1 | tail -3
ZK_HIP_LIB=$PWD/zkevm_specs_amd/libzkevm_hip_prof.so LOGN=16 python tools/state_wave_timeline.py 2>&1 | grep "tag [246]\|kernel_ms\|alive\|exit  time"
run() { python bench.py --workload state --log-rows $1 --no-cpu-baseline --no-cold-leg --no-fresh-leg --steps 50 --warmup 5 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('   rows 2^$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
"; }
echo "== dma"; run 16; run 18; run 20
echo "== quad"; ZK_STATE_DMA=0 run 16; ZK_STATE_DMA=0 run 20
