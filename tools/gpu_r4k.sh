#!/bin/bash
# A/B: warm + cold EVM builds on the device's side stream (ZK_EVM_SIDE_STREAM=1) vs all on the session's stream (=0)
set -u
out=gpurun_out/r4k; mkdir -p $out
timeout 900 python -m pytest tests/test_evm_gpu.py tests/test_super_circuit.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for sd in 1 0 1 0; do
ZK_EVM_SIDE_STREAM=$sd timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/evm_s$sd.json 2>/dev/null
ZK_EVM_SIDE_STREAM=$sd timeout 600 python bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 > $out/super_s$sd.json 2>/dev/null
python - $sd <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4k/evm_s{sys.argv[1]}.json')); r=d['roofline']
print("side", sys.argv[1], "oneshot ms", round(d['ms_per_step'],4), "span", round(r['kernel_ms'],4), "pass", round(r['pass_kernel_ms'],4), "batch", round(r['batch_ms_per_witness'],4), "resident", round(r['resident_ms_per_pass'],4))
d=json.load(open(f'gpurun_out/r4k/super_s{sys.argv[1]}.json'))
pc=d['roofline']['per_circuit']; print("   super", round(d['ms_per_step'],4), {k:round(v['kernel_ms'],4) for k,v in pc.items()})
PY
done
