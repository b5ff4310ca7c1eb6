#!/bin/bash
set -u
out=gpurun_out/r4d; mkdir -p $out
ZK_HIP_LIB=$PWD/tools/micro/libzk_warm_inline.so timeout 300 python tools/evm_warm_timeline.py > $out/warm_inline2.txt 2>&1; tail -7 $out/warm_inline2.txt
timeout 900 python -m pytest tests/test_evm_gpu.py tests/test_super_circuit.py -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-fresh-leg --no-cold-leg --no-batch-leg > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4d/bench.json'))
print("evm value", d['value'], "ms/step", d['ms_per_step'])
r=d['roofline']
print({k:r[k] for k in ('kernel_ms','open_ms','pass_kernel_ms','frac','resident_ms_per_pass','resident_hot_kernel_ms','host_us_in_collect')})
o=d['other_configs']
for k,v in o.items(): print(k, v['value'], v['ms_per_step'], v.get('roofline',{}).get('kernel'), v.get('roofline',{}).get('kernel_ms'))
pc=o['super_2p20']['roofline']['per_circuit']; print({k:(v['rows'], round(v['kernel_ms'],4)) for k,v in pc.items()})
PY
