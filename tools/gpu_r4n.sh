#!/bin/bash
# A/B: result block published by the cold launch's last block (ZK_EVM_PUBLISH=1) vs copied by zk_collect (=0)
set -u
out=gpurun_out/r4n; mkdir -p $out
timeout 900 python -m pytest tests/test_evm_gpu.py tests/test_dropin_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log
for pb in 1 0 1 0; do
ZK_EVM_PUBLISH=$pb timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-session-leg --steps 30 --warmup 5 > $out/evm_pb$pb.json 2>/dev/null
python - $pb <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4n/evm_pb{sys.argv[1]}.json')); r=d['roofline']
print("publish", sys.argv[1], "oneshot ms", round(d['ms_per_step'],4), "span", round(r['kernel_ms'],4), "open", round(r['open_ms'],4), "pass", round(r['pass_kernel_ms'],4), "batch", round(r['batch_ms_per_witness'],4), "host us", r['host_us_in_open'], r['host_us_in_launch'], r['host_us_in_collect'], r['host_us_in_close'])
PY
done
