#!/bin/bash
# A/B: latency-bound blocks of evm_open_phase1_kernel dealt into the RW range (ZK_P1_LAT_SPREAD = 0 off / 1 / 2 / 4)
set -u
out=gpurun_out/r4m; mkdir -p $out
timeout 600 python -m pytest tests/test_evm_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log
for sp in 0 2 1 4 0 2; do
ZK_P1_LAT_SPREAD=$sp timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-session-leg --no-batch-leg --steps 30 --warmup 5 > $out/evm_sp$sp.json 2>/dev/null
python - $sp <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4m/evm_sp{sys.argv[1]}.json')); r=d['roofline']
print("spread", sys.argv[1], "oneshot ms", round(d['ms_per_step'],4), "span", round(r['kernel_ms'],4), "open", round(r['open_ms'],4), "pass", round(r['pass_kernel_ms'],4))
PY
done
timeout 120 tools/micro/rw_pack > $out/rw_pack_micro.txt 2>&1; cat $out/rw_pack_micro.txt
