#!/bin/bash
set -u
out=gpurun_out/r4zf; mkdir -p $out
for st in 20 50 20 100 50; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-session-leg --no-batch-leg --no-live-pmc --steps $st --warmup 5 > $out/evm_$st.json 2>/dev/null
python - $st <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4zf/evm_{sys.argv[1]}.json')); r=d['roofline']
print("steps", sys.argv[1], "oneshot ms", round(d['ms_per_step'],4), "span", round(r['kernel_ms'],4), "open", round(r['open_ms'],4), "pass", round(r['pass_kernel_ms'],4), "host", r['host_us_in_open'], r['host_us_in_launch'], r['host_us_in_collect'])
PY
done
