#!/bin/bash
set -u
out=gpurun_out/r4h; mkdir -p $out
for w in 1 0; do
ZK_EVM_WIRE=$w timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-fresh-leg --no-cold-leg --no-session-leg > $out/bench_w$w.json 2> $out/bench_w$w.err; echo "bench wire=$w rc=$?"
python - $w <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4h/bench_w{sys.argv[1]}.json'))
r=d['roofline']
print("evm value", d['value'], "ms/step", d['ms_per_step'], {k:r[k] for k in ('kernel_ms','open_ms','pass_kernel_ms','frac','host_us_in_open','host_us_in_launch','host_us_in_collect','batch_ms_per_witness')})
PY
done
timeout 1200 python -m pytest tests/test_dropin_gpu.py tests/test_evm_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -6 $out/pytest.log
