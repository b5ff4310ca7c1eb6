#!/bin/bash
set -u
out=gpurun_out/r4b; mkdir -p $out
tools/micro/rw_pack > $out/rw_pack.txt 2>&1; cat $out/rw_pack.txt
timeout 600 python bench.py --no-other-configs --no-cpu-baseline --no-fresh-leg --no-cold-leg > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -c 800 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b/bench.json'))
print("evm value", d['value'], "ms/step", d['ms_per_step'])
print({k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list))})
print(d.get('batch'))
PY
timeout 900 python -m pytest tests/test_evm_gpu.py tests/test_dropin_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
