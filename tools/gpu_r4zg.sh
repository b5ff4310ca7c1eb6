#!/bin/bash
set -u
out=gpurun_out/r4zg; mkdir -p $out
for rep in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-live-pmc > $out/b$rep.json 2>/dev/null
python - $rep <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4zg/b{sys.argv[1]}.json')); r=d['roofline']
print("no-flags run", sys.argv[1], d['steps'], "oneshot ms", round(d['ms_per_step'],4), "span", round(r['kernel_ms'],4), "open", round(r['open_ms'],4), "pass", round(r['pass_kernel_ms'],4), "batch", round(r['batch_ms_per_witness'],4), "resident", round(r['resident_ms_per_pass'],4))
PY
done
timeout 900 python bench.py --no-cpu-baseline --no-live-pmc --steps 20 > $out/b3.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4zg/b3.json')); r=d['roofline']
print("steps 20 with legs", "oneshot ms", round(d['ms_per_step'],4), "span", round(r['kernel_ms'],4))
PY
