#!/bin/bash
set -u
out=gpurun_out/r4zd; mkdir -p $out
s=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4zd/bench_default.json')); r=d['roofline']; c=d['cpu_baseline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'][:40])
print(c['value'], c['kind'], c['cores'], c['reference_measured_on'])
print(c['sample'])
print({k:round(v['value'],2) for k,v in c['legs'].items()})
PY
tail -2 $out/bench.err
