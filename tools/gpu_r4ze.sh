#!/bin/bash
set -u
out=gpurun_out/r4ze; mkdir -p $out
s=$(date +%s)
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench.err; echo "bench (no flags) rc=$? wall $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4ze/bench_default.json')); r=d['roofline']; c=d['cpu_baseline']
print(d['steps'], d['warmup'], d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'][:30], c['value'], c['kind'])
PY
timeout 900 python -m pytest tests/test_bench_multi_gpu_dryrun.py -m gpu -x -q 2>&1 | tail -1
