#!/bin/bash
set -u
out=gpurun_out/r4w; mkdir -p $out
s=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4w/bench_default.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_committed_profile'], r['traffic_over_algorithmic'])
print(r['traffic_source'])
print(d['config']['super_2p20_ms_per_step'], d['config']['tx_2p14_ms_per_step'])
PY
tail -3 $out/bench.err
