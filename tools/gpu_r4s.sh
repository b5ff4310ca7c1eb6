#!/bin/bash
# Super circuit inside the default bench line (many streams alive) vs alone; GPU_MAX_HW_QUEUES 8 / 16 / 24
set -u
out=gpurun_out/r4s; mkdir -p $out
for q in 8 16 24; do
GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_q$q.json 2>/dev/null
GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 > $out/super_q$q.json 2>/dev/null
python - $q <<'PY'
import json,sys
q=sys.argv[1]
d=json.load(open(f'gpurun_out/r4s/bench_q{q}.json')); c=d['config']; r=d['roofline']
s=json.load(open(f'gpurun_out/r4s/super_q{q}.json'))
print(f"queues {q}: default line: oneshot {d['ms_per_step']:.4f} batch {r['batch_ms_per_witness']:.4f} state16 {c['state_2p16_ms_per_step']:.4f} tx {c['tx_2p14_ms_per_step']:.4f} super {c['super_2p20_ms_per_step']:.4f} | super alone {s['ms_per_step']:.4f}")
PY
done
