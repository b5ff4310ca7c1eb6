#!/bin/bash
out=gpurun_out/full; mkdir -p $out
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -c 600 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/full/bench.json'))
print("evm", d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], "fresh", d['fresh_witness']['rows_per_s'], d['fresh_witness']['one_shot_rows_per_s'])
print({k:(round(v['value']),v.get('cores')) for k,v in d['cpu_baseline']['legs'].items()})
for k,v in d['other_configs'].items():
    print(k, v['value'], v['ms_per_step'], v['roofline']['kernel'], v['roofline']['kernel_ms'], {kk:round(vv['value']) for kk,vv in v.get('cpu_baseline',{}).get('legs',{}).items()})
    if 'per_circuit' in v['roofline']: print("   ", {kk:(vv['rows'], round(vv['kernel_ms'],4)) for kk,vv in v['roofline']['per_circuit'].items()})
PY
