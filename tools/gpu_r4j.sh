#!/bin/bash
set -u
out=gpurun_out/r4j; mkdir -p $out
for pr in 1 0 1 0; do
ZK_SUPER_EVM_PRIORITY=$pr timeout 600 python bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 > $out/super_p$pr.json 2>/dev/null
python - $pr <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4j/super_p{sys.argv[1]}.json'))
pc=d['roofline']['per_circuit']; print("prio", sys.argv[1], d['ms_per_step'], {k:round(v['kernel_ms'],4) for k,v in pc.items()})
PY
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$?"
