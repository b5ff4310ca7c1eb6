#!/bin/bash
set -u
out=gpurun_out/r4zh; mkdir -p $out
s=$(date +%s)
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench.err; echo "bench (no flags, first command on the box) rc=$? wall $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4zh/bench_default.json')); r=d['roofline']; c=d['config']
print(d['steps'], d['value'], d['ms_per_step'], r['frac'], r['kernel_ms'], c['state_2p16_ms_per_step'], c['tx_2p14_ms_per_step'], c['super_2p20_ms_per_step'])
PY
for wl in state tx super; do timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],4))"; done
timeout 600 python -m pytest tests/test_bench_multi_gpu_dryrun.py -m gpu -x -q 2>&1 | tail -1
