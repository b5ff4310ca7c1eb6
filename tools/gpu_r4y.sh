#!/bin/bash
# State kernel at one block per CU (LDS pad) so that the hot EVM build can run beside it in the block's pass
set -u
out=gpurun_out/r4y; mkdir -p $out
for pad in 0 16384 0 16384; do
ZK_STATE_LDS_PAD=$pad timeout 600 python bench.py --workload super --no-cpu-baseline --steps 20 --warmup 3 > $out/super_$pad.json 2>/dev/null
python - $pad <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4y/super_{sys.argv[1]}.json'))
pc=d['roofline']['per_circuit']; print("pad", sys.argv[1], round(d['ms_per_step'],4), {k:round(v['kernel_ms'],4) for k,v in pc.items()})
PY
done
for pad in 0 16384; do
ZK_STATE_LDS_PAD=$pad timeout 600 python bench.py --workload state --log-rows 20 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('state 2^20 pad $pad', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done
